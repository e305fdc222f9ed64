import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stereo_visual_slam_amd.pipeline import KeyframePipeline
B = 24
pipe = KeyframePipeline(B, anms_num=1500, unique_frames=24, seed=11, ba_windows="tracks")
pipe.stage_orb(); pipe.stage_stereo_match(); pipe.stage_track(); pipe.stage_build_windows()
T0 = pipe.ba_T.clone(); inl0 = pipe.ba_inl.clone()
res = []
for lanes in (256, 512, -1, 256):
    pipe.ba_T.copy_(T0); pipe.ba_inl.copy_(inl0)
    pipe.vo.set_tuning(ba_lanes=lanes)
    pipe.vo.ba_batch_dev(pipe.ba_batch, schedule=1)
    passes = pipe.vo.ba_schedule_passes(B)
    torch.cuda.synchronize()
    res.append((pipe.ba_T.cpu().numpy().copy(), pipe.ba_inl.cpu().numpy().copy(), passes.copy()))
o = pipe.download()
lo = o["ba_lm_off"]
for i in (1, 2, 3):
    dT = (res[0][0].view(np.uint64) != res[i][0].view(np.uint64)).reshape(B, -1).sum(1)
    dI = [(int((res[0][1][lo[w]:lo[w+1]] != res[i][1][lo[w]:lo[w+1]]).sum())) for w in range(B)]
    print("run", i, "passes equal", np.array_equal(res[0][2], res[i][2]), "windows with pose diff", np.nonzero(dT > 0)[0].tolist(), "nk", [int(x) for x in o["ba_nkf"][:B]] if "ba_nkf" in o else None, "flag diffs", [w for w in range(B) if dI[w]])
print("landmarks per window", np.diff(lo)[:B].tolist())
pipe.close()
