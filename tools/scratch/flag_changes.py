"""How often does a pass of the BA schedule change the landmark flags of a window?  (If pass k flags nothing new, pass k + 1 repeats it.)"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from importlib import import_module
KeyframePipeline = import_module("stereo-visual-slam_amd.pipeline").KeyframePipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
depth = sys.argv[2] if len(sys.argv) > 2 else "match"
pipe = KeyframePipeline(B, anms_num=1500, unique_frames=64, ba_windows="tracks", depth=depth, pose="ransac" if depth == "sgbm" else "lm")
pipe.stage_orb(); pipe.stage_stereo_match(); pipe.stage_track(); pipe.stage_build_windows()
pipe.vo.sync(); torch.cuda.synchronize()
off = pipe.ba_lm_off.cpu().numpy()
def flags():
    pipe.vo.sync(); torch.cuda.synchronize()
    return pipe.ba_inl.cpu().numpy().copy()
f0 = flags()
res = []
prev = f0
for k, its in enumerate((5, 5, 10)):
    pipe.vo.ba_batch_dev(pipe.ba_batch, schedule=0, mode=0, iters=its, update_poses=1 if k == 2 else 0, update_lms=0)
    f = flags()
    ch = np.array([(f[off[b]:off[b + 1]] != prev[off[b]:off[b + 1]]).sum() for b in range(B)])
    print("pass %d (%d its): windows with changed flags %d / %d; changed flags per window mean %.1f max %d; inliers now %.4f" % (
        k + 1, its, int((ch > 0).sum()), B, ch.mean(), ch.max(), f[:off[-1]].mean()))
    prev = f
