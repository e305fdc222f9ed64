# edges per keyframe inside the built BA windows (how unbalanced is one-wave-per-keyframe in pose_only_wave_kernel?)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stereo_visual_slam_amd.pipeline import KeyframePipeline
pipe = KeyframePipeline(64, anms_num=1500, unique_frames=64, ba_windows="tracks")
pipe.stage_orb(); pipe.stage_stereo_match(); pipe.stage_track(); pipe.stage_build_windows()
o = pipe.download()
eo = o["ba_e_off"]; kf = o["ba_kf"]
rows = []
for w in range(12, 64):
    k = kf[eo[w]:eo[w + 1]]
    c = np.bincount(k, minlength=10)
    rows.append(c)
rows = np.array(rows)
print("mean edges per keyframe position:", rows.mean(0).round(0))
print("per window: max / mean = %.2f (avg), total edges %.0f" % ((rows.max(1) / rows.mean(1)).mean(), rows.sum(1).mean()))
pipe.close()
