#!/usr/bin/env python3
"""micro-benchmark: ORB stage only (2B images), for kernel tuning"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_visual_slam_amd.pipeline import KeyframePipeline
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--anms", type=int, default=1500)
a = ap.parse_args()
pipe = KeyframePipeline(a.batch, anms_num=a.anms, with_ba=False)
for _ in range(2): pipe.stage_orb()
torch.cuda.synchronize(); pipe.vo.profile_enable(True); pipe.vo.profile_read()
t0 = time.perf_counter()
for _ in range(a.reps): pipe.stage_orb()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
pr = pipe.vo.profile_read()
print("B=%d images=%d  %.3f ms  %.1f images/s  %.2f us/image" % (a.batch, 2 * a.batch, dt * 1e3, 2 * a.batch / dt, dt * 1e6 / (2 * a.batch)))
print({k: round(v[0] / a.reps, 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])})
import numpy as np
cc = pipe.vo  # corner stats
pipe.close()
