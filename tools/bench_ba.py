#!/usr/bin/env python3
"""micro-benchmark: local-BA schedule only (W windows of 10 KF x L landmarks), for kernel tuning"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stereo_visual_slam_amd.pipeline import KeyframePipeline

ap = argparse.ArgumentParser(); ap.add_argument("--windows", type=int, nargs="+", default=[256]); ap.add_argument("--landmarks", type=int, default=3000)
ap.add_argument("--reps", type=int, default=5); ap.add_argument("--unique", type=int, default=None, help="unique windows (default: one per batch item)")
a = ap.parse_args()
for W in a.windows:
    p = KeyframePipeline.__new__(KeyframePipeline)
    # build only the BA part: reuse the constructor with B tiny images is wasteful, so construct manually
    pipe = KeyframePipeline(W, anms_num=500, n_lm=a.landmarks, unique_frames=2, unique_windows=a.unique)
    pipe.vo.profile_enable(True)
    for _ in range(2): pipe.stage_ba()
    torch.cuda.synchronize(); pipe.vo.profile_read()
    t0 = time.perf_counter()
    for _ in range(a.reps): pipe.stage_ba()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
    pr = pipe.vo.profile_read()
    print("W=%d  %.3f ms/schedule-batch  %.1f windows/s  kernel=%.3f ms" % (W, dt * 1e3, W / dt, pr["lm_window_kernel"][0] / a.reps), flush=True)
    pipe.close()
