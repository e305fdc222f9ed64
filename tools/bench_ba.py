#!/usr/bin/env python3
"""micro-benchmark: local-BA schedule only (W windows of 10 KF x L landmarks), for kernel tuning"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stereo_visual_slam_amd.pipeline import KeyframePipeline

ap = argparse.ArgumentParser(); ap.add_argument("--windows", type=int, nargs="+", default=[256]); ap.add_argument("--landmarks", type=int, default=3000)
ap.add_argument("--reps", type=int, default=5); ap.add_argument("--unique", type=int, default=None, help="unique windows (default: one per batch item)")
ap.add_argument("--tracks", action="store_true", help="windows built on the device from a rendered sequence's own tracks (the default bench step's BA) instead of the config-4 shape")
ap.add_argument("--anms", type=int, default=1500)
a = ap.parse_args()
for W in a.windows:
    p = KeyframePipeline.__new__(KeyframePipeline)
    # build only the BA part: reuse the constructor with B tiny images is wasteful, so construct manually
    if a.tracks:
        pipe = KeyframePipeline(W, anms_num=a.anms, unique_frames=min(W, 24), ba_windows="tracks")
        pipe.stage_orb(); pipe.stage_stereo_match(); pipe.stage_track()
    else:
        pipe = KeyframePipeline(W, anms_num=500, n_lm=a.landmarks, unique_frames=2, unique_windows=a.unique)
    pipe.vo.profile_enable(True)
    for _ in range(2): pipe.stage_ba()
    torch.cuda.synchronize(); pipe.vo.profile_read()
    t0 = time.perf_counter()
    for _ in range(a.reps): pipe.stage_ba()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
    pr = pipe.vo.profile_read()
    print("W=%d  %.3f ms/schedule-batch  %.1f windows/s  kernel=%.3f ms%s" % (W, dt * 1e3, W / dt, pr["lm_window_kernel"][0] / a.reps,
          ("  build_windows=%.3f ms" % (pr["build_windows_kernels"][0] / a.reps)) if a.tracks else ""), flush=True)
    pipe.close()
