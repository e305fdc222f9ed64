#!/usr/bin/env python3
"""micro-benchmark: SGBM disparity stage only (B stereo pairs, device-resident), for kernel tuning"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stereo_visual_slam_amd as pkg
from stereo_visual_slam_amd import synth
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--unique", type=int, default=2)
a = ap.parse_args()
w, h, pitch = synth.W_KITTI, synth.H_KITTI, 1280
seq = synth.stereo_sequence(a.unique, seed=0)
buf = np.zeros((2, a.batch, h, pitch), np.uint8)
for b in range(a.batch):
    L, R = seq[b % a.unique][:2]
    buf[0, b, :, :w] = L; buf[1, b, :, :w] = R
vo = pkg.VO(device=0, max_batch=1)
d = torch.from_numpy(buf).cuda()
out = torch.empty((a.batch, h, w), dtype=torch.float32, device="cuda")
run = lambda: vo.disparity_map_dev(d[0].data_ptr(), d[1].data_ptr(), h * pitch, pitch, w, h, a.batch, out.data_ptr())
run(); vo.sync()
vo.profile_enable(True); vo.profile_read()
t0 = time.perf_counter()
for _ in range(a.reps): run()
vo.sync(); dt = (time.perf_counter() - t0) / a.reps
pr = vo.profile_read()
print("B=%d  %.3f ms  %.1f pairs/s  %.3f ms/pair  device MB %.0f" % (a.batch, dt * 1e3, a.batch / dt, dt * 1e3 / a.batch, vo.device_bytes / 1e6))
print({k: round(v[0] / a.reps, 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])})
print("valid fraction", float((out[:, :, 96:] >= 0).float().mean()))
vo.close()
