#!/usr/bin/env python3
"""micro-benchmark: batched cross-check matcher only (B items of N x N random descriptors), for kernel tuning"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_visual_slam_amd as pkg
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--rows", type=int, default=1500)
ap.add_argument("--cap", type=int, default=4096); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
B, N, cap = a.batch, a.rows, a.cap
vo = pkg.VO(device=0, max_batch=B)
g = torch.Generator(device="cuda"); g.manual_seed(0)
q = torch.randint(0, 256, (B, cap, 32), dtype=torch.uint8, device="cuda", generator=g)
t = torch.randint(0, 256, (B, cap, 32), dtype=torch.uint8, device="cuda", generator=g)
n = torch.full((2 * B,), N, dtype=torch.int32, device="cuda")
gap = torch.ones(B, dtype=torch.float64, device="cuda")
out = torch.empty((B, cap, 16), dtype=torch.uint8, device="cuda"); nout = torch.zeros(B, dtype=torch.int32, device="cuda")
run = lambda: vo.feature_matching_dev(q.data_ptr(), cap * 32, n.data_ptr(), t.data_ptr(), cap * 32, n.data_ptr() + 4 * B, gap.data_ptr(), 1, B, cap,
                                      out.data_ptr(), cap, nout.data_ptr())
run(); vo.sync(); vo.profile_enable(True); vo.profile_read()
t0 = time.perf_counter()
for _ in range(a.reps): run()
vo.sync(); dt = (time.perf_counter() - t0) / a.reps
pr = vo.profile_read()
print("B=%d N=%d  %.3f ms/call  %.2f T pair-distances/s" % (B, N, dt * 1e3, B * N * N / dt / 1e12))
print({k: round(v[0] / a.reps, 4) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])})
vo.close()
