#!/usr/bin/env python3
"""host-buffer tier, one item per call (what the reference's VO loop would see): wall-clock latency including PCIe"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_visual_slam_amd as pkg
from stereo_visual_slam_amd import synth
vo = pkg.VO(device=0, max_batch=1, anms_num=500)
seq = synth.stereo_sequence(2, seed=0)
L, R = seq[0][0], seq[0][1]
def t(f, reps=20):
    f(); vo.sync()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    vo.sync()
    return (time.perf_counter() - t0) / reps * 1e3
k0, d0 = vo.feature_detection(L); k1, d1 = vo.feature_detection(seq[1][0])
p = synth.pnp_problem(M=300, seed=1)
w = synth.ba_window(n_kf=10, n_lm=3000, seed=2)
rows = [("feature_detection (ORB 3000 -> ANMS 500 -> rBRIEF), 1241x376", t(lambda: vo.feature_detection(L))),
        ("feature_matching %d x %d" % (len(d0), len(d1)), t(lambda: vo.feature_matching(d0, d1, 1.0))),
        ("disparity_map (SGBM), 1241x376", t(lambda: vo.disparity_map(L, R), 10)),
        ("motion_estimation (LM, 300 points)", t(lambda: vo.motion_estimation(p["xyz"], p["uv"], p["T0"]))),
        ("motion_estimation_ransac (100 hypotheses, 300 points)", t(lambda: vo.motion_estimation_ransac(p["xyz"], p["uv"], p["T0"]))),
        ("optimize_map (10 KF x 3000 landmarks, 10 its)", t(lambda: vo.optimize_map(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], True, False, 10), 5)),
        ("optimize_pose_only (10 its)", t(lambda: vo.optimize_pose_only(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], True, 10), 5))]
for name, ms in rows: print("%-62s %8.3f ms" % (name, ms))
vo.close()
# kernel-time split of the two slowest calls
vo = pkg.VO(device=0, max_batch=1, anms_num=500)
for name, f in (("feature_detection", lambda: vo.feature_detection(L)), ("disparity_map", lambda: vo.disparity_map(L, R))):
    f(); vo.sync(); vo.profile_enable(True); vo.profile_read()
    t0 = time.perf_counter(); f(); vo.sync(); wall = (time.perf_counter() - t0) * 1e3
    pr = vo.profile_read(); vo.profile_enable(False)
    print(name, "wall %.3f ms, kernels %.3f ms:" % (wall, sum(v[0] for v in pr.values())), {k: round(v[0], 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])})
vo.close()
