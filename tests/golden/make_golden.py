#!/usr/bin/env python3
"""Regenerates the committed golden fixtures in tests/golden/ from seeded synthetic inputs through the CPU oracle.

The reference ships no tests, fixtures or golden vectors and cannot be built or imported here, so these
are ORACLE outputs, not reference outputs: they freeze the oracle's behaviour (regression pin for tests/test_golden.py on
CPU) and give the GPU tests fixed vectors that do not depend on the oracle being rebuilt on the GPU box.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from stereo_visual_slam_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    O.build()
    # ORB: 320x200, budget 600, ANMS 150
    img = synth.noise_image(21, 320, 200)
    det = O.orb_detect(img, 600)
    kps, desc = O.feature_detection(img, 600, 150)
    np.savez_compressed(os.path.join(HERE, "orb_320x200.npz"), img=img, det=det, kps=kps, desc=desc)
    # matcher: planted matches + ties
    q, t = synth.random_descriptors(220, 190, seed=31, tie_frac=0.1)
    np.savez_compressed(os.path.join(HERE, "match_220x190.npz"), q=q, t=t, raw=O.bf_match_xcheck(q, t), gated=O.feature_matching(q, t, 1.0),
                        gated_gap3=O.feature_matching(q, t, 3.0))
    # geometry
    rng = np.random.default_rng(41)
    n = 64
    Z = rng.uniform(3, 500, n); uL = rng.uniform(0, 1241, n); v = rng.uniform(0, 376, n)
    uvL = np.stack([uL, v], 1).astype(np.float32); uvR = np.stack([uL - synth.FX * synth.BASELINE / Z, v], 1).astype(np.float32)
    T = O.se3_exp(rng.normal(0, 0.2, 6))
    xyz, valid, rel = O.triangulate_dlt(uvL, uvR, T)
    np.savez_compressed(os.path.join(HERE, "triangulate_64.npz"), uvL=uvL, uvR=uvR, T=T, xyz=xyz, valid=valid, rel=rel)
    # motion-only pose
    p = synth.pnp_problem(M=120, seed=51)
    Tp, inl, ninl, st = O.pnp_motion_only(p["xyz"], p["uv"], p["T0"], iters=10)
    np.savez_compressed(os.path.join(HERE, "pnp_120.npz"), xyz=p["xyz"], uv=p["uv"], T0=p["T0"], T=Tp, inlier=inl, n_inliers=ninl,
                        chi2_iter=np.array(st["chi2_iter"]), chi2_init=st["chi2_init"])
    # local BA + pose-only window
    w = synth.ba_window(n_kf=10, n_lm=200, seed=61)
    Tb, xb, chi2, st = O.local_ba(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], iters=10, update_poses=True, update_lms=True)
    th, inlb, ni, no = O.chi2_classify(chi2, w["lm_idx"], np.ones(200, np.uint8))
    To, chi2o, sto = O.pose_only_window(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], iters=10)
    np.savez_compressed(os.path.join(HERE, "ba_10x200.npz"), T0=w["T0"], xyz=w["xyz"], kf_idx=w["kf_idx"], lm_idx=w["lm_idx"], uv=w["uv"],
                        T_ba=Tb, xyz_ba=xb, chi2_ba=chi2, chi2_iter_ba=np.array(st["chi2_iter"]), thr_ba=th, inlier_ba=inlb,
                        T_po=To, chi2_po=chi2o, chi2_iter_po=np.array(sto["chi2_iter"]))
    # dense stereo: rendered 360x120 pair with sensor noise on the right view (speckles, LR rejections)
    sc = synth.Scene(71); Tc = synth.trajectory(1, 71)[0]
    L, _ = sc.render(Tc, 360, 120); R, _ = sc.render(Tc, 360, 120, x_offset=synth.BASELINE)
    R = np.clip(R.astype(int) + np.random.default_rng(71).integers(-8, 9, R.shape), 0, 255).astype(np.uint8)
    d16, raw = O.sgbm_compute(L, R, return_raw=True)
    np.savez_compressed(os.path.join(HERE, "sgbm_360x120.npz"), left=L, right=R, disp16=d16, raw16=raw, disparity=O.disparity_map(L, R))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
