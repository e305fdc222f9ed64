"""oracle/windows.c against a hand-worked map (CPU).  The scenario exercises every rule of VO::insert_key_frame's bookkeeping
(visual_odometry.cpp:363-424): an observation added through a frame-to-frame match the pose stage kept, a match it rejected, a match whose
last-frame keypoint has no depth (never an input of the pose stage), landmark creation, the unreliable -> reliable depth update (:391-401),
and the sliding window (map.hpp:22) with the landmark order / window-local indices vslam_ba_batch requires."""
import numpy as np
import pytest


def _scenario(oracle):
    F, cap = 3, 4
    kps = np.zeros((F, cap), oracle.KEYPOINT_DTYPE)
    for f in range(F):
        kps["x"][f] = 100 * f + 10 * np.arange(cap); kps["y"][f] = 100 * f + 10 * np.arange(cap) + 1
    lr = np.zeros((F, cap), oracle.DMATCH_DTYPE); nlr = np.zeros(F, np.int32)
    xyz = np.zeros((F, cap, 3), np.float32); valid = np.zeros((F, cap), np.uint8); rel = np.zeros((F, cap), np.uint8)
    def depth(f, m, q, v, r):
        lr[f, m]["queryIdx"] = q; lr[f, m]["trainIdx"] = q; valid[f, m] = v; rel[f, m] = r
        xyz[f, m] = (f + 0.25 * q, 1 + q, 10 + 5 * f + q)
    depth(0, 0, 0, 1, 1); depth(0, 1, 1, 1, 0); nlr[0] = 2
    depth(1, 0, 0, 1, 1); depth(1, 1, 2, 1, 1); nlr[1] = 2
    depth(2, 0, 0, 1, 1); nlr[2] = 1
    f2f = np.zeros((F - 1, cap), oracle.DMATCH_DTYPE); nf2f = np.zeros(F - 1, np.int32)
    for i, ms in enumerate([[(0, 1), (1, 0), (2, 3)], [(0, 0), (1, 2), (2, 1)]]):
        for k, (q, t) in enumerate(ms):
            f2f[i, k]["queryIdx"] = q; f2f[i, k]["trainIdx"] = t
        nf2f[i] = len(ms)
    inl = np.zeros((F - 1, cap), np.uint8)
    inl[0, :2] = (1, 1)   # pair 0: inputs = matches 0, 1 (match 2's query has no depth)
    inl[1, :2] = (1, 0)   # pair 1: inputs = matches 0, 2 (match 1's query (1, 1) has no depth of its own); the second is an outlier
    T_rel = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], np.float64), (F - 1, 1))
    return kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel


def test_build_windows_hand_worked(oracle):
    kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel = _scenario(oracle)
    w = oracle.build_windows(kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel, n_kf=2)
    assert w["status"] == 0
    assert w["n_kf"].tolist() == [1, 2, 2]
    assert w["lm_off"].tolist() == [0, 2, 5, 8] and w["edge_off"].tolist() == [0, 2, 7, 11]
    # window 0: landmarks A (0, 0) and B (0, 1); B's depth is unreliable at this time
    assert w["kf_idx"][0:2].tolist() == [0, 0] and w["lm_idx"][0:2].tolist() == [0, 1]
    assert w["reliable"][0:2].tolist() == [1, 0]
    assert np.array_equal(w["xyz"][0], xyz[0, 0]) and np.array_equal(w["xyz"][1], xyz[0, 1])
    # window 1 = keyframes 0, 1: A {(0,0), (1,1)}, B {(0,1), (1,0)} now reliable with the point of (1, 0), C created at (1, 2).
    # Landmark order: by observation count, then by the first observation inside the window -> C (1 obs), A, B (2 obs each)
    assert w["lm_idx"][2:7].tolist() == [0, 1, 1, 2, 2] and w["kf_idx"][2:7].tolist() == [1, 0, 1, 0, 1]
    assert w["reliable"][2:5].tolist() == [1, 1, 1]
    assert np.array_equal(w["xyz"][2], xyz[1, 1]) and np.array_equal(w["xyz"][3], xyz[0, 0]) and np.array_equal(w["xyz"][4], xyz[1, 0])
    uv = w["uv"][2:7]
    assert uv[:, 0].tolist() == [kps["x"][1, 2], kps["x"][0, 0], kps["x"][1, 1], kps["x"][0, 1], kps["x"][1, 0]]
    # window 2 = keyframes 1, 2: A (1,1) and C (1,2) with one observation, then B (1,0) [+ (2,0)]; the rejected match (1,2) -> (2,1) adds
    # nothing, and (2,1) has no depth: not a feature
    assert w["lm_idx"][7:11].tolist() == [0, 1, 2, 2] and w["kf_idx"][7:11].tolist() == [0, 0, 0, 1]
    assert np.array_equal(w["xyz"][5], xyz[0, 0]) and np.array_equal(w["xyz"][6], xyz[1, 1]) and np.array_equal(w["xyz"][7], xyz[1, 0])
    assert (w["lm_inlier"][:8] == 1).all()
    assert np.allclose(w["T"][:, :, 3], 1) and np.allclose(w["T"][:, :, :3], 0)


def test_build_windows_world_frame_and_capacity(oracle):
    """poses are the chained relative poses; landmark positions go through T_c_w^-1 of their source frame; a too small capacity
    empties the windows from the first one that does not fit and reports it"""
    kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel = _scenario(oracle)
    T_rel[0] = oracle.se3_exp(np.array([0.1, -0.2, 1.0, 0.01, 0.02, -0.015]))
    T_rel[1] = oracle.se3_exp(np.array([-0.05, 0.1, 0.9, -0.02, 0.01, 0.01]))
    w = oracle.build_windows(kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel, n_kf=2)
    G1 = T_rel[0]; G2 = oracle.se3_mul(T_rel[1], T_rel[0])
    assert np.allclose(w["T"][1, 1], G1) and np.allclose(w["T"][2, 0], G1) and np.allclose(w["T"][2, 1], G2)
    pw = oracle.se3_act(oracle.se3_inv(G1), xyz[1, 0].astype(np.float64))
    assert np.allclose(w["xyz"][4], pw, rtol=1e-6)          # landmark B in window 1: the point seen from frame 1, in the world frame
    small = oracle.build_windows(kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel, n_kf=2, lm_capacity=6, edge_capacity=64)
    assert small["status"] == 1 and small["lm_off"].tolist() == [0, 2, 5, 5] and small["edge_off"].tolist() == [0, 2, 7, 7]
