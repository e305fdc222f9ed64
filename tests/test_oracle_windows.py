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


@pytest.mark.parametrize("rule", [0, 1])
def test_build_windows_hand_worked(oracle, rule):
    """(both track rules: the one keypoint that is tracked into a frame without a depth of its own -- (1, 1), landmark A -- has a match into frame 2, but A's
    map position projects hundreds of pixels from that keypoint, so the reference's rule rejects the link the old convention never considered)"""
    kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel = _scenario(oracle)
    w = oracle.build_windows(kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel, n_kf=2, track_rule=rule)
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


def test_track_survives_frames_without_depth(oracle):
    """[r6] VO::tracking matches the current frame against EVERY feature of the last frame (visual_odometry.cpp:568-599), and motion_estimation hands
    solvePnPRansac the landmark's map position (:260-270).  Hand-worked: camera moving 1 m per frame along z, K = KITTI's.
      A  created UNRELIABLE at (0, 0) -> tracked (pose inlier) to (1, 0), which has NO depth -> its map position (the creation point) reprojects 2.5 px from (2, 2):
         the track goes on; (2, 2) has a reliable depth: A takes that point (:391-401) -> (3, 0) through the pose stage (inlier), no depth there -> (4, 0) lies 1.4 px
         from the projection of the UPDATED position and 13 px from that of the creation point: the track goes on only because pt_3d_ is what the map holds then
      B  created reliable at (0, 1) -> (1, 1) inlier -> its match into frame 2 is a pose-stage OUTLIER: (2, 3) founds landmark C instead (it has a depth)
      C  (2, 3) -> (3, 1) inlier, no depth -> (4, 1) 0.7 px from C's projection: goes on
      D  created at (0, 2) -> (1, 2) inlier, no depth -> (2, 1) lies 6 px from D's projection: rejected, and (2, 1) has no depth: not a feature
    Under the convention of rounds 4-5 (track_rule 0) every track ends at its first keypoint without depth."""
    K = oracle.K_KITTI
    fx, fy, cx, cy = K
    F, cap = 5, 4
    proj = lambda pw, f: np.array([fx * pw[0] / (pw[2] - f) + cx, fy * pw[1] / (pw[2] - f) + cy])   # T_c_w of frame f: p_c = p_w - (0, 0, f)
    A0 = np.array([1.0, 0.5, 20.0]); A1w = np.array([1.3, 0.5, 20.1]); Bw = np.array([-2.0, 0.3, 15.0]); Dw = np.array([4.0, -1.0, 30.0]); Cw = np.array([-3.0, -1.0, 27.0])
    kps = np.zeros((F, cap), oracle.KEYPOINT_DTYPE)
    for f in range(F):
        kps["x"][f] = 50 + 10 * np.arange(cap) + f; kps["y"][f] = 20 + np.arange(cap)
    def put(f, i, uv):
        kps["x"][f, i], kps["y"][f, i] = np.float32(uv[0]), np.float32(uv[1])
    put(2, 2, proj(A0, 2) + (1.5, -2.0)); put(4, 0, proj(A1w, 4) + (1.0, 1.0)); put(4, 1, proj(Cw, 4) + (0.5, -0.5)); put(2, 1, proj(Dw, 2) + (6.0, 0.0))
    assert np.linalg.norm(proj(A0, 4) - (proj(A1w, 4) + (1.0, 1.0))) > 10
    lr = np.zeros((F, cap), oracle.DMATCH_DTYPE); nlr = np.zeros(F, np.int32)
    xyz = np.zeros((F, cap, 3), np.float32); valid = np.zeros((F, cap), np.uint8); rel = np.zeros((F, cap), np.uint8)
    def depth(f, q, pw, r):
        m = int(nlr[f]); lr[f, m]["queryIdx"] = q; lr[f, m]["trainIdx"] = q; valid[f, m] = 1; rel[f, m] = r
        xyz[f, m] = (pw[0], pw[1], pw[2] - f); nlr[f] += 1          # camera coordinates of frame f
    depth(0, 0, A0, 0); depth(0, 1, Bw, 1); depth(0, 2, Dw, 1)
    depth(1, 1, Bw + (-0.1, 0, 0), 1)
    depth(2, 2, A1w, 1); depth(2, 3, Cw, 1)
    f2f = np.zeros((F - 1, cap), oracle.DMATCH_DTYPE); nf2f = np.zeros(F - 1, np.int32)
    for i, ms in enumerate([[(0, 0), (1, 1), (2, 2)], [(0, 2), (1, 3), (2, 1)], [(2, 0), (3, 1)], [(0, 0), (1, 1)]]):
        for k, (q, t) in enumerate(ms):
            f2f[i, k]["queryIdx"] = q; f2f[i, k]["trainIdx"] = t
        nf2f[i] = len(ms)
    inl = np.zeros((F - 1, cap), np.uint8)
    inl[0, :3] = 1          # pair 0: all three queries own a depth: inputs 0, 1, 2
    inl[1, 0] = 0           # pair 1: the only input is (1, 1) -> (2, 3): an outlier
    inl[2, :2] = 1          # pair 2: inputs (2, 2) -> (3, 0) and (2, 3) -> (3, 1)
    T_rel = np.tile(np.array([0, 0, 0, 1, 0, 0, -1.0]), (F - 1, 1))
    w = oracle.build_windows(kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel, n_kf=10, K=K, reproj_thr=4.0, track_rule=1)
    assert w["status"] == 0 and np.allclose(w["T"][4, :5, 6], -np.arange(5))
    # the map after keyframe 4: B and D with two observations (order: first observation (0, 1) before (0, 2)), C with three, A with five
    l0, e0 = int(w["lm_off"][4]), int(w["edge_off"][4])
    assert int(w["lm_off"][5]) - l0 == 4 and int(w["edge_off"][5]) - e0 == 12
    assert w["lm_idx"][e0:e0 + 12].tolist() == [0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 3]
    assert w["kf_idx"][e0:e0 + 12].tolist() == [0, 1, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4]
    uv = w["uv"][e0:e0 + 12]
    assert np.array_equal(uv[4:7], np.stack([[kps["x"][2, 3], kps["y"][2, 3]], [kps["x"][3, 1], kps["y"][3, 1]], [kps["x"][4, 1], kps["y"][4, 1]]]))
    assert np.array_equal(uv[7:], np.stack([[kps["x"][f, i], kps["y"][f, i]] for f, i in ((0, 0), (1, 0), (2, 2), (3, 0), (4, 0))]))
    assert np.allclose(w["xyz"][l0:l0 + 4], np.stack([Bw, Dw, Cw, A1w]), atol=1e-5) and w["reliable"][l0:l0 + 4].tolist() == [1, 1, 1, 1]
    # ... and as of keyframe 1 (window 1) A still sits at its creation point, unreliable
    l1 = int(w["lm_off"][1]); n1 = int(w["lm_off"][2]) - l1
    assert n1 == 3 and np.allclose(w["xyz"][l1:l1 + 3], np.stack([A0, Bw, Dw]), atol=1e-5) and w["reliable"][l1:l1 + 3].tolist() == [0, 1, 1]
    # the old convention: A ends at (1, 0); (2, 2) founds a landmark of its own that reaches (3, 0); C ends at (3, 1)
    w0 = oracle.build_windows(kps, lr, nlr, xyz, valid, rel, f2f, nf2f, inl, T_rel, n_kf=10, K=K, reproj_thr=4.0, track_rule=0, lm_capacity=64, edge_capacity=128)
    assert w0["status"] == 0
    l0, e0 = int(w0["lm_off"][4]), int(w0["edge_off"][4])
    assert int(w0["lm_off"][5]) - l0 == 5 and int(w0["edge_off"][5]) - e0 == 10
