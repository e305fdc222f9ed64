"""GPU parity: K8 depth -> world landmarks (find_3d contract and rectified-stereo DLT) vs the CPU oracle.
Reference path: Frame::find_3d types_def.cpp:9-18, VO::set_ref_3d_position visual_odometry.cpp:176-217."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-4  # north_star: landmarks within 1e-4 relative


def test_triangulate_parity(vo, oracle, synth):
    rng = np.random.default_rng(0)
    n = 2000
    Z = rng.uniform(2, 600, n)
    uL = rng.uniform(0, 1241, n); v = rng.uniform(0, 376, n)
    uvL = np.stack([uL, v], 1).astype(np.float32)
    uvR = np.stack([uL - synth.FX * synth.BASELINE / Z + rng.normal(0, 0.3, n), v + rng.normal(0, 0.3, n)], 1).astype(np.float32)
    uvR[:5] = uvL[:5]  # zero disparity -> invalid
    T = synth.perturb_pose(synth.se3_from_Rt(np.eye(3), [0.3, -0.1, 2.0]), rng, 0.2)
    gx, gv, gr = vo.triangulate(uvL, uvR, T)
    wx, wv, wr = oracle.triangulate_dlt(uvL, uvR, T)
    assert (gv == wv).all() and (gr == wr).all()
    ok = wv.astype(bool)
    assert np.allclose(gx[ok], wx[ok], rtol=RTOL, atol=1e-5)
    assert 0 < ok.sum() < n


def test_find_3d_disparity_parity(vo, oracle, synth):
    rng = np.random.default_rng(1)
    h, w = 376, 1241
    disp = rng.uniform(0.5, 90, (h, w)).astype(np.float32)
    disp[rng.random((h, w)) < 0.1] = -1.0   # invalid SGBM pixels
    disp[rng.random((h, w)) < 0.02] = 0.0
    kps = np.zeros(3000, oracle.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, w - 1, len(kps)); kps["y"] = rng.uniform(0, h - 1, len(kps))
    T = synth.perturb_pose(synth.se3_from_Rt(np.eye(3), [0, 0, 0]), rng, 0.3)
    gx, gv, gr = vo.find_3d_disparity(kps, disp, T)
    wx, wv, wr = oracle.find_3d_disparity(kps, disp, T)
    assert (gv == wv).all() and (gr == wr).all()
    ok = wv.astype(bool)
    assert np.allclose(gx[ok], wx[ok], rtol=RTOL, atol=1e-5)


def test_check_motion(vo, oracle, synth):
    rng = np.random.default_rng(2)
    for i in range(50):
        T = synth.perturb_pose(synth.se3_from_Rt(np.eye(3), [0, 0, 0]), rng, rng.uniform(0.1, 4.0))
        n = int(rng.integers(0, 30)); gap = float(rng.integers(1, 4))
        assert vo.check_motion_estimation(n, T, gap) == oracle.check_motion(n, T, gap)
