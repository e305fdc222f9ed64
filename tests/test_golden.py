"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from seeded inputs through the oracle).
CPU: the oracle still reproduces them (regression pin).  GPU: the HIP path reproduces them through the C-ABI."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def _kps_equal(a, b):
    assert len(a) == len(b)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(a[f], b[f]), f


# ---------------------------------------------------------------- CPU: oracle regression pin
def test_oracle_reproduces_golden_orb(oracle):
    g = _load("orb_320x200.npz")
    _kps_equal(oracle.orb_detect(g["img"], 600), g["det"])
    k, d = oracle.feature_detection(g["img"], 600, 150)
    _kps_equal(k, g["kps"]); assert np.array_equal(d, g["desc"])


def test_oracle_reproduces_golden_match_geom_lm(oracle):
    g = _load("match_220x190.npz")
    assert np.array_equal(oracle.bf_match_xcheck(g["q"], g["t"]), g["raw"])
    assert np.array_equal(oracle.feature_matching(g["q"], g["t"], 1.0), g["gated"])
    assert np.array_equal(oracle.feature_matching(g["q"], g["t"], 3.0), g["gated_gap3"])
    g = _load("triangulate_64.npz")
    x, v, r = oracle.triangulate_dlt(g["uvL"], g["uvR"], g["T"])
    assert np.array_equal(v, g["valid"]) and np.array_equal(r, g["rel"]) and np.allclose(x, g["xyz"], rtol=1e-6)
    g = _load("pnp_120.npz")
    T, inl, n, st = oracle.pnp_motion_only(g["xyz"], g["uv"], g["T0"], iters=10)
    assert np.allclose(T, g["T"], rtol=1e-9, atol=1e-12) and n == int(g["n_inliers"]) and np.array_equal(inl, g["inlier"])
    g = _load("ba_10x200.npz")
    T, x, chi2, st = oracle.local_ba(g["T0"], g["xyz"], g["kf_idx"], g["lm_idx"], g["uv"], iters=10, update_poses=True, update_lms=True)
    assert np.allclose(T, g["T_ba"], rtol=1e-8, atol=1e-10) and np.allclose(x, g["xyz_ba"], rtol=1e-6)
    T, chi2, st = oracle.pose_only_window(g["T0"], g["xyz"], g["kf_idx"], g["lm_idx"], g["uv"], iters=10)
    assert np.allclose(T, g["T_po"], rtol=1e-8, atol=1e-10)


def test_oracle_reproduces_golden_sgbm(oracle):
    g = _load("sgbm_360x120.npz")
    d16, raw = oracle.sgbm_compute(g["left"], g["right"], return_raw=True)
    assert np.array_equal(d16, g["disp16"]) and np.array_equal(raw, g["raw16"])
    assert np.array_equal(oracle.disparity_map(g["left"], g["right"]), g["disparity"])
    assert (g["disp16"] != g["raw16"]).any() and (g["disp16"][:, 96:] >= 0).mean() > 0.5  # the fixture exercises the post-filters


# ---------------------------------------------------------------- GPU: HIP path vs the committed vectors
@pytest.mark.gpu
def test_hip_reproduces_golden_orb(pkg):
    g = _load("orb_320x200.npz")
    ctx = pkg.VO(device=0, max_batch=1, img_w=320, img_h=200, orb_nfeatures=600, anms_num=150)
    try:
        _kps_equal(ctx.orb_detect(g["img"]), g["det"])
        k, d = ctx.feature_detection(g["img"])
        _kps_equal(k, g["kps"]); assert np.array_equal(d, g["desc"])
    finally:
        ctx.close()


@pytest.mark.gpu
def test_hip_reproduces_golden_match_geom_lm(vo):
    g = _load("match_220x190.npz")
    assert np.array_equal(vo.feature_matching(g["q"], g["t"], 1.0, gate=False), g["raw"])
    assert np.array_equal(vo.feature_matching(g["q"], g["t"], 1.0), g["gated"])
    assert np.array_equal(vo.feature_matching(g["q"], g["t"], 3.0), g["gated_gap3"])
    g = _load("triangulate_64.npz")
    x, v, r = vo.triangulate(g["uvL"], g["uvR"], g["T"])
    ok = g["valid"].astype(bool)
    assert np.array_equal(v, g["valid"]) and np.array_equal(r, g["rel"]) and np.allclose(x[ok], g["xyz"][ok], rtol=1e-4, atol=1e-5)
    g = _load("pnp_120.npz")
    T, inl, n, st = vo.motion_estimation(g["xyz"], g["uv"], g["T0"], iters=10)
    assert np.allclose(T, g["T"], rtol=1e-4, atol=1e-7) and n == int(g["n_inliers"]) and np.array_equal(inl, g["inlier"])
    assert np.isclose(st["chi2_init"], float(g["chi2_init"]), rtol=1e-9)
    g = _load("ba_10x200.npz")
    r = vo.optimize_map(g["T0"], g["xyz"], g["kf_idx"], g["lm_idx"], g["uv"], True, True, 10)
    assert np.allclose(r["T"], g["T_ba"], rtol=1e-4, atol=1e-6) and np.allclose(r["xyz"], g["xyz_ba"], rtol=1e-4, atol=1e-4)
    assert r["threshold"] == float(g["thr_ba"]) and np.array_equal(r["lm_inlier"], g["inlier_ba"])
    r = vo.optimize_pose_only(g["T0"], g["xyz"], g["kf_idx"], g["lm_idx"], g["uv"], True, 10)
    assert np.allclose(r["T"], g["T_po"], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_hip_reproduces_golden_sgbm(vo):
    g = _load("sgbm_360x120.npz")
    f, d16, raw = vo.disparity_map(g["left"], g["right"], return_i16=True)
    assert np.array_equal(raw, g["raw16"]) and np.array_equal(d16, g["disp16"]) and np.array_equal(f, g["disparity"])
