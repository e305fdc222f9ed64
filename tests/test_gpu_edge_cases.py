"""GPU edge cases of the hot path through the C-ABI: empty inputs flowing through the batched pipeline, capacity overflow reported
as an error instead of silently truncated results (the reference has undefined behaviour on most of these, quirk Q7)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


def test_flat_image_yields_no_keypoints(vo, oracle):
    img = np.full((376, 1241), 128, np.uint8)
    k, d = vo.feature_detection(img)
    wk, wd = oracle.feature_detection(img)
    assert len(k) == 0 and len(wk) == 0 and d.shape[0] == 0
    assert len(vo.orb_detect(img)) == 0


def test_pipeline_with_an_empty_item(oracle, synth):
    """batch of 3 stereo keyframes whose middle item is a flat image: zero keypoints, zero matches, zero pose points for the two
    frame-to-frame pairs that touch it, and the poses of those pairs stay at the initial guess (finite); the items around it are
    unaffected (same outputs as the oracle)"""
    import torch
    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    B, anms = 3, 500
    pipe = KeyframePipeline(B, anms_num=anms, unique_frames=B, seed=9, with_ba=False)
    try:
        pipe.h_imgs[1] = 90; pipe.h_imgs[B + 1] = 90
        pipe.d_imgs.copy_(torch.from_numpy(pipe.h_imgs))
        pipe.step()
        out = pipe.download()
        assert (pipe.vo.orb_status(2 * B) == 0).all()
        assert out["cnt"][1] == 0 and out["cnt"][B + 1] == 0 and out["nlr"][1] == 0
        assert out["nf2f"][0] == 0 and out["nf2f"][1] == 0 and out["pn"][0] == 0 and out["pn"][1] == 0
        assert np.isfinite(out["Tpnp"][:2]).all() and np.allclose(out["Tpnp"][:2], IDENT)
        assert out["ninl"][0] == 0 and out["ninl"][1] == 0
        w = pipe.w
        for b in (0, 2):
            kL, dL = oracle.feature_detection(pipe.h_imgs[b][:, :w], 3000, anms)
            kR, dR = oracle.feature_detection(pipe.h_imgs[B + b][:, :w], 3000, anms)
            m = oracle.feature_matching(dL, dR, 1.0)
            assert out["cnt"][b] == len(kL) > 0 and out["nlr"][b] == len(m) > 0
            assert (out["desc"][b][:len(kL)] == dL).all() and (out["lr"][b][:len(m)]["trainIdx"] == m["trainIdx"]).all()
    finally:
        pipe.close()


def test_corner_capacity_overflow_is_an_error(pkg):
    """salt-and-pepper noise produces more FAST corners than the per-level lists hold (capacity = area / 16): the host tier returns
    VSLAM_ERR_CAPACITY, the device tier raises the per-image status flag -- never a silently truncated keypoint set"""
    rng = np.random.default_rng(0)
    img = np.where(rng.random((376, 1241)) < 0.5, 0, 255).astype(np.uint8)
    ctx = pkg.VO(device=0, max_batch=1)
    try:
        with pytest.raises(pkg.VslamError) as e:
            ctx.feature_detection(img)
        assert "capacity" in str(e.value).lower()
    finally:
        ctx.close()


def test_window_with_unobserved_landmarks(vo, oracle, synth):
    """landmarks without observations (the flag-only rows ba_host.cpp appends for quirk Q1) are carried through optimize_map untouched"""
    w = synth.ba_window(n_kf=6, n_lm=120, seed=17)
    xyz = np.concatenate([w["xyz"], np.array([[1, 2, 30], [-3, 1, 25]], np.float32)])   # two landmarks nobody observes
    r = vo.optimize_map(w["T0"], xyz, w["kf_idx"], w["lm_idx"], w["uv"], True, True, 8)
    T, x, chi2, st = oracle.local_ba(w["T0"], xyz, w["kf_idx"], w["lm_idx"], w["uv"], iters=8, update_poses=True, update_lms=True)
    assert np.allclose(r["T"], T, rtol=1e-4, atol=1e-6)
    assert np.array_equal(r["xyz"][-2:], xyz[-2:]) and np.array_equal(x[-2:], xyz[-2:])
    assert np.allclose(r["xyz"], x, rtol=1e-4, atol=1e-4)
    assert r["lm_inlier"][-2:].tolist() == [1, 1]   # no edge writes their flag


@pytest.mark.parametrize("shape", [(376, 1241), (257, 333)])
def test_saturated_background_and_odd_sizes(pkg, oracle, shape):
    """white (255) background with dark rectangles: the 8-bit Gaussian blur of cv::GaussianBlur saturates there (its taps add up to 257
    per pass, so sum >> 16 reaches 257 before saturate_cast) and the rBRIEF tests read saturated pixels; (257, 333) is a size
    whose pyramid levels end in partial tiles / partial 16-byte chunks in every tiled kernel"""
    h, w = shape
    rng = np.random.default_rng(5)
    img = np.full((h, w), 255, np.uint8)
    for _ in range(160 if w > 1000 else 40):
        x0, y0 = int(rng.integers(0, w - 12)), int(rng.integers(0, h - 12))
        img[y0:y0 + int(rng.integers(6, 40)), x0:x0 + int(rng.integers(6, 60))] = int(rng.integers(0, 60))
    ctx = pkg.VO(device=0, max_batch=1, img_w=w, img_h=h, orb_nfeatures=3000, anms_num=500)
    try:
        k, d = ctx.feature_detection(img)
        wk, wd = oracle.feature_detection(img, 3000, 500)
        assert len(wk) > 20
        assert len(k) == len(wk) and all(np.array_equal(k[f], wk[f]) for f in ("x", "y", "angle", "response", "octave"))
        assert np.array_equal(d, wd)
    finally:
        ctx.close()
