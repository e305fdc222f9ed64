"""The fused kernels of round 3 (orb_pyrblur_kernel, sgbm_down_kernel, sgbm_forward_kernel) are chosen by batch size (large batches only:
below ~300 images / 8-16 stereo pairs the separate kernels are faster).  The parity suite runs small batches, so every check here is
executed with the thresholds forced down to 1 (fused kernels; the SGBM forward sweep once with 64-row and once with 32-row slabs) and
forced up (separate kernels) through VSLAM_ORB_FUSE_MIN / VSLAM_SGBM_FUSE_MIN / VSLAM_SGBM_FWD_MIN / VSLAM_SGBM_FW_ROWS, which seed a
context's overrides when it is CREATED (every test below creates its context after the fixture ran; vslam_set_tuning changes them later)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[("1", "64"), ("1", "32"), ("1000000", "64")], ids=["fused-slab64", "fused-slab32", "separate"])
def fuse_mode(request, monkeypatch):
    thr, rows = request.param
    monkeypatch.setenv("VSLAM_ORB_FUSE_MIN", thr)
    monkeypatch.setenv("VSLAM_SGBM_FUSE_MIN", thr)
    monkeypatch.setenv("VSLAM_SGBM_FWD_MIN", thr)
    monkeypatch.setenv("VSLAM_SGBM_FW_ROWS", rows)
    return thr


def _kps_equal(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(a[f], b[f]), f


def test_feature_detection_both_paths(fuse_mode, pkg, oracle, synth):
    seq = synth.stereo_sequence(3, seed=0)
    for img in (synth.noise_image(3), seq[1][0], seq[2][1]):
        ctx = pkg.VO(device=0, max_batch=1, anms_num=1500)
        try:
            gk, gd = ctx.feature_detection(img)
            wk, wd = oracle.feature_detection(img, 3000, 1500)
            _kps_equal(gk, wk)
            assert np.array_equal(gd, wd)
        finally:
            ctx.close()


@pytest.mark.parametrize("shape", [(257, 333), (200, 1324), (376, 1034)])
def test_odd_sizes_both_paths(fuse_mode, pkg, oracle, synth, shape):
    h, w = shape
    img = synth.noise_image(17, w, h)
    ctx = pkg.VO(device=0, max_batch=1, img_w=w, img_h=h, orb_nfeatures=1000, anms_num=300)
    try:
        gk, gd = ctx.feature_detection(img)
        wk, wd = oracle.feature_detection(img, 1000, 300)
        _kps_equal(gk, wk)
        assert np.array_equal(gd, wd)
        # user keypoints near the level borders (the descriptor's slow path reads the blurred level right up to its edge)
        kps = oracle.orb_detect(img, 1000)[:48].copy()
        kps["x"] = np.linspace(31, w - 32, len(kps)).astype(np.float32)
        kps["y"] = np.where(np.arange(len(kps)) % 2 == 0, 31.0, h - 32.0).astype(np.float32)
        kps["octave"] = np.arange(len(kps)) % 8
        kps["angle"] = np.linspace(0, 359, len(kps)).astype(np.float32)
        gk, gd = ctx.orb_compute(img, kps)
        wk, wd = oracle.orb_compute(img, kps)
        _kps_equal(gk, wk)
        assert np.array_equal(gd, wd)
    finally:
        ctx.close()


def test_sgbm_both_paths(fuse_mode, vo, oracle, synth):
    rng = np.random.default_rng(5)
    for (w, h, sh, noise) in ((640, 200, 11, 0), (333, 97, 5, 0), (1241, 120, 30, 0), (640, 200, 11, 15), (140, 186, 4, 15)):
        L = synth.noise_image(w % 97, w + 40, h)
        Lc = np.ascontiguousarray(L[:, :w]); R = np.ascontiguousarray(L[:, sh:sh + w])
        if noise:  # a right view that is not an exact shift: no zero-cost disparity, every path carries real values across the slab boundaries
            R = np.clip(R.astype(int) + rng.integers(-noise, noise + 1, R.shape), 0, 255).astype(np.uint8)
        gf, gi, graw = vo.disparity_map(Lc, R, return_i16=True)
        wi, wraw = oracle.sgbm_compute(Lc, R, return_raw=True)
        assert np.array_equal(graw, wraw) and np.array_equal(gi, wi)


def test_sgbm_full_size_both_paths(fuse_mode, vo, oracle, synth):
    """BASELINE image size: six 64-row (twelve 32-row) slabs chained through the boundary buffer, the last one partly empty"""
    left, right = synth.stereo_sequence(1, seed=4)[0][:2]
    gf, gi, graw = vo.disparity_map(left, right, return_i16=True)
    wi, wraw = oracle.sgbm_compute(left, right, return_raw=True)
    assert np.array_equal(graw, wraw) and np.array_equal(gi, wi)
    assert (gi >= 0).mean() > 0.5


def test_sgbm_batched_both_paths(fuse_mode, pkg, oracle, synth):
    import torch
    w, h, pitch, B = 400, 120, 448, 3
    buf = np.zeros((2, B, h, pitch), np.uint8)
    pairs = []
    for b in range(B):
        L = synth.noise_image(30 + b, w + 40, h)
        Lc = np.ascontiguousarray(L[:, :w]); R = np.ascontiguousarray(L[:, 7 + b:7 + b + w])
        buf[0, b, :, :w] = Lc; buf[1, b, :, :w] = R
        pairs.append((Lc, R))
    ctx = pkg.VO(device=0, max_batch=B)
    try:
        d = torch.from_numpy(buf).cuda()
        out = torch.empty((B, h, w), dtype=torch.float32, device="cuda")
        ctx.disparity_map_dev(d[0].data_ptr(), d[1].data_ptr(), h * pitch, pitch, w, h, B, out.data_ptr())
        ctx.sync()
        got = out.cpu().numpy()
        for b, (Lc, R) in enumerate(pairs):
            assert np.array_equal(got[b], oracle.disparity_map(Lc, R)), b
    finally:
        ctx.close()


@pytest.mark.parametrize("shape", [(376, 1241), (257, 333), (200, 1324), (64, 64), (97, 300)])
def test_pyramid_and_blur_levels_pixel_exact(fuse_mode, pkg, oracle, synth, shape):
    """Every pixel of every pyramid level (cv::resize INTER_LINEAR 8U chain) and of every blurred level (GaussianBlur 7x7, 8-bit fixed point,
    reflect-101) against the oracle -- vslam_orb_level reads back what the ORB kernels sample.  Covers the tile loader's border handling
    (chunks shifted into place at the left edge / at the end of the pitch, reflect-101 columns filled from the staged tile; image sizes that
    end anywhere inside a 16-byte chunk, tiles narrower than the 256-column workgroup tile) and, on the white image, the saturation of the
    blur (taps add up to 257 per pass)."""
    h, w = shape
    imgs = [synth.noise_image(23, w, h), np.full((h, w), 255, np.uint8), np.zeros((h, w), np.uint8)]
    g = np.add.outer(np.arange(h) * 3, np.arange(w) * 5) % 256
    imgs.append(g.astype(np.uint8))
    ctx = pkg.VO(device=0, max_batch=1, img_w=w, img_h=h, orb_nfeatures=500, anms_num=0)
    try:
        for img in imgs:
            ctx.feature_detection(img)
            lv = oracle.build_pyramid(img, 8, 500)
            for l in range(8):
                if l > 0:
                    got = ctx.orb_level(0, l, False)
                    assert got.shape == lv[l].shape and np.array_equal(got, lv[l]), ("pyramid", l, int((got != lv[l]).sum()))
                got = ctx.orb_level(0, l, True)
                want = oracle.gaussian_blur7(lv[l])
                assert got.shape == want.shape, (got.shape, want.shape)
                bad = np.argwhere(got != want)
                assert len(bad) == 0, ("blur", l, len(bad), bad[:5].tolist(), got[tuple(bad[0])], want[tuple(bad[0])])
    finally:
        ctx.close()
