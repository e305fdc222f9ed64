"""GPU parity: dense stereo disparity (SGBM) vs the CPU oracle -- bit-exact (integer path).
Reference path: VO::disparity_map visual_odometry.cpp:159-174 (cv::StereoSGBM 3.2 MODE_SGBM + medianBlur + filterSpeckles
+ convertTo(CV_32F, 1/16))."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rendered(synth, seed, w, h, noise=0):
    sc = synth.Scene(seed)
    T = synth.trajectory(1, seed)[0]
    L, depth = sc.render(T, w, h)
    R, _ = sc.render(T, w, h, x_offset=synth.BASELINE)
    if noise:
        rng = np.random.default_rng(seed)
        R = np.clip(R.astype(int) + rng.integers(-noise, noise + 1, R.shape), 0, 255).astype(np.uint8)
    return L, R, depth


def _check(vo, oracle, L, R):
    gf, gi, graw = vo.disparity_map(L, R, return_i16=True)
    wi, wraw = oracle.sgbm_compute(L, R, return_raw=True)
    assert np.array_equal(graw, wraw), f"raw SGBM differs at {(graw != wraw).sum()} px"
    assert np.array_equal(gi, wi), f"filtered map differs at {(gi != wi).sum()} px"
    assert np.array_equal(gf, oracle.disparity_map(L, R))
    return gf


@pytest.mark.parametrize("w,h,noise", [(200, 40, 0), (333, 77, 10), (640, 200, 0)])
def test_sgbm_small_sizes(vo, oracle, synth, w, h, noise):
    L, R, _ = _rendered(synth, 7 + w, w, h, noise)
    _check(vo, oracle, L, R)


def test_sgbm_kitti_size_rendered(vo, oracle, synth):
    L, R, depth = _rendered(synth, 1, 1241, 376)
    f = _check(vo, oracle, L, R)
    gt = synth.FX * synth.BASELINE / depth
    m = f >= 0
    assert m[:, 96:].mean() > 0.8 and np.median(np.abs(f[m] - gt[m])) < 0.75


def test_sgbm_noise_pair_with_speckles(vo, oracle, synth):
    # blocky noise, unrelated left/right halves mixed with a shifted copy: heavy uniqueness / LR-check / speckle traffic
    base = synth.noise_image(3, 700 + 31, 120)
    L = np.ascontiguousarray(base[:, :700]); R = np.ascontiguousarray(base[:, 31:]).copy()
    R[40:80] = synth.noise_image(4, 700, 40)
    f = _check(vo, oracle, L, R)
    assert (f[:, 96:] < 0).mean() > 0.05 and (f[:, 96:] >= 0).mean() > 0.3


def test_sgbm_flat_and_saturated(vo, oracle):
    L = np.full((30, 150), 200, np.uint8)
    _check(vo, oracle, L, L)
    rng = np.random.default_rng(0)
    L = (rng.integers(0, 2, (50, 260)) * 255).astype(np.uint8); R = (rng.integers(0, 2, (50, 260)) * 255).astype(np.uint8)
    _check(vo, oracle, L, R)   # maximum pixel costs: exercises the sat16 sums


def test_sgbm_batched_device_path(vo, oracle, synth):
    import torch
    w, h, B = 500, 90, 5   # B >= 4 takes the fused winner-take-all path
    pairs = [_rendered(synth, 20 + b, w, h, noise=6 * b)[:2] for b in range(B)]
    pitch = 512
    buf = np.zeros((2, B, h, pitch), np.uint8)
    for b, (L, R) in enumerate(pairs):
        buf[0, b, :, :w] = L; buf[1, b, :, :w] = R
    d = torch.from_numpy(buf).cuda()
    out = torch.empty((B, h, w), dtype=torch.float32, device="cuda")
    i16 = torch.empty((B, h, w), dtype=torch.int16, device="cuda")
    vo.disparity_map_dev(d[0].data_ptr(), d[1].data_ptr(), h * pitch, pitch, w, h, B, out.data_ptr(), i16.data_ptr())
    vo.sync()
    for b, (L, R) in enumerate(pairs):
        assert np.array_equal(i16[b].cpu().numpy(), oracle.sgbm_compute(L, R))
        assert np.array_equal(out[b].cpu().numpy(), oracle.disparity_map(L, R))


@pytest.mark.parametrize("B", [17, 35])
def test_sgbm_large_batch_default_kernels(pkg, oracle, synth, B):
    """a batch large enough that the library picks the fused kernels ITSELF (no environment override): the top-down sweep and the forward
    wavefront sweep with 32-row slabs (17 pairs: 5 chained slabs per pair) and 64-row slabs (35 pairs: 3 slabs), noisy pairs"""
    import torch
    w, h, pitch = 300, 130, 320
    pairs = [_rendered(synth, 40 + (b % 6), w, h, noise=3 + 2 * (b % 5))[:2] for b in range(B)]
    pairs = [(L, np.roll(R, b % 3, axis=1)) for b, (L, R) in enumerate(pairs)]   # every pair different
    buf = np.zeros((2, B, h, pitch), np.uint8)
    for b, (L, R) in enumerate(pairs):
        buf[0, b, :, :w] = L; buf[1, b, :, :w] = R
    ctx = pkg.VO(device=0, max_batch=B)
    try:
        d = torch.from_numpy(buf).cuda()
        out = torch.empty((B, h, w), dtype=torch.float32, device="cuda")
        i16 = torch.empty((B, h, w), dtype=torch.int16, device="cuda")
        ctx.disparity_map_dev(d[0].data_ptr(), d[1].data_ptr(), h * pitch, pitch, w, h, B, out.data_ptr(), i16.data_ptr())
        ctx.sync()
        got = i16.cpu().numpy()
        for b, (L, R) in enumerate(pairs):
            assert np.array_equal(got[b], oracle.sgbm_compute(np.ascontiguousarray(L), np.ascontiguousarray(R))), b
    finally:
        ctx.close()


def test_sgbm_rejects_bad_arguments(vo, pkg):
    with pytest.raises(pkg.VslamError):
        vo.disparity_map(np.zeros((40, 90), np.uint8), np.zeros((40, 90), np.uint8))   # w <= 96 disparities
    with pytest.raises(pkg.VslamError):
        vo.disparity_map(np.zeros((40, 100), np.uint8), np.zeros((40, 100), np.uint8))  # width1 <= SW2: undefined in OpenCV
    with pytest.raises(pkg.VslamError):
        vo.disparity_map(np.zeros((8, 200), np.uint8), np.zeros((8, 200), np.uint8))   # h <= block size
