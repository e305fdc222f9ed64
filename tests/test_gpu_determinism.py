"""Run-twice determinism of the HIP path: "fixed summation order, no floating-point atomics"  as a test.
The same device-resident step (ORB on 2 x 4 images, both matchers, DLT, motion-only LM, the BA schedule on 8 windows) is executed
twice from identical inputs; EVERY output must be identical to the bit, including the f64 poses."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b, name):
    assert a.dtype == b.dtype and a.shape == b.shape, name
    assert np.array_equal(a.view(np.uint8) if a.dtype.kind == "f" else a, b.view(np.uint8) if b.dtype.kind == "f" else b), name


def test_step_twice_bit_identical(synth):
    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    pipe = KeyframePipeline(8, device=0, anms_num=1500, n_lm=3000, unique_frames=4, seed=3)
    try:
        pipe.step(); a = pipe.download()
        chi_a = None
        pipe.step(); b = pipe.download()
        assert (pipe.vo.orb_status(16) == 0).all() and (pipe.vo.ba_status(8) == 0).all()
        cnt = a["cnt"]
        _same(a["cnt"], b["cnt"], "keypoint counts")
        for i in range(16):   # outputs beyond the count are scratch
            _same(a["kps"][i][:cnt[i]], b["kps"][i][:cnt[i]], "keypoints %d" % i)
            _same(a["desc"][i][:cnt[i]], b["desc"][i][:cnt[i]], "descriptors %d" % i)
        for k in ("nlr", "nf2f", "pn", "ninl", "Tpnp", "ba_T", "ba_inl"):
            _same(a[k], b[k], k)
        for i in range(8):
            _same(a["lr"][i][:a["nlr"][i]], b["lr"][i][:a["nlr"][i]], "L/R matches %d" % i)
            _same(a["xyz"][i][:a["nlr"][i]], b["xyz"][i][:a["nlr"][i]], "triangulated points %d" % i)
        for i in range(7):
            _same(a["f2f"][i][:a["nf2f"][i]], b["f2f"][i][:a["nf2f"][i]], "frame-to-frame matches %d" % i)
            _same(a["inl"][i][:a["pn"][i]], b["inl"][i][:a["pn"][i]], "PnP inlier flags %d" % i)
        assert a["cnt"].min() > 500 and a["nlr"].min() > 50
    finally:
        pipe.close()


def test_ba_window_twice_bit_identical(vo, synth):
    """host-buffer tier: optimize_map on one window twice (different arena state in between)"""
    w = synth.ba_window(n_kf=10, n_lm=1200, seed=77)
    r1 = vo.optimize_map(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], True, True, 10)
    vo.optimize_pose_only(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], True, 3)
    r2 = vo.optimize_map(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], True, True, 10)
    for k in ("T", "xyz", "chi2", "lm_inlier"):
        assert np.array_equal(r1[k], r2[k]), k
    assert r1["stats"] == r2["stats"]


def test_sgbm_twice_bit_identical(vo, synth):
    L = synth.noise_image(5, 640, 200); R = np.roll(L, -11, axis=1)
    a = vo.disparity_map(L, R); b = vo.disparity_map(L, R)
    assert np.array_equal(a, b)


def test_sgbm_forward_sweep_twice_and_vs_path_kernels(pkg, synth, monkeypatch):
    """the forward wavefront sweep hands rows from workgroup to workgroup through memory with flags: run a 20-pair batch three times with it
    (workgroups meet in a different order every time) and once with the per-path kernels -- all four results must be identical"""
    import torch
    w, h, pitch, B = 420, 150, 448, 20
    rng = np.random.default_rng(11)
    buf = np.zeros((2, B, h, pitch), np.uint8)
    for b in range(B):
        L = synth.noise_image(60 + b, w + 40, h)
        buf[0, b, :, :w] = L[:, :w]
        buf[1, b, :, :w] = np.clip(L[:, 5 + b % 9:5 + b % 9 + w].astype(int) + rng.integers(-12, 13, (h, w)), 0, 255).astype(np.uint8)
    ctx = pkg.VO(device=0, max_batch=B)
    try:
        d = torch.from_numpy(buf).cuda()
        def run():
            out = torch.empty((B, h, w), dtype=torch.float32, device="cuda")
            ctx.disparity_map_dev(d[0].data_ptr(), d[1].data_ptr(), h * pitch, pitch, w, h, B, out.data_ptr())
            ctx.sync()
            return out.cpu().numpy()
        runs = [run() for _ in range(3)]
        ctx.set_tuning(sgbm_fwd_min=1000000)
        ref = run()
        for r in runs:
            assert np.array_equal(r, ref)
        assert (ref >= 0).mean() > 0.3
    finally:
        ctx.close()


def test_sgbm_forward_sweep_under_contention(pkg, synth, monkeypatch):
    """the slabs of the forward sweep take their logical index from an atomic ticket (not blockIdx), so "the lowest unfinished
    slab is resident" holds whatever else shares the device.  Run the sweep while a second stream keeps every CU busy with long filler
    kernels (workgroups then start late and out of order): the maps must equal the per-path kernels' and the status word must stay 0."""
    import torch
    w, h, pitch, B = 420, 150, 448, 24
    rng = np.random.default_rng(12)
    buf = np.zeros((2, B, h, pitch), np.uint8)
    for b in range(B):
        L = synth.noise_image(160 + b, w + 40, h)
        buf[0, b, :, :w] = L[:, :w]
        buf[1, b, :, :w] = np.clip(L[:, 5 + b % 9:5 + b % 9 + w].astype(int) + rng.integers(-12, 13, (h, w)), 0, 255).astype(np.uint8)
    ctx = pkg.VO(device=0, max_batch=B)
    try:
        d = torch.from_numpy(buf).cuda()
        def run():
            out = torch.empty((B, h, w), dtype=torch.float32, device="cuda")
            ctx.disparity_map_dev(d[0].data_ptr(), d[1].data_ptr(), h * pitch, pitch, w, h, B, out.data_ptr())
            assert ctx.sgbm_status() == 0
            return out.cpu().numpy()
        ctx.set_tuning(sgbm_fwd_min=1000000)
        ref = run()
        ctx.set_tuning(sgbm_fwd_min=1)
        for rows in (32, 64):
            ctx.set_tuning(sgbm_fw_rows=rows)
            filler = torch.cuda.Stream()
            a = torch.randn(4096, 4096, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(filler):       # ~100+ ms of matrix products occupying all CUs, queued before the sweep starts
                for _ in range(40):
                    a = torch.tanh(a @ a * 1e-3)
            got = run()
            filler.synchronize()
            assert np.array_equal(got, ref), rows
        assert (ref >= 0).mean() > 0.3
    finally:
        ctx.close()
