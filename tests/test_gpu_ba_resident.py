"""GPU parity of the LDS-resident optimize_map kernel (csrc/ba_resident.hip) -- forced on with vslam_set_tuning("ba_resident", 1), so that the
dense synthetic windows the library would hand to lm_window_kernel by default go through it too -- against the CPU oracle, against the
general kernel on the same inputs, and against itself (bit reproducibility: its sums are integer atomics on a fixed-point scale).
Reference: optimize_map optimization.cpp:103-288, the per-keyframe schedule run_vslam.cpp:58-71."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_lm import RTOL, _stats_close

pytestmark = pytest.mark.gpu


@pytest.fixture()
def vo_res(pkg):
    ctx = pkg.VO(device=0, max_batch=2)
    ctx.set_tuning(ba_resident=1)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("n_kf,n_lm,seed,iters,min_obs,max_obs", [(10, 300, 2, 5, 2, 5), (10, 3000, 7, 10, 2, 5), (10, 1000, 9, 10, 1, 3), (12, 800, 21, 8, 2, 5),
                                                                   (3, 120, 21, 8, 2, 3), (1, 60, 21, 8, 1, 1), (10, 2500, 5, 10, 1, 10)])
def test_resident_local_ba_parity(vo_res, oracle, synth, n_kf, n_lm, seed, iters, min_obs, max_obs):
    """one optimize_map call (host tier): LM trajectory, poses, landmarks, per-edge chi2, threshold and flags vs the oracle; shuffled edge order;
    windows with single-observation landmarks (the Woodbury path), 12 keyframes, one keyframe, tracks through the whole window"""
    w = synth.ba_window(n_kf=n_kf, n_lm=n_lm, seed=seed, min_obs=min_obs, max_obs=min(max_obs, n_kf))
    perm = np.random.default_rng(seed).permutation(len(w["kf_idx"]))
    kf, lm, uv = w["kf_idx"][perm], w["lm_idx"][perm], w["uv"][perm]
    r = vo_res.optimize_map(w["T0"], w["xyz"], kf, lm, uv, True, True, iters)
    T, xyz, chi2, st = oracle.local_ba(w["T0"], w["xyz"], kf, lm, uv, iters=iters, update_poses=True, update_lms=True)
    _stats_close(r["stats"], st)
    assert np.allclose(r["T"], T, rtol=RTOL, atol=1e-6)
    assert np.allclose(r["xyz"], xyz, rtol=RTOL, atol=1e-4)
    assert np.allclose(r["chi2"], chi2, rtol=1e-4, atol=1e-6)
    th, inl, ni, no = oracle.chi2_classify(chi2, lm, np.ones(n_lm, np.uint8))
    assert r["threshold"] == th and (r["lm_inlier"] == inl).all()


def test_resident_no_writeback_and_bad_graph(vo_res, pkg, synth):
    w = synth.ba_window(n_kf=10, n_lm=200, seed=4)
    r = vo_res.optimize_map(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], False, False, 5)
    assert (r["T"] == w["T0"]).all() and (r["xyz"] == w["xyz"]).all()
    bad = w["kf_idx"].copy(); bad[0] = 99
    with pytest.raises(pkg.VslamError):
        vo_res.optimize_map(w["T0"], w["xyz"], bad, w["lm_idx"], w["uv"])
    dup_kf = np.concatenate([w["kf_idx"], w["kf_idx"][:1]]); dup_lm = np.concatenate([w["lm_idx"], w["lm_idx"][:1]])
    dup_uv = np.concatenate([w["uv"], w["uv"][:1]])
    with pytest.raises(pkg.VslamError):
        vo_res.optimize_map(w["T0"], w["xyz"], dup_kf, dup_lm, dup_uv)


def test_resident_twice_bit_identical_and_close_to_general_kernel(pkg, synth):
    """the same call twice gives the same bits (rows are handed to waves dynamically: the sums are integer); the general kernel on the same
    window agrees to rounding (same algorithm, different summation order)"""
    w = synth.ba_window(n_kf=10, n_lm=2000, seed=13, min_obs=1, max_obs=6)
    outs = []
    for resident in (1, 1, 0):
        ctx = pkg.VO(device=0, max_batch=1)
        try:
            ctx.set_tuning(ba_resident=resident)
            outs.append(ctx.optimize_map(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], True, True, 10))
        finally:
            ctx.close()
    a, b, g = outs
    for k in ("T", "xyz", "chi2", "lm_inlier"):
        assert np.array_equal(a[k], b[k]), k
    assert a["stats"]["chi2_iter"] == b["stats"]["chi2_iter"] and a["stats"]["lambda_iter"] == b["stats"]["lambda_iter"]
    assert np.allclose(a["T"], g["T"], rtol=1e-7, atol=1e-9) and np.allclose(a["xyz"], g["xyz"], rtol=1e-5, atol=1e-5)
    assert np.allclose(a["stats"]["chi2_iter"], g["stats"]["chi2_iter"], rtol=1e-8)


def _schedule(pkg, pipe, B, resident, adaptive):
    import torch
    pipe.vo.set_tuning(ba_resident=resident, ba_adaptive=adaptive)
    pipe.vo.ba_batch_dev(pipe.ba_batch, schedule=1)
    passes = pipe.vo.ba_schedule_passes(B)
    assert (pipe.vo.ba_status(B) == 0).all()
    torch.cuda.synchronize()
    return passes


@pytest.mark.parametrize("kind", ["synthetic", "outliers", "tracks"])
def test_resident_schedule_vs_general_kernel_and_plain_vs_adaptive(pkg, synth, kind):
    """the schedule of run_vslam.cpp:61-70 on a batch: (a) adaptive == plain, bit for bit, on the resident kernel; (b) resident == general kernel
    to rounding: poses 1e-7, the same passes executed, the same landmark flags except where a chi2 sits within 1e-9 of the threshold"""
    import torch
    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    B = 16
    if kind == "tracks":
        pipe = KeyframePipeline(B, anms_num=1500, unique_frames=12, seed=41, ba_windows="tracks")
        pipe.stage_orb(); pipe.stage_stereo_match(); pipe.stage_track(); pipe.stage_build_windows()
    else:
        pipe = KeyframePipeline(B, anms_num=500, n_lm=900, unique_frames=2, seed=42, ba_windows="synthetic")
        if kind == "outliers":
            uv = pipe.ba_uv.cpu().numpy().copy()
            rng = np.random.default_rng(6)
            bad = rng.random(len(uv)) < 0.02
            uv[bad] += rng.normal(0, 25, (int(bad.sum()), 2)).astype(np.float32)
            pipe.ba_uv.copy_(torch.from_numpy(uv))
    try:
        pipe.vo.sync(); torch.cuda.synchronize()
        T0, inl0 = pipe.ba_T.clone(), pipe.ba_inl.clone()
        chi2 = torch.zeros(int(pipe.ba_batch.total_edge), dtype=torch.float64, device=pipe.dev)
        pipe.ba_batch.d_chi2 = chi2.data_ptr()
        got = {}
        for key in ((1, 1), (1, 0), (0, 1)):
            pipe.ba_T.copy_(T0); pipe.ba_inl.copy_(inl0); chi2.zero_()
            torch.cuda.synchronize()
            passes = _schedule(pkg, pipe, B, *key)
            got[key] = (pipe.ba_T.cpu().numpy().copy(), pipe.ba_inl.cpu().numpy().copy(), chi2.cpu().numpy().copy(), passes)
        ra, rp, ga = got[(1, 1)], got[(1, 0)], got[(0, 1)]
        assert (rp[3] == 3).all()
        for x, y in zip(ra[:3], rp[:3]):
            assert np.array_equal(x, y)                       # (a)
        assert np.allclose(ra[0], ga[0], rtol=1e-6, atol=1e-8)  # (b) poses after the whole schedule (optimize_map x3 + pose-only)
        assert np.array_equal(ra[3], ga[3]) or (ra[3] != ga[3]).mean() < 0.1
        assert (ra[1] != ga[1]).mean() < 2e-3
    finally:
        pipe.vo.set_tuning(ba_adaptive=-1, ba_resident=-1)
        pipe.close()


def test_resident_defers_windows_that_do_not_fit(pkg, synth):
    """a batch that mixes a window beyond the kernel's LDS budget (7000 landmarks) with small ones: the big one is taken by lm_window_kernel in the
    same call, every window's result equals the all-general-kernel run to rounding, every status is 0"""
    import torch
    sizes = [400, 7000, 900]
    ws = [synth.ba_window_fast(n_kf=10, n_lm=n, seed=50 + i, min_obs=1, max_obs=4) for i, n in enumerate(sizes)]
    lm_off = np.concatenate([[0], np.cumsum([len(w["xyz"]) for w in ws])]).astype(np.int32)
    e_off = np.concatenate([[0], np.cumsum([len(w["kf_idx"]) for w in ws])]).astype(np.int32)
    dev = torch.device("cuda", 0)
    res = {}
    for resident in (1, 0):
        ctx = pkg.VO(device=0, max_batch=1)
        try:
            ctx.set_tuning(ba_resident=resident)
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
            T = t(np.stack([w["T0"] for w in ws]), np.float64); xyz = t(np.concatenate([w["xyz"] for w in ws]), np.float32)
            inl = torch.ones(int(lm_off[-1]), dtype=torch.uint8, device=dev)
            kf = t(np.concatenate([w["kf_idx"] for w in ws]), np.int32); lm = t(np.concatenate([w["lm_idx"] for w in ws]), np.int32)
            uv = t(np.concatenate([w["uv"] for w in ws]), np.float32)
            d_lo, d_eo = t(lm_off, np.int32), t(e_off, np.int32)
            b = pkg.BaBatch()
            b.n_windows = len(ws); b.n_kf = 10; b.d_lm_off = d_lo.data_ptr(); b.d_edge_off = d_eo.data_ptr(); b.d_T_c_w = T.data_ptr(); b.d_xyz = xyz.data_ptr()
            b.d_reliable = None; b.d_lm_inlier = inl.data_ptr(); b.d_kf_idx = kf.data_ptr(); b.d_lm_idx = lm.data_ptr(); b.d_uv = uv.data_ptr()
            b.total_lm = int(lm_off[-1]); b.total_edge = int(e_off[-1])
            ctx.ba_batch_dev(b, schedule=1)
            ctx.sync()
            assert (ctx.ba_status(len(ws)) == 0).all()
            res[resident] = (T.cpu().numpy().copy(), inl.cpu().numpy().copy())
        finally:
            ctx.close()
    assert np.allclose(res[1][0], res[0][0], rtol=1e-6, atol=1e-8)
    assert np.array_equal(res[1][0][1], res[0][0][1])        # the deferred window ran on the same kernel both times: same bits
    assert (res[1][1] != res[0][1]).mean() < 2e-3



def test_both_widths_bit_identical(pkg, synth):
    """ba_resident_kernel has two widths -- 512 lanes and a whole CU's LDS per window (small launches: latency), 256 lanes and half of it (two windows per
    CU: throughput) -- and the launcher picks by the number of windows in the launch, so they must give the SAME BITS (a short last chunk of a sharded
    sequence must not differ from the unsharded run): every floating-point sum that crosses waves runs over eight row streams in both.  One optimize_map
    call on windows with single- and multi-observation landmarks, and the whole schedule on windows built from a rendered sequence's tracks."""
    import torch
    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    for n_lm, seed, mx in ((2000, 13, 6), (3400, 5, 2), (600, 3, 4)):
        w = synth.ba_window(n_kf=10, n_lm=n_lm, seed=seed, min_obs=1, max_obs=mx)
        outs = []
        for lanes in (256, 512):
            ctx = pkg.VO(device=0, max_batch=1)
            try:
                ctx.set_tuning(ba_resident=1, ba_lanes=lanes)
                outs.append(ctx.optimize_map(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], True, True, 10))
            finally:
                ctx.close()
        a, b = outs
        for k in ("T", "xyz", "chi2", "lm_inlier"):
            assert np.array_equal(a[k], b[k]), (n_lm, k)
        assert a["stats"]["chi2_iter"] == b["stats"]["chi2_iter"] and a["stats"]["lambda_iter"] == b["stats"]["lambda_iter"] and a["threshold"] == b["threshold"]
    B = 24
    pipe = KeyframePipeline(B, anms_num=1500, unique_frames=24, seed=11, ba_windows="tracks")
    try:
        pipe.stage_orb(); pipe.stage_stereo_match(); pipe.stage_track(); pipe.stage_build_windows()
        pipe.vo.sync(); torch.cuda.synchronize()
        T0 = pipe.ba_T.clone(); inl0 = pipe.ba_inl.clone()
        res = []
        for lanes in (256, 512, -1):
            pipe.ba_T.copy_(T0); pipe.ba_inl.copy_(inl0)
            torch.cuda.synchronize()  # (the copies run on torch's stream, the library on its own)
            pipe.vo.set_tuning(ba_lanes=lanes)
            pipe.vo.ba_batch_dev(pipe.ba_batch, schedule=1)
            passes = pipe.vo.ba_schedule_passes(B)
            assert (pipe.vo.ba_status(B) == 0).all()
            torch.cuda.synchronize()
            res.append((pipe.ba_T.cpu().numpy().view(np.uint64).copy(), pipe.ba_inl.cpu().numpy().copy(), passes.copy()))  # (bit patterns: unused pose slots of the growing-map windows may hold anything)
        for other in res[1:]:
            assert np.array_equal(res[0][0], other[0]) and np.array_equal(res[0][1], other[1]) and np.array_equal(res[0][2], other[2])
    finally:
        pipe.vo.set_tuning(ba_lanes=-1)
        pipe.close()
