"""GPU parity of the whole throughput-mode keyframe pipeline (device-resident, batched) against the oracle composite."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


# (B, ANMS, landmarks per window): a small case, and the bench configuration itself (bench.py defaults: ANMS 1500, 10 x 3000
# windows, the 5+5+10+10 schedule in one vslam_ba_batch_dev call) at a batch the oracle finishes in seconds
@pytest.mark.parametrize("B,anms,n_lm", [(3, 500, 400), (4, 1500, 3000)])
def test_pipeline_matches_oracle_composite(oracle, synth, B, anms, n_lm):
    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    pipe = KeyframePipeline(B, anms_num=anms, n_lm=n_lm, unique_frames=B, seed=3)
    try:
        pipe.step()
        out = pipe.download()
        assert (pipe.vo.orb_status(2 * B) == 0).all()
        w = pipe.w
        prev = None
        for b in range(B):
            kL, dL = oracle.feature_detection(pipe.h_imgs[b][:, :w], 3000, anms)
            kR, dR = oracle.feature_detection(pipe.h_imgs[B + b][:, :w], 3000, anms)
            assert out["cnt"][b] == len(kL) and out["cnt"][B + b] == len(kR)
            assert (out["desc"][b][:len(kL)] == dL).all() and (out["desc"][B + b][:len(kR)] == dR).all()
            m = oracle.feature_matching(dL, dR, 1.0)
            assert out["nlr"][b] == len(m)
            assert (out["lr"][b][:len(m)]["trainIdx"] == m["trainIdx"]).all() and (out["lr"][b][:len(m)]["queryIdx"] == m["queryIdx"]).all()
            uvL = np.stack([kL["x"][m["queryIdx"]], kL["y"][m["queryIdx"]]], 1)
            uvR = np.stack([kR["x"][m["trainIdx"]], kR["y"][m["trainIdx"]]], 1)
            xyz, valid, rel = oracle.triangulate_dlt(uvL, uvR, IDENT)
            assert (out["valid"][b][:len(m)] == valid).all() and (out["rel"][b][:len(m)] == rel).all()
            ok = valid.astype(bool)
            assert np.allclose(out["xyz"][b][:len(m)][ok], xyz[ok], rtol=1e-4, atol=1e-5)
            if prev is not None:
                i = b - 1
                pk, pd, pm, pxyz, pvalid = prev
                f = oracle.feature_matching(pd, dL, 1.0)
                assert out["nf2f"][i] == len(f) and (out["f2f"][i][:len(f)]["trainIdx"] == f["trainIdx"]).all()
                kp2lr = -np.ones(len(pk), np.int64); kp2lr[pm["queryIdx"]] = np.arange(len(pm))
                li = kp2lr[f["queryIdx"]]
                okm = (li >= 0) & (pvalid[np.maximum(li, 0)] != 0)
                assert out["pn"][i] == okm.sum()
                p3 = pxyz[li[okm]]; p2 = np.stack([kL["x"][f["trainIdx"][okm]], kL["y"][f["trainIdx"][okm]]], 1)
                assert np.array_equal(out["pxyz"][i][:okm.sum()], p3) and np.array_equal(out["puv"][i][:okm.sum()], p2)
                if okm.sum() >= 6:
                    T, inl, n, st = oracle.pnp_motion_only(p3, p2, IDENT, iters=10)
                    assert np.allclose(out["Tpnp"][i], T, rtol=1e-4, atol=1e-6)
                    assert out["ninl"][i] == n
            prev = (kL, dL, m, xyz, valid)
        # local-BA schedule per window (run_vslam.cpp:58-71)
        lm_off = pipe.ba_lm_off.cpu().numpy(); e_off = pipe.ba_e_off.cpu().numpy()
        kf_all = pipe.ba_kf.cpu().numpy(); lm_all = pipe.ba_lm.cpu().numpy(); uv_all = pipe.ba_uv.cpu().numpy()
        xyz_all = pipe.ba_xyz.cpu().numpy(); T0 = pipe.ba_T0.cpu().numpy()
        for b in range(B):
            kf, lm, uv = kf_all[e_off[b]:e_off[b + 1]], lm_all[e_off[b]:e_off[b + 1]], uv_all[e_off[b]:e_off[b + 1]]
            xyz = xyz_all[lm_off[b]:lm_off[b + 1]]
            T = T0[b].copy(); inl = np.ones(len(xyz), np.uint8)
            for iters, upd in ((5, False), (5, False), (10, True)):
                act = inl.astype(bool)[lm]
                T2, _, chi2, _ = oracle.local_ba(T, xyz, kf[act], lm[act], uv[act], iters=iters)
                _, inl, _, _ = oracle.chi2_classify(chi2, lm[act], inl)
                if upd:
                    T = T2
            act = inl.astype(bool)[lm]
            T2, chi2, _ = oracle.pose_only_window(T, xyz, kf[act], lm[act], uv[act], iters=10)
            _, inl, _, _ = oracle.chi2_classify(chi2, lm[act], inl)
            assert np.allclose(out["ba_T"][b], T2, rtol=1e-4, atol=1e-6), np.abs(out["ba_T"][b] - T2).max()
            got = out["ba_inl"][lm_off[b]:lm_off[b + 1]]
            assert np.array_equal(got, inl), ("landmark inlier flags differ", np.nonzero(got != inl)[0][:10], int((got != inl).sum()))
        assert (pipe.vo.ba_status(B) == 0).all()
    finally:
        pipe.close()


@pytest.mark.parametrize("pose", ["lm", "ransac"])
def test_pipeline_sgbm_depth_matches_oracle_composite(oracle, synth, pose):
    """throughput pipeline with the reference's own depth path: batched SGBM + Frame::find_3d on the left keypoints, then the
    frame-to-frame stage on those landmarks (visual_odometry.cpp:159-217, :253-314); pose = "ransac": with the reference's own pose
    stage too, cv::solvePnPRansac(..., 100, 4.0, 0.99) batched on the device (vslam_pnp_ransac_dev, :277) -- the REFERENCE pipeline"""
    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    B, anms = 3, 500
    pipe = KeyframePipeline(B, anms_num=anms, unique_frames=B, seed=4, with_ba=False, depth="sgbm", pose=pose)
    try:
        pipe.step()
        out = pipe.download()
        w = pipe.w
        disp = pipe.d_disp.cpu().numpy()
        prev = None
        for b in range(B):
            L = pipe.h_imgs[b][:, :w]; R = pipe.h_imgs[B + b][:, :w]
            kL, dL = oracle.feature_detection(L, 3000, anms)
            assert out["cnt"][b] == len(kL) and (out["desc"][b][:len(kL)] == dL).all()
            wd = oracle.disparity_map(L, R)
            assert np.array_equal(disp[b], wd)
            xyz, valid, rel = oracle.find_3d_disparity(kL, wd, IDENT)
            n = len(kL)
            assert out["nlr"][b] == n
            assert (out["valid"][b][:n] == valid).all() and (out["rel"][b][:n] == rel).all()
            ok = valid.astype(bool)
            assert np.allclose(out["xyz"][b][:n][ok], xyz[ok], rtol=1e-4, atol=1e-5) and ok.sum() > 50
            if prev is not None:
                i = b - 1
                pk, pd, pxyz, pvalid = prev
                f = oracle.feature_matching(pd, dL, 1.0)
                assert out["nf2f"][i] == len(f)
                okm = pvalid[f["queryIdx"]] != 0
                assert out["pn"][i] == okm.sum()
                p3 = pxyz[f["queryIdx"][okm]]; p2 = np.stack([kL["x"][f["trainIdx"][okm]], kL["y"][f["trainIdx"][okm]]], 1)
                assert np.allclose(out["pxyz"][i][:okm.sum()], p3, rtol=1e-4, atol=1e-5) and np.array_equal(out["puv"][i][:okm.sum()], p2)
                if pose == "ransac":   # on the GPU's own (f32, 1e-4-equal) points: the discrete outputs of RANSAC are not continuous in them
                    g3 = out["pxyz"][i][:okm.sum()]
                    wT, winl, wn, wit = oracle.pnp_ransac(g3, p2, lm_iters=0)
                    assert out["ninl"][i] == wn and np.array_equal(out["inl"][i][:okm.sum()], winl)
                    assert np.allclose(out["Tpnp"][i], wT, rtol=1e-9, atol=1e-12) and wn > 30
            prev = (kL, dL, xyz, valid)
    finally:
        pipe.close()


@pytest.mark.parametrize("depth,pose", [("match", "lm"), ("sgbm", "ransac")])
def test_pipelines_in_flight_together_give_the_same_bits(synth, depth, pose):
    """pipeline.PipelineRing: P contexts on P HIP streams with their steps queued back to back (what bench.py times by default).  The
    kernels of the batches share CUs, L2 and -- in the SGBM sweep -- compete for residency while they wait on each other's slabs; none
    of that may change a result: every pipeline must produce exactly what one pipeline stepping alone produces on the same images."""
    from stereo_visual_slam_amd.pipeline import KeyframePipeline, PipelineRing
    B, anms = 24, 1500
    seq = synth.stereo_sequence(8, seed=21, w=1241, h=376)
    kw = dict(anms_num=anms, unique_frames=8, sequence=seq, ba_windows="tracks", depth=depth, pose=pose)
    alone = KeyframePipeline(B, **kw)
    try:
        alone.step()
        ref = alone.download()
    finally:
        alone.close()
    ring = PipelineRing(3, B, **kw)
    try:
        for _ in range(7):   # steps 0..6: pipelines 0, 1, 2, 0, 1, 2, 0 -- queued without a synchronisation in between
            ring.step()
        ring.sync()
        keys = ["cnt", "kps", "desc", "nlr", "lr", "nf2f", "f2f", "xyz", "valid", "rel", "pn", "Tpnp", "inl", "ninl", "ba_lm_off", "ba_e_off", "ba_nkf",
                "ba_kf", "ba_lm", "ba_uv", "ba_xyz", "ba_rel", "ba_T", "ba_inl"]
        for p in ring.pipes:
            out = p.download()
            assert int(out["ba_build_status"][0]) == 0
            for k in keys:
                assert np.array_equal(out[k], ref[k]), k
    finally:
        ring.close()
