#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep.  Sizes and seeds are drawn at random; any mismatch reports the reproducer.
Stand-alone (minutes): python tests/fuzz_parity.py [--seconds 120] [--seed 0] [--only kind]; a fixed-seed slice of it runs inside
`pytest -m gpu` (tests/test_gpu_fuzz.py)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

KINDS = ["match", "sgbm", "orb", "ba", "pnp", "ransac", "windows", "ransac_dev", "ba_schedule", "ba_resident"]


class Mismatch(AssertionError):
    pass


def kps_equal(x, y):
    return len(x) == len(y) and all(np.array_equal(x[f], y[f]) for f in ("x", "y", "size", "angle", "response", "octave", "class_id"))


def one_case(kind, rng, vo, pkg, O, synth):
    def fail(what, **kw):
        raise Mismatch("MISMATCH %s %r (fuse_min %s fwd_min %s rows %s)" % (what, kw, forced, fwd, rows))
    seed = int(rng.integers(1 << 30))
    # the fused ORB / SGBM kernels are picked by batch size (large batches only); the fuzz cases are single items, so half of them force the fused path
    forced = "1" if rng.random() < 0.5 else "1000000"
    fwd = forced if rng.random() < 0.8 else "1000000"; rows = "32" if rng.random() < 0.5 else "64"
    vo.set_tuning(orb_fuse_min=int(forced), sgbm_fuse_min=int(forced), sgbm_fwd_min=int(fwd), sgbm_fw_rows=int(rows))
    if kind == "match":
        nq, nt = int(rng.integers(1, 2200)), int(rng.integers(1, 2200))
        if rng.random() < 0.2: nq = int(rng.integers(1, 70))
        if rng.random() < 0.2: nt = int(rng.integers(1, 70))
        q, t = synth.random_descriptors(nq, nt, seed=seed, tie_frac=float(rng.choice([0.0, 0.05, 0.5])))
        gap = float(rng.integers(1, 4))
        for gate in (False, True):
            g = vo.feature_matching(q, t, gap, gate=gate)
            w = O.feature_matching(q, t, gap) if gate else O.bf_match_xcheck(q, t)
            if len(g) != len(w) or any(not np.array_equal(g[f], w[f]) for f in ("queryIdx", "trainIdx", "distance")):
                fail("match", nq=nq, nt=nt, seed=seed, gate=gate)
    elif kind == "sgbm":
        w, h = int(rng.integers(101, 700)), int(rng.integers(10, 200))
        L = synth.noise_image(seed % 1000, w + 40, h)
        sh = int(rng.integers(0, 40))
        Lc = np.ascontiguousarray(L[:, :w]); R = np.ascontiguousarray(L[:, sh:sh + w]).copy()
        if rng.random() < 0.5:
            R = np.clip(R.astype(int) + rng.integers(-15, 16, R.shape), 0, 255).astype(np.uint8)
        gf, gi, graw = vo.disparity_map(Lc, R, return_i16=True)
        wi, wraw = O.sgbm_compute(Lc, R, return_raw=True)
        if not (np.array_equal(graw, wraw) and np.array_equal(gi, wi)):
            fail("sgbm", w=w, h=h, seed=seed, shift=sh)
    elif kind == "orb":
        w, h = int(rng.integers(96, 1400)), int(rng.integers(96, 700))
        img = synth.noise_image(seed % 100000, w, h)
        nf = int(rng.choice([300, 1000, 3000])); an = int(rng.choice([50, 100, 500, 1500]))
        ctx = pkg.VO(device=0, max_batch=1, img_w=w, img_h=h, orb_nfeatures=nf, anms_num=an)
        try:
            if not kps_equal(ctx.orb_detect(img), O.orb_detect(img, nf)):
                fail("orb_detect", w=w, h=h, seed=seed, nf=nf)
            k, d = ctx.feature_detection(img); wk, wd = O.feature_detection(img, nf, an)
            if not (kps_equal(k, wk) and np.array_equal(d, wd)):
                fail("feature_detection", w=w, h=h, seed=seed, nf=nf, an=an)
        finally:
            ctx.close()
    elif kind == "ba":
        nk = int(rng.integers(1, 13)); nl = int(rng.choice([40, 300, 1500, 2600]))
        win = synth.ba_window(n_kf=nk, n_lm=nl, seed=seed, max_obs=min(5, nk), min_obs=min(2, nk))
        it = int(rng.integers(1, 11))
        r = vo.optimize_map(win["T0"], win["xyz"], win["kf_idx"], win["lm_idx"], win["uv"], True, True, it)
        T, x, chi2, st = O.local_ba(win["T0"], win["xyz"], win["kf_idx"], win["lm_idx"], win["uv"], iters=it, update_poses=True, update_lms=True)
        if not (np.allclose(r["T"], T, rtol=1e-4, atol=1e-6) and np.allclose(r["xyz"], x, rtol=1e-4, atol=1e-4)):
            fail("local_ba", nk=nk, nl=nl, seed=seed, iters=it)
    elif kind == "ba_resident":
        # optimize_map on the LDS-resident kernel (forced: ba_resident = 1) vs the oracle, random window shapes incl. single-observation landmarks
        # (the Woodbury path), tracks through the whole window, 1..12 keyframes; and the same call twice: the same bits (round 5)
        nk = int(rng.integers(1, 13)); nl = int(rng.choice([40, 300, 1500, 2600, 4000]))
        lo_obs = int(rng.integers(1, 3)); hi_obs = int(rng.integers(lo_obs, 11))
        win = synth.ba_window(n_kf=nk, n_lm=nl, seed=seed, max_obs=min(hi_obs, nk), min_obs=min(lo_obs, nk))
        it = int(rng.integers(1, 11))
        try:
            vo.set_tuning(ba_resident=1)
            r = vo.optimize_map(win["T0"], win["xyz"], win["kf_idx"], win["lm_idx"], win["uv"], True, True, it)
            r2 = vo.optimize_map(win["T0"], win["xyz"], win["kf_idx"], win["lm_idx"], win["uv"], True, True, it)
        finally:
            vo.set_tuning(ba_resident=-1)
        T, x, chi2, st = O.local_ba(win["T0"], win["xyz"], win["kf_idx"], win["lm_idx"], win["uv"], iters=it, update_poses=True, update_lms=True)
        if not (np.allclose(r["T"], T, rtol=1e-4, atol=1e-6) and np.allclose(r["xyz"], x, rtol=1e-4, atol=1e-4) and np.allclose(r["chi2"], chi2, rtol=1e-4, atol=1e-6)):
            fail("ba_resident", nk=nk, nl=nl, seed=seed, iters=it, obs=(lo_obs, hi_obs))
        if not (np.array_equal(r["T"], r2["T"]) and np.array_equal(r["xyz"], r2["xyz"]) and np.array_equal(r["chi2"], r2["chi2"])):
            fail("ba_resident twice", nk=nk, nl=nl, seed=seed, iters=it, obs=(lo_obs, hi_obs))
    elif kind == "ba_schedule":
        # the adaptive BA schedule (a pass that flags nothing new is continued instead of repeated) against the plain one -- every optimize_map pass for
        # every window -- on random batches: bit-identical poses, flags and per-edge chi2 (round 4)
        import torch
        nw = int(rng.integers(1, 10)); nk = int(rng.integers(2, 11)); nl = int(rng.choice([40, 300, 900, 1500]))
        frac = float(rng.choice([0.0, 0.0, 0.005, 0.02, 0.05]))   # gross observation errors: passes that do flag, windows that need two or three of them
        wins = [synth.ba_window_fast(n_kf=nk, n_lm=nl, seed=seed + 17 * i, outlier_frac=frac, max_obs=min(5, nk), min_obs=min(2, nk)) for i in range(nw)]
        lm_off = np.cumsum([0] + [len(wn["xyz"]) for wn in wins]).astype(np.int32); e_off = np.cumsum([0] + [len(wn["kf_idx"]) for wn in wins]).astype(np.int32)
        d = "cuda"
        T0 = torch.from_numpy(np.stack([wn["T0"] for wn in wins])).to(d)
        xyz = torch.from_numpy(np.concatenate([wn["xyz"] for wn in wins])).to(d); kf = torch.from_numpy(np.concatenate([wn["kf_idx"] for wn in wins])).to(d)
        lm = torch.from_numpy(np.concatenate([wn["lm_idx"] for wn in wins])).to(d); uv = torch.from_numpy(np.concatenate([wn["uv"] for wn in wins])).to(d)
        t_lm, t_e = torch.from_numpy(lm_off).to(d), torch.from_numpy(e_off).to(d)
        outs = []
        try:
            for adaptive in (1, 0):
                T = T0.clone(); inl = torch.ones(int(lm_off[-1]), dtype=torch.uint8, device=d); chi = torch.zeros(int(e_off[-1]), dtype=torch.float64, device=d)
                bb = pkg.BaBatch()
                bb.n_windows = nw; bb.n_kf = nk
                bb.d_lm_off = t_lm.data_ptr(); bb.d_edge_off = t_e.data_ptr(); bb.d_T_c_w = T.data_ptr(); bb.d_xyz = xyz.data_ptr()
                bb.d_reliable = None; bb.d_lm_inlier = inl.data_ptr(); bb.d_kf_idx = kf.data_ptr(); bb.d_lm_idx = lm.data_ptr(); bb.d_uv = uv.data_ptr()
                bb.d_chi2 = chi.data_ptr(); bb.d_stats = None; bb.total_lm = int(lm_off[-1]); bb.total_edge = int(e_off[-1])
                torch.cuda.synchronize()
                vo.set_tuning(ba_adaptive=adaptive)
                vo.ba_batch_dev(bb, schedule=1)
                passes = vo.ba_schedule_passes(nw)
                outs.append((T.cpu().numpy(), inl.cpu().numpy(), chi.cpu().numpy(), passes, vo.ba_status(nw)))
        finally:
            vo.set_tuning(ba_adaptive=-1)
        (Ta, ia, ca, pa, sa), (Tb, ib, cb, pb, sb) = outs
        if not ((sa == sb).all() and (pb[sb == 0] == 3).all() and np.array_equal(Ta, Tb) and np.array_equal(ia, ib) and np.array_equal(ca, cb)):
            fail("ba_schedule", nw=nw, nk=nk, nl=nl, frac=frac, seed=seed, passes=pa.tolist())
    elif kind == "pnp":
        M = int(rng.integers(4, 1500))
        p = synth.pnp_problem(M=M, seed=seed, outlier_frac=float(rng.choice([0.0, 0.15, 0.4])))
        gT, gi, gn, _ = vo.motion_estimation(p["xyz"], p["uv"], p["T0"], iters=10)
        wT, wi, wn, _ = O.pnp_motion_only(p["xyz"], p["uv"], p["T0"], iters=10)
        if not (np.allclose(gT, wT, rtol=1e-4, atol=1e-6) and gn == wn):
            fail("pnp", M=M, seed=seed)
    elif kind == "windows":
        # vslam_build_windows_dev on random association / match / flag tables vs oracle/windows.c (round 4)
        import torch
        F = int(rng.integers(1, 24)); cap = int(rng.choice([64, 128, 256])); n_kf = int(rng.integers(1, 13))
        kps = np.zeros((F, cap), O.KEYPOINT_DTYPE); nk = rng.integers(1, cap + 1, F).astype(np.int32)
        for f in range(F):
            kps["x"][f, :nk[f]] = rng.uniform(0, 1241, nk[f]).astype(np.float32); kps["y"][f, :nk[f]] = rng.uniform(0, 376, nk[f]).astype(np.float32)
        lr = np.zeros((F, cap), O.DMATCH_DTYPE); nlr = np.zeros(F, np.int32)
        xyz = rng.uniform(-20, 20, (F, cap, 3)).astype(np.float32); xyz[..., 2] = rng.uniform(5, 60, (F, cap)).astype(np.float32)
        valid = (rng.random((F, cap)) < rng.uniform(0.3, 0.95)).astype(np.uint8); rel = (rng.random((F, cap)) < rng.uniform(0.1, 0.9)).astype(np.uint8)
        for f in range(F):
            n = int(rng.integers(0, nk[f] + 1)); lr["queryIdx"][f, :n] = rng.permutation(nk[f])[:n]; nlr[f] = n
        Fm = max(F - 1, 1)
        f2f = np.zeros((Fm, cap), O.DMATCH_DTYPE); nf2f = np.zeros(Fm, np.int32)
        for i in range(F - 1):
            n = int(rng.integers(0, min(nk[i], nk[i + 1]) + 1))
            f2f["queryIdx"][i, :n] = np.sort(rng.permutation(nk[i])[:n]); f2f["trainIdx"][i, :n] = rng.permutation(nk[i + 1])[:n]; nf2f[i] = n
        inl = (rng.random((Fm, cap)) < rng.uniform(0.2, 1.0)).astype(np.uint8)
        T_rel = np.stack([O.se3_exp(np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 0.02, 3)])) for _ in range(Fm)])
        lm_cap, e_cap = F * cap * (n_kf + 1), 2 * F * cap * (n_kf + 1)
        w = O.build_windows(kps, lr, nlr, xyz, valid, rel, f2f[:F - 1], nf2f[:F - 1], inl[:F - 1], T_rel[:F - 1], n_kf=n_kf, lm_capacity=lm_cap, edge_capacity=e_cap)
        dd = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        t = dict(kps=dd(kps.view(np.uint8)), lr=dd(lr.view(np.uint8)), nlr=dd(nlr), xyz=dd(xyz), valid=dd(valid), rel=dd(rel), f2f=dd(f2f.view(np.uint8)), nf2f=dd(nf2f),
                 inl=dd(inl), T=dd(T_rel), nk=dd(nk))
        tr = pkg.TracksIn()
        tr.n_frames = F; tr.kp_capacity = cap; tr.lr_capacity = cap; tr.match_capacity = cap; tr.pnp_capacity = cap
        tr.d_kps = t["kps"].data_ptr(); tr.d_lr = t["lr"].data_ptr(); tr.d_nlr = t["nlr"].data_ptr(); tr.d_xyz = t["xyz"].data_ptr(); tr.d_valid = t["valid"].data_ptr()
        tr.d_reliable = t["rel"].data_ptr(); tr.d_f2f = t["f2f"].data_ptr(); tr.d_nf2f = t["nf2f"].data_ptr(); tr.d_pose_inlier = t["inl"].data_ptr()
        tr.d_T_rel = t["T"].data_ptr(); tr.d_nkps = t["nk"].data_ptr() if rng.random() < 0.5 else None
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
        o = dict(lm_off=z(F + 1, torch.int32), e_off=z(F + 1, torch.int32), nkf=z(F, torch.int32), T=z((F, n_kf, 7), torch.float64), xyz=z((lm_cap, 3), torch.float32),
                 rel=z(lm_cap, torch.uint8), inl=z(lm_cap, torch.uint8), kf=z(e_cap, torch.int32), lm=z(e_cap, torch.int32), uv=z((e_cap, 2), torch.float32), st=z(1, torch.int32))
        bb = pkg.BaBatch()
        bb.d_lm_off = o["lm_off"].data_ptr(); bb.d_edge_off = o["e_off"].data_ptr(); bb.d_T_c_w = o["T"].data_ptr(); bb.d_xyz = o["xyz"].data_ptr()
        bb.d_reliable = o["rel"].data_ptr(); bb.d_lm_inlier = o["inl"].data_ptr(); bb.d_kf_idx = o["kf"].data_ptr(); bb.d_lm_idx = o["lm"].data_ptr()
        bb.d_uv = o["uv"].data_ptr(); bb.d_n_kf = o["nkf"].data_ptr()
        torch.cuda.synchronize()
        vo.build_windows_dev(tr, n_kf, lm_cap, e_cap, bb, o["st"].data_ptr()); vo.sync()
        g = {k: v.cpu().numpy() for k, v in o.items()}
        nl, ne = int(w["lm_off"][F]), int(w["edge_off"][F])
        ok = (g["st"][0] == w["status"] == 0 and np.array_equal(g["lm_off"], w["lm_off"]) and np.array_equal(g["e_off"], w["edge_off"]) and np.array_equal(g["nkf"], w["n_kf"])
              and np.array_equal(g["kf"][:ne], w["kf_idx"][:ne]) and np.array_equal(g["lm"][:ne], w["lm_idx"][:ne]) and np.array_equal(g["uv"][:ne], w["uv"][:ne])
              and np.array_equal(g["rel"][:nl], w["reliable"][:nl]) and np.allclose(g["xyz"][:nl], w["xyz"][:nl], rtol=3e-6, atol=2e-5) and np.allclose(g["T"], w["T"], rtol=1e-9, atol=1e-11))
        if not ok:
            fail("windows", F=F, cap=cap, n_kf=n_kf, seed=seed)
    elif kind == "ransac_dev":
        # vslam_pnp_ransac_dev: a batch of problems of random sizes in one call vs oracle/ransac.c problem by problem (round 4)
        import torch
        B = int(rng.integers(1, 7)); cap = 1280
        Ms = [int(rng.choice([0, 3, 5, 6, int(rng.integers(7, 1200))])) for _ in range(B)]
        xyz = np.zeros((B, cap, 3), np.float32); uv = np.zeros((B, cap, 2), np.float32); probs = []
        for b, M in enumerate(Ms):
            p = synth.pnp_problem(M=max(M, 1), seed=seed + b, outlier_frac=float(rng.choice([0.0, 0.2, 0.45])), sigma_px=0.4) if M > 0 else None
            if M > 0:
                xyz[b, :M] = p["xyz"][:M]; uv[b, :M] = p["uv"][:M]
            probs.append(p)
        d_xyz, d_uv, d_n = torch.from_numpy(xyz).cuda(), torch.from_numpy(uv).cuda(), torch.tensor(Ms, dtype=torch.int32, device="cuda")
        d_T = torch.zeros((B, 7), dtype=torch.float64, device="cuda"); d_inl = torch.zeros((B, cap), dtype=torch.uint8, device="cuda")
        d_ni = torch.zeros(B, dtype=torch.int32, device="cuda"); d_it = torch.zeros(B, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        vo.pnp_ransac_dev(d_xyz.data_ptr(), d_uv.data_ptr(), d_n.data_ptr(), cap, B, d_T.data_ptr(), 100, 4.0, 0.99, d_inl.data_ptr(), d_ni.data_ptr(), d_it.data_ptr()); vo.sync()
        T, inl, ni, itr = d_T.cpu().numpy(), d_inl.cpu().numpy(), d_ni.cpu().numpy(), d_it.cpu().numpy()
        for b, M in enumerate(Ms):
            if M < 5:
                if ni[b] != 0 or inl[b].any():
                    fail("ransac_dev small", Ms=Ms, seed=seed)
                continue
            wT, wi, wn, wit = O.pnp_ransac(probs[b]["xyz"][:M], probs[b]["uv"][:M], lm_iters=0)
            if not (itr[b] == wit and ni[b] == wn and np.array_equal(inl[b][:M], wi) and not inl[b][M:].any() and (wn == 0 or np.allclose(T[b], wT, rtol=1e-12, atol=1e-14))):
                fail("ransac_dev", Ms=Ms, b=b, seed=seed, got=(int(ni[b]), int(itr[b])), want=(wn, wit))
    else:
        M = int(rng.integers(5, 1200))
        p = synth.pnp_problem(M=M, seed=seed, outlier_frac=float(rng.choice([0.0, 0.2, 0.45])), sigma_px=0.4)
        lmi = int(rng.choice([0, 10]))   # 0: the RANSAC model itself (OpenCV 3.2.0), 10: refined on the inliers
        gT, gi, gn, git = vo.motion_estimation_ransac(p["xyz"], p["uv"], p["T0"], lm_iters=lmi)
        wT, wi, wn, wit = O.pnp_ransac(p["xyz"], p["uv"], p["T0"], lm_iters=lmi)
        if not (git == wit and gn == wn and np.array_equal(gi, wi) and np.allclose(gT, wT, rtol=1e-4, atol=1e-6)):
            fail("ransac", M=M, seed=seed, got=(gn, git), want=(wn, wit))


def run(seconds=120.0, seed=0, only="", vo=None, max_cases=None, schedule=None):
    """returns the per-kind case counts; raises Mismatch with the reproducer on the first difference.  schedule: explicit list of
    kinds to cycle through (the pytest slice uses it so that every stage gets its share inside a short budget)"""
    import oracle as O
    import stereo_visual_slam_amd as pkg
    from stereo_visual_slam_amd import synth
    O.build()
    rng = np.random.default_rng(seed)
    own = vo is None
    if own:
        vo = pkg.VO(device=0, max_batch=2)
    t_end = time.time() + seconds
    n = {k: 0 for k in KINDS}
    i = 0
    try:
        while time.time() < t_end and (max_cases is None or i < max_cases):
            kind = only or (schedule[i % len(schedule)] if schedule else rng.choice(["match", "match", "sgbm", "orb", "ba", "pnp", "ransac", "windows", "ransac_dev", "ba_schedule", "ba_resident"]))
            one_case(kind, rng, vo, pkg, O, synth)
            n[kind] += 1
            i += 1
    finally:
        if not own:
            vo.set_tuning(orb_fuse_min=-1, sgbm_fuse_min=-1, sgbm_fwd_min=-1, sgbm_fw_rows=-1)
        if own:
            vo.close()
    return n


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120); ap.add_argument("--seed", type=int, default=0); ap.add_argument("--only", default="")
    a = ap.parse_args()
    try:
        print("fuzz ok:", run(a.seconds, a.seed, a.only))
    except Mismatch as e:
        print(e); sys.exit(1)
