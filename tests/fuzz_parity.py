#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep.  Sizes and seeds are drawn at random; any mismatch reports the reproducer.
Stand-alone (minutes): python tests/fuzz_parity.py [--seconds 120] [--seed 0] [--only kind]; a fixed-seed slice of it runs inside
`pytest -m gpu` (tests/test_gpu_fuzz.py)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

KINDS = ["match", "sgbm", "orb", "ba", "pnp", "ransac"]


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
    elif kind == "pnp":
        M = int(rng.integers(4, 1500))
        p = synth.pnp_problem(M=M, seed=seed, outlier_frac=float(rng.choice([0.0, 0.15, 0.4])))
        gT, gi, gn, _ = vo.motion_estimation(p["xyz"], p["uv"], p["T0"], iters=10)
        wT, wi, wn, _ = O.pnp_motion_only(p["xyz"], p["uv"], p["T0"], iters=10)
        if not (np.allclose(gT, wT, rtol=1e-4, atol=1e-6) and gn == wn):
            fail("pnp", M=M, seed=seed)
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
            kind = only or (schedule[i % len(schedule)] if schedule else rng.choice(["match", "match", "sgbm", "orb", "ba", "pnp", "ransac"]))
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
