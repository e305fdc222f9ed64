#!/usr/bin/env python3
"""How much would each UNPINNED reading of an OpenCV / g2o detail change?  (VERDICT r2 next #8.)

The oracle (and therefore the HIP path) restates third-party arithmetic that cannot be checked against the real libraries here
("parity unpinned").  For every documented ambiguity this tool flips ONE reading inside the CPU oracle
(oracle/vo_oracle.h VO_VAR_*), re-runs the front end of the 50-pair rendered sequence (BASELINE config 1 shape) and counts what
changes against the default reading: keypoints, orientation, descriptor bits, L/R and frame-to-frame matches, the motion-only
pose.  It cannot make parity green; it tells a maintainer with a real OpenCV 3.2 / g2o build which choices matter and where to look.

CPU only, TEST INFRASTRUCTURE (executes the oracle):  python tests/oracle_sensitivity.py [--frames 50] [--anms 500] [--out profiles/...json]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

VARIANTS = [
    ("cosf_sinf_in_rbrief", 1, "computeOrbDescriptors rotates the pattern with cos/sin of a float angle: the double overloads rounded to float (default) vs cosf/sinf"),
    ("retainBest_exact_n", 2, "KeyPointsFilter::retainBest keeps all ties at the n-th response (default: the documented intent) vs exactly n survivors"),
    ("resize_single_rounding", 4, "8-bit INTER_LINEAR vertical pass: the library's >>4, >>16, (+2)>>2 chain (default) vs one rounding of the 22-bit product"),
    ("atan2f_instead_of_fastAtan2", 8, "IC angle through cv::fastAtan2's polynomial (default, <= 0.009 deg off) vs libm atan2f"),
    ("stale_update_on_failed_solve", 16, "g2o applies the previous solve's x when the linear solve fails; the oracle uses x_p = 0 (the trial is rejected either way)"),
]


def front_end(O, seq, anms, n_frames):
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
    out, prev = [], None
    for f in range(n_frames):
        L, R = seq[f][0], seq[f][1]
        kL, dL = O.feature_detection(L, 3000, anms); kR, dR = O.feature_detection(R, 3000, anms)
        m = O.feature_matching(dL, dR, 1.0)
        uvL = np.stack([kL["x"][m["queryIdx"]], kL["y"][m["queryIdx"]]], 1); uvR = np.stack([kR["x"][m["trainIdx"]], kR["y"][m["trainIdx"]]], 1)
        xyz, valid, _ = O.triangulate_dlt(uvL, uvR, ident)
        T, fm, ninl = None, None, 0
        if prev is not None:
            pk, pd, pm, pxyz, pvalid = prev
            fm = O.feature_matching(pd, dL, 1.0)
            kp2lr = -np.ones(len(pk), np.int64); kp2lr[pm["queryIdx"]] = np.arange(len(pm))
            li = kp2lr[fm["queryIdx"]]
            ok = (li >= 0) & (pvalid[np.maximum(li, 0)] != 0)
            if ok.sum() >= 6:
                T, _, ninl, _ = O.pnp_motion_only(pxyz[li[ok]], np.stack([kL["x"][fm["trainIdx"][ok]], kL["y"][fm["trainIdx"][ok]]], 1), ident, iters=10)
        out.append(dict(kL=kL, dL=dL, kR=kR, dR=dR, lr=m, f2f=fm, T=T, ninl=ninl))
        prev = (kL, dL, m, xyz, valid)
    return out


def kp_key(k):
    return {(float(x), float(y), int(o)): i for i, (x, y, o) in enumerate(zip(k["x"], k["y"], k["octave"]))}


def match_set(k_q, k_t, m):
    return {((float(k_q["x"][q]), float(k_q["y"][q]), int(k_q["octave"][q])), (float(k_t["x"][t]), float(k_t["y"][t]), int(k_t["octave"][t]))) for q, t in zip(m["queryIdx"], m["trainIdx"])}


def compare(base, var):
    r = dict(keypoints_total=0, keypoints_changed=0, angle_changed=0, angle_max_abs_deg=0.0, descriptors_compared=0, descriptors_changed=0, descriptor_bits_flipped=0,
             lr_matches_total=0, lr_matches_changed=0, f2f_matches_total=0, f2f_matches_changed=0, poses_compared=0, pose_translation_rmse_m=0.0, pose_max_translation_m=0.0,
             pnp_inlier_count_changed=0)
    tsq = []
    for b, v in zip(base, var):
        for kb, db, kv, dv in ((b["kL"], b["dL"], v["kL"], v["dL"]), (b["kR"], b["dR"], v["kR"], v["dR"])):
            A, Bm = kp_key(kb), kp_key(kv)
            common = [(A[k], Bm[k]) for k in A if k in Bm]
            r["keypoints_total"] += len(A)
            r["keypoints_changed"] += len(A) - len(common) + (len(Bm) - len(common))
            if common:
                ia = np.array([c[0] for c in common]); ib = np.array([c[1] for c in common])
                da = np.abs(kb["angle"][ia] - kv["angle"][ib]); da = np.minimum(da, 360 - da)
                r["angle_changed"] += int((da > 0).sum()); r["angle_max_abs_deg"] = max(r["angle_max_abs_deg"], float(da.max()))
                x = np.bitwise_xor(db[ia], dv[ib])
                bits = np.unpackbits(x, axis=1).sum(1)
                r["descriptors_compared"] += len(ia); r["descriptors_changed"] += int((bits > 0).sum()); r["descriptor_bits_flipped"] += int(bits.sum())
        sb, sv = match_set(b["kL"], b["kR"], b["lr"]), match_set(v["kL"], v["kR"], v["lr"])
        r["lr_matches_total"] += len(sb); r["lr_matches_changed"] += len(sb ^ sv)
        if b["f2f"] is not None and v["f2f"] is not None:
            r["f2f_matches_total"] += len(b["f2f"]); r["f2f_matches_changed"] += abs(len(b["f2f"]) - len(v["f2f"]))
        if b["T"] is not None and v["T"] is not None:
            d = b["T"][4:] - v["T"][4:]
            tsq.append(float(d @ d)); r["poses_compared"] += 1
            r["pose_max_translation_m"] = max(r["pose_max_translation_m"], float(np.sqrt(d @ d)))
            r["pnp_inlier_count_changed"] += int(b["ninl"] != v["ninl"])
    if tsq:
        r["pose_translation_rmse_m"] = float(np.sqrt(np.mean(tsq)))
    return r


def ba_sensitivity(O, synth, flag):
    """local-BA windows (config 4 shape, reduced): poses after 10 iterations, default vs variant"""
    worst, fails = 0.0, 0
    for seed in range(4):
        w = synth.ba_window(n_kf=10, n_lm=600, seed=40 + seed)
        O.lib().vo_set_variant(0)
        T0, _, _, st0 = O.local_ba(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], iters=10)
        O.lib().vo_set_variant(flag)
        T1, _, _, st1 = O.local_ba(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], iters=10)
        O.lib().vo_set_variant(0)
        worst = max(worst, float(np.abs(T0 - T1).max()))
    return dict(windows=4, max_abs_pose_diff=worst, note="a failed linear solve never occurred on these windows" if worst == 0.0 else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=50); ap.add_argument("--anms", type=int, default=500)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import oracle as O
    from stereo_visual_slam_amd import synth
    O.build()
    seq = synth.stereo_sequence(a.frames, seed=5, workers=min(8, os.cpu_count() or 1))
    t0 = time.time()
    O.lib().vo_set_variant(0)
    base = front_end(O, seq, a.anms, a.frames)
    res = {"frames": a.frames, "anms": a.anms, "images": 2 * a.frames,
           "baseline": {"keypoints": int(sum(len(b["kL"]) + len(b["kR"]) for b in base)), "lr_matches": int(sum(len(b["lr"]) for b in base)),
                        "f2f_matches": int(sum(len(b["f2f"]) for b in base if b["f2f"] is not None))},
           "variants": {}}
    for name, flag, what in VARIANTS:
        if flag == 16:
            res["variants"][name] = dict(what=what, local_ba=ba_sensitivity(O, synth, flag))
            continue
        O.lib().vo_set_variant(flag)
        var = front_end(O, seq, a.anms, a.frames)
        O.lib().vo_set_variant(0)
        res["variants"][name] = dict(what=what, **compare(base, var))
        print(name, json.dumps(res["variants"][name]), flush=True)
    # the returned pose of solvePnPRansac: best RANSAC model (OpenCV 3.2.0, default) vs refined on the inliers (3.4.2+)
    d = []
    for seed in range(20):
        p = synth.pnp_problem(M=300, seed=100 + seed, outlier_frac=0.3)
        Ta = O.pnp_ransac(p["xyz"], p["uv"], lm_iters=0)[0]; Tb = O.pnp_ransac(p["xyz"], p["uv"], lm_iters=10)[0]
        d.append((float(np.linalg.norm(Ta[4:] - p["T_true"][4:])), float(np.linalg.norm(Tb[4:] - p["T_true"][4:])), float(np.linalg.norm(Ta[4:] - Tb[4:]))))
    d = np.array(d)
    res["variants"]["solvePnPRansac_returns_refined_pose"] = dict(
        what="OpenCV 3.2.0 assigns _local_model (best 5-point EPnP model) to rvec/tvec (default, lm_iters = 0) vs the pose refined on the inliers (3.4.2+, lm_iters > 0)",
        problems=20, translation_error_model_m=float(d[:, 0].mean()), translation_error_refined_m=float(d[:, 1].mean()), mean_translation_between_them_m=float(d[:, 2].mean()))
    res["seconds"] = round(time.time() - t0, 1)
    s = json.dumps(res, indent=1)
    if a.out:
        open(a.out, "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
