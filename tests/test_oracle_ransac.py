"""CPU known-answer / property tests of the RANSAC oracle (oracle/ransac.c): the restated control flow of
cv::solvePnPRansac(pts3d, pts2d, K, Mat(), rvec, tvec, false, 100, 4.0, 0.99, inliers) (visual_odometry.cpp:277)."""
import numpy as np


def _mwc(n_draws, state=0xFFFFFFFFFFFFFFFF):
    """cv::RNG::next restated in Python ints (independent of the C code)"""
    out = []
    for _ in range(n_draws):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        out.append(state & 0xFFFFFFFF)
    return out


def test_subset_sequence_follows_cv_rng(oracle):
    count, iters = 97, 40
    got = oracle.ransac_subsets(count, iters)
    draws = iter(_mwc(10000))
    for it in range(iters):
        idx = []
        while len(idx) < 5:
            v = next(draws) % count
            if v not in idx:
                idx.append(v)       # a duplicate is redrawn in place (getSubset)
        assert list(got[it]) == idx, it
    assert all(len(set(r)) == 5 for r in got)


def test_update_num_iters_closed_form(oracle):
    # niters = log(1 - p) / log(1 - (1 - ep)^m), rounded; capped at max_iters; 0 for an all-inlier model
    for p, ep, m, cap in [(0.99, 0.5, 5, 100), (0.99, 0.3, 5, 100), (0.99, 0.1, 5, 100), (0.999, 0.2, 4, 1000), (0.99, 0.95, 5, 100)]:
        want = min(cap, int(np.rint(np.log(1 - p) / np.log(1 - (1 - ep) ** m))))
        assert oracle.ransac_update_num_iters(p, ep, m, cap) == want
    assert oracle.ransac_update_num_iters(0.99, 0.0, 5, 100) == 0
    assert oracle.ransac_update_num_iters(0.99, 1.0, 5, 100) == 100


def test_ransac_rejects_gross_outliers_and_stops_early(oracle, synth):
    p = synth.pnp_problem(M=400, seed=9, outlier_frac=0.35, sigma_px=0.4)
    T, inl, n, iters = oracle.pnp_ransac(p["xyz"], p["uv"], p["T0"], lm_iters=10)   # refined on the inliers (OpenCV 3.4.2+)
    truth = ~p["outlier"]
    assert n == inl.sum() and 0.55 * 400 < n < 0.75 * 400
    # lm_iters = 0 (the default: OpenCV 3.2.0, the reference's pinned version, returns `_local_model`): the accepted 5-point EPnP model itself,
    # same mask / count / iterations; an algebraic 5-point estimate, so further from the truth than its refinement
    T0, inl0, n0, iters0 = oracle.pnp_ransac(p["xyz"], p["uv"])
    assert (n0, iters0) == (n, iters) and np.array_equal(inl0, inl)
    hyp = [oracle.pnp_ransac_hypothesis(p["xyz"], p["uv"], h) for h in range(iters)]
    best = max(range(iters), key=lambda h: (hyp[h][1], -h))
    assert hyp[best][1] == n and np.allclose(T0, hyp[best][0], atol=1e-12)
    assert 1 <= iters < 100                                  # 35 % outliers: the adaptive rule stops before the cap
    # plain (non-robust, non-RANSAC) least squares on the contaminated set lands elsewhere
    Tls, _, _, _ = oracle.pnp_motion_only(p["xyz"], p["uv"], p["T0"], iters=10, huber_delta=1e300)
    ang = lambda A, B: np.linalg.norm(oracle.se3_log(oracle.se3_mul(A, oracle.se3_inv(B))))
    assert ang(T, p["T_true"]) < 0.02 and ang(T, p["T_true"]) < 0.5 * ang(Tls, p["T_true"])
    assert ang(T, p["T_true"]) < ang(T0, p["T_true"]) < 0.2
    assert (inl.astype(bool) & ~truth).sum() <= 0.03 * 400    # a gross outlier lands within 4 px only by chance


def test_ransac_degenerate_inputs(oracle, synth):
    p = synth.pnp_problem(M=4, seed=1)
    T, inl, n, iters = oracle.pnp_ransac(p["xyz"], p["uv"], p["T0"])
    assert n == 0 and iters == 0 and np.array_equal(T, p["T0"])          # fewer points than the minimal set
    p = synth.pnp_problem(M=60, seed=2, outlier_frac=0.0, sigma_px=0.1)
    T, inl, n, iters = oracle.pnp_ransac(p["xyz"], p["uv"], p["T0"])
    assert n == 60 and iters == 1                                        # an all-inlier model ends the loop at once
