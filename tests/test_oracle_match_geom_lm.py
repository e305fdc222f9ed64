"""CPU known-answer / cross-check tests pinning the matcher, SE3, residual and LM parts of the oracle.
PARITY UNPINNED against OpenCV / g2o / Sophus themselves (absent); pinned against independent numpy / scipy restatements."""
import numpy as np
import pytest
from scipy.linalg import expm
from scipy.optimize import least_squares


# ------------------------------------------------------------------ matcher
def _xcheck_numpy(q, t):
    """BFMatcher(crossCheck=true) as OpenCV 3.2 batchDistance implements it, from the definition"""
    if len(q) == 0 or len(t) == 0:
        return []
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)   # nq x nt
    dist = np.full(len(q), np.iinfo(np.int32).max); idx = -np.ones(len(q), int)
    for j in range(len(t)):
        i = int(np.argmin(D[:, j]))          # first minimum
        if D[i, j] < dist[i]:
            dist[i], idx[i] = D[i, j], j
    return [(i, idx[i], dist[i]) for i in range(len(q)) if idx[i] >= 0]


@pytest.mark.parametrize("nq,nt,seed", [(60, 80, 0), (200, 150, 1), (1, 5, 2), (300, 300, 3)])
def test_xcheck_matches_numpy(oracle, synth, nq, nt, seed):
    q, t = synth.random_descriptors(nq, nt, seed=seed, tie_frac=0.2)
    got = oracle.bf_match_xcheck(q, t)
    want = _xcheck_numpy(q, t)
    assert [(m["queryIdx"], m["trainIdx"], int(m["distance"])) for m in got] == want
    assert (np.diff(got["queryIdx"]) > 0).all()


def test_xcheck_is_not_symmetric_mutual_nn(oracle):
    """the reference's cross-check is the asymmetric batchDistance rule: a query may win a train row that is not the
    query's own nearest neighbour"""
    z = np.zeros((1, 32), np.uint8)
    q = np.concatenate([z, z]); q[1, 0] = 0b1            # q1 at distance 1 from t0
    t = np.concatenate([z, z]); t[1, 0] = 0b111          # t1: nearest query is q1 (d=2) ; t0: nearest is q0 (d=0)
    m = oracle.bf_match_xcheck(q, t)
    assert [(a["queryIdx"], a["trainIdx"], a["distance"]) for a in m] == [(0, 0, 0.0), (1, 1, 2.0)]  # q1's own NN is t0 (d=1)


def test_gate(oracle, synth):
    q, t = synth.random_descriptors(300, 300, seed=5, flip_p=0.1)
    raw = oracle.bf_match_xcheck(q, t)
    for gap in (1.0, 2.0):
        thr = max(2.0 * raw["distance"].min(), 30.0 * gap)
        want = raw[raw["distance"] <= thr]
        got = oracle.feature_matching(q, t, gap)
        assert np.array_equal(got, want)
    assert len(oracle.feature_matching(q[:0], t)) == 0 and len(oracle.feature_matching(q, t[:0])) == 0


# ------------------------------------------------------------------ SE3
def _T4(T, oracle):
    M = np.eye(4); M[:3, :3] = oracle.se3_rotmat(T); M[:3, 3] = T[4:]
    return M


def test_se3_exp_matches_matrix_exponential(oracle):
    rng = np.random.default_rng(0)
    for s in (1e-12, 1e-6, 0.3, 2.5):
        xi = rng.normal(0, 1, 6) * s
        if np.linalg.norm(xi[3:]) > 3.0:
            xi[3:] *= 3.0 / np.linalg.norm(xi[3:])     # log is the principal branch: keep |omega| < pi
        tw = np.zeros((4, 4)); w = xi[3:]
        tw[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]; tw[:3, 3] = xi[:3]
        assert np.allclose(_T4(oracle.se3_exp(xi), oracle), expm(tw), atol=1e-12)
        assert np.allclose(oracle.se3_log(oracle.se3_exp(xi)), xi, atol=1e-9 * max(1, s))


def test_se3_group_ops(oracle):
    rng = np.random.default_rng(1)
    A = oracle.se3_exp(rng.normal(0, 0.5, 6)); B = oracle.se3_exp(rng.normal(0, 0.5, 6)); p = rng.normal(0, 3, 3)
    assert np.allclose(_T4(oracle.se3_mul(A, B), oracle), _T4(A, oracle) @ _T4(B, oracle), atol=1e-12)
    assert np.allclose(_T4(oracle.se3_inv(A), oracle), np.linalg.inv(_T4(A, oracle)), atol=1e-12)
    assert np.allclose(oracle.se3_act(A, p), (_T4(A, oracle) @ np.append(p, 1))[:3], atol=1e-12)
    R = _T4(A, oracle)[:3, :3]
    assert np.isclose(oracle.se3_angle_y(A), np.arctan2(-R[2, 0], np.hypot(R[0, 0], R[1, 0])))


# ------------------------------------------------------------------ residuals / Jacobians
def test_jacobians_match_finite_differences(oracle, synth):
    rng = np.random.default_rng(2)
    K = synth.K4
    for _ in range(20):
        T = oracle.se3_exp(rng.normal(0, 0.3, 6)); pw = np.array([rng.normal(0, 2), rng.normal(0, 1), rng.uniform(8, 30)])
        pw = oracle.se3_act(oracle.se3_inv(T), pw); z = rng.uniform(0, 300, 2)
        e, Jp, Jl = oracle.projection_residual(T, pw, z, K)
        e2, J2 = oracle.pose_only_residual(T, pw, z, K)
        assert np.allclose(e, e2) and np.allclose(Jp, J2, rtol=1e-12)
        h = 1e-6
        for a in range(6):       # left perturbation: T <- exp(d) T, tangent [translation; rotation]
            d = np.zeros(6); d[a] = h
            ep, _, _ = oracle.projection_residual(oracle.se3_mul(oracle.se3_exp(d), T), pw, z, K)
            em, _, _ = oracle.projection_residual(oracle.se3_mul(oracle.se3_exp(-d), T), pw, z, K)
            assert np.allclose((ep - em) / (2 * h), Jp[:, a], rtol=1e-5, atol=1e-5)
        for a in range(3):
            d = np.zeros(3); d[a] = h
            ep, _, _ = oracle.projection_residual(T, pw + d, z, K); em, _, _ = oracle.projection_residual(T, pw - d, z, K)
            assert np.allclose((ep - em) / (2 * h), Jl[:, a], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ geometry
def test_triangulate_exact_on_consistent_stereo(oracle, synth):
    rng = np.random.default_rng(3)
    n = 500
    P = np.stack([rng.uniform(-10, 10, n), rng.uniform(-3, 3, n), rng.uniform(5, 300, n)], 1)
    uL = synth.FX * P[:, 0] / P[:, 2] + synth.CX; v = synth.FY * P[:, 1] / P[:, 2] + synth.CY
    uR = synth.FX * (P[:, 0] - synth.BASELINE) / P[:, 2] + synth.CX
    T = oracle.se3_exp(rng.normal(0, 0.2, 6))
    xyz, valid, rel = oracle.triangulate_dlt(np.stack([uL, v], 1).astype(np.float64), np.stack([uR, v], 1).astype(np.float64), T)
    want = np.array([oracle.se3_act(oracle.se3_inv(T), p) for p in P])
    ok = valid.astype(bool)
    assert (ok == ((P[:, 2] > 10) & (P[:, 2] < 400))).mean() > 0.99          # f32 pixel rounding moves a few across the gate
    assert np.allclose(xyz[ok], want[ok], rtol=2e-2)                           # f32 pixels -> depth error grows with Z^2
    near = ok & (P[:, 2] < 30)
    assert np.allclose(xyz[near], want[near], rtol=2e-3, atol=2e-2)
    assert (rel[ok] == (P[ok, 2] < 40)).mean() > 0.99


def test_find_3d_contract(oracle, synth):
    disp = np.full((376, 1241), 8.0, np.float32); disp[100, 200] = -1; disp[101, 200] = 0
    kps = np.zeros(4, oracle.KEYPOINT_DTYPE)
    kps["x"] = [300.7, 200.9, 200.2, 600.0]; kps["y"] = [50.9, 100.9, 101.5, 200.0]   # float -> int truncation (quirk Q3)
    T = np.array([0, 0, 0, 1, 0, 0, 0], float)
    xyz, valid, rel = oracle.find_3d_disparity(kps, disp, T)
    Z = synth.FX * synth.BASELINE / 8.0                                               # 51.49 m: valid, not reliable
    assert list(valid) == [1, 0, 0, 1] and list(rel) == [0, 0, 0, 0]
    assert np.allclose(xyz[0], [(300.7 - synth.CX) / synth.FX * Z, (50.9 - synth.CY) / synth.FY * Z, Z], rtol=1e-6)
    disp[:] = 20.0                                                                    # 20.6 m: reliable
    _, valid, rel = oracle.find_3d_disparity(kps, disp, T)
    assert list(valid) == [1, 1, 1, 1] and list(rel) == [1, 1, 1, 1]


def test_check_motion(oracle):
    I = np.array([0, 0, 0, 1, 0, 0, 0], float)
    assert not oracle.check_motion(9, I, 1) and oracle.check_motion(10, I, 1)
    T = oracle.se3_exp([0, 0, 4.9, 0, 0, 0]); assert oracle.check_motion(50, T, 1)
    T = oracle.se3_exp([0, 0, 5.1, 0, 0, 0]); assert not oracle.check_motion(50, T, 1) and oracle.check_motion(50, T, 2)


# ------------------------------------------------------------------ LM vs scipy
def _huber_cost(e2, delta):
    return np.where(e2 <= delta ** 2, e2, 2 * np.sqrt(e2) * delta - delta ** 2).sum()


def test_pose_only_optimum_matches_scipy(oracle, synth):
    p = synth.pnp_problem(M=200, seed=11, outlier_frac=0.1)
    T, inl, n, st = oracle.pnp_motion_only(p["xyz"], p["uv"], p["T0"], iters=30)
    assert st["chi2_final"] < st["chi2_init"] and st["chi2_iter"] == sorted(st["chi2_iter"], reverse=True)
    xyz = p["xyz"].astype(np.float64); uv = p["uv"].astype(np.float64)

    def res(x):
        Tx = oracle.se3_mul(oracle.se3_exp(x), p["T0"])
        pc = xyz @ oracle.se3_rotmat(Tx).T + Tx[4:]
        return np.stack([uv[:, 0] - (synth.FX * pc[:, 0] / pc[:, 2] + synth.CX), uv[:, 1] - (synth.FY * pc[:, 1] / pc[:, 2] + synth.CY)], 1)

    # scipy's huber acts per scalar residual; g2o's acts on the 2-vector norm.  Compare at the optimum of OUR cost:
    def cost(x):
        r = res(x); return _huber_cost((r ** 2).sum(1), 5.991)
    from scipy.optimize import minimize
    x_or = oracle.se3_log(oracle.se3_mul(T, oracle.se3_inv(p["T0"])))
    sol = minimize(cost, x_or, method="BFGS", options=dict(gtol=1e-10))
    assert np.isclose(cost(x_or), st["chi2_final"], rtol=1e-9)
    assert cost(x_or) <= sol.fun * (1 + 1e-9)
    assert np.allclose(sol.x, x_or, atol=2e-6)
    assert n == int(((res(x_or) ** 2).sum(1) <= 16.0).sum())
    assert np.allclose(T, p["T_true"], atol=2e-2)


def test_local_ba_reaches_least_squares_optimum(oracle, synth):
    w = synth.ba_window(n_kf=4, n_lm=60, seed=3, outlier_frac=0.0, sigma_px=0.3, max_obs=4)
    T, xyz, chi2, st = oracle.local_ba(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], iters=30, huber_delta=1e9,
                                       update_poses=True, update_lms=True)
    assert st["chi2_final"] < 1e-2 * st["chi2_init"]
    assert np.isclose(chi2.sum(), st["chi2_final"], rtol=1e-6)
    nk, nl = 4, 60

    def res(x):
        r = []
        Ts = [oracle.se3_mul(oracle.se3_exp(x[6 * k:6 * k + 6]), w["T0"][k]) for k in range(nk)]
        P = w["xyz"].astype(np.float64) + x[6 * nk:].reshape(nl, 3)
        for k, l, z in zip(w["kf_idx"], w["lm_idx"], w["uv"].astype(np.float64)):
            pc = oracle.se3_rotmat(Ts[k]) @ P[l] + Ts[k][4:]
            r += [z[0] - (synth.FX * pc[0] / pc[2] + synth.CX), z[1] - (synth.FY * pc[1] / pc[2] + synth.CY)]
        return np.array(r)
    sol = least_squares(res, np.zeros(6 * nk + 3 * nl), method="lm", xtol=1e-14, ftol=1e-14, max_nfev=4000)
    assert st["chi2_final"] <= 2 * sol.cost * (1 + 1e-3) + 1e-9      # gauge-free: compare the optimal COST (2*cost = sum r^2)
    assert st["chi2_final"] >= 2 * sol.cost * (1 - 1e-3) - 1e-9


def test_lm_bookkeeping(oracle, synth):
    w = synth.ba_window(n_kf=10, n_lm=300, seed=2)
    T, xyz, chi2, st = oracle.local_ba(w["T0"], w["xyz"], w["kf_idx"], w["lm_idx"], w["uv"], iters=10, update_poses=False)
    assert (T == w["T0"]).all() and (xyz == w["xyz"]).all()          # no write-back (optimization.cpp:272)
    assert st["iterations"] == len(st["chi2_iter"]) <= 10 and sum(st["trials_iter"]) == st["total_trials"]
    assert all(1 <= t <= 10 for t in st["trials_iter"])
    # lambda0 = 1e-5 * max diag; an accepted first step multiplies lambda by a factor in [1/3, 2/3]
    if st["trials_iter"][0] == 1 and len(st["lambda_iter"]) > 1 and st["trials_iter"][1] == 1:
        r = st["lambda_iter"][1] / st["lambda_iter"][0]
        assert 1 / 3 - 1e-12 <= r <= 2 / 3 + 1e-12
    th, inl, ni, no = oracle.chi2_classify(chi2, w["lm_idx"], np.ones(300, np.uint8))
    assert th in (5.991, 2 * 5.991, 4 * 5.991, 8 * 5.991, 16 * 5.991, 32 * 5.991) and ni + no == len(chi2) and ni > no
    # last edge of a landmark (ascending edge index) decides its flag
    last = {}
    for e, l in enumerate(w["lm_idx"]):
        last[l] = e
    assert all(inl[l] == (chi2[e] <= th) for l, e in last.items())


def test_chi2_threshold_doubling(oracle):
    chi2 = np.array([100.0] * 8 + [1.0] * 2)
    th, inl, ni, no = oracle.chi2_classify(chi2, np.arange(10, dtype=np.int32), np.ones(10, np.uint8))
    # 5 doublings -> 191.7; the counts are those of the LAST evaluated threshold (95.9): the reference never recounts
    assert th == 5.991 * 32 and ni == 2 and list(inl) == [1] * 10
    chi2 = np.array([1e6] * 8 + [1.0] * 2)
    th, inl, ni, no = oracle.chi2_classify(chi2, np.arange(10, dtype=np.int32), np.ones(10, np.uint8))
    assert th == 5.991 * 32 and ni == 2 and list(inl) == [0] * 8 + [1] * 2
