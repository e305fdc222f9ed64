"""CPU known-answer tests of the EPnP restatement (oracle/epnp.c: the minimal solver cv::solvePnPRansac runs on every 5-point
hypothesis, visual_odometry.cpp:277).  OpenCV is not available, so the restatement is pinned by (1) an independent numpy
restatement of the published algorithm (numpy's LAPACK eigen / SVD / lstsq instead of the oracle's Jacobi / QR) on point sets
whose null space is one-dimensional (n >= 6), (2) exact pose recovery on noise-free sets including the 5-point case RANSAC uses,
(3) agreement with the least-squares optimum on noisy sets."""
import numpy as np
import pytest

K = np.array([718.856, 718.856, 607.1928, 185.2157])


def _problem(synth, n, seed, sigma=0.0):
    p = synth.pnp_problem(M=n, seed=seed, sigma_px=0.0, outlier_frac=0.0)
    uv, _ = synth.project(p["T_true"], p["xyz"].astype(np.float64))
    uv = uv + np.random.default_rng(seed + 99).normal(0, sigma, uv.shape)
    return p["xyz"], uv.astype(np.float32), synth.R_from_quat(p["T_true"][:4]), p["T_true"][4:]


def _epnp_numpy(xyz, uv):
    """Lepetit / Moreno-Noguer / Fua 2009 with numpy linear algebra (structure of OpenCV's epnp.cpp, independent arithmetic)"""
    fu, fv, uc, vc = K
    pws = xyz.astype(np.float64); n = len(pws)
    xn = ((uv[:, 0].astype(np.float64) - uc) * (1.0 / fu)).astype(np.float32); yn = ((uv[:, 1].astype(np.float64) - vc) * (1.0 / fv)).astype(np.float32)
    us = np.stack([xn.astype(np.float64) * fu + uc, yn.astype(np.float64) * fv + vc], 1)
    c0 = pws.mean(0); d = pws - c0
    w, V = np.linalg.eigh(d.T @ d)
    V = V * np.where(V[np.abs(V).argmax(0), np.arange(3)] < 0, -1.0, 1.0)   # sign convention: largest-magnitude component positive
    cws = np.vstack([c0] + [c0 + np.sqrt(max(w[i], 0) / n) * V[:, i] for i in (2, 1, 0)])
    al = np.linalg.solve((cws[1:] - cws[0]).T, (pws - cws[0]).T).T
    al = np.hstack([1 - al.sum(1, keepdims=True), al])
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = al[:, j] * fu; M[0::2, 3 * j + 2] = al[:, j] * (uc - us[:, 0])
        M[1::2, 3 * j + 1] = al[:, j] * fv; M[1::2, 3 * j + 2] = al[:, j] * (vc - us[:, 1])
    ew, ev = np.linalg.eigh(M.T @ M)
    v = [ev[:, i].reshape(4, 3) for i in range(4)]
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    dv = [[v[i][a] - v[i][b] for (a, b) in pairs] for i in range(4)]
    L = np.array([[dv[0][i] @ dv[0][i], 2 * dv[0][i] @ dv[1][i], dv[1][i] @ dv[1][i], 2 * dv[0][i] @ dv[2][i], 2 * dv[1][i] @ dv[2][i], dv[2][i] @ dv[2][i],
                   2 * dv[0][i] @ dv[3][i], 2 * dv[1][i] @ dv[3][i], 2 * dv[2][i] @ dv[3][i], dv[3][i] @ dv[3][i]] for i in range(6)])
    rho = np.array([((cws[a] - cws[b]) ** 2).sum() for (a, b) in pairs])

    def approx(N):
        cols = {1: [0, 1, 3, 6], 2: [0, 1, 2], 3: [0, 1, 2, 3, 4]}[N]
        x = np.linalg.lstsq(L[:, cols], rho, rcond=None)[0]
        b = np.zeros(4)
        if N == 1:
            s = -1.0 if x[0] < 0 else 1.0
            b[0] = np.sqrt(s * x[0]); b[1:] = s * x[1:] / b[0]
        else:
            if x[0] < 0:
                b[0] = np.sqrt(-x[0]); b[1] = np.sqrt(-x[2]) if x[2] < 0 else 0.0
            else:
                b[0] = np.sqrt(x[0]); b[1] = np.sqrt(x[2]) if x[2] > 0 else 0.0
            if x[1] < 0:
                b[0] = -b[0]
            if N == 3:
                b[2] = x[3] / b[0]
        return b

    def quad(b):
        return np.array([b[0] * b[0], b[0] * b[1], b[1] * b[1], b[0] * b[2], b[1] * b[2], b[2] * b[2], b[0] * b[3], b[1] * b[3], b[2] * b[3], b[3] * b[3]])

    best = None
    for N in (1, 2, 3):
        b = approx(N)
        for _ in range(5):
            J = np.stack([2 * L[:, 0] * b[0] + L[:, 1] * b[1] + L[:, 3] * b[2] + L[:, 6] * b[3], L[:, 1] * b[0] + 2 * L[:, 2] * b[1] + L[:, 4] * b[2] + L[:, 7] * b[3],
                          L[:, 3] * b[0] + L[:, 4] * b[1] + 2 * L[:, 5] * b[2] + L[:, 8] * b[3], L[:, 6] * b[0] + L[:, 7] * b[1] + L[:, 8] * b[2] + 2 * L[:, 9] * b[3]], 1)
            b = b + np.linalg.lstsq(J, rho - L @ quad(b), rcond=None)[0]
        ccs = sum(b[i] * v[i] for i in range(4))
        pcs = al @ ccs
        if pcs[0, 2] < 0:
            pcs = -pcs
        pc0, pw0 = pcs.mean(0), pws.mean(0)
        U, D, Vt = np.linalg.svd((pcs - pc0).T @ (pws - pw0))
        R = U @ Vt
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0
        pc = pws @ R.T + t
        err = np.mean(np.hypot(us[:, 0] - (uc + fu * pc[:, 0] / pc[:, 2]), us[:, 1] - (vc + fv * pc[:, 1] / pc[:, 2])))
        if best is None or err < best[2]:
            best = (R, t, err)
    return best


def test_jacobi_eig12_matches_lapack(oracle):
    rng = np.random.default_rng(0)
    for trial in range(5):
        A = rng.normal(size=(12, 12 if trial else 10)); A = A @ A.T        # trial 0: rank 10, a 2-dimensional null space like the 5-point M^T M
        w, V = oracle.jacobi_eig12(A)
        assert np.allclose(np.sort(w), np.linalg.eigvalsh(A), atol=1e-11 * np.abs(A).max())
        assert np.allclose(V @ np.diag(w) @ V.T, A, atol=1e-11 * np.abs(A).max()) and np.allclose(V.T @ V, np.eye(12), atol=1e-13)


@pytest.mark.parametrize("n", [6, 8, 20, 60])
def test_epnp_matches_numpy_restatement(oracle, synth, n):
    for seed in range(4):
        xyz, uv, Rt, tt = _problem(synth, n, 10 * n + seed, sigma=0.5)
        R, t, err = oracle.epnp(xyz, uv)
        Rn, tn, en = _epnp_numpy(xyz, uv)
        assert err >= 0 and abs(err - en) < 1e-6 * max(1.0, en)
        assert np.allclose(R, Rn, atol=1e-7) and np.allclose(t, tn, atol=1e-6)


@pytest.mark.parametrize("n", [5, 6, 12, 50])
def test_epnp_exact_on_noise_free_sets(oracle, synth, n):
    for seed in range(6):
        xyz, uv, Rt, tt = _problem(synth, n, 100 * n + seed)
        R, t, err = oracle.epnp(xyz, uv)
        # inputs are f32 pixels (and f32 normalised coordinates inside solvePnP): exact up to that rounding
        assert 0 <= err < 2e-4, err
        assert np.allclose(R, Rt, atol=2e-6) and np.allclose(t, tt, atol=2e-4) and abs(np.linalg.det(R) - 1) < 1e-12


def test_epnp_agrees_with_least_squares_optimum(oracle, synth):
    for seed in range(4):
        xyz, uv, Rt, tt = _problem(synth, 80, 500 + seed, sigma=0.5)
        R, t, err = oracle.epnp(xyz, uv)
        T0 = np.concatenate([synth.quat_from_R(R), t])
        Tls, _, _, st = oracle.pnp_motion_only(xyz, uv, T0, iters=10, huber_delta=1e300)
        d = oracle.se3_log(oracle.se3_mul(T0, oracle.se3_inv(Tls)))
        assert np.linalg.norm(d[3:]) < 3e-3 and np.linalg.norm(d[:3]) < 0.1   # EPnP is an algebraic estimate: close to, not at, the optimum
        assert st["chi2_iter"][-1] <= st["chi2_init"]


def test_ransac_needs_no_pose_guess(oracle, synth):
    p = synth.pnp_problem(M=300, seed=21, outlier_frac=0.3)
    a = oracle.pnp_ransac(p["xyz"], p["uv"])
    b = oracle.pnp_ransac(p["xyz"], p["uv"], synth.perturb_pose(p["T_true"], np.random.default_rng(0), 0.5))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    assert (a[1].astype(bool) == ~p["outlier"]).mean() > 0.97


def test_ransac_five_points_is_the_single_model(oracle, synth):
    xyz, uv, Rt, tt = _problem(synth, 5, 7)
    T, inl, n, iters = oracle.pnp_ransac(xyz, uv)
    assert n == 5 and inl.all() and iters == 0                  # ptsetreg.cpp: count == modelPoints
    assert np.allclose(synth.R_from_quat(T[:4]), Rt, atol=1e-5) and np.allclose(T[4:], tt, atol=1e-3)
