"""Rows A10 / A11 on their own: the residual and the Jacobians of PoseOnlyEdgeProjection (optimization.cpp:75-101) and EdgeProjection (:41-73)
as the DEVICE FUNCTIONS of the LM kernels evaluate them (vslam_edge_jacobians runs cam_norm / eval_obs / jac_norm / jac_point_norm, the code
every window kernel linearises with), against (a) oracle/lm.c's line-by-line restatement of the reference and (b) central differences of the
GPU's own residual -- so these two rows no longer rest on end-to-end chi2 agreement only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases(oracle, synth, n, seed):
    rng = np.random.default_rng(seed)
    T = oracle.se3_exp(rng.normal(0, 0.3, 6))
    pc = np.stack([rng.normal(0, 4, n), rng.normal(0, 1.5, n), rng.uniform(4, 60, n)], 1)           # camera frame, in front of the camera
    pw = np.array([oracle.se3_act(oracle.se3_inv(T), p) for p in pc]).astype(np.float32)               # f32 at rest, like cv::Point3f
    uv = np.stack([synth.FX * pc[:, 0] / pc[:, 2] + synth.CX, synth.FY * pc[:, 1] / pc[:, 2] + synth.CY], 1) + rng.normal(0, 3, (n, 2))
    uv[::7] += rng.uniform(-40, 40, (len(uv[::7]), 2))                                                 # some beyond the Huber threshold
    return T, pw, uv.astype(np.float32)


@pytest.mark.parametrize("n,seed", [(257, 1), (64, 2)])
def test_device_residual_and_jacobians_match_the_oracle(vo, oracle, synth, n, seed):
    T, pw, uv = _cases(oracle, synth, n, seed)
    g = vo.edge_jacobians(pw, uv, T)
    delta = vo.params.huber_delta if hasattr(vo, "params") else 5.991
    for i in range(n):
        e, Jp, Jl = oracle.projection_residual(T, pw[i].astype(np.float64), uv[i].astype(np.float64), synth.K4)
        e2, Jp2 = oracle.pose_only_residual(T, pw[i].astype(np.float64), uv[i].astype(np.float64), synth.K4)
        assert np.allclose(g["err"][i], e, rtol=1e-10, atol=1e-9)
        assert np.allclose(g["J_pose"][i], Jp, rtol=1e-10, atol=1e-9) and np.allclose(g["J_pose"][i], Jp2, rtol=1e-10, atol=1e-9)
        assert np.allclose(g["J_point"][i], Jl, rtol=1e-10, atol=1e-9)
        chi = float(e @ e)
        assert np.isclose(g["chi2"][i], chi, rtol=1e-10)
        w = 1.0 if chi <= delta * delta else delta / np.sqrt(chi)                                      # g2o RobustKernelHuber: rho'(chi)
        assert np.isclose(g["huber_w"][i], w, rtol=1e-10)
    assert (g["huber_w"] < 1).any() and (g["huber_w"] == 1).any()


def test_device_jacobians_match_central_differences_of_the_device_residual(vo, oracle, synth):
    n = 96
    T, pw, uv = _cases(oracle, synth, n, 3)
    g = vo.edge_jacobians(pw, uv, T)
    h = 1e-5
    for a in range(6):   # left perturbation T <- exp(d) T, tangent [translation; rotation] (optimization.cpp:26-32)
        d = np.zeros(6); d[a] = h
        ep = vo.edge_jacobians(pw, uv, oracle.se3_mul(oracle.se3_exp(d), T))["err"]
        em = vo.edge_jacobians(pw, uv, oracle.se3_mul(oracle.se3_exp(-d), T))["err"]
        assert np.allclose((ep - em) / (2 * h), g["J_pose"][:, :, a], rtol=2e-5, atol=2e-4), a
    # the landmark is f32 at rest: step on a grid the f32 representation carries exactly
    hp = np.float32(2.0 ** -8)
    for a in range(3):
        d = np.zeros(3, np.float32); d[a] = hp
        base = (pw + d) - d                                                                               # (rounded so that +-d are exact)
        ep = vo.edge_jacobians(base + d, uv, T)["err"]; em = vo.edge_jacobians(base - d, uv, T)["err"]
        g0 = vo.edge_jacobians(base, uv, T)
        assert np.allclose((ep - em) / (2 * float(hp)), g0["J_point"][:, :, a], rtol=2e-4, atol=2e-3), a
