// lm_device.h -- device helpers shared by the Levenberg-Marquardt kernels (lm_kernels.hip: the general persistent window kernel, the wave kernels;
// ba_resident.hip: the LDS-resident window kernel): wave reductions, the residual / Jacobian factors of EdgeProjection and PoseOnlyEdgeProjection
// (/root/reference/src/stereo_visual_slam_main/optimization.cpp:41-101) in normalised image coordinates, the Huber kernel, small dense solves.
#pragma once
#include "vslam_internal.h"
#include "se3_device.h"

namespace vslam {

__device__ inline double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
// Sum N per-lane values over the 64 lanes of a wave with a halving butterfly: at every step a lane sends one half of its
// values to its partner and keeps (and accumulates) the other half, so N values cost about N shuffles instead of 6 N.
// On return v[0] holds the wave-wide sum of value wave_slot<N>(lane) (a fixed tree: deterministic).
// The two widest steps (partner lane ^ 32, lane ^ 16) are the gfx950 half-wave / row swaps: with X = a, Y = b,
// v_permlane32_swap leaves {keep, received} in {X, Y} of the lower lanes and {received, keep} in the upper ones, so the step is
// two swaps (one per dword) and the add -- no select, no LDS crossbar; same pairs, same sums as the generic step.
template <int M>
__device__ inline double swap_add(double a, double b) {
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    unsigned xl, yl, xh, yh;
    if constexpr (M == 32) {
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
        xl = lo[0]; yl = lo[1]; xh = hi[0]; yh = hi[1];
    } else {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
        xl = lo[0]; yl = lo[1]; xh = hi[0]; yh = hi[1];
    }
    return __builtin_bit_cast(double, (unsigned long long)xl | ((unsigned long long)xh << 32)) +
           __builtin_bit_cast(double, (unsigned long long)yl | ((unsigned long long)yh << 32));
}
template <int N>
__device__ inline void wave_reduce_scatter(double (&v)[N], int lane) {
    int n = N;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const int h = (n + 1) / 2;
        const bool upper = (lane & m) != 0;
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const double a = v[i], b = (i + h < n) ? v[i + h] : 0.0;
            if (m == 32) v[i] = swap_add<32>(a, b);
            else if (m == 16) v[i] = swap_add<16>(a, b);
            else {
                const double send = upper ? a : b, keep = upper ? b : a;
                v[i] = keep + __shfl_xor(send, m);
            }
        }
        n = h;
    }
}
// which of the N values lane `lane` ends up with (-1: none)
template <int N>
__device__ inline int wave_slot(int lane) {
    int sizes[7];
    sizes[0] = N;
    for (int k = 1; k <= 6; ++k) sizes[k] = (sizes[k - 1] + 1) / 2;
    int pos = 0;
    bool ok = true;
    for (int k = 6; k >= 1; --k) { // undo the steps, last first: step k used mask 64 >> k and half size sizes[k]
        if (lane & (64 >> k)) pos += sizes[k];
        if (pos >= sizes[k - 1]) ok = false;
    }
    return ok ? pos : -1;
}

// 1/x and 1/sqrt(x) to ~1 ulp without the IEEE division / sqrt sequences (v_div_scale, v_div_fmas, v_div_fixup and the sqrt
// rescaling): hardware seed + two Newton steps.  The LM outputs are tolerance-checked (1e-4), not bit-exact; 0, inf and NaN
// inputs still give non-finite results, which is all the failure checks below rely on.
__device__ inline double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ inline double rsqrt_nr(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = fma(r * 0.5, fma(-x * r, r, 1.0), r);
    r = fma(r * 0.5, fma(-x * r, r, 1.0), r);
    return r;
}

// A wave-uniform double moved into scalar registers: values read from LDS land in VGPRs even when every lane reads the same
// address; pinning the loop-invariant rotation of the current keyframe in SGPRs frees 2 VGPRs per value in the hot loops.
__device__ inline double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ inline double readlane_f64(double v, int l) { // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ inline void expand_pose(const double* T, double* Rt) {
    se3::rotmat(T, Rt);
    Rt[9] = T[4]; Rt[10] = T[5]; Rt[11] = T[6];
}

__device__ inline void huber(double e, double delta, double& rho, double& w) {
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho = e; w = 1.0; }
    else { const double r = rsqrt_nr(e), s = e * r; rho = 2 * s * delta - dsqr; w = delta * r; }
}

// The pose Jacobian has two structural zeros, A[1] = A[6] = 0.  Without fast-math the compiler must keep 0 * x (NaN / signed-zero
// semantics), so the hot loops spell the sparsity out: the helpers below are called with loop indices that are constants after
// unrolling, and the branches fold away.  (Dropping an exact 0 * finite term does not change any sum.)
// A[i] * x + A[6 + i] * y
__device__ inline double a_dot2(const double* A, int i, double x, double y) {
    if (i == 0) return A[0] * x;
    if (i == 1) return A[7] * y;
    return A[i] * x + A[6 + i] * y;
}
// acc + A[i] * x + A[6 + i] * y
__device__ inline double a_fma2(const double* A, int i, double x, double y, double acc) {
    if (i == 0) return fma(A[0], x, acc);
    if (i == 1) return fma(A[7], y, acc);
    return fma(x, A[i], fma(y, A[6 + i], acc));
}
// acc + U[r] * V[c] + U[6 + r] * V[6 + c] for two arrays with the pose Jacobian's zero pattern
__device__ inline double a_fma_pair(const double* U, int r, const double* V, int c, double acc) {
    if (r != 0 && c != 0) acc = fma(U[6 + r], V[6 + c], acc);
    if (r != 1 && c != 1) acc = fma(U[r], V[c], acc);
    return acc;
}
template <bool FAST = false>
__device__ inline void project_err(const double* Rt, const double* K, double px, double py, double pz, float u, float v, double& X,
                                   double& Y, double& Z, double& ex, double& ey, double* rz_out = nullptr) {
    X = Rt[0] * px + Rt[1] * py + Rt[2] * pz + Rt[9];
    Y = Rt[3] * px + Rt[4] * py + Rt[5] * pz + Rt[10];
    Z = Rt[6] * px + Rt[7] * py + Rt[8] * pz + Rt[11];
    const double rz = FAST ? rcp_nr(Z) : 1.0 / Z; // one reciprocal instead of two divisions (K*(T*p) / z, optimization.cpp:46-49); tolerance-checked
    ex = (double)u - (K[0] * X * rz + K[2]);
    ey = (double)v - (K[1] * Y * rz + K[3]);
    if (rz_out) *rz_out = rz;
}

// ---- linearisation in NORMALISED image coordinates.  With the camera-frame point (X, Y, Z): rho = 1/Z, x = X rho, y = Y rho.  The 2x6
// pose Jacobian of the reprojection error (optimization.cpp:52-72, :84-100) factors as  A = diag(fx, fy) At  with
//   At = [ -rho   0    x rho   x y      -(1 + x^2)   y ]
//        [  0    -rho  y rho   1 + y^2  -x y        -x ]
// (7 flops from x, y, rho instead of 18 for A), the landmark Jacobian as  B = diag(fx, fy) Bt,  Bt = At[:, 0:3] R, and the
// error as  e = diag(fx, fy) en,  en = (z - c) / f - (x, y).  Every block of the normal equations then carries the focal lengths
// only through the two per-edge weights  l0 = w fx^2, l1 = w fy^2  (w = Huber weight):
//   H_pp = At^T L At,  b_p = -At^T L en,  H_ll = Bt^T L Bt,  b_l = -Bt^T L en,  W = At^T L Bt      (L = diag(l0, l1)).
// EdgeProjection uses 1/(Z + 1e-18) (optimization.cpp:66), PoseOnlyEdgeProjection 1/Z (:96-100): the two differ by less than one
// ulp for every Z > 0.01 m and by a relative 1e-16 / Z below that -- one reciprocal serves both.
struct CamK { double fx, fy, ifx, ify, kx, ky, fx2, fy2; }; // kx = -cx / fx, ky = -cy / fy
__device__ inline CamK make_camk(const double* K) {
    CamK c;
    c.fx = K[0]; c.fy = K[1]; c.ifx = 1.0 / K[0]; c.ify = 1.0 / K[1]; c.kx = -K[2] * c.ifx; c.ky = -K[3] * c.ify; c.fx2 = K[0] * K[0]; c.fy2 = K[1] * K[1];
    return c;
}
__device__ inline void cam_norm(const double* Rt, double px, double py, double pz, double& x, double& y, double& rho) {
    const double X = Rt[0] * px + Rt[1] * py + Rt[2] * pz + Rt[9];
    const double Y = Rt[3] * px + Rt[4] * py + Rt[5] * pz + Rt[10];
    const double Z = Rt[6] * px + Rt[7] * py + Rt[8] * pz + Rt[11];
    rho = rcp_nr(Z);
    x = X * rho; y = Y * rho;
}
__device__ inline void jac_norm(double x, double y, double rho, double A[12]) {
    A[0] = -rho; A[1] = 0; A[2] = x * rho; A[3] = x * y; A[4] = -fma(x, x, 1.0); A[5] = y;
    A[6] = 0; A[7] = -rho; A[8] = y * rho; A[9] = fma(y, y, 1.0); A[10] = -A[3]; A[11] = -x;
}
// Bt = At[:, 0:3] R = rho (x R2 - R0 ; y R2 - R1)   (R0, R1, R2: rows of the rotation)
__device__ inline void jac_point_norm(double x, double y, double rho, const double* R, double B[6]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        B[c] = rho * fma(x, R[6 + c], -R[c]);
        B[3 + c] = rho * fma(y, R[6 + c], -R[3 + c]);
    }
}
// the same without the factor rho (the Schur passes fold it into their scalar weights: six multiplies less per observation)
__device__ inline void jac_point_unit(double x, double y, const double* R, double B[6]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        B[c] = fma(x, R[6 + c], -R[c]);
        B[3 + c] = fma(y, R[6 + c], -R[3 + c]);
    }
}
// evaluation of one observation: normalised error en, pixel chi2, robust rho, Huber weight
__device__ inline void eval_obs(const CamK& ck, double x, double y, float2 z, double delta, double& enx, double& eny, double& chi, double& rob, double& wgt) {
    enx = fma((double)z.x, ck.ifx, ck.kx) - x;
    eny = fma((double)z.y, ck.ify, ck.ky) - y;
    chi = ck.fx2 * enx * enx + ck.fy2 * eny * eny;
    huber(chi, delta, rob, wgt);
}

__device__ inline bool inv3_sym(double a, double b, double c, double d, double e, double f, double Di[6]) {
    // symmetric [[a b c],[b d e],[c e f]] -> unique entries of the inverse (00 01 02 11 12 22)
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = rcp_nr(det);
    Di[0] = c00 * id; Di[1] = c01 * id; Di[2] = c02 * id;
    Di[3] = (a * f - c * c) * id; Di[4] = (b * c - a * e) * id; Di[5] = (a * d - b * b) * id;
    return isfinite(id);
}

// 6x6 SPD solve by one thread (pose-only mode); returns false if not positive definite
__device__ inline bool chol6_solve(const double* H, double lambda, const double* b, double* x) {
    double L[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) L[i] = H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) L[7 * i] += lambda;
    bool ok = true;
    double rd[6]; // reciprocal diagonal of L
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = L[j * 6 + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0.0) || !isfinite(d)) ok = false;
        rd[j] = rsqrt_nr(d);
        L[j * 6 + j] = d * rd[j];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i > j) {
                double s = L[i * 6 + j];
#pragma unroll
                for (int k = 0; k < 6; ++k) if (k < j) s -= L[i * 6 + k] * L[j * 6 + k];
                L[i * 6 + j] = s * rd[j];
            }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < i) s -= L[i * 6 + k] * y[k];
        y[i] = s * rd[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k > i) s -= L[k * 6 + i] * x[k];
        x[i] = s * rd[i];
    }
    return ok;
}

} // namespace vslam
