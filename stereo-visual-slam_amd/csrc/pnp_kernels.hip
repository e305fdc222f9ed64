// pnp_kernels.hip -- the minimal solver and the scoring of the RANSAC pose stage.
//
// Replaces what cv::solvePnPRansac(pts3d, pts2d, K, Mat(), rvec, tvec, false, 100, 4.0, 0.99, inliers) does per hypothesis
// (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:277): OpenCV 3.2 solves every 5-point subset with EPnP
// (solvepnp.cpp: SOLVEPNP_EPNP for more than four points; epnp.cpp = Lepetit / Moreno-Noguer / Fua 2009) and counts the points
// whose f32 squared reprojection error is <= (float)(4 * 4) (PnPRansacCallback::computeError, ptsetreg.cpp findInliers).
//
// gfx950 mapping: the hypotheses do not depend on each other, so ALL of them are solved at once, in three launches that each give the
// work the shape it has (one wave per hypothesis with lane 0 doing the sequential algebra -- the first version -- left 63 lanes idle for
// two thirds of its ~100 k wave-instructions: 8.8 ms for the 25.5 k hypotheses of a 256-keyframe batch):
//   epnp_front_kernel   ONE LANE PER HYPOTHESIS: control points (3 x 3 Jacobi), barycentric coordinates, the rows of M and M^T M
//                       (144 sums of 10 products); hands M^T M and the per-hypothesis constants on through a structure-of-arrays
//                       workspace (hypothesis index fastest: coalesced for lane-per-hypothesis kernels);
//   epnp_jacobi_kernel  the 12 x 12 symmetric eigenproblem, the only O(n^3) piece, EIGHT HYPOTHESES PER WAVE in LDS: cyclic Jacobi with a
//                       round-robin ordering, 6 disjoint rotations per hypothesis and round; a lane owns one rotation (48 of 64 lanes) and applies
//                       it to its two rows (J^T A), then to its two columns of A and V ((.) J, V J): two barriers per round;
//   epnp_back_kernel    ONE LANE PER HYPOTHESIS again: L (6 x 10), rho, and for N = 1, 2, 3 the betas (Householder least squares), five
//                       Gauss-Newton steps, absolute orientation (3 x 3 Jacobi) and the reprojection error; best of the three.
// Every floating-point operation is performed in the same order as the CPU oracle's restatement, with IEEE division / sqrt and
// no FMA contraction (this file is compiled with -ffp-contract=off): the hypothesis poses agree to the bit, so the inlier masks
// (integers) can be compared exactly.  The basis of the 2-dimensional null space of a 5-point system is a property of the
// eigen-solver (OpenCV's SVD would return another one); see DESIGN.md.
#include "vslam_internal.h"

namespace vslam {

constexpr int kSweeps12 = 10, kSweeps3 = 8;

__device__ inline void jacobi_cs(double app, double aqq, double apq, double& c, double& s) {
    if (apq == 0.0) { c = 1.0; s = 0.0; return; }
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    c = 1.0 / sqrt(t * t + 1.0);
    s = t * c;
}

__device__ inline void rr_pair(int round, int k, int& p, int& q) {
    int a, b;
    if (k == 0) { a = 11; b = round; }
    else { a = (round + k) % 11; b = (round - k + 11) % 11; }
    p = a < b ? a : b; q = a < b ? b : a;
}

// symmetric 3 x 3, sequential cyclic Jacobi (one lane)
__device__ inline void jacobi_eig3(double* A, double* V) {
    for (int i = 0; i < 9; ++i) V[i] = (i / 3 == i % 3) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < kSweeps3; ++sweep)
        for (int r = 0; r < 3; ++r) {
            const int p = r == 2 ? 1 : 0, q = r == 0 ? 1 : 2;
            double c, s;
            jacobi_cs(A[p * 3 + p], A[q * 3 + q], A[p * 3 + q], c, s);
            for (int j = 0; j < 3; ++j) {
                const double ap = A[p * 3 + j], aq = A[q * 3 + j];
                A[p * 3 + j] = c * ap - s * aq; A[q * 3 + j] = s * ap + c * aq;
            }
            for (int i = 0; i < 3; ++i) {
                const double ap = A[i * 3 + p], aq = A[i * 3 + q];
                A[i * 3 + p] = c * ap - s * aq; A[i * 3 + q] = s * ap + c * aq;
                const double vp = V[i * 3 + p], vq = V[i * 3 + q];
                V[i * 3 + p] = c * vp - s * vq; V[i * 3 + q] = s * vp + c * vq;
            }
        }
}

// least squares of an m x n system (m = 6, n <= 5), Householder QR, one lane
__device__ inline void qr_solve(double* A, double* b, int m, int n, double* x) {
    for (int k = 0; k < n; ++k) {
        double norm2 = 0;
        for (int i = k; i < m; ++i) norm2 += A[i * n + k] * A[i * n + k];
        const double norm = sqrt(norm2);
        if (norm == 0.0) continue;
        const double alpha = A[k * n + k] > 0 ? -norm : norm;
        A[k * n + k] -= alpha;
        double vtv = 0;
        for (int i = k; i < m; ++i) vtv += A[i * n + k] * A[i * n + k];
        if (vtv != 0.0) {
            for (int j = k + 1; j < n; ++j) {
                double dot = 0;
                for (int i = k; i < m; ++i) dot += A[i * n + k] * A[i * n + j];
                const double f = 2.0 * dot / vtv;
                for (int i = k; i < m; ++i) A[i * n + j] -= f * A[i * n + k];
            }
            double dot = 0;
            for (int i = k; i < m; ++i) dot += A[i * n + k] * b[i];
            const double f = 2.0 * dot / vtv;
            for (int i = k; i < m; ++i) b[i] -= f * A[i * n + k];
        }
        A[k * n + k] = alpha;
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = b[k];
        for (int j = k + 1; j < n; ++j) s -= A[k * n + j] * x[j];
        x[k] = A[k * n + k] != 0.0 ? s / A[k * n + k] : 0.0;
    }
}

__device__ inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ inline double dist2(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

constexpr int kMp = 5; // model points of solvePnPRansac for n > 4

// per-hypothesis state of the lane-per-hypothesis kernels (private memory)
struct EpnpLane {
    double pws[kMp * 3], us[kMp * 2], alphas[kMp * 4], pcs[kMp * 3];
    double cws[4][3], ccs[4][3];
    double v[4][12];
    double L[60], rho[6];
};
// workspace between the three launches, structure of arrays: field f of hypothesis h at ws[f * Hs + h] (Hs = H rounded up to 64)
constexpr int kWsMtM = 0, kWsPws = 144, kWsUs = kWsPws + 15, kWsAlphas = kWsUs + 10, kWsCws = kWsAlphas + 20, kWsV = kWsCws + 12, kWsFields = kWsV + 48;
constexpr int kJacG = 8; // hypotheses per wave in epnp_jacobi_kernel: 8 x 144 elements = 18 per lane, 48 of 64 lanes own a rotation

__device__ inline void find_betas(int N, const double* L, const double* rho, double* betas) {
    const int nc = N == 1 ? 4 : (N == 2 ? 3 : 5);
    double A[30], b[6], x[5];
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < nc; ++j) {
            const int col = N == 1 ? (j == 0 ? 0 : (j == 1 ? 1 : (j == 2 ? 3 : 6))) : j;
            A[i * nc + j] = L[10 * i + col];
        }
        b[i] = rho[i];
    }
    qr_solve(A, b, 6, nc, x);
    if (N == 1) {
        if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = -x[1] / betas[0]; betas[2] = -x[2] / betas[0]; betas[3] = -x[3] / betas[0]; }
        else { betas[0] = sqrt(x[0]); betas[1] = x[1] / betas[0]; betas[2] = x[2] / betas[0]; betas[3] = x[3] / betas[0]; }
    } else {
        if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
        else { betas[0] = sqrt(x[0]); betas[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) betas[0] = -betas[0];
        betas[2] = N == 3 ? x[3] / betas[0] : 0.0;
        betas[3] = 0.0;
    }
}

__device__ inline void gauss_newton(const double* L, const double* rho, double* betas) {
    for (int it = 0; it < 5; ++it) {
        double A[24], b[6], x[4];
        for (int i = 0; i < 6; ++i) {
            const double* r = L + 10 * i;
            A[i * 4 + 0] = 2 * r[0] * betas[0] + r[1] * betas[1] + r[3] * betas[2] + r[6] * betas[3];
            A[i * 4 + 1] = r[1] * betas[0] + 2 * r[2] * betas[1] + r[4] * betas[2] + r[7] * betas[3];
            A[i * 4 + 2] = r[3] * betas[0] + r[4] * betas[1] + 2 * r[5] * betas[2] + r[8] * betas[3];
            A[i * 4 + 3] = r[6] * betas[0] + r[7] * betas[1] + r[8] * betas[2] + 2 * r[9] * betas[3];
            b[i] = rho[i] - (r[0] * betas[0] * betas[0] + r[1] * betas[0] * betas[1] + r[2] * betas[1] * betas[1] + r[3] * betas[0] * betas[2] +
                             r[4] * betas[1] * betas[2] + r[5] * betas[2] * betas[2] + r[6] * betas[0] * betas[3] + r[7] * betas[1] * betas[3] +
                             r[8] * betas[2] * betas[3] + r[9] * betas[3] * betas[3]);
        }
        qr_solve(A, b, 6, 4, x);
        for (int i = 0; i < 4; ++i) betas[i] += x[i];
    }
}

__device__ inline void estimate_R_and_t(const EpnpLane& e, double fu, double fv, double uc, double vc, double R[9], double t[3]) {
    const int n = kMp;
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) { pc0[j] += e.pcs[3 * i + j]; pw0[j] += e.pws[3 * i + j]; }
    for (int j = 0; j < 3; ++j) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) abt[3 * j + k] += (e.pcs[3 * i + j] - pc0[j]) * (e.pws[3 * i + k] - pw0[k]);
    double S[9], V[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a * 3 + b] = abt[a] * abt[b] + abt[3 + a] * abt[3 + b] + abt[6 + a] * abt[6 + b];
    jacobi_eig3(S, V);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2 - a; ++b) if (S[ord[b + 1] * 4] > S[ord[b] * 4]) { const int tt = ord[b]; ord[b] = ord[b + 1]; ord[b + 1] = tt; }
    double Vs[9], U[9];
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Vs[r * 3 + c] = V[r * 3 + ord[c]];
    Vs[2] = Vs[3] * Vs[7] - Vs[6] * Vs[4]; Vs[5] = Vs[6] * Vs[1] - Vs[0] * Vs[7]; Vs[8] = Vs[0] * Vs[4] - Vs[3] * Vs[1];
    for (int c = 0; c < 2; ++c) {
        double u[3], nrm = 0;
        for (int r = 0; r < 3; ++r) { u[r] = abt[r * 3] * Vs[c] + abt[r * 3 + 1] * Vs[3 + c] + abt[r * 3 + 2] * Vs[6 + c]; nrm += u[r] * u[r]; }
        nrm = sqrt(nrm);
        for (int r = 0; r < 3; ++r) U[r * 3 + c] = nrm > 0 ? u[r] / nrm : (r == c ? 1.0 : 0.0);
    }
    U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] = U[i * 3] * Vs[j * 3] + U[i * 3 + 1] * Vs[j * 3 + 1] + U[i * 3 + 2] * Vs[j * 3 + 2];
    for (int i = 0; i < 3; ++i) t[i] = pc0[i] - dot3(R + 3 * i, pw0);
}

__device__ inline double compute_R_and_t(EpnpLane& e, const double* betas, double fu, double fv, double uc, double vc, double R[9], double t[3]) {
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) e.ccs[j][k] = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 3; ++k) e.ccs[j][k] += betas[i] * e.v[i][3 * j + k];
    for (int i = 0; i < kMp; ++i) {
        const double* a = e.alphas + 4 * i;
        for (int j = 0; j < 3; ++j) e.pcs[3 * i + j] = a[0] * e.ccs[0][j] + a[1] * e.ccs[1][j] + a[2] * e.ccs[2][j] + a[3] * e.ccs[3][j];
    }
    if (e.pcs[2] < 0.0) {
        for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) e.ccs[j][k] = -e.ccs[j][k];
        for (int i = 0; i < 3 * kMp; ++i) e.pcs[i] = -e.pcs[i];
    }
    estimate_R_and_t(e, fu, fv, uc, vc, R, t);
    double sum = 0;
    for (int i = 0; i < kMp; ++i) {
        const double* pw = e.pws + 3 * i;
        const double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        const double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
        const double du = e.us[2 * i] - ue, dv = e.us[2 * i + 1] - ve;
        sum += sqrt(du * du + dv * dv);
    }
    return sum / kMp;
}

__device__ inline void rotmat_to_quat(const double R[9], double q[4]) {
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) { const double s = sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; }
    else if (R[0] > R[4] && R[0] > R[8]) { const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; }
    else if (R[4] > R[8]) { const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s; }
    else { const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double sgn = q[3] < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 4; ++i) q[i] = sgn * q[i] / n;
}

// hx / hu hold the 5 points of every subset (H x 5 x 3 f32, H x 5 x 2 f32).  Batched form: problem h / h_per draws only nh_of[.] of its h_per
// hypothesis slots (0: fewer than 5 points, 1: exactly 5); the other slots get ok = 0 and are skipped by the later launches.
__global__ __launch_bounds__(64) void epnp_front_kernel(const float* __restrict__ hx, const float* __restrict__ hu, int H, int Hs, double fu, double fv, double uc,
                                                       double vc, double* __restrict__ ws, int32_t* __restrict__ ok_out, const int32_t* __restrict__ nh_of,
                                                       int h_per) {
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H) return;
    if (nh_of && (h % h_per) >= nh_of[h / h_per]) { ok_out[h] = 0; return; }
    const float* xyz = hx + (size_t)h * kMp * 3;
    const float* uv = hu + (size_t)h * kMp * 2;
    double pws[kMp * 3], us[kMp * 2], alphas[kMp * 4], cws[4][3];
    double m[kMp][2][12]; // rows of M
    int ok = 1;
    for (int i = 0; i < kMp; ++i) {
        for (int j = 0; j < 3; ++j) pws[3 * i + j] = (double)xyz[3 * i + j];
        const float xn = (float)(((double)uv[2 * i] - uc) * (1.0 / fu)), yn = (float)(((double)uv[2 * i + 1] - vc) * (1.0 / fv));
        us[2 * i] = (double)xn * fu + uc;
        us[2 * i + 1] = (double)yn * fv + vc;
    }
    // choose_control_points
    cws[0][0] = cws[0][1] = cws[0][2] = 0;
    for (int i = 0; i < kMp; ++i) for (int j = 0; j < 3; ++j) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; ++j) cws[0][j] /= kMp;
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, V3[9];
    for (int i = 0; i < kMp; ++i) {
        double d[3];
        for (int j = 0; j < 3; ++j) d[j] = pws[3 * i + j] - cws[0][j];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[a * 3 + b] += d[a] * d[b];
    }
    jacobi_eig3(C, V3);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2 - a; ++b) if (C[ord[b + 1] * 4] > C[ord[b] * 4]) { const int tt = ord[b]; ord[b] = ord[b + 1]; ord[b + 1] = tt; }
    for (int i = 1; i < 4; ++i) {
        const int col = ord[i - 1];
        const double ev = C[col * 4] > 0 ? C[col * 4] : 0.0;
        const double k = sqrt(ev / kMp);
        int big = 0; // sign convention of the principal directions: largest-magnitude component positive (first on ties)
        for (int j = 1; j < 3; ++j) if (fabs(V3[j * 3 + col]) > fabs(V3[big * 3 + col])) big = j;
        const double sg = V3[big * 3 + col] < 0 ? -1.0 : 1.0;
        for (int j = 0; j < 3; ++j) cws[i][j] = cws[0][j] + k * (sg * V3[j * 3 + col]);
    }
    // compute_barycentric_coordinates
    double cc[9], ci[9];
    for (int i = 0; i < 3; ++i) for (int j = 1; j < 4; ++j) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
    const double c00 = cc[4] * cc[8] - cc[5] * cc[7], c01 = cc[5] * cc[6] - cc[3] * cc[8], c02 = cc[3] * cc[7] - cc[4] * cc[6];
    const double det = cc[0] * c00 + cc[1] * c01 + cc[2] * c02;
    if (det == 0.0 || !isfinite(det)) ok = 0;
    ok_out[h] = ok; // (the back-end clears it again when the pose is not finite)
    if (!ok) return;
    const double id = 1.0 / det;
    ci[0] = c00 * id; ci[1] = (cc[2] * cc[7] - cc[1] * cc[8]) * id; ci[2] = (cc[1] * cc[5] - cc[2] * cc[4]) * id;
    ci[3] = c01 * id; ci[4] = (cc[0] * cc[8] - cc[2] * cc[6]) * id; ci[5] = (cc[2] * cc[3] - cc[0] * cc[5]) * id;
    ci[6] = c02 * id; ci[7] = (cc[1] * cc[6] - cc[0] * cc[7]) * id; ci[8] = (cc[0] * cc[4] - cc[1] * cc[3]) * id;
    for (int i = 0; i < kMp; ++i) {
        const double* pi = pws + 3 * i;
        double* a = alphas + 4 * i;
        for (int j = 0; j < 3; ++j)
            a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
        for (int j = 0; j < 4; ++j) { // fill_M
            m[i][0][3 * j] = a[j] * fu; m[i][0][3 * j + 1] = 0.0; m[i][0][3 * j + 2] = a[j] * (uc - us[2 * i]);
            m[i][1][3 * j] = 0.0; m[i][1][3 * j + 1] = a[j] * fv; m[i][1][3 * j + 2] = a[j] * (vc - us[2 * i + 1]);
        }
    }
    double* w = ws + h;
    // M^T M: per element the sum over the points in their order
    for (int a = 0; a < 12; ++a)
        for (int b = 0; b < 12; ++b) {
            double acc = 0;
            for (int i = 0; i < kMp; ++i) acc += m[i][0][a] * m[i][0][b] + m[i][1][a] * m[i][1][b];
            w[(size_t)(kWsMtM + a * 12 + b) * Hs] = acc;
        }
    for (int i = 0; i < 15; ++i) w[(size_t)(kWsPws + i) * Hs] = pws[i];
    for (int i = 0; i < 10; ++i) w[(size_t)(kWsUs + i) * Hs] = us[i];
    for (int i = 0; i < 20; ++i) w[(size_t)(kWsAlphas + i) * Hs] = alphas[i];
    for (int i = 0; i < 12; ++i) w[(size_t)(kWsCws + i) * Hs] = cws[i / 3][i % 3];
}

// one wave = kJacG hypotheses; lane 6 g + k owns rotation k of hypothesis g for the round (48 of the 64 lanes).  A round: the owner computes (c, s)
// from its pair's rows, then B = J^T A on ITS two rows (nobody else touches them: no exchange, (c, s) stay in registers); barrier; A = B J and
// V = V J on ITS two columns; barrier.  Each element sees the operations of the sequential solver in its order (an element is touched by
// exactly one rotation per side and round): B[p][j] = c A[p][j] - s A[q][j], B[q][j] = c A[q][j] + s A[p][j], likewise for the columns.
// (First version of this kernel: all 64 lanes, 18 flat elements each, the rotations handed over through LDS tables -- 324 LDS instructions and five
// barriers per round against 120 and two here; 1.44 ms per 25.5 k hypotheses.)
__global__ __launch_bounds__(64) void epnp_jacobi_kernel(int H, int Hs, double* __restrict__ ws, const int32_t* __restrict__ ok) {
    __shared__ __attribute__((aligned(16))) double A[kJacG * 144];
    __shared__ __attribute__((aligned(16))) double V[kJacG * 144];
    const int lane = threadIdx.x, h0 = blockIdx.x * kJacG;
    constexpr int kPer = kJacG * 144 / 64; // 18
    static_assert(kJacG * 144 % 64 == 0 && 6 * kJacG <= 64, "elements per lane / one lane per rotation");
    int any = 0;
    if (lane < kJacG && h0 + lane < H) any = ok[h0 + lane];
    if (__ballot(any != 0) == 0) return; // (uniform) nothing to solve in this group
#pragma unroll
    for (int u = 0; u < kPer; ++u) { // flat = lane + 64 u -> hypothesis flat / 144, element flat % 144
        const int flat = lane + 64 * u, g = flat / 144, e = flat - 144 * g;
        const int h = min(h0 + g, H - 1);
        A[flat] = ws[(size_t)(kWsMtM + e) * Hs + h];
        V[flat] = (e / 12) == (e % 12) ? 1.0 : 0.0;
    }
    __syncthreads();
    const bool own = lane < 6 * kJacG;
    const int g = own ? lane / 6 : 0, k = lane - 6 * (lane / 6);
    double* Ag = A + 144 * g;
    double* Vg = V + 144 * g;
    for (int sweep = 0; sweep < kSweeps12; ++sweep)
        for (int round = 0; round < 11; ++round) {
            int p, q; double c = 1.0, s = 0.0;
            rr_pair(round, k, p, q);
            if (own) {
                jacobi_cs(Ag[p * 12 + p], Ag[q * 12 + q], Ag[p * 12 + q], c, s);
                double rp[12], rq[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) { rp[j] = Ag[p * 12 + j]; rq[j] = Ag[q * 12 + j]; }
#pragma unroll
                for (int j = 0; j < 12; ++j) { Ag[p * 12 + j] = c * rp[j] - s * rq[j]; Ag[q * 12 + j] = c * rq[j] + s * rp[j]; }
            }
            __syncthreads();
            if (own) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const double bp = Ag[i * 12 + p], bq = Ag[i * 12 + q], vp = Vg[i * 12 + p], vq = Vg[i * 12 + q];
                    Ag[i * 12 + p] = c * bp - s * bq; Ag[i * 12 + q] = c * bq + s * bp;
                    Vg[i * 12 + p] = c * vp - s * vq; Vg[i * 12 + q] = c * vq + s * vp;
                }
            }
            __syncthreads();
        }
    // the four eigenvectors of the smallest eigenvalues, in ascending order (stable selection)
    if (lane < kJacG && h0 + lane < H) {
        const double* Al = A + 144 * lane;
        const double* Vl = V + 144 * lane;
        int ord[12];
        for (int i = 0; i < 12; ++i) ord[i] = i;
        for (int a = 0; a < 4; ++a) {
            int best = a;
            for (int b = a + 1; b < 12; ++b) if (Al[ord[b] * 13] < Al[ord[best] * 13]) best = b;
            const int tmp = ord[best];
            for (int b = best; b > a; --b) ord[b] = ord[b - 1];
            ord[a] = tmp;
        }
        double* w = ws + h0 + lane;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 12; ++r) w[(size_t)(kWsV + 12 * i + r) * Hs] = Vl[r * 12 + ord[i]];
    }
}

__global__ __launch_bounds__(64) void epnp_back_kernel(int H, int Hs, double fu, double fv, double uc, double vc, const double* __restrict__ ws,
                                                      double* __restrict__ Rt /* H x 12: R row-major, t */, double* __restrict__ T /* H x 7 */,
                                                      int32_t* __restrict__ ok_out) {
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H || !ok_out[h]) return;
    EpnpLane e;
    const double* w = ws + h;
    for (int i = 0; i < 15; ++i) e.pws[i] = w[(size_t)(kWsPws + i) * Hs];
    for (int i = 0; i < 10; ++i) e.us[i] = w[(size_t)(kWsUs + i) * Hs];
    for (int i = 0; i < 20; ++i) e.alphas[i] = w[(size_t)(kWsAlphas + i) * Hs];
    for (int i = 0; i < 12; ++i) e.cws[i / 3][i % 3] = w[(size_t)(kWsCws + i) * Hs];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 12; ++r) e.v[i][r] = w[(size_t)(kWsV + 12 * i + r) * Hs];
    // compute_L_6x10, compute_rho
    {
        double dv[4][6][3];
        for (int i = 0; i < 4; ++i) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; ++j) {
                for (int c = 0; c < 3; ++c) dv[i][j][c] = e.v[i][3 * a + c] - e.v[i][3 * b + c];
                b++;
                if (b > 3) { a++; b = a + 1; }
            }
        }
        for (int i = 0; i < 6; ++i) {
            double* row = e.L + 10 * i;
            row[0] = dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0 * dot3(dv[0][i], dv[1][i]);
            row[2] = dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0 * dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0 * dot3(dv[1][i], dv[2][i]);
            row[5] = dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0 * dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0 * dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0 * dot3(dv[2][i], dv[3][i]);
            row[9] = dot3(dv[3][i], dv[3][i]);
        }
    }
    e.rho[0] = dist2(e.cws[0], e.cws[1]); e.rho[1] = dist2(e.cws[0], e.cws[2]); e.rho[2] = dist2(e.cws[0], e.cws[3]);
    e.rho[3] = dist2(e.cws[1], e.cws[2]); e.rho[4] = dist2(e.cws[1], e.cws[3]); e.rho[5] = dist2(e.cws[2], e.cws[3]);
    double best_err = -1, R[9], t[3];
    for (int N = 1; N <= 3; ++N) {
        double betas[4], Rn[9], tn[3];
        find_betas(N, e.L, e.rho, betas);
        gauss_newton(e.L, e.rho, betas);
        const double err = compute_R_and_t(e, betas, fu, fv, uc, vc, Rn, tn);
        if (N == 1 || err < best_err) {
            best_err = err;
            for (int i = 0; i < 9; ++i) R[i] = Rn[i];
            for (int i = 0; i < 3; ++i) t[i] = tn[i];
        }
    }
    bool fin = true;
    for (int i = 0; i < 9; ++i) fin = fin && isfinite(R[i]);
    for (int i = 0; i < 3; ++i) fin = fin && isfinite(t[i]);
    double q[4] = {0, 0, 0, 1};
    if (fin) rotmat_to_quat(R, q);
    for (int i = 0; i < 9; ++i) Rt[(size_t)h * 12 + i] = R[i];
    for (int i = 0; i < 3; ++i) Rt[(size_t)h * 12 + 9 + i] = t[i];
    for (int i = 0; i < 4; ++i) T[(size_t)h * 7 + i] = q[i];
    for (int i = 0; i < 3; ++i) T[(size_t)h * 7 + 4 + i] = t[i];
    ok_out[h] = fin ? 1 : 0;
}

size_t pnp_epnp_ws_bytes(int H) { return (size_t)kWsFields * (size_t)((H + 63) & ~63) * sizeof(double); }

// the three launches for H hypotheses (ws: pnp_epnp_ws_bytes(H) bytes)
static void launch_epnp_stages(const float* d_hx, const float* d_hu, int H, const double K[4], double* ws, double* d_Rt, double* d_T, int32_t* d_ok,
                               const int32_t* nh_of, int h_per, hipStream_t stream) {
    const int Hs = (H + 63) & ~63;
    hipLaunchKernelGGL(epnp_front_kernel, dim3(Hs / 64), dim3(64), 0, stream, d_hx, d_hu, H, Hs, K[0], K[1], K[2], K[3], ws, d_ok, nh_of, h_per);
    hipLaunchKernelGGL(epnp_jacobi_kernel, dim3((H + kJacG - 1) / kJacG), dim3(64), 0, stream, H, Hs, ws, (const int32_t*)d_ok);
    hipLaunchKernelGGL(epnp_back_kernel, dim3(Hs / 64), dim3(64), 0, stream, H, Hs, K[0], K[1], K[2], K[3], (const double*)ws, d_Rt, d_T, d_ok);
}

// PnPRansacCallback::computeError + findInliers: inliers of every hypothesis over the shared point set (block = hypothesis);
// mask (optional) receives the per-point flags of hypothesis `mask_hyp`
__global__ __launch_bounds__(256) void pnp_count_inliers_kernel(const float* __restrict__ xyz, const float* __restrict__ uv, int n, const double* __restrict__ Rt,
                                                               const int32_t* __restrict__ ok, double fx, double fy, double cx, double cy, float thr2,
                                                               int32_t* __restrict__ counts, int hyp0, uint8_t* __restrict__ mask) {
    const int hyp = hyp0 + blockIdx.x;
    __shared__ double sR[12];
    __shared__ int cnt;
    if (threadIdx.x < 12) sR[threadIdx.x] = Rt[(size_t)hyp * 12 + threadIdx.x];
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    if (ok[hyp])
        for (int i = threadIdx.x; i < n; i += 256) {
            const double X = xyz[3 * i], Y = xyz[3 * i + 1], Z = xyz[3 * i + 2];
            double x = sR[0] * X + sR[1] * Y + sR[2] * Z + sR[9];
            double y = sR[3] * X + sR[4] * Y + sR[5] * Z + sR[10];
            double z = sR[6] * X + sR[7] * Y + sR[8] * Z + sR[11];
            z = z ? 1. / z : 1;
            x *= z; y *= z;
            const float pu = (float)(x * fx + cx), pv = (float)(y * fy + cy);
            const float du = uv[2 * i] - pu, dv = uv[2 * i + 1] - pv;
            float err = 0.f;
            err += du * du;
            err += dv * dv;
            const bool in = err <= thr2;
            if (mask) mask[i] = in;
            mine += in;
        }
    else if (mask)
        for (int i = threadIdx.x; i < n; i += 256) mask[i] = 0;
    atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && counts) counts[hyp] = cnt;
}

int launch_pnp_epnp(const float* d_hx, const float* d_hu, int H, const double K[4], double* d_Rt, double* d_T, int32_t* d_ok, uint8_t* ws, hipStream_t stream) {
    if (H <= 0) return VSLAM_OK;
    ProfScope prof__(stream, "pnp_epnp_kernels", 3);
    launch_epnp_stages(d_hx, d_hu, H, K, (double*)ws, d_Rt, d_T, d_ok, nullptr, 1, stream);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

// ------------------------------------------------------------------------------------------- batched RANSAC pose (vslam_pnp_ransac_dev)
// cv::solvePnPRansac(..., useExtrinsicGuess = false, 100, 4.0, 0.99) of VO::motion_estimation (visual_odometry.cpp:277) for B independent
// problems whose points are already in device memory -- the reference's own pose stage in throughput mode.  Same restatement as the host
// tier (api.hip pnp_ransac_impl, oracle/ransac.c): every problem draws the SAME cv::RNG sequence (seed -1) of 5-point subsets modulo its
// own point count, all B x max_iters hypotheses are solved (EPnP, one wave each) and scored at once, and the sequential acceptance rule
// with its adaptive stopping (RANSACUpdateNumIters) is replayed per problem over the counts.  Returns the best RANSAC model itself (OpenCV
// 3.2.0, the reference's pinned version: the refined pose is discarded) and its inlier mask.
__device__ inline unsigned cv_rng_next_dev(unsigned long long& state) {
    state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
    return (unsigned)state;
}
// one lane per problem: the subset sequence (a chain of RNG draws: ~600 dependent multiply-adds) and the gather of its points
__global__ __launch_bounds__(64) void pnp_ransac_subsets_kernel(const float* __restrict__ xyz, const float* __restrict__ uv, const int32_t* __restrict__ d_n,
                                                               int capacity, int B, int H, float* __restrict__ hx, float* __restrict__ hu,
                                                               int32_t* __restrict__ nh_of) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int n = min(max(d_n[b], 0), capacity);
    const int nh = n < kMp ? 0 : (n == kMp ? 1 : H); // ptsetreg.cpp: count == modelPoints -> one model from all points
    nh_of[b] = nh;
    const float* px = xyz + 3 * (size_t)b * capacity;
    const float* pu = uv + 2 * (size_t)b * capacity;
    unsigned long long state = 0xFFFFFFFFFFFFFFFFULL;
    for (int it = 0; it < nh; ++it) {
        int idx[kMp];
#pragma unroll
        for (int i = 0; i < kMp; ++i) {
            if (nh == 1) { idx[i] = i; continue; }
            for (;;) {
                const int v = (int)(cv_rng_next_dev(state) % (unsigned)n);
                bool dup = false;
#pragma unroll
                for (int j = 0; j < kMp; ++j) if (j < i && idx[j] == v) dup = true;
                idx[i] = v;
                if (!dup) break;
            }
        }
        float* ox = hx + ((size_t)b * H + it) * kMp * 3;
        float* ou = hu + ((size_t)b * H + it) * kMp * 2;
#pragma unroll
        for (int i = 0; i < kMp; ++i) {
            ox[3 * i] = px[3 * idx[i]]; ox[3 * i + 1] = px[3 * idx[i] + 1]; ox[3 * i + 2] = px[3 * idx[i] + 2];
            ou[2 * i] = pu[2 * idx[i]]; ou[2 * i + 1] = pu[2 * idx[i] + 1];
        }
    }
}

// PnPRansacCallback::computeError of one point under one model: f64 projection with one reciprocal, f32 squared error (the arithmetic of
// pnp_count_inliers_kernel above, shared by the two batched kernels below)
__device__ inline bool ransac_point_is_inlier(const double* sR, const float* xyz, const float* uv, int i, double fx, double fy, double cx, double cy, float thr2) {
    const double X = xyz[3 * i], Y = xyz[3 * i + 1], Z = xyz[3 * i + 2];
    double x = sR[0] * X + sR[1] * Y + sR[2] * Z + sR[9];
    double y = sR[3] * X + sR[4] * Y + sR[5] * Z + sR[10];
    double z = sR[6] * X + sR[7] * Y + sR[8] * Z + sR[11];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    const float pu = (float)(x * fx + cx), pv = (float)(y * fy + cy);
    const float du = uv[2 * i] - pu, dv = uv[2 * i + 1] - pv;
    float err = 0.f;
    err += du * du;
    err += dv * dv;
    return err <= thr2;
}
// block (h, b): inliers of problem b's points under its hypothesis h
__global__ __launch_bounds__(128) void pnp_ransac_count_kernel(const float* __restrict__ xyz, const float* __restrict__ uv, const int32_t* __restrict__ d_n,
                                                              int capacity, int H, const double* __restrict__ Rt, const int32_t* __restrict__ ok,
                                                              double fx, double fy, double cx, double cy, float thr2, int32_t* __restrict__ counts) {
    const int h = blockIdx.x, b = blockIdx.y, hyp = b * H + h;
    __shared__ double sR[12];
    __shared__ int cnt;
    if (!ok[hyp]) { if (threadIdx.x == 0) counts[hyp] = 0; return; } // (uniform)
    if (threadIdx.x < 12) sR[threadIdx.x] = Rt[(size_t)hyp * 12 + threadIdx.x];
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int n = min(max(d_n[b], 0), capacity);
    const float* px = xyz + 3 * (size_t)b * capacity;
    const float* pu = uv + 2 * (size_t)b * capacity;
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += 128) mine += ransac_point_is_inlier(sR, px, pu, i, fx, fy, cx, cy, thr2);
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[hyp] = cnt;
}
// cv::RANSACUpdateNumIters (ptsetreg.cpp)
__device__ inline int ransac_update_num_iters_dev(double p, double ep, int model_points, int max_iters) {
    p = fmin(fmax(p, 0.), 1.); ep = fmin(fmax(ep, 0.), 1.);
    double num = fmax(1. - p, 2.2250738585072014e-308), denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}
// block per problem: thread 0 replays the sequential loop over the counts (strict improvement, adaptive iteration count), then all
// threads write the best model's inlier mask; pose = the best model (identity when no model was accepted)
__global__ __launch_bounds__(128) void pnp_ransac_select_kernel(const float* __restrict__ xyz, const float* __restrict__ uv, const int32_t* __restrict__ d_n,
                                                               int capacity, int H, const double* __restrict__ Rt, const double* __restrict__ hT,
                                                               const int32_t* __restrict__ ok, const int32_t* __restrict__ counts,
                                                               const int32_t* __restrict__ nh_of, double confidence, double fx, double fy, double cx,
                                                               double cy, float thr2, double* __restrict__ T_out, uint8_t* __restrict__ inlier,
                                                               int32_t* __restrict__ n_inl, int32_t* __restrict__ iters_run) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ int s_best, s_good;
    __shared__ double sR[12];
    const int n = min(max(d_n[b], 0), capacity), nh = nh_of[b];
    if (tid == 0) {
        int best = -1, max_good = 0, it = 0;
        if (nh == 1) { if (ok[(size_t)b * H]) { best = 0; max_good = n; } }
        else {
            int niters = nh;
            for (it = 0; it < niters; ++it) {
                const int hyp = b * H + it;
                if (ok[hyp] && counts[hyp] > max(max_good, kMp - 1)) {
                    best = it; max_good = counts[hyp];
                    niters = ransac_update_num_iters_dev(confidence, (double)(n - max_good) / n, kMp, niters);
                }
            }
        }
        s_best = best; s_good = max_good;
        if (iters_run) iters_run[b] = it;
        if (n_inl) n_inl[b] = best >= 0 ? max_good : 0;
    }
    __syncthreads();
    const int best = s_best;
    uint8_t* m = inlier ? inlier + (size_t)b * capacity : nullptr;
    if (best < 0) {
        if (tid < 7) T_out[7 * (size_t)b + tid] = tid == 3 ? 1.0 : 0.0;
        if (m) for (int i = tid; i < capacity; i += 128) m[i] = 0;
        return;
    }
    const int hyp = b * H + best;
    if (tid < 7) T_out[7 * (size_t)b + tid] = hT[(size_t)hyp * 7 + tid];
    if (!m) return;
    if (tid < 12) sR[tid] = Rt[(size_t)hyp * 12 + tid];
    __syncthreads();
    const float* px = xyz + 3 * (size_t)b * capacity;
    const float* pu = uv + 2 * (size_t)b * capacity;
    for (int i = tid; i < capacity; i += 128)
        m[i] = i < n ? (nh == 1 ? (uint8_t)1 : (uint8_t)ransac_point_is_inlier(sR, px, pu, i, fx, fy, cx, cy, thr2)) : (uint8_t)0;
}

size_t pnp_ransac_scratch_bytes(int B, int H) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nh = (size_t)B * H;
    return al(nh * kMp * 3 * 4) + al(nh * kMp * 2 * 4) + al(nh * 12 * 8) + al(nh * 7 * 8) + 2 * al(nh * 4) + al((size_t)B * 4) + al(pnp_epnp_ws_bytes((int)nh));
}

int launch_pnp_ransac_batch(const float* d_xyz, const float* d_uv, const int32_t* d_n, int capacity, int B, int H, const double K[4], double reproj_err,
                            double confidence, uint8_t* scratch, double* d_T, uint8_t* d_inlier, int32_t* d_n_inl, int32_t* d_iters, hipStream_t stream) {
    if (B <= 0) return VSLAM_OK;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nh = (size_t)B * H;
    float* hx = (float*)scratch; scratch += al(nh * kMp * 3 * 4);
    float* hu = (float*)scratch; scratch += al(nh * kMp * 2 * 4);
    double* Rt = (double*)scratch; scratch += al(nh * 12 * 8);
    double* hT = (double*)scratch; scratch += al(nh * 7 * 8);
    int32_t* ok = (int32_t*)scratch; scratch += al(nh * 4);
    int32_t* cnt = (int32_t*)scratch; scratch += al(nh * 4);
    int32_t* nh_of = (int32_t*)scratch; scratch += al((size_t)B * 4);
    double* ws = (double*)scratch;
    const float thr2 = (float)(reproj_err * reproj_err);
    { ProfScope p(stream, "pnp_ransac_subsets_kernel");
      hipLaunchKernelGGL(pnp_ransac_subsets_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, d_xyz, d_uv, d_n, capacity, B, H, hx, hu, nh_of); }
    { ProfScope p(stream, "pnp_epnp_kernels", 3);
      launch_epnp_stages(hx, hu, (int)nh, K, ws, Rt, hT, ok, nh_of, H, stream); }
    { ProfScope p(stream, "pnp_ransac_count_kernel");
      hipLaunchKernelGGL(pnp_ransac_count_kernel, dim3(H, B), dim3(128), 0, stream, d_xyz, d_uv, d_n, capacity, H, Rt, ok, K[0], K[1], K[2], K[3], thr2, cnt); }
    { ProfScope p(stream, "pnp_ransac_select_kernel");
      hipLaunchKernelGGL(pnp_ransac_select_kernel, dim3(B), dim3(128), 0, stream, d_xyz, d_uv, d_n, capacity, H, Rt, hT, ok, cnt, nh_of, confidence, K[0], K[1], K[2],
                         K[3], thr2, d_T, d_inlier, d_n_inl, d_iters); }
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

int launch_pnp_count_inliers(const float* d_xyz, const float* d_uv, int n, const double* d_Rt, const int32_t* d_ok, int hyp0, int n_hyp, const double K[4],
                             double reproj_thr, int32_t* d_counts, uint8_t* d_mask, hipStream_t stream) {
    if (n_hyp <= 0) return VSLAM_OK;
    ProfScope prof__(stream, "pnp_count_inliers_kernel");
    hipLaunchKernelGGL(pnp_count_inliers_kernel, dim3(n_hyp), dim3(256), 0, stream, d_xyz, d_uv, n, d_Rt, d_ok, K[0], K[1], K[2], K[3],
                       (float)(reproj_thr * reproj_thr), d_counts, hyp0, d_mask);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

} // namespace vslam
