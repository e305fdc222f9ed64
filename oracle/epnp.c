/* epnp.c -- CPU ORACLE (test infrastructure only; PARITY UNPINNED) for the minimal solver inside cv::solvePnPRansac.
 *
 * The reference calls cv::solvePnPRansac(pts3d, pts2d, K, Mat(), rvec, tvec, false, 100, 4.0, 0.99, inliers)
 * (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:277).  For more than four points OpenCV 3.2 solves every
 * 5-point RANSAC hypothesis with EPnP (solvepnp.cpp: model_points = 5, ransac_kernel_method = SOLVEPNP_EPNP; epnp.cpp is the
 * authors' implementation of Lepetit, Moreno-Noguer, Fua, "EPnP: An Accurate O(n) Solution to the PnP Problem", IJCV 2009).
 * OpenCV is not in this image; this file restates the published algorithm in the structure of epnp.cpp:
 *   choose_control_points      centroid + the three principal directions of the points, scaled by sqrt(eigenvalue / n)
 *   compute_barycentric_coordinates   alphas = CC^-1 (p - c0), alpha0 = 1 - sum
 *   fill_M, M^T M (12 x 12), its eigenvectors for the four smallest eigenvalues (epnp.cpp: cvSVD of M^T M, rows 11..8 of U^T)
 *   compute_L_6x10, compute_rho, find_betas_approx_1/2/3, gauss_newton (5 iterations, 6x4 least squares),
 *   compute_ccs / compute_pcs / solve_for_sign / estimate_R_and_t (SVD of the 3x3 correlation, det fix) / reprojection_error,
 *   best of N = 1, 2, 3 by mean reprojection error.
 * and the image-point round trip of solvePnP's EPNP branch (undistortPoints to f32 normalised coordinates, then u = x*fu + uc).
 *
 * What cannot be pinned: OpenCV's SVD.  With 5 points M is 10 x 12, so M^T M has a 2-dimensional null space and the basis a
 * particular SVD returns for it is arbitrary; the N = 2 / N = 3 candidates are basis-independent in exact arithmetic, the Gauss-
 * Newton start values are not.  Defined here: a cyclic Jacobi eigen-solver with a fixed round-robin ordering and a fixed number of
 * sweeps, eigenpairs sorted ascending with ties broken by index.  The HIP kernel (csrc/pnp_kernels.hip) performs the SAME
 * floating-point operations in the same order (no FMA contraction on either side, IEEE division and sqrt), so the two agree to
 * the bit and the RANSAC masks are compared exactly.  Least-squares solves use Householder QR (OpenCV: cvSolve(CV_SVD) for the
 * beta approximations, its own Householder QR for Gauss-Newton): same solutions up to rounding for full-rank systems.
 */
#include <math.h>
#include <string.h>

#include "vo_oracle.h"

#define EPNP_SWEEPS12 10
#define EPNP_SWEEPS3 8

/* rotation (c, s) that annihilates a_pq of a symmetric matrix (Rutishauser's formulas) */
static void jacobi_cs(double app, double aqq, double apq, double* c, double* s) {
    if (apq == 0.0) { *c = 1.0; *s = 0.0; return; }
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    *c = 1.0 / sqrt(t * t + 1.0);
    *s = t * (*c);
}

/* round-robin ("chess tournament") ordering for 12 indices: 11 rounds of 6 disjoint pairs */
static void rr_pair(int round, int k, int* p, int* q) {
    int a, b;
    if (k == 0) { a = 11; b = round; }
    else { a = (round + k) % 11; b = (round - k + 11) % 11; }
    *p = a < b ? a : b; *q = a < b ? b : a;
}

/* eigen-decomposition of a symmetric 12 x 12 matrix: A (row-major, destroyed: eigenvalues end on the diagonal), V = eigenvectors
 * in columns.  One round: the six rotations of the round are computed from the current A, then applied as  B = J^T A  (all rows),
 * A = B J (all columns), V = V J -- the order the GPU wave uses (three element-parallel phases). */
void vo_jacobi_eig12(double* A, double* V) {
    for (int i = 0; i < 144; ++i) V[i] = (i / 12 == i % 12) ? 1.0 : 0.0;
    double B[144];
    for (int sweep = 0; sweep < EPNP_SWEEPS12; ++sweep)
        for (int round = 0; round < 11; ++round) {
            int partner[12]; double cc[12], ss[12]; /* per index: its partner, cos, signed sin (sign: +s for the p side, -s for the q side) */
            for (int k = 0; k < 6; ++k) {
                int p, q; double c, s;
                rr_pair(round, k, &p, &q);
                jacobi_cs(A[p * 12 + p], A[q * 12 + q], A[p * 12 + q], &c, &s);
                partner[p] = q; partner[q] = p; cc[p] = c; cc[q] = c; ss[p] = -s; ss[q] = s;
            }
            /* J has J_pp = J_qq = c, J_pq = s, J_qp = -s.  (J^T A)_ij = c_i A_ij + ss_i A_partner(i),j with ss_p = -s, ss_q = +s */
            for (int i = 0; i < 12; ++i)
                for (int j = 0; j < 12; ++j) B[i * 12 + j] = cc[i] * A[i * 12 + j] + ss[i] * A[partner[i] * 12 + j];
            for (int i = 0; i < 12; ++i)
                for (int j = 0; j < 12; ++j) A[i * 12 + j] = cc[j] * B[i * 12 + j] + ss[j] * B[i * 12 + partner[j]];
            for (int i = 0; i < 12; ++i)
                for (int j = 0; j < 12; ++j) B[i * 12 + j] = cc[j] * V[i * 12 + j] + ss[j] * V[i * 12 + partner[j]];
            memcpy(V, B, sizeof(B));
        }
}

/* symmetric 3 x 3: sequential cyclic Jacobi (0,1), (0,2), (1,2); eigenvalues on the diagonal of A, eigenvectors in the columns of V */
void vo_jacobi_eig3(double* A, double* V) {
    for (int i = 0; i < 9; ++i) V[i] = (i / 3 == i % 3) ? 1.0 : 0.0;
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int sweep = 0; sweep < EPNP_SWEEPS3; ++sweep)
        for (int r = 0; r < 3; ++r) {
            const int p = P[r], q = Q[r];
            double c, s;
            jacobi_cs(A[p * 3 + p], A[q * 3 + q], A[p * 3 + q], &c, &s);
            for (int j = 0; j < 3; ++j) { /* rows p, q of J^T A */
                const double ap = A[p * 3 + j], aq = A[q * 3 + j];
                A[p * 3 + j] = c * ap - s * aq; A[q * 3 + j] = s * ap + c * aq;
            }
            for (int i = 0; i < 3; ++i) { /* columns p, q of (.) J and of V J */
                const double ap = A[i * 3 + p], aq = A[i * 3 + q];
                A[i * 3 + p] = c * ap - s * aq; A[i * 3 + q] = s * ap + c * aq;
                const double vp = V[i * 3 + p], vq = V[i * 3 + q];
                V[i * 3 + p] = c * vp - s * vq; V[i * 3 + q] = s * vp + c * vq;
            }
        }
}

/* least squares min |A x - b| for an m x n system (m <= 6, n <= 5), Householder QR; A and b are destroyed */
static void qr_solve(double* A, double* b, int m, int n, double* x) {
    for (int k = 0; k < n; ++k) {
        double norm2 = 0;
        for (int i = k; i < m; ++i) norm2 += A[i * n + k] * A[i * n + k];
        const double norm = sqrt(norm2);
        if (norm == 0.0) continue;
        const double alpha = A[k * n + k] > 0 ? -norm : norm;
        /* v = a_k - alpha e_k (stored in place), beta = 2 / v^T v */
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
        A[k * n + k] = alpha; /* R_kk */
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = b[k];
        for (int j = k + 1; j < n; ++j) s -= A[k * n + j] * x[j];
        x[k] = A[k * n + k] != 0.0 ? s / A[k * n + k] : 0.0;
    }
}

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dist2(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

#define EPNP_MAXN 512

typedef struct {
    int n;
    double fu, fv, uc, vc;
    double pws[EPNP_MAXN * 3], us[EPNP_MAXN * 2], alphas[EPNP_MAXN * 4], pcs[EPNP_MAXN * 3];
    double cws[4][3], ccs[4][3];
} epnp_t;

static void choose_control_points(epnp_t* e) {
    const int n = e->n;
    e->cws[0][0] = e->cws[0][1] = e->cws[0][2] = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) e->cws[0][j] += e->pws[3 * i + j];
    for (int j = 0; j < 3; ++j) e->cws[0][j] /= n;
    double C[9] = {0}, V[9];
    for (int i = 0; i < n; ++i) {
        double d[3];
        for (int j = 0; j < 3; ++j) d[j] = e->pws[3 * i + j] - e->cws[0][j];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[a * 3 + b] += d[a] * d[b];
    }
    vo_jacobi_eig3(C, V);
    /* principal directions in descending order of eigenvalue (epnp.cpp reads the SVD's rows in that order); ties: lower index first */
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2 - a; ++b) if (C[ord[b + 1] * 4] > C[ord[b] * 4]) { int t = ord[b]; ord[b] = ord[b + 1]; ord[b + 1] = t; }
    for (int i = 1; i < 4; ++i) {
        const int col = ord[i - 1];
        const double ev = C[col * 4] > 0 ? C[col * 4] : 0.0;
        const double k = sqrt(ev / n);
        /* an eigenvector's sign is the eigen-solver's choice (and with noisy data the EPnP estimate depends on where the control
         * points sit): fixed here -- the component of largest magnitude (first on ties) is positive */
        int big = 0;
        for (int j = 1; j < 3; ++j) if (fabs(V[j * 3 + col]) > fabs(V[big * 3 + col])) big = j;
        const double sg = V[big * 3 + col] < 0 ? -1.0 : 1.0;
        for (int j = 0; j < 3; ++j) e->cws[i][j] = e->cws[0][j] + k * (sg * V[j * 3 + col]);
    }
}

static int compute_barycentric(epnp_t* e) {
    double cc[9], ci[9];
    for (int i = 0; i < 3; ++i) for (int j = 1; j < 4; ++j) cc[3 * i + j - 1] = e->cws[j][i] - e->cws[0][i];
    /* 3 x 3 inverse by the adjugate (epnp.cpp: cvInvert(CV_SVD); the same matrix for a non-singular CC) */
    const double c00 = cc[4] * cc[8] - cc[5] * cc[7], c01 = cc[5] * cc[6] - cc[3] * cc[8], c02 = cc[3] * cc[7] - cc[4] * cc[6];
    const double det = cc[0] * c00 + cc[1] * c01 + cc[2] * c02;
    if (det == 0.0 || !isfinite(det)) return 0;
    const double id = 1.0 / det;
    ci[0] = c00 * id; ci[1] = (cc[2] * cc[7] - cc[1] * cc[8]) * id; ci[2] = (cc[1] * cc[5] - cc[2] * cc[4]) * id;
    ci[3] = c01 * id; ci[4] = (cc[0] * cc[8] - cc[2] * cc[6]) * id; ci[5] = (cc[2] * cc[3] - cc[0] * cc[5]) * id;
    ci[6] = c02 * id; ci[7] = (cc[1] * cc[6] - cc[0] * cc[7]) * id; ci[8] = (cc[0] * cc[4] - cc[1] * cc[3]) * id;
    for (int i = 0; i < e->n; ++i) {
        const double* pi = e->pws + 3 * i;
        double* a = e->alphas + 4 * i;
        for (int j = 0; j < 3; ++j)
            a[1 + j] = ci[3 * j] * (pi[0] - e->cws[0][0]) + ci[3 * j + 1] * (pi[1] - e->cws[0][1]) + ci[3 * j + 2] * (pi[2] - e->cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    return 1;
}

static void compute_L_6x10(const double v[4][12], double* L) {
    double dv[4][6][3];
    for (int i = 0; i < 4; ++i) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; ++j) {
            for (int c = 0; c < 3; ++c) dv[i][j][c] = v[i][3 * a + c] - v[i][3 * b + c];
            b++;
            if (b > 3) { a++; b = a + 1; }
        }
    }
    for (int i = 0; i < 6; ++i) {
        double* row = L + 10 * i;
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

static void find_betas(int N, const double* L, const double* rho, double* betas) {
    static const int cols1[4] = {0, 1, 3, 6}, cols2[3] = {0, 1, 2}, cols3[5] = {0, 1, 2, 3, 4};
    const int nc = N == 1 ? 4 : (N == 2 ? 3 : 5);
    const int* cols = N == 1 ? cols1 : (N == 2 ? cols2 : cols3);
    double A[30], b[6], x[5];
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < nc; ++j) A[i * nc + j] = L[10 * i + cols[j]]; b[i] = rho[i]; }
    qr_solve(A, b, 6, nc, x);
    if (N == 1) { /* [B11 B12 B13 B14] */
        if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = -x[1] / betas[0]; betas[2] = -x[2] / betas[0]; betas[3] = -x[3] / betas[0]; }
        else { betas[0] = sqrt(x[0]); betas[1] = x[1] / betas[0]; betas[2] = x[2] / betas[0]; betas[3] = x[3] / betas[0]; }
    } else {       /* [B11 B12 B22 (B13 B23)] */
        if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
        else { betas[0] = sqrt(x[0]); betas[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) betas[0] = -betas[0];
        betas[2] = N == 3 ? x[3] / betas[0] : 0.0;
        betas[3] = 0.0;
    }
}

static void gauss_newton(const double* L, const double* rho, double* betas) {
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

/* R, t from the camera-frame points (Arun / Horn via the SVD of the 3 x 3 correlation; epnp.cpp estimate_R_and_t) */
static void estimate_R_and_t(const epnp_t* e, double R[9], double t[3]) {
    const int n = e->n;
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) { pc0[j] += e->pcs[3 * i + j]; pw0[j] += e->pws[3 * i + j]; }
    for (int j = 0; j < 3; ++j) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) abt[3 * j + k] += (e->pcs[3 * i + j] - pc0[j]) * (e->pws[3 * i + k] - pw0[k]);
    /* SVD abt = U D V^T through the symmetric eigenproblem of abt^T abt (= V D^2 V^T); U = abt V D^-1, the column of the smallest
     * singular value completed as the cross product of the other two (rank-2 safe); then R = U V^T with the sign fix of epnp.cpp */
    double S[9], V[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a * 3 + b] = abt[a] * abt[b] + abt[3 + a] * abt[3 + b] + abt[6 + a] * abt[6 + b];
    vo_jacobi_eig3(S, V);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2 - a; ++b) if (S[ord[b + 1] * 4] > S[ord[b] * 4]) { int tt = ord[b]; ord[b] = ord[b + 1]; ord[b + 1] = tt; }
    double Vs[9], U[9];
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Vs[r * 3 + c] = V[r * 3 + ord[c]];
    /* make V a proper rotation basis first: third column = v0 x v1 (keeps the eigenvector up to sign) */
    Vs[2] = Vs[3] * Vs[7] - Vs[6] * Vs[4]; Vs[5] = Vs[6] * Vs[1] - Vs[0] * Vs[7]; Vs[8] = Vs[0] * Vs[4] - Vs[3] * Vs[1];
    for (int c = 0; c < 2; ++c) {
        double u[3], nrm = 0;
        for (int r = 0; r < 3; ++r) { u[r] = abt[r * 3] * Vs[c] + abt[r * 3 + 1] * Vs[3 + c] + abt[r * 3 + 2] * Vs[6 + c]; nrm += u[r] * u[r]; }
        nrm = sqrt(nrm);
        for (int r = 0; r < 3; ++r) U[r * 3 + c] = nrm > 0 ? u[r] / nrm : (r == c ? 1.0 : 0.0);
    }
    U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
    /* with both bases right-handed R = U V^T has det +1; epnp.cpp reaches the same matrix by flipping the third row when det < 0
     * (its SVD may return a reflection), except for correlations whose best orthogonal fit IS a reflection (noise-dominated sets) */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] = U[i * 3] * Vs[j * 3] + U[i * 3 + 1] * Vs[j * 3 + 1] + U[i * 3 + 2] * Vs[j * 3 + 2];
    for (int i = 0; i < 3; ++i) t[i] = pc0[i] - dot3(R + 3 * i, pw0);
}

static double reprojection_error(const epnp_t* e, const double R[9], const double t[3]) {
    double sum = 0;
    for (int i = 0; i < e->n; ++i) {
        const double* pw = e->pws + 3 * i;
        const double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        const double ue = e->uc + e->fu * Xc * inv_Zc, ve = e->vc + e->fv * Yc * inv_Zc;
        const double du = e->us[2 * i] - ue, dv = e->us[2 * i + 1] - ve;
        sum += sqrt(du * du + dv * dv);
    }
    return sum / e->n;
}

static double compute_R_and_t(epnp_t* e, const double v[4][12], const double* betas, double R[9], double t[3]) {
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) e->ccs[j][k] = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 3; ++k) e->ccs[j][k] += betas[i] * v[i][3 * j + k];
    for (int i = 0; i < e->n; ++i) {
        const double* a = e->alphas + 4 * i;
        for (int j = 0; j < 3; ++j) e->pcs[3 * i + j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
    }
    if (e->pcs[2] < 0.0) { /* solve_for_sign */
        for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) e->ccs[j][k] = -e->ccs[j][k];
        for (int i = 0; i < 3 * e->n; ++i) e->pcs[i] = -e->pcs[i];
    }
    estimate_R_and_t(e, R, t);
    return reprojection_error(e, R, t);
}

/* EPnP pose of n (4 <= n <= 512) 3D-2D correspondences.  xyz: n x 3 f32 (cv::Point3f), uv: n x 2 f32 pixels (cv::Point2f), K4 = fx fy cx cy.
 * Output R (row-major 3 x 3) and t of the world-to-camera transform; returns the mean reprojection error of the chosen candidate,
 * or -1 for a degenerate configuration (singular control-point basis / non-finite result). */
double vo_epnp(const float* xyz, const float* uv, int n, const double K4[4], double R[9], double t[3]) {
    if (n < 4 || n > EPNP_MAXN) return -1.0;
    epnp_t e;
    e.n = n; e.fu = K4[0]; e.fv = K4[1]; e.uc = K4[2]; e.vc = K4[3];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < 3; ++j) e.pws[3 * i + j] = (double)xyz[3 * i + j];
        /* solvePnP(EPNP): undistortPoints -> normalised coordinates STORED AS f32 (the input type), epnp::init_points: x * fu + uc */
        const float xn = (float)(((double)uv[2 * i] - e.uc) * (1.0 / e.fu)), yn = (float)(((double)uv[2 * i + 1] - e.vc) * (1.0 / e.fv));
        e.us[2 * i] = (double)xn * e.fu + e.uc;
        e.us[2 * i + 1] = (double)yn * e.fv + e.vc;
    }
    choose_control_points(&e);
    if (!compute_barycentric(&e)) return -1.0;
    double MtM[144] = {0}, V[144];
    for (int i = 0; i < n; ++i) { /* fill_M rows 2i, 2i+1 accumulated straight into M^T M (cvMulTransposed(M, MtM, 1)) */
        double m1[12], m2[12];
        const double* as = e.alphas + 4 * i;
        for (int j = 0; j < 4; ++j) {
            m1[3 * j] = as[j] * e.fu; m1[3 * j + 1] = 0.0; m1[3 * j + 2] = as[j] * (e.uc - e.us[2 * i]);
            m2[3 * j] = 0.0; m2[3 * j + 1] = as[j] * e.fv; m2[3 * j + 2] = as[j] * (e.vc - e.us[2 * i + 1]);
        }
        for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) MtM[a * 12 + b] += m1[a] * m1[b] + m2[a] * m2[b];
    }
    vo_jacobi_eig12(MtM, V);
    int ord[12];
    for (int i = 0; i < 12; ++i) ord[i] = i;
    for (int a = 0; a < 4; ++a) { /* the four smallest eigenvalues, ascending, ties by index (selection sort: stable) */
        int best = a;
        for (int b = a + 1; b < 12; ++b) if (MtM[ord[b] * 13] < MtM[ord[best] * 13]) best = b;
        const int tmp = ord[best];
        for (int b = best; b > a; --b) ord[b] = ord[b - 1];
        ord[a] = tmp;
    }
    double v[4][12];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 12; ++r) v[i][r] = V[r * 12 + ord[i]];
    double L[60], rho[6];
    compute_L_6x10(v, L);
    rho[0] = dist2(e.cws[0], e.cws[1]); rho[1] = dist2(e.cws[0], e.cws[2]); rho[2] = dist2(e.cws[0], e.cws[3]);
    rho[3] = dist2(e.cws[1], e.cws[2]); rho[4] = dist2(e.cws[1], e.cws[3]); rho[5] = dist2(e.cws[2], e.cws[3]);
    double best_err = -1;
    for (int N = 1; N <= 3; ++N) {
        double betas[4], Rn[9], tn[3];
        find_betas(N, L, rho, betas);
        gauss_newton(L, rho, betas);
        const double err = compute_R_and_t(&e, v, betas, Rn, tn);
        /* epnp.cpp: N = 1; if (err2 < err1) N = 2; if (err3 < err_N) N = 3 -- NaN never wins a '<' */
        if (N == 1 || err < best_err) { best_err = err; memcpy(R, Rn, sizeof(Rn)); memcpy(t, tn, sizeof(tn)); }
    }
    for (int i = 0; i < 9; ++i) if (!isfinite(R[i])) return -1.0;
    for (int i = 0; i < 3; ++i) if (!isfinite(t[i])) return -1.0;
    return best_err;
}

/* rotation matrix -> unit quaternion (x, y, z, w), w >= 0 (Shepperd's method): the pose storage of this package */
void vo_rotmat_to_quat(const double R[9], double q[4]) {
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) { const double s = sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; }
    else if (R[0] > R[4] && R[0] > R[8]) { const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; }
    else if (R[4] > R[8]) { const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s; }
    else { const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double sgn = q[3] < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 4; ++i) q[i] = sgn * q[i] / n;
}
