/*
 * oracle/lm.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see vo_oracle.h) for rows A8, A10, A11, A12, A13 of
 * the reprojection residuals/Jacobians and the two g2o Levenberg-Marquardt passes of
 * /root/reference/src/stereo_visual_slam_main/optimization.cpp (optimize_map :103-288, optimize_pose_only
 * :290-436), plus the north_star motion-only pose stage that stands in for cv::solvePnPRansac
 * (visual_odometry.cpp:277).
 *
 * [REF] residuals, Jacobians and vertex updates follow optimization.cpp:26-101 line by line (in meaning).
 * [UPSTREAM, PARITY UNPINNED] the optimiser is a restatement of g2o (not in the image):
 *   core/optimization_algorithm_levenberg.cpp (solve, computeLambdaInit tau=1e-5, computeScale, lambda
 *   schedule 1/3..2/3, ni doubling, <=10 trials), core/block_solver.hpp (buildSystem, Schur complement,
 *   back-substitution), core/base_binary_edge.hpp / base_unary_edge.hpp (constructQuadraticForm with robust
 *   kernel: weight rho'), core/robust_kernel_impl.cpp (Huber), core/sparse_optimizer.cpp (optimize loop).
 *   Linear solves are dense Cholesky (g2o: CSparse Cholesky / Eigen LDLT; same solution up to rounding).
 * Pinned by: finite-difference Jacobian tests, scipy.optimize.least_squares optima, per-iteration invariants.
 */
#include "vo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ residuals ----------------------- */

void vo_pose_only_residual(const double T[7], const double pw[3], const double z[2], const double K[4],
                           double e[2], double J[12]) {
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    double pc[3];
    vo_se3_act(T, pw, pc);
    double X = pc[0], Y = pc[1], Z = pc[2];
    /* computeError optimization.cpp:75-82: pos_pixel = K*(T*p); pos_pixel /= pos_pixel[2]; e = z - pixel */
    double px = fx * X + cx * Z, py = fy * Y + cy * Z;
    e[0] = z[0] - px / Z;
    e[1] = z[1] - py / Z;
    if (J) { /* linearizeOplus optimization.cpp:84-101 */
        double Z2 = Z * Z;
        J[0] = -fx / Z; J[1] = 0; J[2] = fx * X / Z2; J[3] = fx * X * Y / Z2; J[4] = -fx - fx * X * X / Z2; J[5] = fx * Y / Z;
        J[6] = 0; J[7] = -fy / Z; J[8] = fy * Y / (Z * Z); J[9] = fy + fy * Y * Y / Z2; J[10] = -fy * X * Y / Z2; J[11] = -fy * X / Z;
    }
}

void vo_projection_residual(const double T[7], const double pw[3], const double z[2], const double K[4],
                            double e[2], double Jp[12], double Jl[6]) {
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    double pc[3];
    vo_se3_act(T, pw, pc);
    double X = pc[0], Y = pc[1], Z = pc[2];
    double px = fx * X + cx * Z, py = fy * Y + cy * Z; /* optimization.cpp:41-50 */
    e[0] = z[0] - px / Z;
    e[1] = z[1] - py / Z;
    if (Jp) { /* optimization.cpp:52-73 */
        double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;
        Jp[0] = -fx * Zinv; Jp[1] = 0; Jp[2] = fx * X * Zinv2; Jp[3] = fx * X * Y * Zinv2; Jp[4] = -fx - fx * X * X * Zinv2; Jp[5] = fx * Y * Zinv;
        Jp[6] = 0; Jp[7] = -fy * Zinv; Jp[8] = fy * Y * Zinv2; Jp[9] = fy + fy * Y * Y * Zinv2; Jp[10] = -fy * X * Y * Zinv2; Jp[11] = -fy * X * Zinv;
        if (Jl) {
            double R[9];
            vo_se3_rotmat(T, R);
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 3; ++c)
                    Jl[r * 3 + c] = Jp[r * 6 + 0] * R[0 + c] + Jp[r * 6 + 1] * R[3 + c] + Jp[r * 6 + 2] * R[6 + c];
        }
    }
}

/* ------------------------------------------------------------------ helpers ------------------------- */

static void huber(double e, double delta, double rho[2]) {
    double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.0; }
    else { double s = sqrt(e); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; }
}

/* dense Cholesky solve A x = b (A n x n symmetric, row-major, destroyed). returns 0 if not positive definite */
static int chol_solve(double* A, int n, const double* b, double* x) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return 0;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[i * n + k] * x[k];
        x[i] = s / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * x[k];
        x[i] = s / A[i * n + i];
    }
    return 1;
}

static int inv3_sym(const double D[9], double Di[9]) {
    double a = D[0], b = D[1], c = D[2], d = D[4], e = D[5], f = D[8];
    double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    double det = a * c00 + b * c01 + c * c02;
    double id = 1.0 / det;
    Di[0] = c00 * id; Di[1] = c01 * id; Di[2] = c02 * id;
    Di[3] = Di[1]; Di[4] = (a * f - c * c) * id; Di[5] = (b * c - a * e) * id;
    Di[6] = Di[2]; Di[7] = Di[5]; Di[8] = (a * d - b * b) * id;
    return isfinite(id);
}

typedef struct {
    int n_kf, n_lm, n_edge, with_lm;
    const int32_t *kf_idx, *lm_idx;
    const float* uv;
    double K[4], delta;
    double *T, *P;           /* current estimates */
    double *err, *chi2;      /* per edge */
    int *lm_ptr, *lm_edges;  /* CSR by landmark */
} lm_problem;

static double compute_errors(lm_problem* p) {
    double total = 0;
    for (int e = 0; e < p->n_edge; ++e) {
        double z[2] = {(double)p->uv[2 * e], (double)p->uv[2 * e + 1]};
        vo_pose_only_residual(p->T + 7 * p->kf_idx[e], p->P + 3 * p->lm_idx[e], z, p->K, p->err + 2 * e, NULL);
        double c = p->err[2 * e] * p->err[2 * e] + p->err[2 * e + 1] * p->err[2 * e + 1];
        p->chi2[e] = c;
        double rho[2];
        huber(c, p->delta, rho);
        total += rho[0];
    }
    return total;
}

static void oplus_pose(double T[7], const double d[6]) {
    double E[7], C[7];
    vo_se3_exp(d, E);      /* optimization.cpp:26-32: estimate = exp(update) * estimate */
    vo_se3_mul(E, T, C);
    memcpy(T, C, sizeof(C));
}

static int run_lm(lm_problem* p, int iters, vo_lm_stats* st) {
    const int nk = p->n_kf, nl = p->with_lm ? p->n_lm : 0, ne = p->n_edge, np = 6 * nk;
    double* Hpp = (double*)calloc((size_t)nk * 36, sizeof(double));
    double* bp = (double*)calloc((size_t)np, sizeof(double));
    double* Hll = (double*)calloc((size_t)(nl ? nl : 1) * 9, sizeof(double));
    double* bl = (double*)calloc((size_t)(nl ? nl : 1) * 3, sizeof(double));
    double* Hpl = (double*)calloc((size_t)(ne ? ne : 1) * 18, sizeof(double));
    double* S = (double*)malloc(sizeof(double) * (size_t)np * np);
    double* bs = (double*)malloc(sizeof(double) * (size_t)np);
    double* xp = (double*)calloc((size_t)np, sizeof(double));
    double* xl = (double*)calloc((size_t)(nl ? nl : 1) * 3, sizeof(double));
    double* Dinv = (double*)malloc(sizeof(double) * (size_t)(nl ? nl : 1) * 9);
    double* W = (double*)malloc(sizeof(double) * (size_t)nk * 18);
    int* has = (int*)malloc(sizeof(int) * (size_t)nk);
    double* Tbak = (double*)malloc(sizeof(double) * (size_t)nk * 7);
    double* Pbak = (double*)malloc(sizeof(double) * (size_t)(p->n_lm ? p->n_lm : 1) * 3);
    double lambda = 0, ni = 2;
    int it = 0, total_trials = 0;
    if (st) memset(st, 0, sizeof(*st));
    for (it = 0; it < iters; ++it) {
        double currentChi = compute_errors(p);
        if (it == 0 && st) st->chi2_init = currentChi;
        /* ---- buildSystem ---- */
        memset(Hpp, 0, sizeof(double) * (size_t)nk * 36);
        memset(bp, 0, sizeof(double) * (size_t)np);
        if (nl) { memset(Hll, 0, sizeof(double) * (size_t)nl * 9); memset(bl, 0, sizeof(double) * (size_t)nl * 3); }
        for (int e = 0; e < ne; ++e) {
            int k = p->kf_idx[e], l = p->lm_idx[e];
            double z[2] = {(double)p->uv[2 * e], (double)p->uv[2 * e + 1]}, er[2], A[12], B[6];
            if (p->with_lm) vo_projection_residual(p->T + 7 * k, p->P + 3 * l, z, p->K, er, A, B);
            else vo_pose_only_residual(p->T + 7 * k, p->P + 3 * l, z, p->K, er, A);
            /* constructQuadraticForm: the error is the one stored by computeActiveErrors (same state) */
            er[0] = p->err[2 * e]; er[1] = p->err[2 * e + 1];
            double rho[2];
            huber(p->chi2[e], p->delta, rho);
            double w = rho[1];
            for (int a = 0; a < 6; ++a) {
                bp[6 * k + a] += -w * (A[a] * er[0] + A[6 + a] * er[1]);
                for (int b = 0; b < 6; ++b) Hpp[36 * k + 6 * a + b] += w * (A[a] * A[b] + A[6 + a] * A[6 + b]);
            }
            if (p->with_lm) {
                for (int a = 0; a < 3; ++a) {
                    bl[3 * l + a] += -w * (B[a] * er[0] + B[3 + a] * er[1]);
                    for (int b = 0; b < 3; ++b) Hll[9 * l + 3 * a + b] += w * (B[a] * B[b] + B[3 + a] * B[3 + b]);
                }
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 3; ++b) Hpl[18 * e + 3 * a + b] = w * (A[a] * B[b] + A[6 + a] * B[3 + b]);
            }
        }
        if (it == 0) { /* computeLambdaInit */
            double md = 0;
            for (int k = 0; k < nk; ++k) for (int a = 0; a < 6; ++a) md = fmax(fabs(Hpp[36 * k + 7 * a]), md);
            for (int l = 0; l < nl; ++l) for (int a = 0; a < 3; ++a) md = fmax(fabs(Hll[9 * l + 4 * a]), md);
            lambda = 1e-5 * md;
            ni = 2;
        }
        double rho_gain = 0, tempChi = currentChi;
        int qmax = 0;
        do {
            memcpy(Tbak, p->T, sizeof(double) * (size_t)nk * 7); /* push */
            if (p->with_lm) memcpy(Pbak, p->P, sizeof(double) * (size_t)p->n_lm * 3);
            /* ---- solve with lambda on every diagonal ---- */
            memset(S, 0, sizeof(double) * (size_t)np * np);
            for (int k = 0; k < nk; ++k)
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) S[(6 * k + a) * np + 6 * k + b] = Hpp[36 * k + 6 * a + b] + (a == b ? lambda : 0.0);
            memcpy(bs, bp, sizeof(double) * (size_t)np);
            int ok2 = 1;
            for (int l = 0; l < nl; ++l) {
                int e0 = p->lm_ptr[l], e1 = p->lm_ptr[l + 1];
                if (e0 == e1) { memset(Dinv + 9 * l, 0, 72); continue; }
                double D[9];
                memcpy(D, Hll + 9 * l, 72);
                D[0] += lambda; D[4] += lambda; D[8] += lambda;
                if (!inv3_sym(D, Dinv + 9 * l)) ok2 = 0;
                const double* Di = Dinv + 9 * l;
                double db[3];
                for (int a = 0; a < 3; ++a) db[a] = Di[3 * a] * bl[3 * l] + Di[3 * a + 1] * bl[3 * l + 1] + Di[3 * a + 2] * bl[3 * l + 2];
                memset(has, 0, sizeof(int) * (size_t)nk);
                memset(W, 0, sizeof(double) * (size_t)nk * 18);
                for (int j = e0; j < e1; ++j) {
                    int e = p->lm_edges[j], k = p->kf_idx[e];
                    has[k] = 1;
                    for (int a = 0; a < 18; ++a) W[18 * k + a] += Hpl[18 * e + a];
                }
                for (int k1 = 0; k1 < nk; ++k1) {
                    if (!has[k1]) continue;
                    double BD[18];
                    for (int a = 0; a < 6; ++a)
                        for (int b = 0; b < 3; ++b)
                            BD[3 * a + b] = W[18 * k1 + 3 * a] * Di[b] + W[18 * k1 + 3 * a + 1] * Di[3 + b] + W[18 * k1 + 3 * a + 2] * Di[6 + b];
                    for (int a = 0; a < 6; ++a)
                        bs[6 * k1 + a] -= W[18 * k1 + 3 * a] * db[0] + W[18 * k1 + 3 * a + 1] * db[1] + W[18 * k1 + 3 * a + 2] * db[2];
                    for (int k2 = 0; k2 < nk; ++k2) {
                        if (!has[k2]) continue;
                        for (int a = 0; a < 6; ++a)
                            for (int b = 0; b < 6; ++b)
                                S[(6 * k1 + a) * np + 6 * k2 + b] -= BD[3 * a] * W[18 * k2 + 3 * b] + BD[3 * a + 1] * W[18 * k2 + 3 * b + 1] + BD[3 * a + 2] * W[18 * k2 + 3 * b + 2];
                    }
                }
            }
            if (ok2) ok2 = chol_solve(S, np, bs, xp);
            if (!ok2 && !(vo_variant_flags & VO_VAR_STALE_UPDATE)) memset(xp, 0, sizeof(double) * (size_t)np); /* variant: g2o's stale x of the previous solve */
            for (int l = 0; l < nl; ++l) {
                double c[3] = {bl[3 * l], bl[3 * l + 1], bl[3 * l + 2]};
                for (int j = p->lm_ptr[l]; j < p->lm_ptr[l + 1]; ++j) {
                    int e = p->lm_edges[j], k = p->kf_idx[e];
                    for (int b = 0; b < 3; ++b)
                        for (int a = 0; a < 6; ++a) c[b] -= Hpl[18 * e + 3 * a + b] * xp[6 * k + a];
                }
                const double* Di = Dinv + 9 * l;
                for (int a = 0; a < 3; ++a) xl[3 * l + a] = Di[3 * a] * c[0] + Di[3 * a + 1] * c[1] + Di[3 * a + 2] * c[2];
            }
            /* ---- update ---- */
            for (int k = 0; k < nk; ++k) oplus_pose(p->T + 7 * k, xp + 6 * k);
            for (int l = 0; l < nl; ++l)
                if (p->lm_ptr[l] != p->lm_ptr[l + 1]) for (int a = 0; a < 3; ++a) p->P[3 * l + a] += xl[3 * l + a]; /* :34-39 */
            tempChi = compute_errors(p);
            if (!ok2) tempChi = DBL_MAX;
            rho_gain = currentChi - tempChi;
            double scale = 0; /* computeScale */
            for (int i = 0; i < np; ++i) scale += xp[i] * (lambda * xp[i] + bp[i]);
            for (int l = 0; l < nl; ++l)
                if (p->lm_ptr[l] != p->lm_ptr[l + 1]) for (int a = 0; a < 3; ++a) scale += xl[3 * l + a] * (lambda * xl[3 * l + a] + bl[3 * l + a]);
            scale += 1e-3;
            rho_gain /= scale;
            if (rho_gain > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho_gain - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                double sf = fmax(1. / 3., alpha);
                lambda *= sf;
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                memcpy(p->T, Tbak, sizeof(double) * (size_t)nk * 7); /* pop */
                if (p->with_lm) memcpy(p->P, Pbak, sizeof(double) * (size_t)p->n_lm * 3);
            }
            qmax++;
        } while (rho_gain < 0 && qmax < 10);
        total_trials += qmax;
        if (st && it < 32) { st->chi2_iter[it] = currentChi; st->lambda_iter[it] = lambda; st->trials_iter[it] = qmax; }
        if (st) { st->chi2_final = currentChi; st->lambda_final = lambda; }
        if (qmax == 10 || rho_gain == 0) { ++it; break; } /* Terminate: optimize() stops after this iteration */
    }
    if (st) { st->iterations = it; st->total_trials = total_trials; }
    free(Hpp); free(bp); free(Hll); free(bl); free(Hpl); free(S); free(bs); free(xp); free(xl); free(Dinv);
    free(W); free(has); free(Tbak); free(Pbak);
    return 0;
}

static int setup(lm_problem* p, int n_kf, const double* T, int n_lm, const float* xyz, int n_edge,
                 const int32_t* kf_idx, const int32_t* lm_idx, const float* uv, const double K[4], double delta,
                 int with_lm) {
    if (n_kf <= 0 || n_lm < 0 || n_edge < 0) return -1;
    for (int e = 0; e < n_edge; ++e)
        if (kf_idx[e] < 0 || kf_idx[e] >= n_kf || lm_idx[e] < 0 || lm_idx[e] >= n_lm) return -2;
    memset(p, 0, sizeof(*p));
    p->n_kf = n_kf; p->n_lm = n_lm; p->n_edge = n_edge; p->with_lm = with_lm;
    p->kf_idx = kf_idx; p->lm_idx = lm_idx; p->uv = uv; p->delta = delta;
    memcpy(p->K, K, sizeof(double) * 4);
    p->T = (double*)malloc(sizeof(double) * (size_t)n_kf * 7);
    memcpy(p->T, T, sizeof(double) * (size_t)n_kf * 7);
    p->P = (double*)malloc(sizeof(double) * (size_t)(n_lm ? n_lm : 1) * 3);
    for (int i = 0; i < 3 * n_lm; ++i) p->P[i] = (double)xyz[i]; /* Landmark::to_vector_3d types_def.hpp:115-120 */
    p->err = (double*)calloc((size_t)(n_edge ? n_edge : 1) * 2, sizeof(double));
    p->chi2 = (double*)calloc((size_t)(n_edge ? n_edge : 1), sizeof(double));
    p->lm_ptr = (int*)calloc((size_t)n_lm + 2, sizeof(int));
    p->lm_edges = (int*)malloc(sizeof(int) * (size_t)(n_edge ? n_edge : 1));
    for (int e = 0; e < n_edge; ++e) p->lm_ptr[lm_idx[e] + 1]++;
    for (int l = 0; l < n_lm; ++l) p->lm_ptr[l + 1] += p->lm_ptr[l];
    int* fill = (int*)calloc((size_t)n_lm + 1, sizeof(int));
    for (int e = 0; e < n_edge; ++e) { int l = lm_idx[e]; p->lm_edges[p->lm_ptr[l] + fill[l]++] = e; }
    free(fill);
    return 0;
}
static void teardown(lm_problem* p) { free(p->T); free(p->P); free(p->err); free(p->chi2); free(p->lm_ptr); free(p->lm_edges); }

int vo_local_ba(int n_kf, double* T_c_w, int n_lm, float* xyz, int n_edge, const int32_t* kf_idx,
                const int32_t* lm_idx, const float* uv, const double K[4], int iters, double huber_delta,
                int update_poses, int update_lms, double* chi2_out, vo_lm_stats* stats) {
    lm_problem p;
    int rc = setup(&p, n_kf, T_c_w, n_lm, xyz, n_edge, kf_idx, lm_idx, uv, K, huber_delta, 1);
    if (rc) return rc;
    run_lm(&p, iters, stats);
    if (chi2_out) memcpy(chi2_out, p.chi2, sizeof(double) * (size_t)n_edge);
    if (update_poses) memcpy(T_c_w, p.T, sizeof(double) * (size_t)n_kf * 7); /* optimization.cpp:272-278 */
    if (update_lms)                                                           /* :279-286 (float cast :284) */
        for (int l = 0; l < n_lm; ++l)
            if (p.lm_ptr[l] != p.lm_ptr[l + 1]) for (int a = 0; a < 3; ++a) xyz[3 * l + a] = (float)p.P[3 * l + a];
    teardown(&p);
    return 0;
}

int vo_pose_only_window(int n_kf, double* T_c_w, int n_lm, const float* xyz, int n_edge, const int32_t* kf_idx,
                        const int32_t* lm_idx, const float* uv, const double K[4], int iters, double huber_delta,
                        int update_poses, double* chi2_out, vo_lm_stats* stats) {
    lm_problem p;
    int rc = setup(&p, n_kf, T_c_w, n_lm, xyz, n_edge, kf_idx, lm_idx, uv, K, huber_delta, 0);
    if (rc) return rc;
    run_lm(&p, iters, stats);
    if (chi2_out) memcpy(chi2_out, p.chi2, sizeof(double) * (size_t)n_edge);
    if (update_poses) memcpy(T_c_w, p.T, sizeof(double) * (size_t)n_kf * 7); /* optimization.cpp:429-435 */
    teardown(&p);
    return 0;
}

double vo_chi2_classify(const double* chi2, int n_edge, const int32_t* flag_lm, uint8_t* lm_inlier, int n_lm,
                        int* n_inlier_edges, int* n_outlier_edges) {
    double chi2_th = 5.991; /* optimization.cpp:154 */
    int cnt_outlier = 0, cnt_inlier = 0, iteration = 0;
    while (iteration < 5) { /* :226-252 */
        cnt_outlier = 0; cnt_inlier = 0;
        for (int e = 0; e < n_edge; ++e) {
            if (chi2[e] > chi2_th) cnt_outlier++;
            else cnt_inlier++;
        }
        double inlier_ratio = cnt_inlier / (double)(cnt_inlier + cnt_outlier);
        if (inlier_ratio > 0.5) break;
        chi2_th *= 2;
        iteration++;
    }
    for (int e = 0; e < n_edge; ++e) { /* :254-266, ascending edge index */
        int l = flag_lm[e];
        if (l < 0 || l >= n_lm) continue;
        lm_inlier[l] = (uint8_t)!(chi2[e] > chi2_th);
    }
    if (n_inlier_edges) *n_inlier_edges = cnt_inlier;
    if (n_outlier_edges) *n_outlier_edges = cnt_outlier;
    return chi2_th;
}

int vo_pnp_motion_only(const float* xyz_w, const float* uv, int n, const double K[4], double T_c_w[7], int iters,
                       double huber_delta, double reproj_thr, uint8_t* inlier, vo_lm_stats* stats) {
    if (n <= 0) return -1;
    int32_t* kf = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    int32_t* lm = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    for (int i = 0; i < n; ++i) lm[i] = i;
    double* chi2 = (double*)malloc(sizeof(double) * (size_t)n);
    vo_pose_only_window(1, T_c_w, n, xyz_w, n, kf, lm, uv, K, iters, huber_delta, 1, NULL, stats);
    /* inliers at the accepted estimate: reprojection error <= reproj_thr (visual_odometry.cpp:277, 4.0 px) */
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        double pw[3] = {xyz_w[3 * i], xyz_w[3 * i + 1], xyz_w[3 * i + 2]}, z[2] = {uv[2 * i], uv[2 * i + 1]}, e[2];
        vo_pose_only_residual(T_c_w, pw, z, K, e, NULL);
        double c = e[0] * e[0] + e[1] * e[1];
        int ok = isfinite(c) && c <= reproj_thr * reproj_thr;
        if (inlier) inlier[i] = (uint8_t)ok;
        cnt += ok;
    }
    free(kf); free(lm); free(chi2);
    return cnt;
}
