/* ransac.c -- CPU ORACLE (test infrastructure only; PARITY UNPINNED) for the RANSAC wrapper of VO::motion_estimation.
 *
 * The reference calls cv::solvePnPRansac(pts3d, pts2d, K, Mat(), rvec, tvec, false, 100, 4.0, 0.99, inliers)
 * (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:277).  OpenCV 3.2 is not present here; this file restates
 * the published control flow of its RANSACPointSetRegistrator::run (modules/calib3d/src/ptsetreg.cpp) as used by
 * solvePnPRansac (solvepnp.cpp):
 *   - RNG: cv::RNG seeded with (uint64)-1, multiply-with-carry, coefficient 4164903690, uniform(a, b) = a + next() % (b - a);
 *   - getSubset: 5 (= model points for n > 4) indices, each redrawn until distinct from the ones before it;
 *   - per iteration: minimal model, inliers = squared reprojection error <= reprojectionError^2, a model replaces the best one
 *     only when it has strictly more inliers than max(best, modelPoints - 1), then niters = RANSACUpdateNumIters(...);
 *   - on success the pose is refined on the inliers of the best model; the returned mask is the RANSAC mask.
 * Documented deviations (the GPU path makes the same ones, so GPU-vs-oracle parity is exact up to floating point):
 *   R1  minimal solver: OpenCV runs EPnP on the 5 points; here a 10-iteration least-squares LM (no robust kernel) on the 5
 *       points started from the caller's pose guess (the reference passes useExtrinsicGuess = false; the VO loop has the
 *       previous pose at hand).  For an outlier-free subset both reach the same reprojection minimum.
 *   R2  final refinement: OpenCV runs solvePnP(ITERATIVE) from a DLT start on the inliers; here the same least-squares LM
 *       from the best RANSAC model.
 *   R3  reprojection errors are evaluated in f64 (OpenCV: projectPoints output rounded to f32).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vo_oracle.h"

static unsigned rng_next(uint64_t* state) {
    *state = (uint64_t)(unsigned)(*state) * 4164903690ULL + (unsigned)(*state >> 32);
    return (unsigned)(*state);
}
static int rng_uniform(uint64_t* state, int a, int b) { return a == b ? a : (int)(rng_next(state) % (unsigned)(b - a)) + a; }

/* cv::RANSACUpdateNumIters */
int vo_ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* the subset sequence: subsets[it * model_points + i]; returns the number of iterations generated (= max_iters) */
int vo_ransac_subsets(int count, int model_points, int max_iters, int32_t* subsets) {
    uint64_t state = 0xFFFFFFFFFFFFFFFFULL;
    for (int it = 0; it < max_iters; ++it) {
        int32_t* idx = subsets + (size_t)it * model_points;
        for (int i = 0; i < model_points; ++i) {
            for (;;) {
                const int v = rng_uniform(&state, 0, count);
                int j = 0;
                for (; j < i; ++j) if (idx[j] == v) break;
                idx[i] = v;
                if (j == i) break;
            }
        }
    }
    return max_iters;
}

static int count_inliers(const float* xyz, const float* uv, int n, const double K[4], const double T[7], double thr2, uint8_t* mask) {
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const double pw[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}, z[2] = {uv[2 * i], uv[2 * i + 1]};
        double e[2];
        vo_pose_only_residual(T, pw, z, K, e, NULL);
        const double c = e[0] * e[0] + e[1] * e[1];
        const int ok = isfinite(c) && c <= thr2;
        if (mask) mask[i] = (uint8_t)ok;
        cnt += ok;
    }
    return cnt;
}

/* returns the number of RANSAC inliers (0 = failure: T_c_w untouched); iters_run = iterations actually evaluated */
int vo_pnp_ransac(const float* xyz_w, const float* uv, int n, const double K[4], double T_c_w[7], int max_iters, double reproj_err,
                  double confidence, int lm_iters, uint8_t* inlier, int* iters_run) {
    const int mp = 5;
    if (iters_run) *iters_run = 0;
    if (inlier) memset(inlier, 0, (size_t)(n > 0 ? n : 0));
    if (n < mp || max_iters <= 0) return 0;
    int32_t* subsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_iters * mp);
    vo_ransac_subsets(n, mp, max_iters, subsets);
    uint8_t* mask = (uint8_t*)malloc((size_t)n);
    uint8_t* best_mask = (uint8_t*)calloc((size_t)n, 1);
    double best_T[7];
    int max_good = 0, niters = max_iters, it = 0;
    const double thr2 = reproj_err * reproj_err;
    for (it = 0; it < niters; ++it) {
        float sx[15], su[10];
        for (int i = 0; i < mp; ++i) {
            const int k = subsets[(size_t)it * mp + i];
            memcpy(sx + 3 * i, xyz_w + 3 * k, 12); memcpy(su + 2 * i, uv + 2 * k, 8);
        }
        double T[7];
        memcpy(T, T_c_w, sizeof(T));
        vo_pnp_motion_only(sx, su, mp, K, T, lm_iters, 1e300, reproj_err, NULL, NULL); /* R1 */
        const int good = count_inliers(xyz_w, uv, n, K, T, thr2, mask);
        if (good > (max_good > mp - 1 ? max_good : mp - 1)) {
            memcpy(best_mask, mask, (size_t)n); memcpy(best_T, T, sizeof(T));
            max_good = good;
            niters = vo_ransac_update_num_iters(confidence, (double)(n - good) / n, mp, niters);
        }
    }
    if (iters_run) *iters_run = it;
    if (max_good > 0) {
        float* ix = (float*)malloc(sizeof(float) * 3 * (size_t)max_good);
        float* iu = (float*)malloc(sizeof(float) * 2 * (size_t)max_good);
        int m = 0;
        for (int i = 0; i < n; ++i) if (best_mask[i]) { memcpy(ix + 3 * m, xyz_w + 3 * i, 12); memcpy(iu + 2 * m, uv + 2 * i, 8); ++m; }
        vo_pnp_motion_only(ix, iu, m, K, best_T, lm_iters, 1e300, reproj_err, NULL, NULL); /* R2 */
        memcpy(T_c_w, best_T, sizeof(best_T));
        if (inlier) memcpy(inlier, best_mask, (size_t)n);
        free(ix); free(iu);
    }
    free(subsets); free(mask); free(best_mask);
    return max_good;
}
