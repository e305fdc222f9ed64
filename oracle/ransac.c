/* ransac.c -- CPU ORACLE (test infrastructure only; PARITY UNPINNED) for the RANSAC wrapper of VO::motion_estimation.
 *
 * The reference calls cv::solvePnPRansac(pts3d, pts2d, K, Mat(), rvec, tvec, false, 100, 4.0, 0.99, inliers)
 * (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:277).  OpenCV 3.2 is not present here; this file restates
 * the published control flow of its RANSACPointSetRegistrator::run (modules/calib3d/src/ptsetreg.cpp) as used by
 * solvePnPRansac (solvepnp.cpp):
 *   - RNG: cv::RNG seeded with (uint64)-1, multiply-with-carry, coefficient 4164903690, uniform(a, b) = a + next() % (b - a);
 *   - getSubset: 5 (= model points for n > 4) indices, each redrawn until distinct from the ones before it;
 *   - per iteration: minimal model = EPnP on the 5 points from scratch (useExtrinsicGuess = false; epnp.c), inliers by
 *     PnPRansacCallback::computeError + findInliers: the projection is computed in f64 with ONE reciprocal of the depth
 *     (cvProjectPoints2: z = 1/z; x *= z; u = x*fx + cx), stored as f32, the squared error is formed in f32
 *     ((u - u')^2 + (v - v')^2 on Point2f / Matx21f) and compared with (float)(4.0 * 4.0); a model replaces the best one only
 *     when it has strictly more inliers than max(best, modelPoints - 1), then niters = RANSACUpdateNumIters(...);
 *   - count == modelPoints: the single model, every point an inlier;
 *   - on success the returned mask is the RANSAC mask; the returned pose is the best RANSAC model (lm_iters = 0, OpenCV 3.2.0) or
 *     that model refined on its inliers (lm_iters > 0, OpenCV 3.4.2+), see R2.
 * Remaining documented deviations (the GPU path makes the same ones, so GPU-vs-oracle parity is exact up to floating point):
 *   R2  final refinement: OpenCV runs solvePnP(ITERATIVE) on the inliers -- a DLT / homography start followed by CvLevMarq
 *       (<= 20 iterations, plain L2); here the same least-squares cost is minimised by the package's LM from the best RANSAC
 *       model (both stop at the same local minimum of the reprojection error; OpenCV's stopping rule is not restated).
 *       Which pose is RETURNED differs between 3.x point releases: 3.2.0 (the reference's pinned version) runs the refinement and
 *       then assigns `_local_model` -- the unrefined best RANSAC model -- to rvec / tvec; 3.4.2+ return the refined pose.  lm_iters = 0
 *       selects the former (the host mirror's default), lm_iters > 0 the latter.
 *   R3' the hypothesis rotation goes R -> rvec -> R through cv::Rodrigues in OpenCV (model = [rvec | tvec]); here R is used
 *       as EPnP returns it (a 1e-16 effect).
 *   SVD: see epnp.c (OpenCV's basis for the 2-dimensional null space of the 5-point system is not reproducible).
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

/* PnPRansacCallback::computeError + RANSACPointSetRegistrator::findInliers for the pose (R row-major, t) */
static int count_inliers(const float* xyz, const float* uv, int n, const double K[4], const double R[9], const double t[3], float thr2, uint8_t* mask) {
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const double X = xyz[3 * i], Y = xyz[3 * i + 1], Z = xyz[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const float pu = (float)(x * K[0] + K[2]), pv = (float)(y * K[1] + K[3]); /* projpoints is CV_32FC2 */
        const float du = uv[2 * i] - pu, dv = uv[2 * i + 1] - pv;
        float err = 0.f; /* normL2Sqr<float, float> */
        err += du * du;
        err += dv * dv;
        const int ok = err <= thr2; /* (a NaN error is not an inlier) */
        if (mask) mask[i] = (uint8_t)ok;
        cnt += ok;
    }
    return cnt;
}

static void pose_from_Rt(const double R[9], const double t[3], double T[7]) {
    vo_rotmat_to_quat(R, T);
    T[4] = t[0]; T[5] = t[1]; T[6] = t[2];
}

/* returns the number of RANSAC inliers (0 = failure: T_c_w untouched); iters_run = iterations actually evaluated.
 * T_c_w is an OUTPUT (the reference passes useExtrinsicGuess = false: no pose guess is consumed). */
int vo_pnp_ransac(const float* xyz_w, const float* uv, int n, const double K[4], double T_c_w[7], int max_iters, double reproj_err,
                  double confidence, int lm_iters, uint8_t* inlier, int* iters_run) {
    const int mp = 5;
    if (iters_run) *iters_run = 0;
    if (inlier) memset(inlier, 0, (size_t)(n > 0 ? n : 0));
    if (n < mp || max_iters <= 0) return 0;
    uint8_t* mask = (uint8_t*)malloc((size_t)n);
    uint8_t* best_mask = (uint8_t*)calloc((size_t)n, 1);
    double best_R[9], best_t[3];
    int max_good = 0, niters = max_iters, it = 0;
    const float thr2 = (float)(reproj_err * reproj_err);
    if (n == mp) { /* ptsetreg.cpp: count == modelPoints -> the model of all points, mask of ones */
        if (vo_epnp(xyz_w, uv, n, K, best_R, best_t) >= 0) { memset(best_mask, 1, (size_t)n); max_good = n; }
    } else {
        int32_t* subsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_iters * mp);
        vo_ransac_subsets(n, mp, max_iters, subsets);
        for (it = 0; it < niters; ++it) {
            float sx[15], su[10];
            for (int i = 0; i < mp; ++i) {
                const int k = subsets[(size_t)it * mp + i];
                memcpy(sx + 3 * i, xyz_w + 3 * k, 12); memcpy(su + 2 * i, uv + 2 * k, 8);
            }
            double R[9], t[3];
            if (vo_epnp(sx, su, mp, K, R, t) < 0) continue; /* degenerate subset: OpenCV carries a non-finite model, which scores 0 inliers */
            const int good = count_inliers(xyz_w, uv, n, K, R, t, thr2, mask);
            if (good > (max_good > mp - 1 ? max_good : mp - 1)) {
                memcpy(best_mask, mask, (size_t)n); memcpy(best_R, R, sizeof(R)); memcpy(best_t, t, sizeof(t));
                max_good = good;
                niters = vo_ransac_update_num_iters(confidence, (double)(n - good) / n, mp, niters);
            }
        }
        free(subsets);
    }
    if (iters_run) *iters_run = it;
    if (max_good > 0) {
        float* ix = (float*)malloc(sizeof(float) * 3 * (size_t)max_good);
        float* iu = (float*)malloc(sizeof(float) * 2 * (size_t)max_good);
        int m = 0;
        for (int i = 0; i < n; ++i) if (best_mask[i]) { memcpy(ix + 3 * m, xyz_w + 3 * i, 12); memcpy(iu + 2 * m, uv + 2 * i, 8); ++m; }
        double best_T[7];
        pose_from_Rt(best_R, best_t, best_T);
        /* lm_iters = 0: the best RANSAC model itself -- OpenCV 3.2.0 assigns `_local_model` to rvec / tvec after (and regardless of) the
         * refinement; lm_iters > 0: the refined pose (3.4.2+), deviation R2 */
        if (lm_iters > 0) vo_pnp_motion_only(ix, iu, m, K, best_T, lm_iters, 1e300, reproj_err, NULL, NULL);
        memcpy(T_c_w, best_T, sizeof(best_T));
        if (inlier) memcpy(inlier, best_mask, (size_t)n);
        free(ix); free(iu);
    }
    free(mask); free(best_mask);
    return max_good;
}

/* the model of ONE hypothesis (subset `it` of the cv::RNG sequence for `n` points): pose as 7 doubles + its inlier count; for tests */
int vo_pnp_ransac_hypothesis(const float* xyz_w, const float* uv, int n, const double K[4], int it, double reproj_err, double T[7], int32_t subset[5]) {
    int32_t* subsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)(it + 1) * 5);
    vo_ransac_subsets(n, 5, it + 1, subsets);
    float sx[15], su[10];
    for (int i = 0; i < 5; ++i) { const int k = subsets[(size_t)it * 5 + i]; subset[i] = k; memcpy(sx + 3 * i, xyz_w + 3 * k, 12); memcpy(su + 2 * i, uv + 2 * k, 8); }
    free(subsets);
    double R[9], t[3];
    if (vo_epnp(sx, su, 5, K, R, t) < 0) return -1;
    pose_from_Rt(R, t, T);
    return count_inliers(xyz_w, uv, n, K, R, t, (float)(reproj_err * reproj_err), NULL);
}
