/*
 * vo_oracle.h -- CPU oracle for the stereo-VO hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a from-scratch, dependency-free, double-precision CPU restatement of the algorithms the
 * reference (shangzhouye/stereo-visual-slam) runs on its hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (libvslam_hip.so) never links, includes or
 * calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and it cannot be built here
 * (needs ROS, OpenCV 3.2, g2o, Sophus, Eigen, CSparse -- none in the image).  Every
 * function that restates a third-party algorithm says so and names the upstream file it follows
 * (OpenCV 3.2 `modules/...`, g2o `core/...`, Sophus `se3.hpp`).  What pins this oracle instead:
 * known-answer tests in tests/test_oracle_*.py and independent numpy/scipy cross-checks.
 *
 * Reference citations are `file:line` into /root/reference.
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ types ---------------------------- */

/* layout-compatible with cv::KeyPoint (28 B): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} vo_keypoint;

/* layout-compatible with cv::DMatch (16 B) */
typedef struct {
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;
} vo_dmatch;

#define VO_ORB_NLEVELS 8
#define VO_ORB_BORDER 32 /* max(edgeThreshold 31, ceil(15*sqrt2)=22, 9/2)+1 */

typedef struct {
    int w[VO_ORB_NLEVELS], h[VO_ORB_NLEVELS];
    float scale[VO_ORB_NLEVELS];
    int nfeat[VO_ORB_NLEVELS];
} vo_orb_layout;

typedef struct {
    int iterations;      /* LM iterations actually run (g2o optimize() return value) */
    int total_trials;    /* total inner trials */
    double chi2_init;    /* robustified chi2 before the first iteration */
    double chi2_final;   /* robustified chi2 of the accepted state */
    double lambda_final;
    double chi2_iter[32];   /* accepted robust chi2 after iteration i */
    double lambda_iter[32]; /* lambda after iteration i */
    int trials_iter[32];
} vo_lm_stats;

/* ------------------------------------------------------------------ ORB (A1..A3) --------------------- */

/* Level sizes, scales and per-level feature budgets of cv::ORB::create(nfeatures) (scale 1.2, 8 levels).
 * [UPSTREAM OpenCV 3.2 orb.cpp: getScale, ORB_Impl::detectAndCompute, computeKeyPoints] */
void vo_orb_layout_init(int w, int h, int nfeatures, vo_orb_layout* L);

/* 8-bit INTER_LINEAR resize, 11-bit fixed point. [UPSTREAM OpenCV 3.2 imgwarp.cpp resizeGeneric_/HResizeLinear/VResizeLinear] */
void vo_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);

/* GaussianBlur 7x7 sigma=2, BORDER_REFLECT_101, 8-bit fixed-point separable kernel.
 * [UPSTREAM OpenCV 3.2 smooth.cpp getGaussianKernel + filter.cpp createSeparableLinearFilter (8 bit path)] */
void vo_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
void vo_gaussian_kernel7_fixed(int k[7]);

/* Build the level chain: level 0 = copy, level l = resize(level l-1).  dst[l] must hold w[l]*h[l] bytes (tight). */
void vo_orb_build_pyramid(const uint8_t* img, int stride, const vo_orb_layout* L, int nlevels, uint8_t* const* dst);

/* FAST-9/16 corner score (largest threshold for which the pixel is still a corner, minus 1).
 * [UPSTREAM OpenCV 3.2 fast_score.cpp cornerScore<16>] */
int vo_fast_corner_score(const uint8_t* p, int stride, int threshold);

/* FAST-9/16 with 3x3 non-max suppression; raster order; response = score.
 * [UPSTREAM OpenCV 3.2 fast.cpp FAST_t<16>] returns count (<= cap; extra corners are dropped and -1 returned) */
int vo_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax, vo_keypoint* out, int cap);

/* Harris response (7x7 block, k = 0.04) at integer pixel (x,y). [UPSTREAM orb.cpp HarrisResponses] */
float vo_harris_response(const uint8_t* img, int stride, int x, int y);

/* cv::fastAtan2 (degrees, [0,360]). [UPSTREAM OpenCV 3.2 mathfuncs_core.cpp atan_f32] */
float vo_fast_atan2(float y, float x);

/* intensity-centroid angle over the radius-15 disc. [UPSTREAM orb.cpp ICAngles] */
float vo_ic_angle(const uint8_t* img, int stride, int x, int y);

/* KeyPointsFilter::retainBest with a defined total order: keeps every keypoint whose response is >= the
 * n-th best response (ties at the cut are kept, as the upstream comment intends), preserving input order. */
int vo_retain_best(vo_keypoint* kps, int n, int npoints);

/* cv::ORB::create(nfeatures)->detect(img, kps).  Output order: level ascending, raster within a level.
 * reference call site: visual_odometry.cpp:80 (ctor :22).  returns count or <0 on overflow of cap. */
int vo_orb_detect(const uint8_t* img, int w, int h, int stride, int nfeatures, vo_keypoint* out, int cap);

/* VO::adaptive_non_maximal_suppresion (visual_odometry.cpp:96-157).  Stable sort on response ties.
 * In place; returns the new count. */
int vo_anms(vo_keypoint* kps, int n, int num);

/* cv::ORB::compute(img, kps, desc) (visual_odometry.cpp:85): border cull, regroup by octave (stable),
 * blur, rBRIEF.  kps is filtered/reordered in place; desc gets 32 bytes per surviving keypoint.
 * returns the new count. */
int vo_orb_compute(const uint8_t* img, int w, int h, int stride, vo_keypoint* kps, int n, uint8_t* desc);

/* VO::feature_detection (visual_odometry.cpp:70-94) without the GUI calls: detect(3000) -> ANMS(anms_num)
 * -> compute. returns count. */
int vo_feature_detection(const uint8_t* img, int w, int h, int stride, int nfeatures, int anms_num,
                         vo_keypoint* kps, int cap, uint8_t* desc);

/* ------------------------------------------------------------------ matcher (A5) --------------------- */

/* cv::BFMatcher(NORM_HAMMING, crossCheck=true)::match(query, train).
 * [UPSTREAM OpenCV 3.2 matchers.cpp BFMatcher::knnMatchImpl + core/batch_distance.cpp batchDistance(crosscheck)]
 * returns the number of matches (ascending queryIdx). */
int vo_bf_match_hamming_xcheck(const uint8_t* q, int nq, const uint8_t* t, int nt, vo_dmatch* out);

/* VO::feature_matching (visual_odometry.cpp:219-251): cross-check match + gate d <= max(2*dmin, 30*gap). */
int vo_feature_matching(const uint8_t* q, int nq, const uint8_t* t, int nt, double frame_gap, vo_dmatch* out);

/* ------------------------------------------------------------------ stereo depth (A6) ---------------- */

/* cv::StereoSGBM (MODE_SGBM) + medianBlur 3x3 + filterSpeckles; disp is h x w int16 in 1/16 px, invalid = -16.
 * [UPSTREAM OpenCV 3.2 calib3d/src/stereosgbm.cpp]  disp_raw (optional) receives the map before median/speckle. */
int vo_sgbm_compute(const uint8_t* left, const uint8_t* right, int w, int h, int stride, int num_disp, int block, int P1, int P2,
                    int disp12_max_diff, int pre_filter_cap, int uniqueness, int speckle_window, int speckle_range, int16_t* disp,
                    int16_t* disp_raw);
/* VO::disparity_map (visual_odometry.cpp:159-174): SGBM(0,96,9,648,2592,1,63,10,100,32) -> f32 disparity, invalid = -1 */
int vo_disparity_map(const uint8_t* left, const uint8_t* right, int w, int h, int stride, float* disparity);

/* ------------------------------------------------------------------ geometry (A7, A9) ---------------- */

/* SE3 stored as 7 doubles: unit quaternion (x,y,z,w) then translation (Sophus::SE3d memory order). */
void vo_se3_exp(const double xi[6], double T[7]);             /* [UPSTREAM Sophus se3.hpp exp] tangent = [upsilon; omega] */
void vo_se3_log(const double T[7], double xi[6]);             /* [UPSTREAM Sophus se3.hpp log] */
void vo_se3_mul(const double A[7], const double B[7], double C[7]);
void vo_se3_inv(const double A[7], double C[7]);
void vo_se3_act(const double T[7], const double p[3], double out[3]);
void vo_se3_rotmat(const double T[7], double R[9]);
double vo_se3_angle_y(const double T[7]);                     /* [UPSTREAM Sophus so3.hpp angleY] */

/* Frame::find_3d + VO::set_ref_3d_position (types_def.cpp:9-18, visual_odometry.cpp:176-217).
 * disparity: f32 h x w map (row stride in elements).  Writes per-keypoint world xyz (f32), valid, reliable.
 * No compaction (caller compacts in order).  returns number valid. */
int vo_find_3d_disparity(const vo_keypoint* kps, int n, const float* disparity, int w, int h, int dstride,
                         const double T_c_w[7], const double cam[5] /*fx,fy,cx,cy,b*/,
                         float* xyz_w, uint8_t* valid, uint8_t* reliable);

/* north_star stage K8: rectified-stereo inhomogeneous DLT (4 equations, 3 unknowns, normal equations)
 * on matched (uL,vL),(uR,vR), then the same gates/outputs as set_ref_3d_position.  row_tol: epipolar gate
 * |vL - vR| <= row_tol and uL > uR (rectified pair); < 0 disables it. */
int vo_triangulate_dlt(const float* uvL, const float* uvR, int n, const double T_c_w[7], const double cam[5], double row_tol,
                       float* xyz_w, uint8_t* valid, uint8_t* reliable);

/* VO::check_motion_estimation (visual_odometry.cpp:316-346) */
int vo_check_motion(int num_inliers, const double T_c_l[7], double frame_gap);

/* ------------------------------------------------------------------ LM back-end (A8, A10-A13) -------- */

/* PoseOnlyEdgeProjection::computeError / linearizeOplus (optimization.cpp:75-101). J is 2x6 row-major. */
void vo_pose_only_residual(const double T[7], const double pw[3], const double z[2], const double K[4],
                           double e[2], double J[12]);
/* EdgeProjection::computeError / linearizeOplus (optimization.cpp:41-73). Jp 2x6, Jl 2x3 row-major. */
void vo_projection_residual(const double T[7], const double pw[3], const double z[2], const double K[4],
                            double e[2], double Jp[12], double Jl[6]);

/* g2o LM (OptimizationAlgorithmLevenberg + BlockSolver_6_3 Schur + Cholesky) on the EdgeProjection graph
 * built by optimize_map (optimization.cpp:103-218).  Poses n_kf x 7 (in/out if update_poses), landmarks
 * n_lm x 3 f32 (in/out if update_lms).  chi2_out[e] = un-robustified e^T e of edge e as left by the LAST
 * computeActiveErrors (g2o semantics: after a rejected last trial it reflects the rejected state).
 * returns 0, or <0 on bad arguments. */
int vo_local_ba(int n_kf, double* T_c_w, int n_lm, float* xyz, int n_edge, const int32_t* kf_idx,
                const int32_t* lm_idx, const float* uv, const double K[4], int iters, double huber_delta,
                int update_poses, int update_lms, double* chi2_out, vo_lm_stats* stats);

/* Same for the PoseOnlyEdgeProjection graph of optimize_pose_only (optimization.cpp:290-377): unary edges,
 * landmarks constant, dense solve, one lambda shared by all poses. */
int vo_pose_only_window(int n_kf, double* T_c_w, int n_lm, const float* xyz, int n_edge, const int32_t* kf_idx,
                        const int32_t* lm_idx, const float* uv, const double K[4], int iters, double huber_delta,
                        int update_poses, double* chi2_out, vo_lm_stats* stats);

/* Adaptive chi2 threshold + inlier flags (optimization.cpp:224-266 / :382-424).  Edges are visited in
 * ascending edge index (defined order; the reference iterates a std::map keyed by pointer).  flag_lm[e] is
 * the landmark whose is_inlier flag edge e writes (reference: feat.landmark_id_).  returns final threshold. */
double vo_chi2_classify(const double* chi2, int n_edge, const int32_t* flag_lm, uint8_t* lm_inlier, int n_lm,
                        int* n_inlier_edges, int* n_outlier_edges);

/* north_star stage K9 (substitute for cv::solvePnPRansac at visual_odometry.cpp:277): motion-only LM on one
 * pose with the PoseOnlyEdgeProjection math, Huber delta, then inlier = reprojection error <= reproj_thr px
 * (solvePnPRansac's reprojectionError contract).  T_c_w in: guess, out: estimate. returns #inliers. */
int vo_pnp_motion_only(const float* xyz_w, const float* uv, int n, const double K[4], double T_c_w[7],
                       int iters, double huber_delta, double reproj_thr, uint8_t* inlier, vo_lm_stats* stats);

/* RANSAC wrapper of VO::motion_estimation (cv::solvePnPRansac(..., 100, 4.0, 0.99), visual_odometry.cpp:277); see ransac.c
 * for the restated control flow (RNG, subset draw, acceptance rule, adaptive iteration count) and the documented deviations. */
int vo_ransac_update_num_iters(double p, double ep, int model_points, int max_iters);
int vo_ransac_subsets(int count, int model_points, int max_iters, int32_t* subsets);
int vo_pnp_ransac(const float* xyz_w, const float* uv, int n, const double K[4], double T_c_w[7] /* out */, int max_iters, double reproj_err,
                  double confidence, int lm_iters, uint8_t* inlier, int* iters_run);
int vo_pnp_ransac_hypothesis(const float* xyz_w, const float* uv, int n, const double K[4], int it, double reproj_err, double T[7], int32_t subset[5]);

/* Map bookkeeping in front of the local BA for a batch of consecutive keyframes (windows.c): VO::tracking / insert_key_frame
 * (visual_odometry.cpp:363-424, :592-599) + the graph build of optimize_map (optimization.cpp:127-214), window b = the map right after
 * keyframe b.  Layouts as in include/vslam_hip.h (vslam_tracks_in / vslam_ba_batch), host arrays.  Returns 0, 1 when a capacity was too
 * small (the windows from the first that did not fit are empty), < 0 on inconsistent input.
 * track_rule 1 (the reference's tracking(), :568-599 + :260-270): a match continues a track whenever the last-frame keypoint is a feature; 0: only
 * when it owns a depth of its own (the convention of rounds 4-5).  K4 = {fx, fy, cx, cy}, reproj_thr in pixels (rule 1 only). */
int vo_build_windows(int n_frames, int kp_cap, int lr_cap, int match_cap, int pnp_cap, const vo_keypoint* kps, const vo_dmatch* lr,
                     const int32_t* nlr, const float* xyz, const uint8_t* valid, const uint8_t* reliable, const vo_dmatch* f2f,
                     const int32_t* nf2f, const uint8_t* pose_inlier, const double* T_rel, int n_kf, int lm_capacity, int edge_capacity,
                     int32_t* lm_off, int32_t* edge_off, int32_t* n_kf_out, double* T_out, float* xyz_out, uint8_t* rel_out,
                     uint8_t* inl_out, int32_t* kf_out, int32_t* lm_out, float* uv_out, const double K4[4], double reproj_thr, int track_rule);

/* EPnP (the minimal solver of solvePnPRansac; OpenCV 3.2 modules/calib3d/src/epnp.cpp restated, see epnp.c).  R row-major 3 x 3,
 * t: world -> camera.  Returns the mean reprojection error of the chosen candidate, < 0 for a degenerate configuration. */
double vo_epnp(const float* xyz, const float* uv, int n, const double K4[4], double R[9], double t[3]);
void vo_jacobi_eig12(double* A /* 12 x 12 symmetric, destroyed: eigenvalues on the diagonal */, double* V /* eigenvectors in columns */);
void vo_jacobi_eig3(double* A, double* V);
void vo_rotmat_to_quat(const double R[9], double q[4]);

#ifdef __cplusplus
}
#endif
/* [UPSTREAM] ambiguity switches (tests/oracle_sensitivity.py); 0 = the documented readings */
#define VO_VAR_COSF 1u          /* rBRIEF rotation: cosf / sinf instead of (float)cos((double)) */
#define VO_VAR_RETAIN_EXACT 2u  /* KeyPointsFilter::retainBest: exactly n survivors instead of all ties at the cut */
#define VO_VAR_RESIZE_ROUND 4u  /* 8-bit INTER_LINEAR: one rounding of the 22-bit product instead of the >>4, >>16, +2 >>2 chain */
#define VO_VAR_ATAN2F 8u        /* IC angle: libm atan2f instead of cv::fastAtan2 */
#define VO_VAR_STALE_UPDATE 16u /* g2o failed linear solve: stale update instead of x_p = 0 */
extern unsigned vo_variant_flags;
void vo_set_variant(unsigned flags);
unsigned vo_get_variant(void);

#endif /* VO_ORACLE_H */
