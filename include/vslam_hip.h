/*
 * vslam_hip.h -- C-ABI of libvslam_hip.so: the MI355X (gfx950) build of the stereo-VO hot path of
 * shangzhouye/stereo-visual-slam (front-end ORB/ANMS/rBRIEF, cross-checked Hamming matching, stereo
 * triangulation, motion-only pose refinement, 10-keyframe local bundle adjustment).
 *
 * The reference has no FFI/plugin layer: its seam is the public C++ surface of visual_odometry.hpp,
 * optimization.hpp and map.hpp.  Each entry point below replaces the arithmetic behind
 * ONE reference method; `file:line` citations are into /root/reference.  The C++ host mirror that keeps the
 * reference's method names on top of this ABI lives in stereo-visual-slam_amd/host/ (see INTEGRATION.md for the
 * binding a maintainer of the reference would add).
 *
 * Conventions
 *   - plain pointers and sizes, POD structs, no C++/torch types; every call returns an int status
 *     (0 = ok, <0 = error, never throws, never reads out of bounds on bad indices -- quirk Q7);
 *   - `vslam_*`      : HOST buffers in/out, synchronous, one call per reference method (drop-in granularity);
 *   - `vslam_*_dev`  : DEVICE-resident, batched over B independent items (stereo pairs / frames / windows),
 *                      asynchronous on the context's stream -- the throughput path bench.py measures;
 *   - SE3 poses are 7 doubles: unit quaternion (x,y,z,w) then translation (Sophus::SE3d memory order);
 *   - camera = {fx, fy, cx, cy, baseline}; K4 = {fx, fy, cx, cy}.
 *   - a context is thread-compatible, not thread-safe; one context per GPU per thread.
 */
#ifndef VSLAM_HIP_H
#define VSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSLAM_OK 0
#define VSLAM_ERR_ARG (-1)       /* null pointer / bad size / index out of range */
#define VSLAM_ERR_HIP (-2)       /* a HIP runtime call failed; see vslam_last_error() */
#define VSLAM_ERR_CAPACITY (-3)  /* an internal or caller capacity was exceeded (results truncated) */
#define VSLAM_ERR_NO_DEVICE (-4) /* no gfx950-class GPU visible */

/* ABI revision of this header.  Bumped whenever a struct grows or a signature changes position-wise (revision 2 inserted K4 into
 * vslam_local_ba / vslam_pose_only_window; revision 3 added struct_size / abi_version to vslam_params and the alignment contract
 * of vslam_feature_matching_dev; revision 4 appended d_n_kf to vslam_ba_batch and added vslam_build_windows_dev, vslam_pnp_ransac_dev,
 * vslam_set_tuning, vslam_sgbm_status_dev).  vslam_create refuses a vslam_params whose struct_size / abi_version do not match the library's,
 * so a caller compiled against an older header fails with VSLAM_ERR_ARG instead of having its arguments reinterpreted. */
#define VSLAM_ABI_VERSION 5

#define VSLAM_ORB_NLEVELS 8
#define VSLAM_MAX_KF 12          /* keyframes per optimisation window (reference: Map::num_keyframes_ = 10, map.hpp:22) */
#define VSLAM_LM_MAX_ITERS 32

/* layout-compatible with cv::KeyPoint (28 B) -- types_def.hpp:23 `cv::KeyPoint keypoint_` */
typedef struct vslam_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} vslam_keypoint;

/* layout-compatible with cv::DMatch (16 B) -- visual_odometry.hpp:117 */
typedef struct vslam_dmatch {
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;
} vslam_dmatch;

/* every hard-coded constant of the reference's hot path (SURVEY.md section 5 "Config / flags") */
typedef struct vslam_params {
    int32_t img_w, img_h;          /* 1241 x 376 (KITTI-00)                                         */
    int32_t max_batch;             /* largest B the context is sized for                            */
    int32_t orb_nfeatures;         /* 3000  visual_odometry.cpp:22                                  */
    int32_t anms_num;              /* 500   visual_odometry.cpp:82 (BASELINE config 2 uses 1500)    */
    int32_t fast_threshold;        /* 20    cv::ORB default                                         */
    int32_t kp_capacity;           /* per-image capacity of keypoint/descriptor outputs (>= 4096)   */
    double cam[5];                 /* fx fy cx cy b: 718.856 718.856 607.1928 185.2157 0.573  types_def.hpp:53-54 */
    double depth_min, depth_max;   /* 10, 400   visual_odometry.cpp:194                             */
    double depth_reliable;         /* 40        visual_odometry.cpp:201                             */
    double match_ratio;            /* 2.0       visual_odometry.cpp:242                             */
    double match_gap_thr;          /* 30.0      visual_odometry.cpp:242                             */
    double huber_delta;            /* 5.991     optimization.cpp:154,205                            */
    double pnp_reproj_thr;         /* 4.0 px    visual_odometry.cpp:277                             */
    double stereo_row_tol;         /* 2.0 px    epipolar gate of the L/R-match depth stage (vslam_triangulate*): a pair is kept
                                      only if |vL - vR| <= tol and uL > uR; < 0 = off.  No counterpart in the reference, whose
                                      depth is SGBM (row-constrained by construction, visual_odometry.cpp:159-174)         */
    int32_t struct_size;           /* sizeof(vslam_params) of the caller's header; set by vslam_default_params, checked by vslam_create */
    int32_t abi_version;           /* VSLAM_ABI_VERSION of the caller's header; likewise                                  */
} vslam_params;

typedef struct vslam_lm_stats {
    int32_t iterations, total_trials;
    double chi2_init, chi2_final, lambda_final;
    double chi2_iter[VSLAM_LM_MAX_ITERS];
    double lambda_iter[VSLAM_LM_MAX_ITERS];
    int32_t trials_iter[VSLAM_LM_MAX_ITERS];
} vslam_lm_stats;

typedef struct vslam_ctx vslam_ctx;

/* ------------------------------------------------------------------ context --------------------------- */
void vslam_default_params(vslam_params* p);
/* stream: a hipStream_t (as void*) to run on, or NULL to create a private one. */
int vslam_create(const vslam_params* p, int device, void* stream, vslam_ctx** out);
void vslam_destroy(vslam_ctx* ctx);
const char* vslam_last_error(void);
const char* vslam_version(void);
int vslam_abi_version(void);       /* VSLAM_ABI_VERSION the library was built with */
int vslam_sync(vslam_ctx* ctx);
/* bytes of device memory the context holds */
size_t vslam_device_bytes(const vslam_ctx* ctx);
/* name of the GPU kernel families, for profiling cross-reference (NUL separated list not needed: static string) */
const char* vslam_kernel_names(void);

/* ------------------------------------------------------------------ A1+A2+A3: VO::feature_detection --- */
/* Replaces the body of VO::feature_detection (visual_odometry.cpp:70-94) minus the GUI calls:
 * cv::ORB(3000)::detect (:80) -> VO::adaptive_non_maximal_suppresion(kps, anms_num) (:82, :96-157)
 * -> cv::ORB::compute (:85).  img: h rows of `stride` bytes (host).  kps/desc: caller buffers of `cap`
 * entries (desc: cap x 32 bytes).  *n_out = number of keypoints written. */
int vslam_feature_detection(vslam_ctx* ctx, const uint8_t* img, int w, int h, int stride,
                            vslam_keypoint* kps, uint8_t* desc, int cap, int* n_out);

/* The three stages individually, for parity tests (same citations). */
int vslam_orb_detect(vslam_ctx* ctx, const uint8_t* img, int w, int h, int stride, vslam_keypoint* kps, int cap, int* n_out);
int vslam_anms(vslam_ctx* ctx, vslam_keypoint* kps /*in/out*/, int n, int num, int* n_out);
int vslam_orb_compute(vslam_ctx* ctx, const uint8_t* img, int w, int h, int stride,
                      vslam_keypoint* kps /*in/out: filtered + regrouped by octave*/, int n, uint8_t* desc, int* n_out);

/* Batched, device-resident: d_imgs = B images, each h x pitch bytes, image b at d_imgs + b*img_bytes.
 * d_kps: B x kp_capacity keypoints; d_desc: B x kp_capacity x 32; d_count: B int32. */
int vslam_feature_detection_dev(vslam_ctx* ctx, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B,
                                vslam_keypoint* d_kps, uint8_t* d_desc, int32_t* d_count);

/* ------------------------------------------------------------------ A5: VO::feature_matching ---------- */
/* Replaces VO::feature_matching (visual_odometry.cpp:219-251): BFMatcher(NORM_HAMMING, crossCheck=true)::match
 * (:225) + the distance gate d <= max(match_ratio*d_min, match_gap_thr*frame_gap) (:229-246).
 * q: nq x 32 (descriptors_1 = last frame), t: nt x 32 (descriptors_2 = current frame). out: >= nq entries.
 * gate = 0 returns the raw cross-check matches.  An empty match set yields *n_out = 0 (reference: UB, Q7). */
int vslam_feature_matching(vslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double frame_gap,
                           int gate, vslam_dmatch* out, int* n_out);

/* Batched, device-resident: item b matches d_q + b*q_stride_bytes (d_nq[b] rows) against d_t + b*t_stride_bytes
 * (d_nt[b] rows); writes d_out + b*out_capacity (ascending queryIdx) and d_nout[b].  d_gap: per-item frame gap.
 * Alignment contract: d_q, d_t and both strides must be multiples of 16 bytes (the kernel reads descriptors with 16-byte
 * loads); anything else is VSLAM_ERR_ARG.  (The host-buffer call above stages into aligned memory itself.) */
int vslam_feature_matching_dev(vslam_ctx* ctx, const uint8_t* d_q, size_t q_stride_bytes, const int32_t* d_nq,
                               const uint8_t* d_t, size_t t_stride_bytes, const int32_t* d_nt,
                               const double* d_gap, int gate, int B, int max_rows,
                               vslam_dmatch* d_out, int out_capacity, int32_t* d_nout);

/* ------------------------------------------------------------------ A6: dense stereo disparity --------- */
/* Replaces VO::disparity_map (visual_odometry.cpp:159-174): cv::StereoSGBM::create(0, 96, 9, 8*9*9, 32*9*9, 1, 63, 10,
 * 100, 32)->compute(left, right) followed by convertTo(CV_32F, 1/16).  left/right: h x w u8 (row stride in bytes);
 * disparity: h x w f32, tightly packed (invalid pixels = -1.0, like the reference).  Optional outputs (may be NULL):
 * disp_i16 = the CV_16S fixed-point map after median + speckle filtering, disp_raw_i16 = before them.
 * Sizes: 100 < w <= 4096, h > 9 (VSLAM_ERR_ARG otherwise; OpenCV 3.2's own output is undefined for w - 96 <= 4, where its
 * horizontal box sum reads past the pixel-cost row). */
int vslam_disparity_map(vslam_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                        float* disparity, int16_t* disp_i16, int16_t* disp_raw_i16);

/* Batched, device-resident: B stereo pairs at d_left/d_right + b*img_stride_bytes (row pitch in bytes); outputs
 * B x h x w, tightly packed.  The SGBM working set (about 250 MB per 1241x376 pair) is grown on demand and kept. */
int vslam_disparity_map_dev(vslam_ctx* ctx, const uint8_t* d_left, const uint8_t* d_right, size_t img_stride_bytes,
                            int pitch, int w, int h, int B, float* d_disparity, int16_t* d_disp_i16, int16_t* d_disp_raw_i16);

/* ------------------------------------------------------------------ A7: depth -> landmarks ------------ */
/* Replaces Frame::find_3d (types_def.cpp:9-18) + the gating of VO::set_ref_3d_position
 * (visual_odometry.cpp:176-217) for a disparity map (h x w f32, row stride in elements).  No compaction:
 * valid[i]/reliable[i] per keypoint, xyz_w[3i..] world point (f32, cv::Point3f).  *n_valid optional. */
int vslam_find_3d_disparity(vslam_ctx* ctx, const vslam_keypoint* kps, int n, const float* disparity, int w, int h,
                            int dstride, const double T_c_w[7], float* xyz_w, uint8_t* valid, uint8_t* reliable, int* n_valid);

/* north_star stage K8: the same contract from matched left/right pixels (uvL, uvR: n x 2 f32) through the
 * rectified-stereo inhomogeneous DLT instead of an SGBM disparity map. */
int vslam_triangulate(vslam_ctx* ctx, const float* uvL, const float* uvR, int n, const double T_c_w[7],
                      float* xyz_w, uint8_t* valid, uint8_t* reliable, int* n_valid);

/* Batched, device-resident form of the above: item b = d_n[b] keypoints at d_kps + b*kp_capacity, disparity map
 * d_disparity + b*h*w (tightly packed f32, e.g. the output of vslam_disparity_map_dev), pose d_T_c_w + 7*b. */
int vslam_find_3d_disparity_dev(vslam_ctx* ctx, const vslam_keypoint* d_kps, const int32_t* d_n, int kp_capacity, int B,
                                const float* d_disparity, int w, int h, const double* d_T_c_w, float* d_xyz_w,
                                uint8_t* d_valid, uint8_t* d_reliable);

/* Batched, device-resident: item b has d_n[b] pairs at offset b*capacity; pose d_T_c_w + 7*b. */
int vslam_triangulate_dev(vslam_ctx* ctx, const float* d_uvL, const float* d_uvR, const int32_t* d_n, int capacity, int B,
                          const double* d_T_c_w, float* d_xyz_w, uint8_t* d_valid, uint8_t* d_reliable);

/* Gather matched keypoint coordinates (device glue between matcher and triangulation / PnP):
 * uvQ[b][i] = kpsQ[b][match.queryIdx].pt, uvT[b][i] = kpsT[b][match.trainIdx].pt */
int vslam_gather_matched_uv_dev(vslam_ctx* ctx, const vslam_keypoint* d_kpsQ, const vslam_keypoint* d_kpsT, int kp_capacity,
                                const vslam_dmatch* d_matches, const int32_t* d_nmatch, int match_capacity, int B,
                                float* d_uvQ, float* d_uvT);

/* ------------------------------------------------------------------ A8/A10: VO::motion_estimation ----- */
/* north_star stage K9, standing in for cv::solvePnPRansac (visual_odometry.cpp:277): motion-only LM on one
 * pose with PoseOnlyEdgeProjection's residual/Jacobian (optimization.cpp:75-101), g2o LM schedule, Huber
 * delta; inlier[i] = reprojection error <= pnp_reproj_thr at the estimate.  T_c_w in: guess, out: estimate. */
int vslam_pnp_motion_only(vslam_ctx* ctx, const float* xyz_w, const float* uv, int n, double T_c_w[7], int iters,
                          uint8_t* inlier, int* n_inliers, vslam_lm_stats* stats);

/* Batched, device-resident: problem b has d_n[b] points at offset b*capacity; poses d_T + 7*b (in/out). */
int vslam_pnp_motion_only_dev(vslam_ctx* ctx, const float* d_xyz_w, const float* d_uv, const int32_t* d_n, int capacity, int B,
                              double* d_T_c_w, int iters, uint8_t* d_inlier, int32_t* d_n_inliers);

/* The reference's own pose stage: cv::solvePnPRansac(pts3d, pts2d, K, Mat(), rvec, tvec, false, 100, 4.0, 0.99, inliers) at
 * visual_odometry.cpp:277 -- OpenCV's RNG (seed (uint64)-1, multiply-with-carry) and 5-point subset draw, EPnP on every subset
 * from scratch (no pose guess: useExtrinsicGuess = false), f32 squared reprojection errors against (float)(reproj_err^2), the
 * strict "more inliers than max(best, 4)" acceptance, RANSACUpdateNumIters, mask = RANSAC mask.
 * lm_iters = 0: T_c_w = the best RANSAC model itself -- what OpenCV 3.2.0 (the version the reference pins, README.md:72) returns: its
 * solvePnPRansac runs solvePnP on the inliers and then assigns `_local_model` to rvec / tvec (solvepnp.cpp), discarding the refined pose.
 * lm_iters > 0: T_c_w = the pose refined on the inliers of the best model (OpenCV 3.4.2+ behaviour).  All `max_iters` hypotheses are solved (one wave each) and scored in parallel on the device (they do not depend on
 * each other); the adaptive stopping rule is then replayed over the counts in order, so the result is that of the sequential loop.
 * Remaining deviations from OpenCV (the final refinement is this library's least-squares LM, not CvLevMarq; the eigen-solver's
 * basis for the null space of the 5-point system) are listed in oracle/ransac.c / epnp.c.  T_c_w: OUTPUT only (untouched on failure).
 * Returns VSLAM_OK; *n_inliers = 0 means RANSAC found no model (reference: solvePnPRansac returns false). */
int vslam_pnp_ransac(vslam_ctx* ctx, const float* xyz_w, const float* uv, int n, double T_c_w[7], int max_iters,
                     double reproj_err, double confidence, int lm_iters, uint8_t* inlier, int* n_inliers, int* iters_run);

/* The same call, additionally returning every hypothesis: models_Rt = max_iters x 12 doubles ([R row-major | t] of the EPnP model of
 * subset i), models_count = its inlier count over all points (-1: degenerate subset).  For diagnostics and the parity tests. */
int vslam_pnp_ransac_models(vslam_ctx* ctx, const float* xyz_w, const float* uv, int n, double T_c_w[7], int max_iters, double reproj_err,
                            double confidence, int lm_iters, uint8_t* inlier, int* n_inliers, int* iters_run, double* models_Rt,
                            int32_t* models_count);
/* The same call for B independent problems whose points are already on the device (throughput mode: the reference's own pose stage,
 * visual_odometry.cpp:277, batched): problem b owns d_n[b] points at [b * capacity, ...) of d_xyz_w (x 3) / d_uv (x 2).  All B x max_iters
 * hypotheses are solved and scored at once, the sequential acceptance rule with its adaptive stopping is replayed per problem.  Returns
 * what OpenCV 3.2.0 returns: the best RANSAC model itself (no refinement; the host-buffer call's lm_iters = 0), T = identity and 0 inliers
 * when no model was accepted or d_n[b] < 5.  d_inlier (B x capacity), d_n_inliers (B), d_iters_run (B) may be NULL.  Asynchronous. */
int vslam_pnp_ransac_dev(vslam_ctx* ctx, const float* d_xyz_w, const float* d_uv, const int32_t* d_n, int capacity, int B, double* d_T_c_w,
                         int max_iters, double reproj_err, double confidence, uint8_t* d_inlier, int32_t* d_n_inliers, int32_t* d_iters_run);

/* VO::check_motion_estimation (visual_odometry.cpp:316-346); host arithmetic (scalar). returns 1/0. */
int vslam_check_motion(int num_inliers, const double T_c_l[7], double frame_gap);

/* ------------------------------------------------------------------ A12: optimize_map ----------------- */
/* Replaces the optimiser of optimize_map (optimization.cpp:103-288) on the graph the caller built from the map
 * containers (:127-214): n_kf poses (none fixed), n_lm landmarks (f32 at rest, Q4), n_edge EdgeProjection
 * edges {kf_idx, lm_idx, uv}; any edge order.  g2o LM + Schur + Huber(huber_delta), `iters` iterations.
 * flag_lm[e]: landmark whose is_inlier flag edge e writes (reference: feat.landmark_id_, :258-264; quirk Q1),
 * or NULL for flag_lm = lm_idx.  lm_inlier: n_lm flags, in/out (:224-266, edges visited in ascending index).
 * update_poses / update_lms: if_update_map / if_update_landmark (:272-287).  chi2_out (n_edge) optional.
 * K4 = {fx, fy, cx, cy}: the `const cv::Mat& K` argument of optimize_map / optimize_pose_only (optimization.hpp:137-139,
 * :150-152), passed per call; NULL = the context's intrinsics (vslam_params.cam). */
int vslam_local_ba(vslam_ctx* ctx, int n_kf, double* T_c_w, int n_lm, float* xyz, int n_edge,
                   const int32_t* kf_idx, const int32_t* lm_idx, const float* uv, const double* K4, const int32_t* flag_lm,
                   int iters, int update_poses, int update_lms, uint8_t* lm_inlier, double* chi2_out,
                   double* chi2_threshold_out, vslam_lm_stats* stats);

/* ------------------------------------------------------------------ A13: optimize_pose_only ----------- */
/* Replaces optimize_pose_only (optimization.cpp:290-436): unary PoseOnlyEdgeProjection edges, landmarks constant,
 * dense per-pose solve with one shared lambda, same chi2 classification, pose write-back (:429-435). */
int vslam_pose_only_window(vslam_ctx* ctx, int n_kf, double* T_c_w, int n_lm, const float* xyz, int n_edge,
                           const int32_t* kf_idx, const int32_t* lm_idx, const float* uv, const double* K4, const int32_t* flag_lm,
                           int iters, int update_poses, uint8_t* lm_inlier, double* chi2_out,
                           double* chi2_threshold_out, vslam_lm_stats* stats);

/* Batched, device-resident windows (throughput mode).  n_kf = keyframe slots per window (the stride of
 * d_T_c_w); window w uses the first d_n_kf[w] of them (NULL: all windows have n_kf keyframes).
 * Window w owns landmarks [lm_off[w], lm_off[w+1]) and edges [edge_off[w], edge_off[w+1]) of the concatenated
 * arrays; edges MUST be sorted by landmark inside a window (lm_idx ascending) with lm_idx/kf_idx window-local.
 * One call runs the reference's per-keyframe schedule (run_vslam.cpp:58-71) when schedule = 1:
 *   optimize_map(5 its, no write) x2, optimize_map(10 its, poses written), optimize_pose_only(10 its, written),
 * each followed by the chi2 classification that feeds the next pass's landmark filter; schedule = 0 runs a
 * single optimize_map(iters) (mode 0) or optimize_pose_only(iters) (mode 1) pass with update flags. */
typedef struct vslam_ba_batch {
    int32_t n_windows, n_kf;
    const int32_t* d_lm_off;      /* n_windows + 1 */
    const int32_t* d_edge_off;    /* n_windows + 1 */
    double* d_T_c_w;              /* n_windows x n_kf x 7, in/out */
    float* d_xyz;                 /* total_lm x 3, in/out (only with update_lms) */
    const uint8_t* d_reliable;    /* total_lm: Landmark::reliable_depth_ (optimize_map filter :160); NULL = all 1 */
    uint8_t* d_lm_inlier;         /* total_lm, in/out */
    const int32_t* d_kf_idx;      /* total_edge */
    const int32_t* d_lm_idx;      /* total_edge, window-local */
    const float* d_uv;            /* total_edge x 2 */
    double* d_chi2;               /* total_edge, out (last pass), caller's edge order; NULL = not wanted */
    vslam_lm_stats* d_stats;      /* n_windows (last pass) or NULL */
    int32_t total_lm, total_edge;
    const double* K4;             /* HOST pointer to {fx, fy, cx, cy} (the optimisers' `const cv::Mat& K`), NULL = context intrinsics */
    const int32_t* d_n_kf;        /* n_windows: keyframes of window w (1..n_kf), e.g. the growing map at the start of a sequence; NULL = n_kf */
} vslam_ba_batch;
int vslam_ba_batch_dev(vslam_ctx* ctx, const vslam_ba_batch* batch, int schedule, int mode, int iters,
                       int update_poses, int update_lms);

/* ------------------------------------------------------------------ graph construction on the device --
 * The BA half of a throughput-mode step built from the front-end's own output: replaces, for a batch of n_frames CONSECUTIVE
 * keyframes, the landmark / observation bookkeeping of VO::insert_key_frame (visual_odometry.cpp:363-424) and the graph build of
 * optimize_map / optimize_pose_only (optimization.cpp:127-214, :303-361).  Frame f = batch item f.  A frame's features are its
 * keypoints that the pose stage kept as inliers of a frame-to-frame match (they observe the landmark of the matched feature of
 * frame f - 1, :592-599) plus every other keypoint with a valid depth (it creates a landmark, :403-421); a landmark with an
 * unreliable depth takes the point of its first later observation with a reliable one (:391-401).  Window b = the map right after
 * keyframe b: keyframes [max(0, b - n_kf + 1), b], every landmark they observe (position and reliable flag as of frame b, world
 * = frame 0 through the chained relative poses), one edge per observation, edges landmark-major (landmarks ordered by their number
 * of observations inside the window, then by the first of them: frame, then keypoint index), lm_idx / kf_idx window-local, is_inlier = 1.
 * All pointers are device pointers. */
typedef struct vslam_tracks_in {
    int32_t n_frames;
    int32_t kp_capacity, lr_capacity, match_capacity, pnp_capacity; /* kp_capacity <= 65536 (the builder packs a keypoint index into 16 bits;
                                      larger values are refused with VSLAM_ERR_ARG) */
    const vslam_keypoint* d_kps;   /* n_frames x kp_capacity: left keypoints */
    const vslam_dmatch* d_lr;      /* n_frames x lr_capacity: depth association of frame f, queryIdx = left keypoint (L/R matches; the identity
                                      list when the depth comes from the disparity map) */
    const int32_t* d_nlr;          /* n_frames */
    const float* d_xyz;            /* n_frames x lr_capacity x 3: point of association m in the CAMERA frame of frame f (T_c_w = identity) */
    const uint8_t* d_valid;        /* n_frames x lr_capacity: the depth gates of set_ref_3d_position passed (:199) */
    const uint8_t* d_reliable;     /* n_frames x lr_capacity: reliable_depth_ (:201) */
    const vslam_dmatch* d_f2f;     /* (n_frames - 1) x match_capacity: item i = matches frame i (query) -> frame i + 1 (train).  PRECONDITION: one-to-one
                                      inside an item (no two matches share a queryIdx or a trainIdx) -- what the cross-checked matcher of
                                      vslam_feature_matching[_dev] (VO::feature_matching, visual_odometry.cpp:219-251) emits.  Matches that share a
                                      trainIdx (knn / ratio-test output) would make two tracks claim one keypoint: not supported, result undefined */
    const int32_t* d_nf2f;         /* n_frames - 1 */
    const uint8_t* d_pose_inlier;  /* (n_frames - 1) x pnp_capacity: inlier flag of input j of item i's pose problem, inputs in the order
                                      vslam_build_pnp_inputs_dev emitted them */
    const double* d_T_rel;         /* (n_frames - 1) x 7: T_{i+1,i}, the pose stage's estimate with frame i as the world */
    const int32_t* d_nkps;         /* n_frames: keypoints of frame f (no match refers to a keypoint index beyond it), or NULL: every slot up to
                                      kp_capacity is examined (slower, same result) */
    /* ---- ABI rev 5: a batch that is a CHUNK of a longer sequence (sequence mode, BASELINE config 5).  All four may be NULL / 0. */
    const double* d_T_abs;         /* n_frames x 7: T_{f,0} of every frame of the batch in the SEQUENCE's world (the gathered, chained relative poses);
                                      used instead of chaining d_T_rel from the batch's first frame */
    const float* d_carry_in;       /* kp_capacity x 4 floats, for the batch's FIRST frame: {x, y, z, flags} per keypoint slot -- flags (as float) 0: no
                                      track reaches this keypoint from before the batch; 1: one does, and (x, y, z) is its landmark's position so far
                                      (its creation point, no reliable depth seen yet); 3: likewise, position from a reliable depth.  What the rank that
                                      owns the frames before this chunk exports (d_carry_out): tracks and their landmark positions then continue across
                                      the chunk boundary exactly as in one unsharded batch */
    float* d_carry_out;            /* kp_capacity x 4 floats: the same record for frame `carry_out_frame` of THIS batch (the first frame of the next chunk) */
    int32_t carry_out_frame;       /* 1 .. n_frames - 1 (0: no carry-out) */
} vslam_tracks_in;
/* Fills the device arrays of `out` (caller-allocated: d_lm_off / d_edge_off n_frames + 1, d_T_c_w n_frames x n_kf x 7, d_xyz /
 * d_reliable / d_lm_inlier for lm_capacity landmarks, d_kf_idx / d_lm_idx / d_uv for edge_capacity edges, d_n_kf n_frames; the
 * const members are written through) and its scalar members (n_windows = n_frames, n_kf, total_lm = lm_capacity, total_edge =
 * edge_capacity: bounds).  d_status (1 int32): 0, or 1 when a capacity was too small -- the windows from the first one that did
 * not fit are then emitted empty.  Asynchronous on the context stream; `out` can go straight into vslam_ba_batch_dev.
 * Track continuation (round 6; vslam_set_tuning "track_rule", default 1): a frame-to-frame match gives the current keypoint the landmark of the
 * last-frame keypoint whenever that keypoint IS A FEATURE of the last frame -- created there (a valid depth) or tracked into it -- as VO::tracking
 * does (visual_odometry.cpp:568-599: the query set is frame_last_.features_).  If the last-frame keypoint owns a depth, it was input j of the pose
 * stage and d_pose_inlier decides (:306); if it does not, the pose stage's inlier rule (solvePnPRansac's 4 px, params.pnp_reproj_thr) is applied
 * to the landmark's MAP position (:260-270: pt_3d_, the creation point or the first reliable one) through the current frame's chained pose and
 * params.cam.  track_rule 0: only matches whose last-frame keypoint owns a depth continue a track (rounds 4-5). */
int vslam_build_windows_dev(vslam_ctx* ctx, const vslam_tracks_in* in, int n_kf, int lm_capacity, int edge_capacity, vslam_ba_batch* out,
                            int32_t* d_status);

/* per-window status of the most recent window launch on this process (VSLAM_OK or VSLAM_ERR_ARG per window) */
int vslam_ba_status_dev(vslam_ctx* ctx, int n_windows, int32_t* h_status);
/* optimize_map passes the most recent vslam_ba_batch_dev(schedule = 1) call EXECUTED per window: 3 = all of run_vslam.cpp:61-66; 1 or 2 = the
 * window's first / second pass flagged no new landmark, so the following passes would have repeated it bit for bit (all passes start from the same
 * poses and landmarks; only the flags carry over) and it was continued to the last pass's 10 iterations instead -- same result, 10 or 15 LM
 * iterations instead of 20.  vslam_set_tuning(ctx, "ba_adaptive", 0) runs every pass.  Synchronises the context stream. */
int vslam_ba_schedule_passes_dev(vslam_ctx* ctx, int n_windows, int32_t* h_passes);
/* Which kernel took each window of the most recent vslam_ba_batch_dev / vslam_local_ba call (optimization.cpp:103-288): h_deferred[w] = 0 when
 * ba_resident_kernel (landmark state in LDS) ran its optimize_map passes, 1 when it was left to lm_window_kernel (state in HBM: the window does not
 * fit half a CU's LDS, is denser than 2.2 observations per landmark, or the call did not involve the resident kernel).  Diagnostic for the
 * measurement tier (bench.py names the roofline kernel after what ran); synchronises the context stream. */
int vslam_ba_deferred_dev(vslam_ctx* ctx, int n_windows, int32_t* h_deferred);

/* Diagnostic (rows A10 / A11): the residual and the Jacobians of EdgeProjection (optimization.cpp:41-73) and PoseOnlyEdgeProjection
 * (:75-101) as THE DEVICE CODE OF THE LM KERNELS evaluates them -- the same device functions (normalised-coordinate factors At, Bt,
 * en, Huber weight), scaled back to pixels: err = z - K (T p) / Z (2), J_pose = d err / d xi for the left perturbation T <- exp(xi) T,
 * xi = [translation; rotation] (2 x 6, row-major), J_point = d err / d p_w (2 x 3), chi2 = |err|^2, w = Huber weight (delta =
 * params.huber_delta).  n observations of n world points through ONE pose; host buffers, synchronous; any output may be NULL. */
int vslam_edge_jacobians(vslam_ctx* ctx, int n, const float* xyz_w, const float* uv, const double T_c_w[7], const double* K4,
                         double* err, double* J_pose, double* J_point, double* chi2, double* huber_w);
/* Kernel-choice overrides of a context (tuning aid, and how the tests force every kernel path): name in {"orb_fuse_min", "sgbm_fuse_min",
 * "sgbm_fwd_min" (items per call from which the fused kernel is used), "sgbm_fw_rows" (32 | 64), "pose_only_window", "pnp_window", "ba_adaptive" (0 | 1), "ba_lanes" (256 | 512: lanes per window of the
 * LDS-resident optimize_map kernel; default by the number of windows in the call; the results do not depend on it), "track_rule" (0 | 1: see
 * vslam_build_windows_dev; this one changes RESULTS, it is the before / after switch of round 6)};
 * value -1 = the library's batch-size rule.  vslam_create seeds them once from the environment variables VSLAM_<NAME> (an unparsable
 * or out-of-range value makes vslam_create fail with VSLAM_ERR_ARG); nothing reads the environment afterwards. */
int vslam_set_tuning(vslam_ctx* ctx, const char* name, int value);
/* status word of the most recent vslam_disparity_map_dev launch: synchronises the stream; *h_status = 0 and VSLAM_OK, or
 * *h_status != 0 and VSLAM_ERR_HIP when the chained forward sweep gave up waiting for a predecessor slab (its maps are void).
 * The host-buffer call vslam_disparity_map checks it itself. */
int vslam_sgbm_status_dev(vslam_ctx* ctx, int32_t* h_status);
/* per-image ORB capacity flags of the most recent ORB launch (0 = ok) */
int vslam_orb_status_dev(vslam_ctx* ctx, int B, int32_t* h_status);
/* Diagnostic (rows A1 / A3): one level of the scale pyramid (blurred = 0: cv::resize INTER_LINEAR of the level above, what cv::ORB::detect runs
 * FAST / Harris / the IC angle on; level 0 is the caller's image and is not kept) or of the GaussianBlur 7x7 sigma 2 pyramid (blurred = 1: what
 * cv::ORB::compute samples) of image `item` of the most recent ORB launch of this context (vslam_feature_detection[_dev], vslam_orb_compute:
 * the calls that describe fill both; detect-only calls fill the unblurred pyramid), copied to host memory: *w x *h bytes, row stride out_stride.
 * Synchronises the stream.  visual_odometry.cpp:80,85 (inside cv::ORB). */
int vslam_orb_level(vslam_ctx* ctx, int item, int level, int blurred, uint8_t* out, int out_stride, int out_rows, int* w, int* h);

/* Device glue between the frame-to-frame matcher and the motion-only stage (the gather of
 * VO::motion_estimation, visual_odometry.cpp:260-270): for every frame-to-frame match (query = previous frame,
 * train = current frame) whose query keypoint owns a valid triangulated point (found through the previous frame's
 * L/R matches d_lr, whose queryIdx is the left keypoint), emit (xyz, current pixel), in match order.
 * d_kp2lr: B x kp_capacity int32 scratch.  Outputs: B x out_capacity. */
int vslam_build_pnp_inputs_dev(vslam_ctx* ctx, const vslam_dmatch* d_f2f, const int32_t* d_nf2f, int match_capacity,
                               const vslam_dmatch* d_lr, const int32_t* d_nlr, int lr_capacity, const float* d_xyz_lr,
                               const uint8_t* d_valid_lr, const vslam_keypoint* d_kps_cur, int kp_capacity, int B,
                               int32_t* d_kp2lr, float* d_xyz_out, float* d_uv_out, int32_t* d_nout, int out_capacity);

/* ------------------------------------------------------------------ stage profiler --------------------- */
/* hipEvent brackets around every kernel family launched by this thread's calls on this context (the reference only
 * has a commented-out ros::Time stopwatch, visual_odometry.cpp:652,701-702).  vslam_profile_read synchronises the
 * stream, returns accumulated milliseconds per kernel family since the last read, and resets. */
typedef struct vslam_kernel_time {
    char name[48];
    double total_ms;
    int32_t launches; /* kernel launches covered */
    int32_t calls;    /* brackets recorded */
} vslam_kernel_time;
int vslam_profile_enable(vslam_ctx* ctx, int on);
int vslam_profile_read(vslam_ctx* ctx, vslam_kernel_time* out, int cap, int* n_out);
/* The same brackets as INTERVALS on one time axis PER DEVICE (milliseconds since the first vslam_profile_enable(.., 1) of any context of this
 * process on the context's device; contexts on different devices have different origins -- HIP events of two devices share no clock): with several contexts / streams in flight together this is what tells how much their kernel families overlap.  Synchronises
 * the stream, returns up to `cap` brackets recorded since the last read of either kind (in launch order) and resets, like vslam_profile_read. */
typedef struct vslam_stage_interval {
    char name[48];
    double t0_ms, t1_ms;
} vslam_stage_interval;
int vslam_profile_intervals(vslam_ctx* ctx, vslam_stage_interval* out, int cap, int* n_out);

/* Measurement aid: a float4 streaming copy of `bytes` (read + write), `reps` launches timed with hipEvents on the context
 * stream; *gbs_out = moved GB/s.  The achievable-bandwidth figure reported next to the 8 TB/s HBM spec. */
int vslam_hbm_copy_probe(vslam_ctx* ctx, size_t bytes, int reps, double* gbs_out);
/* One shape of that copy (unroll, non-temporal or not, workgroups per CU): variant = 0 .. vslam_hbm_copy_probe_variants() - 1;
 * name_out (>= 64 bytes, may be NULL) receives its description.  vslam_hbm_copy_probe reports the best of them. */
int vslam_hbm_copy_probe_variants(void);
int vslam_hbm_copy_probe_variant(vslam_ctx* ctx, size_t bytes, int reps, int variant, double* gbs_out, char* name_out);

/* ------------------------------------------------------------------ raw device memory helpers ---------- */
/* For hosts without their own device allocator (the C++ mirror in host/); bench.py passes torch tensors. */
int vslam_dev_alloc(void** p, size_t bytes);
int vslam_dev_free(void* p);
int vslam_dev_upload(vslam_ctx* ctx, void* d, const void* h, size_t bytes);
int vslam_dev_download(vslam_ctx* ctx, void* h, const void* d, size_t bytes);
int vslam_dev_memset(vslam_ctx* ctx, void* d, int value, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* VSLAM_HIP_H */
