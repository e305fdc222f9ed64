/*
 * cpu_shim.c -- TEST INFRASTRUCTURE ONLY: the host-buffer tier of include/vslam_hip.h served by the CPU oracle.
 *
 * Purpose: BASELINE.json config 1 ("first 50 stereo pairs, CPU path, plumbing") and the trajectory-level parity of
 * SURVEY.md section 8 row A15.  The C++ host mirror (stereo-visual-slam_amd/host/{vo_host,map_host,ba_host}.cpp +
 * run_vslam_main.cpp) is linked against THIS library instead of libvslam_hip.so to give `oracle/run_vslam_cpu`: the
 * same tracking state machine, keyframe policy, map bookkeeping and BA schedule (run_vslam.cpp:40-82) with every
 * arithmetic step done by oracle/libvo_oracle.so on the CPU.  tests/test_gpu_host_driver.py runs both drivers on the
 * same rendered sequence and compares their per-frame traces.
 *
 * Nothing in the product links, loads or calls this file (tests/test_abi.py enforces it); only the `vslam_*` host
 * entry points the C++ mirror uses are provided -- the `_dev` tier has no CPU counterpart on purpose.
 * PARITY UNPINNED like the rest of oracle/ (see vo_oracle.h).
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vslam_hip.h"
#include "vo_oracle.h"

struct vslam_ctx { vslam_params p; };

static char g_err[256] = "";
static void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void vslam_default_params(vslam_params* p) { /* the reference's constants, same list as the product (api.hip) */
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->img_w = 1241; p->img_h = 376; p->max_batch = 1;
    p->orb_nfeatures = 3000; p->anms_num = 500; p->fast_threshold = 20; p->kp_capacity = 4096;
    p->cam[0] = 718.856; p->cam[1] = 718.856; p->cam[2] = 607.1928; p->cam[3] = 185.2157; p->cam[4] = 0.573;
    p->depth_min = 10; p->depth_max = 400; p->depth_reliable = 40;
    p->match_ratio = 2.0; p->match_gap_thr = 30.0; p->huber_delta = 5.991; p->pnp_reproj_thr = 4.0; p->stereo_row_tol = 2.0;
    p->struct_size = (int32_t)sizeof(vslam_params); p->abi_version = VSLAM_ABI_VERSION;
}
int vslam_abi_version(void) { return VSLAM_ABI_VERSION; }
const char* vslam_last_error(void) { return g_err; }
const char* vslam_version(void) { return "vslam CPU oracle shim (test infrastructure)"; }

int vslam_create(const vslam_params* p, int device, void* stream, vslam_ctx** out) {
    (void)device; (void)stream;
    if (!p || !out) { set_error("null argument"); return VSLAM_ERR_ARG; }
    if (p->struct_size != (int32_t)sizeof(vslam_params) || p->abi_version != VSLAM_ABI_VERSION) { set_error("vslam_params from a different ABI"); return VSLAM_ERR_ARG; }
    /* the oracle hard-codes the constants the reference hard-codes; refuse configurations it cannot honour */
    if (p->fast_threshold != 20 || p->depth_min != 10 || p->depth_max != 400 || p->depth_reliable != 40 || p->match_ratio != 2.0 ||
        p->match_gap_thr != 30.0) { set_error("CPU shim: only the reference's constants are supported"); return VSLAM_ERR_ARG; }
    vslam_ctx* c = (vslam_ctx*)calloc(1, sizeof(*c));
    c->p = *p;
    *out = c;
    return VSLAM_OK;
}
void vslam_destroy(vslam_ctx* ctx) { free(ctx); }
int vslam_sync(vslam_ctx* ctx) { return ctx ? VSLAM_OK : VSLAM_ERR_ARG; }

int vslam_feature_detection(vslam_ctx* c, const uint8_t* img, int w, int h, int stride, vslam_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    if (!c || !img || !kps || !desc || !n_out || cap <= 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    const int big = 16384; /* detection produces up to ~3000 (+ ties) keypoints before ANMS */
    vo_keypoint* k = (vo_keypoint*)malloc(sizeof(vo_keypoint) * (size_t)big);
    uint8_t* d = (uint8_t*)malloc((size_t)big * 32);
    const int n = vo_feature_detection(img, w, h, stride, c->p.orb_nfeatures, c->p.anms_num, k, big, d);
    int rc = VSLAM_OK;
    if (n < 0 || n > cap) { set_error("capacity"); rc = VSLAM_ERR_CAPACITY; *n_out = 0; }
    else { memcpy(kps, k, sizeof(vo_keypoint) * (size_t)n); memcpy(desc, d, (size_t)n * 32); *n_out = n; }
    free(k); free(d);
    return rc;
}

int vslam_anms(vslam_ctx* c, vslam_keypoint* kps, int n, int num, int* n_out) {
    if (!c || !kps || !n_out || n < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    *n_out = n ? vo_anms((vo_keypoint*)kps, n, num) : 0;
    return VSLAM_OK;
}

int vslam_feature_matching(vslam_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, double frame_gap, int gate, vslam_dmatch* out, int* n_out) {
    if (!c || !n_out || nq < 0 || nt < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    *n_out = 0;
    if (nq == 0 || nt == 0) return VSLAM_OK;
    *n_out = gate ? vo_feature_matching(q, nq, t, nt, frame_gap, (vo_dmatch*)out) : vo_bf_match_hamming_xcheck(q, nq, t, nt, (vo_dmatch*)out);
    return VSLAM_OK;
}

int vslam_disparity_map(vslam_ctx* c, const uint8_t* left, const uint8_t* right, int w, int h, int stride, float* disparity, int16_t* disp_i16,
                        int16_t* disp_raw_i16) {
    if (!c || !left || !right || !disparity || disp_i16 || disp_raw_i16) { set_error("CPU shim: f32 disparity only"); return VSLAM_ERR_ARG; }
    return vo_disparity_map(left, right, w, h, stride, disparity) == 0 ? VSLAM_OK : VSLAM_ERR_ARG;
}

int vslam_find_3d_disparity(vslam_ctx* c, const vslam_keypoint* kps, int n, const float* disparity, int w, int h, int dstride, const double T_c_w[7],
                            float* xyz_w, uint8_t* valid, uint8_t* reliable, int* n_valid) {
    if (!c || n < 0 || !disparity || !T_c_w) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    const int k = n ? vo_find_3d_disparity((const vo_keypoint*)kps, n, disparity, w, h, dstride, T_c_w, c->p.cam, xyz_w, valid, reliable) : 0;
    if (n_valid) *n_valid = k;
    return VSLAM_OK;
}

int vslam_triangulate(vslam_ctx* c, const float* uvL, const float* uvR, int n, const double T_c_w[7], float* xyz_w, uint8_t* valid, uint8_t* reliable,
                      int* n_valid) {
    if (!c || n < 0 || !T_c_w) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    const int k = n ? vo_triangulate_dlt(uvL, uvR, n, T_c_w, c->p.cam, c->p.stereo_row_tol, xyz_w, valid, reliable) : 0;
    if (n_valid) *n_valid = k;
    return VSLAM_OK;
}

int vslam_pnp_motion_only(vslam_ctx* c, const float* xyz_w, const float* uv, int n, double T_c_w[7], int iters, uint8_t* inlier, int* n_inliers,
                          vslam_lm_stats* stats) {
    if (!c || !xyz_w || !uv || n <= 0 || !T_c_w || iters < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    uint8_t* tmp = inlier ? NULL : (uint8_t*)malloc((size_t)n);
    const int k = vo_pnp_motion_only(xyz_w, uv, n, c->p.cam, T_c_w, iters, c->p.huber_delta, c->p.pnp_reproj_thr, inlier ? inlier : tmp, (vo_lm_stats*)stats);
    free(tmp);
    if (n_inliers) *n_inliers = k < 0 ? 0 : k;
    return VSLAM_OK;
}

int vslam_pnp_ransac(vslam_ctx* c, const float* xyz_w, const float* uv, int n, double T_c_w[7], int max_iters, double reproj_err, double confidence,
                     int lm_iters, uint8_t* inlier, int* n_inliers, int* iters_run) {
    if (!c || !xyz_w || !uv || n < 0 || !T_c_w) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    const int k = vo_pnp_ransac(xyz_w, uv, n, c->p.cam, T_c_w, max_iters, reproj_err, confidence, lm_iters, inlier, iters_run);
    if (n_inliers) *n_inliers = k;
    return VSLAM_OK;
}

int vslam_check_motion(int num_inliers, const double T_c_l[7], double frame_gap) { return T_c_l ? vo_check_motion(num_inliers, T_c_l, frame_gap) : 0; }

/* optimize_map / optimize_pose_only: the optimiser, then the chi2 classification of optimization.cpp:224-266 in the caller's edge
 * order with the caller's flag_lm (the same contract as the product's host wrapper) */
static int window(vslam_ctx* c, int pose_only, int n_kf, double* T, int n_lm, float* xyz, int n_edge, const int32_t* kf_idx, const int32_t* lm_idx,
                  const float* uv, const double* K4, const int32_t* flag_lm, int iters, int update_poses, int update_lms, uint8_t* lm_inlier,
                  double* chi2_out, double* thr_out, vslam_lm_stats* stats) {
    if (!c || n_kf <= 0 || n_kf > VSLAM_MAX_KF || !T || n_lm <= 0 || !xyz || n_edge <= 0 || !kf_idx || !lm_idx || !uv || iters < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    const double* K = K4 ? K4 : c->p.cam;
    double* chi2 = (double*)calloc((size_t)n_edge, sizeof(double));
    int rc = pose_only ? vo_pose_only_window(n_kf, T, n_lm, xyz, n_edge, kf_idx, lm_idx, uv, K, iters, c->p.huber_delta, update_poses, chi2, (vo_lm_stats*)stats)
                       : vo_local_ba(n_kf, T, n_lm, xyz, n_edge, kf_idx, lm_idx, uv, K, iters, c->p.huber_delta, update_poses, update_lms, chi2, (vo_lm_stats*)stats);
    if (rc) { free(chi2); set_error("oracle rejected the graph (%d)", rc); return VSLAM_ERR_ARG; }
    uint8_t* scratch = lm_inlier ? NULL : (uint8_t*)calloc((size_t)n_lm, 1);
    const double th = vo_chi2_classify(chi2, n_edge, flag_lm ? flag_lm : lm_idx, lm_inlier ? lm_inlier : scratch, n_lm, NULL, NULL);
    free(scratch);
    if (thr_out) *thr_out = th;
    if (chi2_out) memcpy(chi2_out, chi2, sizeof(double) * (size_t)n_edge);
    free(chi2);
    return VSLAM_OK;
}

int vslam_local_ba(vslam_ctx* ctx, int n_kf, double* T_c_w, int n_lm, float* xyz, int n_edge, const int32_t* kf_idx, const int32_t* lm_idx,
                   const float* uv, const double* K4, const int32_t* flag_lm, int iters, int update_poses, int update_lms, uint8_t* lm_inlier,
                   double* chi2_out, double* chi2_threshold_out, vslam_lm_stats* stats) {
    return window(ctx, 0, n_kf, T_c_w, n_lm, xyz, n_edge, kf_idx, lm_idx, uv, K4, flag_lm, iters, update_poses, update_lms, lm_inlier, chi2_out,
                  chi2_threshold_out, stats);
}

int vslam_pose_only_window(vslam_ctx* ctx, int n_kf, double* T_c_w, int n_lm, const float* xyz, int n_edge, const int32_t* kf_idx, const int32_t* lm_idx,
                           const float* uv, const double* K4, const int32_t* flag_lm, int iters, int update_poses, uint8_t* lm_inlier, double* chi2_out,
                           double* chi2_threshold_out, vslam_lm_stats* stats) {
    return window(ctx, 1, n_kf, T_c_w, n_lm, (float*)xyz, n_edge, kf_idx, lm_idx, uv, K4, flag_lm, iters, update_poses, 0, lm_inlier, chi2_out,
                  chi2_threshold_out, stats);
}
