// vslam_internal.h -- private declarations shared by the HIP translation units of libvslam_hip.so.
// gfx950 (MI355X / CDNA4) only: wave64, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vslam_hip.h"

namespace vslam {

void set_error(const char* fmt, ...);

#define VS_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            ::vslam::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return VSLAM_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

// ----------------------------------------------------------------------------------------------- stage profiler
// Optional hipEvent brackets around every kernel family (SURVEY.md section 5 "Tracing / profiling").  Enabled per
// context by vslam_profile_enable(); launch functions call PROF_BEGIN/PROF_END, which are no-ops when disabled.
struct Prof;
Prof* prof_current();
void prof_set_current(Prof* p);
void prof_begin(hipStream_t s, const char* name, int launches);
void prof_end(hipStream_t s);
struct ProfScope {
    hipStream_t s;
    ProfScope(hipStream_t st, const char* name, int launches = 1) : s(st) { prof_begin(s, name, launches); }
    ~ProfScope() { prof_end(s); }
};

constexpr int kWave = 64;
constexpr int kNLevels = VSLAM_ORB_NLEVELS;
constexpr int kEdge = 31;            // ORB edgeThreshold
constexpr int kMaxRows = 4096;       // max descriptors per matcher item / keypoints per image through ANMS

// ----------------------------------------------------------------------------------------------- ORB
struct OrbLevel {
    int w, h;          // level size
    float scale;       // (float)pow(1.2, l)
    int nfeat;         // per-level budget
    int pyr_off;       // byte offset of this level inside one image's pyramid buffer (level 0: unused)
    int corner_cap;    // capacity of the FAST corner list of this level
    int corner_off;    // offset (entries) of this level's corner list inside one image's corner buffer
    int tiles_x, tiles_y, tile_off; // FAST tiling of this level
    int tab_off;       // offset into the resize tables (xofs/ialpha) for this level
};

struct OrbPlan {
    OrbLevel lv[kNLevels];
    int w, h;
    int pyr_bytes;        // per image, levels 1..7
    int corner_total;     // per image
    int total_tiles;      // FAST tiles per image, all levels
    int sel_cap;          // per (image, level) capacity of the selected keypoint staging list
    int blur_off[kNLevels]; // byte offset of each level inside one image's blurred pyramid (level 0 included)
    int blur_bytes;       // per image
};

struct Ctx;

// resize tables (host-computed, OpenCV arithmetic) live in device memory: per level xofs[w], ialpha[2w], yofs[h], ibeta[2h]
struct OrbTables {
    int* d_xofs;      // concatenated over levels 1..7
    short* d_ialpha;
    int* d_yofs;
    short* d_ibeta;
    int x_off[kNLevels], y_off[kNLevels];
    // orb_pyrblur_kernel: source level l is cut into 256 x 64 tiles; tile column tx OWNS the output columns dx of level l + 1 whose
    // left source pixel xofs[dx] falls into it: [tile_dx[tdx_off[l] + tx], tile_dx[tdx_off[l] + tx + 1]); rows likewise
    int* d_tile_dx;
    int* d_tile_dy;
    int tdx_off[kNLevels], tdy_off[kNLevels];
};

void orb_debug_enable();
void orb_debug_dump(hipStream_t stream);
int orb_plan_init(OrbPlan* plan, int w, int h, int nfeatures, int kp_capacity);
int orb_tables_init(const OrbPlan* plan, OrbTables* t);
void orb_tables_free(OrbTables* t);

struct OrbBuffers {
    uint8_t* d_pyr;        // B x pyr_bytes
    uint32_t* d_corners;   // B x corner_total packed (x | y<<12 | score<<24)
    int32_t* d_corner_cnt; // B x 8
    vslam_keypoint* d_sel; // B x 8 x sel_cap  (per level, raster order, level coords scaled to L0, with angle)
    int32_t* d_sel_cnt;    // B x 8
    int32_t* d_status;     // B  (bit flags: overflow conditions)
    vslam_keypoint* d_det; // B x kp_capacity: detect output (level asc, raster) when ANMS is run as a separate call
    uint8_t* d_blur;       // B x blur_bytes: Gaussian-blurred pyramid (rBRIEF samples these)
    float2* d_cs;          // B x kp_capacity: per-keypoint (cos, sin) of the rBRIEF rotation
    int32_t* d_order;      // B x kp_capacity: the output slots in (octave, raster) order -- the walk order of orient / describe
    double* d_rad;         // B x kMaxRows: ANMS suppression radii (f64)
};

// launches (all asynchronous on `stream`)
int launch_orb_pyramid(const OrbPlan& plan, const OrbTables& tab, const uint8_t* d_imgs, size_t img_bytes, int pitch,
                       int B, uint8_t* d_pyr, hipStream_t stream);
// pyramid level l + 1 AND the blurred level l from ONE staging of the level-l tile (8 launches: the levels depend on each other)
// [r6] ... and, with d_corners != nullptr, FAST + NMS of level l from the same tile (then launch_orb_fast is not called)
int launch_orb_pyrblur(const OrbPlan& plan, const OrbTables& tab, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, uint8_t* d_pyr,
                       uint8_t* d_blur, int fast_thr, uint32_t* d_corners, int32_t* d_corner_cnt, int32_t* d_status, hipStream_t stream);
int launch_orb_fast(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr,
                    int fast_thr, uint32_t* d_corners, int32_t* d_corner_cnt, int32_t* d_status, hipStream_t stream);
int launch_orb_select(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr,
                      const uint32_t* d_corners, const int32_t* d_corner_cnt, vslam_keypoint* d_sel, int32_t* d_sel_cnt,
                      int32_t* d_status, hipStream_t stream);
// gather per-level lists (level asc) -> ANMS(num) -> regroup by octave -> d_kps (B x kp_capacity), d_count
// anms_num <= 0: no ANMS (detect order).  regroup: apply cv::ORB::compute's border cull + octave regrouping.
int launch_orb_anms(const OrbPlan& plan, int B, const vslam_keypoint* d_sel, const int32_t* d_sel_cnt, int sel_cap,
                    int anms_num, int regroup, vslam_keypoint* d_kps, float2* d_cs, int32_t* d_order, int kp_capacity, int32_t* d_count, int32_t* d_status,
                    double* d_rad, hipStream_t stream);
// same ANMS kernel on a flat list per image (d_in: B x in_capacity, d_nin[b]) for the stand-alone vslam_anms call
int launch_anms_flat(int B, const vslam_keypoint* d_in, const int32_t* d_nin, int in_capacity, int anms_num, int regroup,
                     int img_w, int img_h, vslam_keypoint* d_kps, float2* d_cs, int32_t* d_order, int kp_capacity, int32_t* d_count, int32_t* d_status,
                     double* d_rad, hipStream_t stream);
int launch_orb_blur(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr, uint8_t* d_blur,
                    hipStream_t stream);
int launch_orb_orient(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr, vslam_keypoint* d_kps,
                      float2* d_cs, const int32_t* d_order, int kp_capacity, const int32_t* d_count, hipStream_t stream);
int launch_orb_describe(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr,
                        const uint8_t* d_blur, const vslam_keypoint* d_kps, const float2* d_cs, const int32_t* d_order, int kp_capacity, const int32_t* d_count,
                        uint8_t* d_desc, hipStream_t stream);

// ----------------------------------------------------------------------------------------------- matcher
struct MatchBuffers {
    uint32_t* d_train_best; // B x max_rows : (dist << 16 | query index) per train row
};
int launch_match(const uint8_t* d_q, size_t q_stride, const int32_t* d_nq, const uint8_t* d_t, size_t t_stride,
                 const int32_t* d_nt, const double* d_gap, int gate, double ratio, double gap_thr, int B, int max_rows,
                 uint32_t* d_train_best, vslam_dmatch* d_out, int out_capacity, int32_t* d_nout, hipStream_t stream);

// ----------------------------------------------------------------------------------------------- geometry
struct CamParams { double fx, fy, cx, cy, b, dmin, dmax, drel, row_tol; };
int launch_find3d_disparity(const vslam_keypoint* d_kps, int n, const float* d_disp, int w, int h, int dstride,
                            const double* d_T, CamParams cam, float* d_xyz, uint8_t* d_valid, uint8_t* d_rel, hipStream_t stream);
int launch_find3d_disparity_batch(const vslam_keypoint* d_kps, const int32_t* d_n, int kp_capacity, int B, const float* d_disp, int w, int h,
                                  const double* d_T, CamParams cam, float* d_xyz, uint8_t* d_valid, uint8_t* d_rel, hipStream_t stream);
int launch_triangulate(const float* d_uvL, const float* d_uvR, const int32_t* d_n, int capacity, int B, const double* d_T,
                       CamParams cam, float* d_xyz, uint8_t* d_valid, uint8_t* d_rel, hipStream_t stream);
// f2f matches + per-keypoint 3-D points of the query frame -> compact (xyz, uv) PnP inputs (ordered, valid only)
int launch_build_pnp_inputs(const vslam_dmatch* d_m, const int32_t* d_nm, int match_capacity, const vslam_dmatch* d_lr,
                            const int32_t* d_nlr, int lr_capacity, const float* d_xyz_lr, const uint8_t* d_valid_lr,
                            const vslam_keypoint* d_kpsT, int kp_capacity, int B, int32_t* d_kp2lr, float* d_xyz_out, float* d_uv_out,
                            int32_t* d_nout, int out_capacity, hipStream_t stream);
int launch_gather_uv(const vslam_keypoint* d_kpsQ, const vslam_keypoint* d_kpsT, int kp_capacity, const vslam_dmatch* d_m,
                     const int32_t* d_nm, int match_capacity, int B, float* d_uvQ, float* d_uvT, hipStream_t stream);

int launch_hbm_copy_probe(const void* src, void* dst, size_t bytes, int variant, hipStream_t stream);
int hbm_copy_probe_variants();
const char* hbm_copy_probe_name(int variant);

// ----------------------------------------------------------------------------------------------- LM
struct LmWindowArgs {
    int n_windows, n_kf;
    const int32_t* n_kf_w;  // n_windows: keyframes of window w (<= n_kf, the pose stride), or null
    const int32_t* lm_off;
    const int32_t* edge_off;
    double* T;              // n_windows x n_kf x 7
    float* xyz;             // total_lm x 3
    const uint8_t* reliable;
    uint8_t* lm_inlier;
    const int32_t* kf_idx;
    const int32_t* lm_idx;
    const float* uv;
    double* chi2;
    vslam_lm_stats* stats;
    // scratch (sized by total_lm / total_edge)
    double* P;      // total_lm x 3   current landmark estimates
    double* Ptrial; // total_lm x 3
    double* Hll;    // total_lm x 6 (only the first trial of a call stores it: its lambda comes out of the same pass)
    double* bl;     // total_lm x 3
    double* Dinv;   // total_lm x 6
    double* lin;    // total_edge x 2 : Huber weight per edge at the current / at the trial state, keyframe-major
    int32_t* lm_ptr;   // total_lm + n_windows (CSR by landmark, window-local edge ids, built in-kernel)
    int32_t* kf_ptr;   // n_windows x (MAX_KF + 1)
    int32_t* kf_edges; // total_edge: landmark id of the edge stored at keyframe-major position j
    int32_t* pair_ptr; // n_windows x (NPAIR + 1)
    int32_t* pair_hits;// Schur hit records of the off-diagonal keyframe pairs: {pos1 | pos2 << 16, landmark} (8 B each)
    int32_t hit_capacity_per_edge; // hits of a window <= n_edges_w * hit_capacity_per_edge
    double K[4];
    double huber_delta;
    double* chi2_thr; // n_windows: final adaptive threshold of the last pass
    size_t total_lm, total_edge;
};
// mode 0 = optimize_map (EdgeProjection + Schur), 1 = optimize_pose_only.
// sgbm_kernels.hip.  *scratch / *scratch_bytes: caller-owned growable device buffer.
// Kernel-choice overrides of a context (tuning aid / tests): -1 = the library's batch-size rule.  Seeded ONCE at vslam_create from the
// VSLAM_* environment variables of the same names (validated there), changed afterwards only through vslam_set_tuning -- no getenv on
// the call path (it races with a host that mutates its environment).
struct Tuning {
    int orb_fuse_min = -1;    // VSLAM_ORB_FUSE_MIN: images per call from which orb_pyrblur_kernel replaces resize + blur
    int sgbm_fuse_min = -1;   // VSLAM_SGBM_FUSE_MIN: pairs per call from which sgbm_down_kernel replaces hsum / vsum / path<0,1>
    int sgbm_fwd_min = -1;    // VSLAM_SGBM_FWD_MIN: pairs per call from which sgbm_forward_kernel replaces three path kernels
    int sgbm_fw_rows = -1;    // VSLAM_SGBM_FW_ROWS: 32 or 64 image rows per slab of the forward sweep
    int pose_only_window = -1; // VSLAM_POSE_ONLY_WINDOW: 1 = the schedule's pose-only pass on lm_window_kernel instead of pose_only_wave_kernel
    int pnp_window = -1;      // VSLAM_PNP_WINDOW: 1 = single-pose problems on lm_window_kernel<pnp> instead of pnp_wave_kernel
    int ba_lanes = -1;        // VSLAM_BA_LANES: 256 | 512 = lanes per window of ba_resident_kernel (default: 512 when a launch has at most as many windows as the device has CUs, else 256 -- two windows per CU; same bits either way)
    int ba_resident = -1;     // VSLAM_BA_RESIDENT: 0 = optimize_map windows always on lm_window_kernel; 1 = on ba_resident_kernel whenever they fit its LDS budget;
                              // default: ba_resident_kernel for the windows that fit and have at most 2.2 observations per landmark (the windows of a real sequence)
    int track_rule = -1;      // VSLAM_TRACK_RULE: which frame-to-frame matches continue a track in vslam_build_windows_dev.  1 (default) = the reference's tracking()
                              // (visual_odometry.cpp:568-599): whenever the last-frame keypoint is a feature -- created there or tracked into it; one without a depth of its own
                              // is judged by the pose stage's inlier rule on its landmark's map position (:260-270, :277).  0 = the convention of rounds 4-5: only when the
                              // last-frame keypoint owns a valid depth (kept for before / after measurements)
    int ba_adaptive = -1;     // VSLAM_BA_ADAPTIVE: 0 = the BA schedule runs all three optimize_map passes for every window (default: a pass that flags nothing new is continued instead of repeated)
};
int launch_sgbm(const Tuning& tune, const uint8_t* d_left, const uint8_t* d_right, size_t img_bytes, int pitch, int w, int h, int B, float* d_disp_f32,
                int16_t* d_disp_i16, int16_t* d_disp_raw, uint8_t** scratch, size_t* scratch_bytes, size_t* dev_bytes, hipStream_t stream);
size_t sgbm_scratch_bytes(int w, int h, int B);

// device scratch of the LM kernels: owned by the context (two contexts / streams must not share it), grown on demand
struct LmScratch {
    void* buf = nullptr; size_t bytes = 0;
    int32_t* status = nullptr; int status_n = 0;
    int32_t* passes = nullptr; // optimize_map passes executed per window by the most recent schedule (lm_fetch_passes)
    const Tuning* tune = nullptr; // the owning context's overrides
    bool lds_opt_in = false; // the > 64 KB dynamic-LDS attribute of lm_window_kernel has been set on this context's device
    bool rs_opt_in = false;  // ... of ba_resident_kernel
    int rs_dyn_bytes = -1;   // dynamic LDS a ba_resident_kernel workgroup may use on this context's device (-1: not queried yet)
    int32_t* defer = nullptr; // per window of the most recent launch: 1 = left to lm_window_kernel by ba_resident_kernel
    bool defer_valid = false; // ... written by that launch (it involved ba_resident_kernel)
};
// ba_resident.hip: optimize_map / the BA schedule on windows whose landmark state fits the LDS of one CU (the rest is marked in `defer`)
struct RsLaunch {
    LmWindowArgs a;
    void* uv_s; int32_t* epos; double* tab; double* xin; double* Pbak; double* Dc; double* blc; // scratch slices (see RsArgs)
    int32_t* status; int32_t* passes; int32_t* defer; const int32_t* order; long long* dbg;
    int dyn_bytes, schedule, adaptive, iters, update_poses, update_lms, dense_to_general;
    int want_chi2;           // the CALLER asked for per-edge chi2 (L.a.chi2 is never null here: carve() substitutes scratch)
    int lanes;               // 256 | 512 forces a width of ba_resident_kernel (Tuning::ba_lanes); 0: by the number of windows in the launch
    bool opt_in_done;
};
int rs_dyn_lds_bytes(int device);
int launch_ba_resident(const RsLaunch& L, hipStream_t stream);
int launch_lm_windows(const LmWindowArgs& a, int schedule, int mode, int iters, int update_poses, int update_lms, LmScratch* scratch,
                      hipStream_t stream);
size_t lm_hits_per_edge();
int lm_fetch_status(const LmScratch* scratch, int n_windows, int32_t* h_status, hipStream_t stream);
int lm_fetch_passes(const LmScratch* scratch, int n_windows, int32_t* h_passes, hipStream_t stream);
int lm_fetch_deferred(const LmScratch* scratch, int n_windows, int32_t* h_defer, hipStream_t stream);

struct PnpArgs {
    const float* xyz; const float* uv; const int32_t* n; int capacity; int B;
    double* T; int iters; double K[4]; double huber_delta; double reproj_thr;
    uint8_t* inlier; int32_t* n_inliers; vslam_lm_stats* stats;
    int n_hint; // points per problem when the host knows it (0 = unknown): picks the kernel in launch_pnp
};
int launch_pnp(const PnpArgs& a, LmScratch* scratch, hipStream_t stream);
int launch_edge_jacobians(int n, const float* d_xyz, const float* d_uv, const double* d_T, const double K[4], double delta, double* d_err, double* d_Jp, double* d_Jl,
                          double* d_chi2, double* d_hw, hipStream_t stream);
// pnp_kernels.hip: EPnP of H 5-point subsets (one wave each) -> R|t (H x 12), pose (H x 7), ok flag; f32 inlier scoring of hypotheses
size_t pnp_epnp_ws_bytes(int H);
int launch_pnp_epnp(const float* d_hx, const float* d_hu, int H, const double K[4], double* d_Rt, double* d_T, int32_t* d_ok, uint8_t* ws, hipStream_t stream);
int launch_pnp_count_inliers(const float* d_xyz, const float* d_uv, int n, const double* d_Rt, const int32_t* d_ok, int hyp0, int n_hyp, const double K[4],
                             double reproj_thr, int32_t* d_counts, uint8_t* d_mask, hipStream_t stream);

size_t pnp_ransac_scratch_bytes(int B, int H);
int launch_pnp_ransac_batch(const float* d_xyz, const float* d_uv, const int32_t* d_n, int capacity, int B, int H, const double K[4], double reproj_err,
                            double confidence, uint8_t* scratch, double* d_T, uint8_t* d_inlier, int32_t* d_n_inl, int32_t* d_iters, hipStream_t stream);
// track_kernels.hip: BA windows of a batch of consecutive keyframes from the front end's device-resident output
size_t track_scratch_bytes(int B, int kp_cap, int lm_capacity);
// K4 = {fx, fy, cx, cy}, reproj_thr (pixels), track_rule: see Tuning::track_rule
int launch_build_windows(const vslam_tracks_in& in, int n_kf, int lm_capacity, int edge_capacity, const double K4[4], double reproj_thr, int track_rule, uint8_t* scratch,
                         int32_t* d_lm_off, int32_t* d_edge_off, int32_t* d_n_kf, double* d_T, float* d_xyz_out, uint8_t* d_rel_out, uint8_t* d_inl_out,
                         int32_t* d_kf_out, int32_t* d_lm_out, float* d_uv_out, int32_t* d_status, hipStream_t stream);

// ----------------------------------------------------------------------------------------------- context
struct Ctx {
    vslam_params p;
    int device;
    hipStream_t stream;
    bool own_stream;
    size_t dev_bytes;
    OrbPlan plan;
    OrbTables tab;
    OrbBuffers orb;
    MatchBuffers match;
    // staging for the host-buffer API (sized for one item)
    uint8_t* d_stage; size_t stage_bytes;
    uint8_t* h_pinned; size_t pinned_bytes;
    // SGBM working set (cost volumes; grown on demand by vslam_disparity_map*)
    uint8_t* d_sgbm; size_t sgbm_bytes;
    bool sgbm_unchecked;  // an asynchronous SGBM launch whose error word nobody has read yet (vslam_sync / vslam_sgbm_status_dev do)
    uint8_t* d_ransac; size_t ransac_bytes; // hypothesis tables of vslam_pnp_ransac_dev (grown on demand)
    uint8_t* d_track; size_t track_bytes; // chain tables of vslam_build_windows_dev (grown on demand)
    LmScratch lm;
    Tuning tune;
    Prof* prof;           // stage profiler of this context (vslam_profile_enable); null until first enabled
};

} // namespace vslam
