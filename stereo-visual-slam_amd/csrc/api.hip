// api.hip -- the C-ABI of libvslam_hip.so (declared in include/vslam_hip.h).
// Host-buffer entry points (`vslam_*`) stage through a growable device arena and call the same launches as the
// device-resident batched entry points (`vslam_*_dev`).  No CPU fallback anywhere: without a HIP device every
// compute call fails with VSLAM_ERR_HIP / VSLAM_ERR_NO_DEVICE.
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <cfloat>
#include <cmath>
#include <vector>

#include "se3_device.h"
#include "vslam_internal.h"

#include <mutex>

namespace vslam {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- stage profiler (hipEvent brackets; see vslam_internal.h)
struct Prof {
    struct Rec { const char* name; int launches; hipEvent_t e0, e1; };
    bool on = false;
    std::vector<Rec> recs, open;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
static thread_local Prof* g_prof = nullptr;
// time origin shared by the brackets of every context ON ONE DEVICE (vslam_profile_intervals; events of different devices have no common clock):
// one event per device, created under a mutex by the first context of that device that profiles
static hipEvent_t g_prof_origin[16] = {nullptr};
static std::mutex g_prof_origin_mu;
Prof* prof_current() { return g_prof; }
void prof_set_current(Prof* p) { g_prof = p; }
void prof_begin(hipStream_t s, const char* name, int launches) {
    Prof* p = g_prof;
    if (!p || !p->on) return;
    Prof::Rec r{name, launches, p->get(), p->get()};
    (void)hipEventRecord(r.e0, s);
    p->open.push_back(r);
}
void prof_end(hipStream_t s) {
    Prof* p = g_prof;
    if (!p || !p->on || p->open.empty()) return;
    Prof::Rec r = p->open.back();
    p->open.pop_back();
    (void)hipEventRecord(r.e1, s);
    p->recs.push_back(r);
}
// every entry point makes its context's device current and routes the stage profiler to that context's recorder
#define VS_ENTER(c)                                                                         \
    do {                                                                                    \
        VS_HIP(hipSetDevice((c)->device));                                                  \
        prof_set_current(((c)->prof && (c)->prof->on) ? (c)->prof : nullptr);               \
    } while (0)

// simple bump arena over one growable device allocation (host-buffer API only)
struct Arena {
    Ctx* c; size_t off;
    explicit Arena(Ctx* ctx) : c(ctx), off(0) {}
};
static int arena_reserve(Ctx* c, size_t bytes) {
    if (c->stage_bytes >= bytes) return VSLAM_OK;
    VS_HIP(hipStreamSynchronize(c->stream));
    if (c->d_stage) { hipFree(c->d_stage); c->dev_bytes -= c->stage_bytes; }
    c->d_stage = nullptr; c->stage_bytes = 0;
    const size_t want = std::max(bytes, (size_t)1 << 20);
    VS_HIP(hipMalloc((void**)&c->d_stage, want));
    c->stage_bytes = want; c->dev_bytes += want;
    return VSLAM_OK;
}
template <typename T>
static T* arena_take(Arena& a, size_t n) {
    a.off = (a.off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(a.c->d_stage + a.off);
    a.off += n * sizeof(T);
    return p;
}
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static CamParams cam_of(const Ctx* c) {
    CamParams cam;
    cam.fx = c->p.cam[0]; cam.fy = c->p.cam[1]; cam.cx = c->p.cam[2]; cam.cy = c->p.cam[3]; cam.b = c->p.cam[4];
    cam.dmin = c->p.depth_min; cam.dmax = c->p.depth_max; cam.drel = c->p.depth_reliable; cam.row_tol = c->p.stereo_row_tol;
    return cam;
}

template <typename T>
static int dev_alloc(Ctx* c, T** p, size_t n) {
    VS_HIP(hipMalloc((void**)p, n * sizeof(T)));
    c->dev_bytes += n * sizeof(T);
    return VSLAM_OK;
}

// ---- kernel-choice overrides (struct Tuning): table of {name, environment variable, field, allowed range}
struct TuneKey { const char* name; const char* env; int Tuning::*field; int lo, hi; };
static const TuneKey kTuneKeys[] = {
    {"orb_fuse_min", "VSLAM_ORB_FUSE_MIN", &Tuning::orb_fuse_min, 0, 1 << 30},
    {"sgbm_fuse_min", "VSLAM_SGBM_FUSE_MIN", &Tuning::sgbm_fuse_min, 0, 1 << 30},
    {"sgbm_fwd_min", "VSLAM_SGBM_FWD_MIN", &Tuning::sgbm_fwd_min, 0, 1 << 30},
    {"sgbm_fw_rows", "VSLAM_SGBM_FW_ROWS", &Tuning::sgbm_fw_rows, 32, 64},
    {"pose_only_window", "VSLAM_POSE_ONLY_WINDOW", &Tuning::pose_only_window, 0, 1},
    {"ba_resident", "VSLAM_BA_RESIDENT", &Tuning::ba_resident, 0, 1},
    {"pnp_window", "VSLAM_PNP_WINDOW", &Tuning::pnp_window, 0, 1},
    {"ba_adaptive", "VSLAM_BA_ADAPTIVE", &Tuning::ba_adaptive, 0, 1},
    {"ba_lanes", "VSLAM_BA_LANES", &Tuning::ba_lanes, 256, 512},
    {"track_rule", "VSLAM_TRACK_RULE", &Tuning::track_rule, 0, 1},
};
static int tune_set(Tuning& t, const TuneKey& k, long v) {
    if (v == -1) { t.*(k.field) = -1; return VSLAM_OK; } // back to the library's rule
    if (v < k.lo || v > k.hi || (k.field == &Tuning::sgbm_fw_rows && v != 32 && v != 64) || (k.field == &Tuning::ba_lanes && v != 256 && v != 512)) {
        set_error("tuning value %s = %ld out of range (%d..%d%s, or -1 = default)", k.name, v, k.lo, k.hi, k.field == &Tuning::sgbm_fw_rows ? ", 32 or 64" : "");
        return VSLAM_ERR_ARG;
    }
    t.*(k.field) = (int)v;
    return VSLAM_OK;
}
// the environment seeds the overrides ONCE, at context creation; an unparsable value is an error there, not a silent 0
static int tune_from_env(Tuning& t) {
    for (const TuneKey& k : kTuneKeys) {
        const char* e = getenv(k.env);
        if (!e || !*e) continue;
        char* end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end == e || *end != 0) { set_error("environment variable %s = \"%s\" is not an integer", k.env, e); return VSLAM_ERR_ARG; }
        int rc = tune_set(t, k, v);
        if (rc) return rc;
    }
    return VSLAM_OK;
}

static int orb_status_check(Ctx* c, int B) {
    std::vector<int32_t> st(B);
    VS_HIP(hipMemcpyAsync(st.data(), c->orb.d_status, sizeof(int32_t) * B, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    for (int b = 0; b < B; ++b)
        if (st[b]) { set_error("ORB capacity exceeded on image %d (status bits 0x%x)", b, st[b]); return VSLAM_ERR_CAPACITY; }
    return VSLAM_OK;
}

// detect + (ANMS) + (describe) on device-resident images
static int orb_pipeline(Ctx* c, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, int anms_num, int regroup, bool describe,
                        vslam_keypoint* d_kps, uint8_t* d_desc, int32_t* d_count) {
    if (B <= 0) return VSLAM_OK;
    if (B > c->p.max_batch) { set_error("batch %d exceeds context max_batch %d", B, c->p.max_batch); return VSLAM_ERR_ARG; }
    int rc;
    // With descriptors wanted and a LARGE batch, every level is staged once for both its successor and its blurred copy
    // (orb_pyrblur_kernel: 3 % faster at 512 images, a third less pyramid traffic).  Small batches keep the separate kernels: eight
    // dependent launches that each do resize AND blur are slower than seven short resize launches + one blur launch over all levels
    // (0.36 vs 0.43 ms for two images).  [r6] With FAST inside the tile pass and the blur on the matrix cores the fused kernel wins from ~64 images on
    // (128 images: 0.89 vs 0.96 ms, 256: 1.35 vs 1.55; round 5's break-even was ~300).  Tuning::orb_fuse_min overrides the threshold (tests run both paths).
    const bool fused = describe && B >= (c->tune.orb_fuse_min >= 0 ? c->tune.orb_fuse_min : 96);
    if (fused) { // [r6] ... and FAST + NMS of every level from the same staged tile
        if ((rc = launch_orb_pyrblur(c->plan, c->tab, d_imgs, img_bytes, pitch, B, c->orb.d_pyr, c->orb.d_blur, c->p.fast_threshold, c->orb.d_corners,
                                     c->orb.d_corner_cnt, c->orb.d_status, c->stream))) return rc;
    } else {
        if ((rc = launch_orb_pyramid(c->plan, c->tab, d_imgs, img_bytes, pitch, B, c->orb.d_pyr, c->stream))) return rc;
        if ((rc = launch_orb_fast(c->plan, d_imgs, img_bytes, pitch, B, c->orb.d_pyr, c->p.fast_threshold, c->orb.d_corners,
                                  c->orb.d_corner_cnt, c->orb.d_status, c->stream))) return rc;
    }
    if ((rc = launch_orb_select(c->plan, d_imgs, img_bytes, pitch, B, c->orb.d_pyr, c->orb.d_corners, c->orb.d_corner_cnt, c->orb.d_sel,
                                c->orb.d_sel_cnt, c->orb.d_status, c->stream))) return rc;
    if ((rc = launch_orb_anms(c->plan, B, c->orb.d_sel, c->orb.d_sel_cnt, c->plan.sel_cap, anms_num, regroup, d_kps, nullptr, c->orb.d_order, c->p.kp_capacity,
                              d_count, c->orb.d_status, c->orb.d_rad, c->stream))) return rc;
    // orientation (and the rBRIEF rotation) only for the keypoints the ANMS kept
    if ((rc = launch_orb_orient(c->plan, d_imgs, img_bytes, pitch, B, c->orb.d_pyr, d_kps, c->orb.d_cs, c->orb.d_order, c->p.kp_capacity, d_count, c->stream))) return rc;
    if (describe) {
        if (!fused && (rc = launch_orb_blur(c->plan, d_imgs, img_bytes, pitch, B, c->orb.d_pyr, c->orb.d_blur, c->stream))) return rc;
        if ((rc = launch_orb_describe(c->plan, d_imgs, img_bytes, pitch, B, c->orb.d_pyr, c->orb.d_blur, d_kps, c->orb.d_cs, c->orb.d_order, c->p.kp_capacity, d_count,
                                      d_desc, c->stream))) return rc;
    }
    if (getenv("VSLAM_ORB_PROFILE")) orb_debug_dump(c->stream);
    return VSLAM_OK;
}

static int check_img(Ctx* c, const void* img, int w, int h, int stride) {
    if (!c || !img) { set_error("null context or image"); return VSLAM_ERR_ARG; }
    if (w != c->p.img_w || h != c->p.img_h || stride < w) { set_error("image %dx%d (stride %d) does not match the context's %dx%d", w, h, stride, c->p.img_w, c->p.img_h); return VSLAM_ERR_ARG; }
    return VSLAM_OK;
}

} // namespace vslam

using namespace vslam;

extern "C" {

void vslam_default_params(vslam_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->img_w = 1241; p->img_h = 376; p->max_batch = 1;
    p->orb_nfeatures = 3000; p->anms_num = 500; p->fast_threshold = 20; p->kp_capacity = 4096;
    p->cam[0] = 718.856; p->cam[1] = 718.856; p->cam[2] = 607.1928; p->cam[3] = 185.2157; p->cam[4] = 0.573;
    p->depth_min = 10; p->depth_max = 400; p->depth_reliable = 40;
    p->match_ratio = 2.0; p->match_gap_thr = 30.0; p->huber_delta = 5.991; p->pnp_reproj_thr = 4.0; p->stereo_row_tol = 2.0;
    p->struct_size = (int32_t)sizeof(vslam_params); p->abi_version = VSLAM_ABI_VERSION;
}

const char* vslam_last_error(void) { return g_err; }
const char* vslam_version(void) { return "vslam_hip 0.3 (gfx950, ABI 3)"; }
int vslam_abi_version(void) { return VSLAM_ABI_VERSION; }
const char* vslam_kernel_names(void) { // the ProfScope names of csrc/*.hip (tests/test_abi.py checks the list against the sources)
    return "orb_resize_kernel orb_pyrblur_kernel orb_fast_kernel orb_select_kernel orb_anms_kernel orb_orient_kernel orb_blur_kernel orb_describe_kernel "
           "match_train_nearest_kernel match_finalize_kernel sgbm_prefilter_kernel sgbm_down_kernel sgbm_forward_kernel sgbm_hsum_kernel sgbm_vsum_kernel sgbm_path_kernel "
           "sgbm_wta_kernel sgbm_lrcheck_kernel sgbm_median3_kernel sgbm_ccl_rows_kernel sgbm_ccl_union_kernel sgbm_ccl_count_kernel "
           "sgbm_ccl_apply_kernel sgbm_ccl_kernels triangulate_kernel find3d_disparity_kernel gather_uv_kernel build_pnp_inputs_kernel lm_window_kernel pose_only_wave_kernel "
           "lm_window_kernel<pnp> pnp_wave_kernel pnp_inlier_kernel pnp_epnp_kernels epnp_front_kernel epnp_jacobi_kernel epnp_back_kernel pnp_count_inliers_kernel hbm_copy_probe_kernel "
           "pnp_ransac_subsets_kernel pnp_ransac_count_kernel pnp_ransac_select_kernel "
           "build_windows_kernels track_init_kernel track_pose_chain_kernel track_link_kernel track_chain_kernel window_count_kernel window_scan_kernel window_rank_kernel window_emit_kernel";
}

int vslam_create(const vslam_params* p, int device, void* stream, vslam_ctx** out) {
    if (!p || !out) { set_error("null argument"); return VSLAM_ERR_ARG; }
    *out = nullptr;
    // (the ABI guard comes first: nothing else of *p may be read from a struct of another revision, and it needs no device)
    if (p->struct_size != (int32_t)sizeof(vslam_params) || p->abi_version != VSLAM_ABI_VERSION) {
        set_error("vslam_params from a different ABI (struct_size %d / abi_version %d, library has %d / %d): rebuild the caller against include/vslam_hip.h and "
                  "fill the struct with vslam_default_params", p->struct_size, p->abi_version, (int)sizeof(vslam_params), VSLAM_ABI_VERSION);
        return VSLAM_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible (libvslam_hip has no CPU path)"); return VSLAM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (0..%d)", device, ndev - 1); return VSLAM_ERR_ARG; }
    if (p->max_batch <= 0 || p->kp_capacity < 64 || p->kp_capacity > kMaxRows || p->orb_nfeatures <= 0) { set_error("bad params (max_batch>0, 64<=kp_capacity<=%d)", kMaxRows); return VSLAM_ERR_ARG; }
    VS_HIP(hipSetDevice(device));
    Ctx* c = new Ctx(); // (value-initialised: zeroes, then the members' default initialisers -- Tuning's -1, LmScratch's)
    c->p = *p; c->device = device;
    c->lm.tune = &c->tune;
    { int rc_t = tune_from_env(c->tune); if (rc_t) { delete c; return rc_t; } }
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete c; return VSLAM_ERR_HIP; }
        c->own_stream = true;
    }
    if (getenv("VSLAM_ORB_PROFILE")) orb_debug_enable();
    int rc = orb_plan_init(&c->plan, p->img_w, p->img_h, p->orb_nfeatures, p->kp_capacity);
    if (rc == VSLAM_OK) rc = orb_tables_init(&c->plan, &c->tab);
    const size_t B = (size_t)p->max_batch;
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_pyr, B * c->plan.pyr_bytes);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_corners, B * c->plan.corner_total);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_corner_cnt, B * kNLevels);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_sel, B * kNLevels * c->plan.sel_cap);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_sel_cnt, B * kNLevels);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_status, B);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_blur, B * c->plan.blur_bytes);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_cs, B * (size_t)p->kp_capacity);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_order, B * (size_t)p->kp_capacity);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->orb.d_rad, B * (size_t)kMaxRows);
    if (rc == VSLAM_OK) rc = dev_alloc(c, &c->match.d_train_best, B * kMaxRows);
    if (rc != VSLAM_OK) { vslam_destroy(reinterpret_cast<vslam_ctx*>(c)); return rc; }
    *out = reinterpret_cast<vslam_ctx*>(c);
    return VSLAM_OK;
}

void vslam_destroy(vslam_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    orb_tables_free(&c->tab);
    if (c->d_sgbm) hipFree(c->d_sgbm);
    if (c->d_track) hipFree(c->d_track);
    if (c->d_ransac) hipFree(c->d_ransac);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->lm.buf) hipFree(c->lm.buf);
    void* ptrs[] = {c->orb.d_pyr, c->orb.d_corners, c->orb.d_corner_cnt, c->orb.d_sel, c->orb.d_sel_cnt, c->orb.d_status, c->orb.d_det, c->orb.d_blur, c->orb.d_cs, c->orb.d_order, c->orb.d_rad,
                    c->match.d_train_best, c->d_stage};
    for (void* q : ptrs) if (q) hipFree(q);
    if (c->prof) {
        if (prof_current() == c->prof) prof_set_current(nullptr);
        for (Prof::Rec& r : c->prof->recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
        for (hipEvent_t e : c->prof->pool) hipEventDestroy(e);
        delete c->prof;
    }
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

int vslam_sync(vslam_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return VSLAM_ERR_ARG;
    VS_ENTER(c);
    if (c->sgbm_unchecked) { int32_t st = 0; return vslam_sgbm_status_dev(ctx, &st); } // (synchronises; a fired backstop voids the maps: VSLAM_ERR_HIP)
    VS_HIP(hipStreamSynchronize(c->stream));
    return VSLAM_OK;
}

size_t vslam_device_bytes(const vslam_ctx* ctx) { return ctx ? reinterpret_cast<const Ctx*>(ctx)->dev_bytes : 0; }

// ---------------------------------------------------------------------------------------------- ORB, host buffers
// Host image -> device image with a 64-byte row pitch.  The rows are repacked on the host into a pinned staging buffer and
// sent as ONE linear copy: hipMemcpy2DAsync from pageable memory degenerates into a copy per row (measured 2.8 ms for a
// 1241x376 image against 0.03 ms for the same bytes as one block).  Several images of one call use consecutive slots.
static int upload_image(Ctx* c, Arena& ar, const uint8_t* img, int w, int h, int stride, uint8_t** d_img, int* pitch, int slot = 0) {
    *pitch = (w + 63) & ~63;
    const size_t bytes = (size_t)*pitch * h;
    *d_img = arena_take<uint8_t>(ar, bytes);
    const size_t need = bytes * (size_t)(slot + 1);
    if (c->pinned_bytes < need) {
        VS_HIP(hipStreamSynchronize(c->stream));
        if (c->h_pinned) (void)hipHostFree(c->h_pinned);
        c->h_pinned = nullptr; c->pinned_bytes = 0;
        const size_t want = std::max(need, 2 * bytes);
        VS_HIP(hipHostMalloc((void**)&c->h_pinned, want, hipHostMallocDefault));
        c->pinned_bytes = want;
    }
    uint8_t* stage = c->h_pinned + bytes * (size_t)slot;
    for (int y = 0; y < h; ++y) {
        memcpy(stage + (size_t)y * *pitch, img + (size_t)y * stride, (size_t)w);
        memset(stage + (size_t)y * *pitch + w, 0, (size_t)(*pitch - w)); // deterministic padding
    }
    VS_HIP(hipMemcpyAsync(*d_img, stage, bytes, hipMemcpyHostToDevice, c->stream));
    return VSLAM_OK;
}

static int orb_host_call(vslam_ctx* ctx, const uint8_t* img, int w, int h, int stride, int anms_num, int regroup, bool describe,
                         vslam_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    int rc = check_img(c, img, w, h, stride);
    if (rc) return rc;
    if (!kps || !n_out || cap <= 0 || (describe && !desc)) { set_error("null output"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    const int kc = c->p.kp_capacity;
    const size_t pitch = (w + 63) & ~63;
    if ((rc = arena_reserve(c, al256(pitch * h) + al256(sizeof(vslam_keypoint) * kc) + al256((size_t)kc * 32) + 1024))) return rc;
    Arena ar(c);
    uint8_t* d_img; int dp;
    if ((rc = upload_image(c, ar, img, w, h, stride, &d_img, &dp))) return rc;
    vslam_keypoint* d_kps = arena_take<vslam_keypoint>(ar, kc);
    uint8_t* d_desc = arena_take<uint8_t>(ar, (size_t)kc * 32);
    int32_t* d_cnt = arena_take<int32_t>(ar, 1);
    if ((rc = orb_pipeline(c, d_img, (size_t)dp * h, dp, 1, anms_num, regroup, describe, d_kps, d_desc, d_cnt))) return rc;
    int32_t n = 0;
    VS_HIP(hipMemcpyAsync(&n, d_cnt, sizeof(n), hipMemcpyDeviceToHost, c->stream));
    if ((rc = orb_status_check(c, 1))) return rc;
    if (n > cap) { *n_out = 0; set_error("caller capacity %d < %d keypoints", cap, n); return VSLAM_ERR_CAPACITY; }
    VS_HIP(hipMemcpyAsync(kps, d_kps, sizeof(vslam_keypoint) * n, hipMemcpyDeviceToHost, c->stream));
    if (describe) VS_HIP(hipMemcpyAsync(desc, d_desc, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    *n_out = n;
    return VSLAM_OK;
}

int vslam_feature_detection(vslam_ctx* ctx, const uint8_t* img, int w, int h, int stride, vslam_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) { set_error("null context"); return VSLAM_ERR_ARG; }
    return orb_host_call(ctx, img, w, h, stride, c->p.anms_num, 1, true, kps, desc, cap, n_out);
}

int vslam_orb_detect(vslam_ctx* ctx, const uint8_t* img, int w, int h, int stride, vslam_keypoint* kps, int cap, int* n_out) {
    return orb_host_call(ctx, img, w, h, stride, 0, 0, false, kps, nullptr, cap, n_out);
}

int vslam_anms(vslam_ctx* ctx, vslam_keypoint* kps, int n, int num, int* n_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !kps || !n_out || n < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    if (n > kMaxRows) { set_error("ANMS input %d > %d", n, kMaxRows); return VSLAM_ERR_CAPACITY; }
    if (n == 0) { *n_out = 0; return VSLAM_OK; }
    VS_ENTER(c);
    int rc;
    if ((rc = arena_reserve(c, 2 * al256(sizeof(vslam_keypoint) * kMaxRows) + 1024))) return rc;
    Arena ar(c);
    vslam_keypoint* d_in = arena_take<vslam_keypoint>(ar, kMaxRows);
    vslam_keypoint* d_out = arena_take<vslam_keypoint>(ar, kMaxRows);
    int32_t* d_n = arena_take<int32_t>(ar, 1);
    int32_t* d_cnt = arena_take<int32_t>(ar, 1);
    int32_t nn = n;
    VS_HIP(hipMemcpyAsync(d_in, kps, sizeof(vslam_keypoint) * n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_n, &nn, sizeof(nn), hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemsetAsync(c->orb.d_status, 0, sizeof(int32_t), c->stream));
    if ((rc = launch_anms_flat(1, d_in, d_n, kMaxRows, num, 0, c->p.img_w, c->p.img_h, d_out, nullptr, nullptr, kMaxRows, d_cnt, c->orb.d_status, c->orb.d_rad, c->stream))) return rc;
    int32_t m = 0;
    VS_HIP(hipMemcpyAsync(&m, d_cnt, sizeof(m), hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    VS_HIP(hipMemcpy(kps, d_out, sizeof(vslam_keypoint) * m, hipMemcpyDeviceToHost));
    *n_out = m;
    return VSLAM_OK;
}

int vslam_orb_compute(vslam_ctx* ctx, const uint8_t* img, int w, int h, int stride, vslam_keypoint* kps, int n, uint8_t* desc, int* n_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    int rc = check_img(c, img, w, h, stride);
    if (rc) return rc;
    if (!kps || !desc || !n_out || n < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    if (n > c->p.kp_capacity) { set_error("compute input %d > kp_capacity %d", n, c->p.kp_capacity); return VSLAM_ERR_CAPACITY; }
    if (n == 0) { *n_out = 0; return VSLAM_OK; }
    for (int i = 0; i < n; ++i)
        if (kps[i].octave < 0 || kps[i].octave >= kNLevels) { set_error("keypoint %d: octave %d out of range", i, kps[i].octave); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    const int kc = c->p.kp_capacity;
    const size_t pitch = (w + 63) & ~63;
    if ((rc = arena_reserve(c, al256(pitch * h) + 2 * al256(sizeof(vslam_keypoint) * kc) + al256((size_t)kc * 32) + 1024))) return rc;
    Arena ar(c);
    uint8_t* d_img; int dp;
    if ((rc = upload_image(c, ar, img, w, h, stride, &d_img, &dp))) return rc;
    vslam_keypoint* d_in = arena_take<vslam_keypoint>(ar, kc);
    vslam_keypoint* d_kps = arena_take<vslam_keypoint>(ar, kc);
    uint8_t* d_desc = arena_take<uint8_t>(ar, (size_t)kc * 32);
    int32_t* d_n = arena_take<int32_t>(ar, 1);
    int32_t* d_cnt = arena_take<int32_t>(ar, 1);
    int32_t nn = n;
    VS_HIP(hipMemcpyAsync(d_in, kps, sizeof(vslam_keypoint) * n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_n, &nn, sizeof(nn), hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemsetAsync(c->orb.d_status, 0, sizeof(int32_t), c->stream));
    const bool fused = 1 >= (c->tune.orb_fuse_min >= 0 ? c->tune.orb_fuse_min : 384); // (one image: the separate kernels unless a test forces the fused one)
    if (fused) { if ((rc = launch_orb_pyrblur(c->plan, c->tab, d_img, (size_t)dp * h, dp, 1, c->orb.d_pyr, c->orb.d_blur, 0, nullptr, nullptr, nullptr, c->stream))) return rc; }
    else {
        if ((rc = launch_orb_pyramid(c->plan, c->tab, d_img, (size_t)dp * h, dp, 1, c->orb.d_pyr, c->stream))) return rc;
        if ((rc = launch_orb_blur(c->plan, d_img, (size_t)dp * h, dp, 1, c->orb.d_pyr, c->orb.d_blur, c->stream))) return rc;
    }
    if ((rc = launch_anms_flat(1, d_in, d_n, kc, 0, 1, w, h, d_kps, c->orb.d_cs, nullptr, kc, d_cnt, c->orb.d_status, c->orb.d_rad, c->stream))) return rc;
    if ((rc = launch_orb_describe(c->plan, d_img, (size_t)dp * h, dp, 1, c->orb.d_pyr, c->orb.d_blur, d_kps, c->orb.d_cs, nullptr, kc, d_cnt, d_desc, c->stream))) return rc;
    int32_t m = 0;
    VS_HIP(hipMemcpyAsync(&m, d_cnt, sizeof(m), hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    VS_HIP(hipMemcpy(kps, d_kps, sizeof(vslam_keypoint) * m, hipMemcpyDeviceToHost));
    VS_HIP(hipMemcpy(desc, d_desc, (size_t)m * 32, hipMemcpyDeviceToHost));
    *n_out = m;
    return VSLAM_OK;
}

int vslam_feature_detection_dev(vslam_ctx* ctx, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, vslam_keypoint* d_kps,
                                uint8_t* d_desc, int32_t* d_count) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !d_imgs || !d_kps || !d_desc || !d_count || pitch < c->p.img_w || img_bytes < (size_t)pitch * c->p.img_h) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    return orb_pipeline(c, d_imgs, img_bytes, pitch, B, c->p.anms_num, 1, true, d_kps, d_desc, d_count);
}

int vslam_orb_status_dev(vslam_ctx* ctx, int B, int32_t* h_status) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !h_status || B <= 0 || B > c->p.max_batch) return VSLAM_ERR_ARG;
    VS_ENTER(c);
    VS_HIP(hipMemcpyAsync(h_status, c->orb.d_status, sizeof(int32_t) * B, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    return VSLAM_OK;
}

int vslam_orb_level(vslam_ctx* ctx, int item, int level, int blurred, uint8_t* out, int out_stride, int out_rows, int* w, int* h) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !out || !w || !h || item < 0 || item >= c->p.max_batch || level < 0 || level >= kNLevels || (level == 0 && !blurred)) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    const OrbLevel& L = c->plan.lv[level];
    if (out_stride < L.w || out_rows < L.h) { set_error("vslam_orb_level: level %d is %d x %d", level, L.w, L.h); return VSLAM_ERR_CAPACITY; }
    const uint8_t* src = blurred ? c->orb.d_blur + (size_t)item * c->plan.blur_bytes + c->plan.blur_off[level]
                                 : c->orb.d_pyr + (size_t)item * c->plan.pyr_bytes + L.pyr_off;
    VS_HIP(hipStreamSynchronize(c->stream));
    VS_HIP(hipMemcpy2D(out, (size_t)out_stride, src, (size_t)((L.w + 63) & ~63), (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost));
    *w = L.w; *h = L.h;
    return VSLAM_OK;
}

// ---------------------------------------------------------------------------------------------- matcher
int vslam_feature_matching_dev(vslam_ctx* ctx, const uint8_t* d_q, size_t q_stride_bytes, const int32_t* d_nq, const uint8_t* d_t,
                               size_t t_stride_bytes, const int32_t* d_nt, const double* d_gap, int gate, int B, int max_rows,
                               vslam_dmatch* d_out, int out_capacity, int32_t* d_nout) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !d_q || !d_t || !d_nq || !d_nt || !d_gap || !d_out || !d_nout || out_capacity <= 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    if (B > c->p.max_batch) { set_error("batch %d exceeds context max_batch %d", B, c->p.max_batch); return VSLAM_ERR_ARG; }
    if ((((uintptr_t)d_q | (uintptr_t)d_t | (uintptr_t)q_stride_bytes | (uintptr_t)t_stride_bytes) & 15) != 0) {
        set_error("vslam_feature_matching_dev: d_q, d_t and both strides must be multiples of 16 bytes (the matcher reads descriptors with 16-byte loads)");
        return VSLAM_ERR_ARG;
    }
    VS_ENTER(c);
    return launch_match(d_q, q_stride_bytes, d_nq, d_t, t_stride_bytes, d_nt, d_gap, gate, c->p.match_ratio, c->p.match_gap_thr, B, max_rows,
                        c->match.d_train_best, d_out, out_capacity, d_nout, c->stream);
}

int vslam_feature_matching(vslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double frame_gap, int gate,
                           vslam_dmatch* out, int* n_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !n_out || nq < 0 || nt < 0 || (nq > 0 && (!q || !out)) || (nt > 0 && !t)) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    if (nq > kMaxRows || nt > kMaxRows) { set_error("matcher supports at most %d rows per side", kMaxRows); return VSLAM_ERR_CAPACITY; }
    *n_out = 0;
    if (nq == 0 || nt == 0) return VSLAM_OK; // empty set: no matches (reference: UB, quirk Q7)
    VS_ENTER(c);
    int rc;
    const int rows = std::max(nq, nt);
    if ((rc = arena_reserve(c, 2 * al256((size_t)rows * 32) + al256(sizeof(vslam_dmatch) * nq) + 2048))) return rc;
    Arena ar(c);
    uint8_t* d_q = arena_take<uint8_t>(ar, (size_t)nq * 32);
    uint8_t* d_t = arena_take<uint8_t>(ar, (size_t)nt * 32);
    vslam_dmatch* d_out = arena_take<vslam_dmatch>(ar, nq);
    int32_t* d_n = arena_take<int32_t>(ar, 4);
    double* d_gap = arena_take<double>(ar, 1);
    int32_t hn[3] = {nq, nt, 0};
    VS_HIP(hipMemcpyAsync(d_q, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_t, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_n, hn, sizeof(hn), hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_gap, &frame_gap, sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_match(d_q, 0, d_n, d_t, 0, d_n + 1, d_gap, gate, c->p.match_ratio, c->p.match_gap_thr, 1, rows, c->match.d_train_best,
                           d_out, nq, d_n + 2, c->stream))) return rc;
    int32_t m = 0;
    VS_HIP(hipMemcpyAsync(&m, d_n + 2, sizeof(m), hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    if (m > 0) VS_HIP(hipMemcpy(out, d_out, sizeof(vslam_dmatch) * m, hipMemcpyDeviceToHost));
    *n_out = m;
    return VSLAM_OK;
}

// ---------------------------------------------------------------------------------------------- geometry
// ---------------------------------------------------------------------------------------------- SGBM
int vslam_disparity_map_dev(vslam_ctx* ctx, const uint8_t* d_left, const uint8_t* d_right, size_t img_stride_bytes, int pitch, int w, int h,
                            int B, float* d_disparity, int16_t* d_disp_i16, int16_t* d_disp_raw_i16) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !d_left || !d_right || w <= 0 || h <= 0 || pitch < w || B < 0 || img_stride_bytes < (size_t)pitch * h ||
        (!d_disparity && !d_disp_i16 && !d_disp_raw_i16)) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    c->sgbm_unchecked = true; // (asynchronous: the forward sweep's error word is looked at by the next vslam_sync / vslam_sgbm_status_dev)
    return launch_sgbm(c->tune, d_left, d_right, img_stride_bytes, pitch, w, h, B, d_disparity, d_disp_i16, d_disp_raw_i16, &c->d_sgbm, &c->sgbm_bytes,
                       &c->dev_bytes, c->stream);
}

int vslam_disparity_map(vslam_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, int stride, float* disparity, int16_t* disp_i16,
                        int16_t* disp_raw_i16) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !left || !right || w <= 0 || h <= 0 || stride < w || (!disparity && !disp_i16 && !disp_raw_i16)) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    int rc;
    const size_t npix = (size_t)w * h;
    if ((rc = arena_reserve(c, 2 * al256((size_t)((w + 63) & ~63) * h) + al256(npix * 4) + 2 * al256(npix * 2) + 1024))) return rc;
    Arena ar(c);
    uint8_t *d_l, *d_r; int pl, pr;
    if ((rc = upload_image(c, ar, left, w, h, stride, &d_l, &pl))) return rc;
    if ((rc = upload_image(c, ar, right, w, h, stride, &d_r, &pr, 1))) return rc;
    float* d_f = arena_take<float>(ar, npix);
    int16_t* d_i = arena_take<int16_t>(ar, npix);
    int16_t* d_raw = arena_take<int16_t>(ar, npix);
    if ((rc = launch_sgbm(c->tune, d_l, d_r, (size_t)pl * h, pl, w, h, 1, d_f, d_i, disp_raw_i16 ? d_raw : nullptr, &c->d_sgbm, &c->sgbm_bytes, &c->dev_bytes,
                          c->stream))) return rc;
    if (disparity) VS_HIP(hipMemcpyAsync(disparity, d_f, npix * 4, hipMemcpyDeviceToHost, c->stream));
    if (disp_i16) VS_HIP(hipMemcpyAsync(disp_i16, d_i, npix * 2, hipMemcpyDeviceToHost, c->stream));
    if (disp_raw_i16) VS_HIP(hipMemcpyAsync(disp_raw_i16, d_raw, npix * 2, hipMemcpyDeviceToHost, c->stream));
    int32_t st = 0;
    return vslam_sgbm_status_dev(ctx, &st); // synchronises; VSLAM_ERR_HIP if the forward sweep's backstop fired
}

int vslam_set_tuning(vslam_ctx* ctx, const char* name, int value) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !name) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    for (const TuneKey& k : kTuneKeys)
        if (!strcmp(k.name, name)) return tune_set(c->tune, k, value);
    set_error("unknown tuning key \"%s\"", name);
    return VSLAM_ERR_ARG;
}

int vslam_sgbm_status_dev(vslam_ctx* ctx, int32_t* h_status) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !h_status) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    *h_status = 0;
    c->sgbm_unchecked = false;
    if (!c->d_sgbm) { VS_HIP(hipStreamSynchronize(c->stream)); return VSLAM_OK; } // no SGBM launch yet
    // header of the SGBM scratch: int32 [0..7] ticket pools, [8] error word of the most recent launch (sgbm_kernels.hip, launch_sgbm)
    VS_HIP(hipMemcpyAsync(h_status, c->d_sgbm + 32, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    if (*h_status != 0) { set_error("sgbm_forward_kernel: a slab waited for its predecessor beyond the spin limit; the disparity maps of this call are void"); return VSLAM_ERR_HIP; }
    return VSLAM_OK;
}

int vslam_find_3d_disparity(vslam_ctx* ctx, const vslam_keypoint* kps, int n, const float* disparity, int w, int h, int dstride,
                            const double T_c_w[7], float* xyz_w, uint8_t* valid, uint8_t* reliable, int* n_valid) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || n < 0 || !disparity || !T_c_w || w <= 0 || h <= 0 || dstride < w || (n > 0 && (!kps || !xyz_w || !valid || !reliable))) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    if (n_valid) *n_valid = 0;
    if (n == 0) return VSLAM_OK;
    VS_ENTER(c);
    int rc;
    if ((rc = arena_reserve(c, al256(sizeof(vslam_keypoint) * n) + al256(sizeof(float) * (size_t)dstride * h) + al256(12 * (size_t)n) + 2 * al256(n) + 1024))) return rc;
    Arena ar(c);
    vslam_keypoint* d_k = arena_take<vslam_keypoint>(ar, n);
    float* d_d = arena_take<float>(ar, (size_t)dstride * h);
    double* d_T = arena_take<double>(ar, 7);
    float* d_x = arena_take<float>(ar, 3 * (size_t)n);
    uint8_t* d_v = arena_take<uint8_t>(ar, n);
    uint8_t* d_r = arena_take<uint8_t>(ar, n);
    VS_HIP(hipMemcpyAsync(d_k, kps, sizeof(vslam_keypoint) * n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_d, disparity, sizeof(float) * (size_t)dstride * h, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_T, T_c_w, sizeof(double) * 7, hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_find3d_disparity(d_k, n, d_d, w, h, dstride, d_T, cam_of(c), d_x, d_v, d_r, c->stream))) return rc;
    VS_HIP(hipMemcpyAsync(xyz_w, d_x, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipMemcpyAsync(valid, d_v, n, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipMemcpyAsync(reliable, d_r, n, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    if (n_valid) { int k = 0; for (int i = 0; i < n; ++i) k += valid[i] != 0; *n_valid = k; }
    return VSLAM_OK;
}

int vslam_find_3d_disparity_dev(vslam_ctx* ctx, const vslam_keypoint* d_kps, const int32_t* d_n, int kp_capacity, int B, const float* d_disparity,
                                int w, int h, const double* d_T_c_w, float* d_xyz_w, uint8_t* d_valid, uint8_t* d_reliable) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !d_kps || !d_n || !d_disparity || !d_T_c_w || !d_xyz_w || !d_valid || !d_reliable || kp_capacity <= 0 || w <= 0 || h <= 0 || B < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    return launch_find3d_disparity_batch(d_kps, d_n, kp_capacity, B, d_disparity, w, h, d_T_c_w, cam_of(c), d_xyz_w, d_valid, d_reliable, c->stream);
}

int vslam_triangulate_dev(vslam_ctx* ctx, const float* d_uvL, const float* d_uvR, const int32_t* d_n, int capacity, int B,
                          const double* d_T_c_w, float* d_xyz_w, uint8_t* d_valid, uint8_t* d_reliable) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !d_uvL || !d_uvR || !d_n || !d_T_c_w || !d_xyz_w || !d_valid || !d_reliable || capacity <= 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    return launch_triangulate(d_uvL, d_uvR, d_n, capacity, B, d_T_c_w, cam_of(c), d_xyz_w, d_valid, d_reliable, c->stream);
}

int vslam_triangulate(vslam_ctx* ctx, const float* uvL, const float* uvR, int n, const double T_c_w[7], float* xyz_w, uint8_t* valid,
                      uint8_t* reliable, int* n_valid) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || n < 0 || !T_c_w || (n > 0 && (!uvL || !uvR || !xyz_w || !valid || !reliable))) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    if (n_valid) *n_valid = 0;
    if (n == 0) return VSLAM_OK;
    VS_ENTER(c);
    int rc;
    if ((rc = arena_reserve(c, 2 * al256(8 * (size_t)n) + al256(12 * (size_t)n) + 2 * al256(n) + 2048))) return rc;
    Arena ar(c);
    float* d_l = arena_take<float>(ar, 2 * (size_t)n);
    float* d_r = arena_take<float>(ar, 2 * (size_t)n);
    double* d_T = arena_take<double>(ar, 7);
    int32_t* d_n = arena_take<int32_t>(ar, 1);
    float* d_x = arena_take<float>(ar, 3 * (size_t)n);
    uint8_t* d_v = arena_take<uint8_t>(ar, n);
    uint8_t* d_rel = arena_take<uint8_t>(ar, n);
    int32_t nn = n;
    VS_HIP(hipMemcpyAsync(d_l, uvL, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_r, uvR, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_T, T_c_w, 56, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_n, &nn, 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_triangulate(d_l, d_r, d_n, n, 1, d_T, cam_of(c), d_x, d_v, d_rel, c->stream))) return rc;
    VS_HIP(hipMemcpyAsync(xyz_w, d_x, 12 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipMemcpyAsync(valid, d_v, n, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipMemcpyAsync(reliable, d_rel, n, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    if (n_valid) { int k = 0; for (int i = 0; i < n; ++i) k += valid[i] != 0; *n_valid = k; }
    return VSLAM_OK;
}

int vslam_gather_matched_uv_dev(vslam_ctx* ctx, const vslam_keypoint* d_kpsQ, const vslam_keypoint* d_kpsT, int kp_capacity,
                                const vslam_dmatch* d_matches, const int32_t* d_nmatch, int match_capacity, int B, float* d_uvQ, float* d_uvT) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !d_kpsQ || !d_kpsT || !d_matches || !d_nmatch || !d_uvQ || !d_uvT || kp_capacity <= 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    return launch_gather_uv(d_kpsQ, d_kpsT, kp_capacity, d_matches, d_nmatch, match_capacity, B, d_uvQ, d_uvT, c->stream);
}

int vslam_check_motion(int num_inliers, const double T_c_l[7], double frame_gap) {
    if (!T_c_l) return 0;
    if (num_inliers < 10) return 0; // visual_odometry.cpp:319
    double xi[6];
    se3::log(T_c_l, xi);            // :327
    double s = 0;
    for (int i = 0; i < 6; ++i) s += xi[i] * xi[i];
    return std::sqrt(s) > 5.0 * frame_gap ? 0 : 1; // :329
}

// ---------------------------------------------------------------------------------------------- motion-only pose
static void fill_K(const Ctx* c, double K[4]) { K[0] = c->p.cam[0]; K[1] = c->p.cam[1]; K[2] = c->p.cam[2]; K[3] = c->p.cam[3]; }

int vslam_pnp_motion_only_dev(vslam_ctx* ctx, const float* d_xyz_w, const float* d_uv, const int32_t* d_n, int capacity, int B,
                              double* d_T_c_w, int iters, uint8_t* d_inlier, int32_t* d_n_inliers) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !d_xyz_w || !d_uv || !d_n || !d_T_c_w || capacity <= 0 || iters < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    PnpArgs p;
    memset(&p, 0, sizeof(p));
    p.xyz = d_xyz_w; p.uv = d_uv; p.n = d_n; p.capacity = capacity; p.B = B; p.T = d_T_c_w; p.iters = iters;
    fill_K(c, p.K); p.huber_delta = c->p.huber_delta; p.reproj_thr = c->p.pnp_reproj_thr;
    p.inlier = d_inlier; p.n_inliers = d_n_inliers; p.stats = nullptr;
    return launch_pnp(p, &c->lm, c->stream);
}

int vslam_pnp_motion_only(vslam_ctx* ctx, const float* xyz_w, const float* uv, int n, double T_c_w[7], int iters, uint8_t* inlier,
                          int* n_inliers, vslam_lm_stats* stats) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !xyz_w || !uv || n <= 0 || !T_c_w || iters < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    int rc;
    if ((rc = arena_reserve(c, al256(12 * (size_t)n) + al256(8 * (size_t)n) + al256(n) + al256(sizeof(vslam_lm_stats)) + 2048))) return rc;
    Arena ar(c);
    float* d_x = arena_take<float>(ar, 3 * (size_t)n);
    float* d_u = arena_take<float>(ar, 2 * (size_t)n);
    double* d_T = arena_take<double>(ar, 7);
    int32_t* d_n = arena_take<int32_t>(ar, 2);
    uint8_t* d_in = arena_take<uint8_t>(ar, n);
    vslam_lm_stats* d_st = arena_take<vslam_lm_stats>(ar, 1);
    int32_t nn = n;
    VS_HIP(hipMemcpyAsync(d_x, xyz_w, 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_u, uv, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_T, T_c_w, 56, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_n, &nn, 4, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemsetAsync(d_st, 0, sizeof(vslam_lm_stats), c->stream));
    PnpArgs p;
    memset(&p, 0, sizeof(p));
    p.xyz = d_x; p.uv = d_u; p.n = d_n; p.capacity = n; p.B = 1; p.T = d_T; p.iters = iters;
    fill_K(c, p.K); p.huber_delta = c->p.huber_delta; p.reproj_thr = c->p.pnp_reproj_thr;
    p.inlier = d_in; p.n_inliers = d_n + 1; p.stats = d_st; p.n_hint = n;
    if ((rc = launch_pnp(p, &c->lm, c->stream))) return rc;
    int32_t ni = 0;
    VS_HIP(hipMemcpyAsync(T_c_w, d_T, 56, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipMemcpyAsync(&ni, d_n + 1, 4, hipMemcpyDeviceToHost, c->stream));
    if (inlier) VS_HIP(hipMemcpyAsync(inlier, d_in, n, hipMemcpyDeviceToHost, c->stream));
    if (stats) VS_HIP(hipMemcpyAsync(stats, d_st, sizeof(vslam_lm_stats), hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    if (n_inliers) *n_inliers = ni;
    return VSLAM_OK;
}

// RANSAC wrapper (see include/vslam_hip.h and oracle/ransac.c for the restated control flow)
static unsigned cv_rng_next(uint64_t& state) {
    state = (uint64_t)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
    return (unsigned)state;
}
static int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) { // cv::RANSACUpdateNumIters
    p = std::min(std::max(p, 0.), 1.); ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN), denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

static int pnp_ransac_impl(vslam_ctx* ctx, const float* xyz_w, const float* uv, int n, double T_c_w[7], int max_iters, double reproj_err,
                           double confidence, int lm_iters, uint8_t* inlier, int* n_inliers, int* iters_run, double* models_Rt, int32_t* models_count) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !xyz_w || !uv || n < 0 || !T_c_w || max_iters < 0 || max_iters > 4096 || lm_iters < 0 || !(reproj_err > 0)) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    if (n_inliers) *n_inliers = 0;
    if (iters_run) *iters_run = 0;
    if (inlier && n > 0) memset(inlier, 0, (size_t)n);
    const int mp = 5;
    if (n < mp || max_iters == 0) return VSLAM_OK;
    VS_ENTER(c);
    const bool single = n == mp;          // ptsetreg.cpp: count == modelPoints -> one model from all points, every point an inlier
    const int H = single ? 1 : max_iters;
    // 1. the subset sequence (host: a few hundred RNG draws)
    std::vector<float> hx((size_t)H * mp * 3), hu((size_t)H * mp * 2);
    uint64_t state = 0xFFFFFFFFFFFFFFFFULL;
    for (int it = 0; it < H; ++it) {
        int idx[5];
        for (int i = 0; i < mp; ++i) {
            if (single) { idx[i] = i; continue; }
            for (;;) {
                const int v = (int)(cv_rng_next(state) % (unsigned)n);
                int j = 0;
                for (; j < i; ++j) if (idx[j] == v) break;
                idx[i] = v;
                if (j == i) break;
            }
        }
        for (int i = 0; i < mp; ++i) {
            memcpy(&hx[((size_t)it * mp + i) * 3], xyz_w + 3 * (size_t)idx[i], 12);
            memcpy(&hu[((size_t)it * mp + i) * 2], uv + 2 * (size_t)idx[i], 8);
        }
    }
    int rc;
    if ((rc = arena_reserve(c, al256(hx.size() * 4) + al256(hu.size() * 4) + al256((size_t)H * 96) + al256((size_t)H * 56) + 2 * al256((size_t)H * 4) +
                                   2 * al256(12 * (size_t)n) + 2 * al256(8 * (size_t)n) + 2 * al256(n) + al256(pnp_epnp_ws_bytes(H)) + 4096))) return rc;
    Arena ar(c);
    float* d_hx = arena_take<float>(ar, hx.size()); float* d_hu = arena_take<float>(ar, hu.size());
    double* d_Rt = arena_take<double>(ar, (size_t)H * 12); double* d_hT = arena_take<double>(ar, (size_t)H * 7);
    int32_t* d_ok = arena_take<int32_t>(ar, H); int32_t* d_cnt = arena_take<int32_t>(ar, H);
    float* d_x = arena_take<float>(ar, 3 * (size_t)n); float* d_u = arena_take<float>(ar, 2 * (size_t)n);
    float* d_ix = arena_take<float>(ar, 3 * (size_t)n); float* d_iu = arena_take<float>(ar, 2 * (size_t)n);
    uint8_t* d_mask = arena_take<uint8_t>(ar, n); double* d_T = arena_take<double>(ar, 7); int32_t* d_n1 = arena_take<int32_t>(ar, 2);
    uint8_t* d_ws = arena_take<uint8_t>(ar, pnp_epnp_ws_bytes(H));
    VS_HIP(hipMemcpyAsync(d_hx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_hu, hu.data(), hu.size() * 4, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_x, xyz_w, 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_u, uv, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    // 2. every hypothesis: EPnP on its 5 points (one wave each), then its inlier count over all points
    double K[4];
    fill_K(c, K);
    if ((rc = launch_pnp_epnp(d_hx, d_hu, H, K, d_Rt, d_hT, d_ok, d_ws, c->stream))) return rc;
    std::vector<int32_t> cnt(H), okv(H);
    if (!single && (rc = launch_pnp_count_inliers(d_x, d_u, n, d_Rt, d_ok, 0, H, K, reproj_err, d_cnt, nullptr, c->stream))) return rc;
    if (!single) VS_HIP(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)H * 4, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipMemcpyAsync(okv.data(), d_ok, (size_t)H * 4, hipMemcpyDeviceToHost, c->stream));
    if (models_Rt) VS_HIP(hipMemcpyAsync(models_Rt, d_Rt, (size_t)H * 96, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    if (models_count) for (int i = 0; i < H; ++i) models_count[i] = okv[i] ? (single ? n : cnt[i]) : -1;
    // 3. replay of the sequential loop over the counts (ptsetreg.cpp: strict improvement, adaptive iteration count)
    int best = -1, max_good = 0, it = 0;
    if (single) { if (okv[0]) { best = 0; max_good = n; } }
    else {
        int niters = H;
        for (it = 0; it < niters; ++it)
            if (okv[it] && cnt[it] > std::max(max_good, mp - 1)) {
                best = it; max_good = cnt[it];
                niters = ransac_update_num_iters(confidence, (double)(n - max_good) / n, mp, niters);
            }
    }
    if (iters_run) *iters_run = it;
    if (best < 0) return VSLAM_OK;
    // 4. mask of the best model; lm_iters > 0: refinement on its inliers (solvePnP on the inliers; deviation R2 of oracle/ransac.c)
    std::vector<uint8_t> mask(n, 1);
    if (!single) {
        if ((rc = launch_pnp_count_inliers(d_x, d_u, n, d_Rt, d_ok, best, 1, K, reproj_err, nullptr, d_mask, c->stream))) return rc;
        VS_HIP(hipMemcpyAsync(mask.data(), d_mask, n, hipMemcpyDeviceToHost, c->stream));
        VS_HIP(hipStreamSynchronize(c->stream));
    }
    std::vector<float> ix, iu;
    for (int i = 0; i < n; ++i) if (mask[i]) { ix.insert(ix.end(), xyz_w + 3 * (size_t)i, xyz_w + 3 * (size_t)i + 3); iu.insert(iu.end(), uv + 2 * (size_t)i, uv + 2 * (size_t)i + 2); }
    const int m = (int)(iu.size() / 2);
    if (m != max_good) { set_error("RANSAC inlier recount mismatch (%d vs %d)", m, max_good); return VSLAM_ERR_HIP; }
    if (lm_iters <= 0) { // OpenCV 3.2.0: solvePnPRansac assigns _local_model (the best RANSAC model) to rvec / tvec; the refined pose is discarded
        VS_HIP(hipMemcpyAsync(T_c_w, d_hT + 7 * (size_t)best, 56, hipMemcpyDeviceToHost, c->stream));
        VS_HIP(hipStreamSynchronize(c->stream));
        if (inlier) memcpy(inlier, mask.data(), n);
        if (n_inliers) *n_inliers = max_good;
        return VSLAM_OK;
    }
    VS_HIP(hipMemcpyAsync(d_ix, ix.data(), ix.size() * 4, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_iu, iu.data(), iu.size() * 4, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_T, d_hT + 7 * (size_t)best, 56, hipMemcpyDeviceToDevice, c->stream));
    int32_t nn = m;
    VS_HIP(hipMemcpyAsync(d_n1, &nn, 4, hipMemcpyHostToDevice, c->stream));
    PnpArgs p;
    memset(&p, 0, sizeof(p));
    p.xyz = d_ix; p.uv = d_iu; p.n = d_n1; p.capacity = m; p.B = 1; p.T = d_T; p.iters = lm_iters;
    fill_K(c, p.K); p.huber_delta = 1e300; p.reproj_thr = reproj_err; p.n_hint = m;
    if ((rc = launch_pnp(p, &c->lm, c->stream))) return rc;
    VS_HIP(hipMemcpyAsync(T_c_w, d_T, 56, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    if (inlier) memcpy(inlier, mask.data(), n);
    if (n_inliers) *n_inliers = max_good;
    return VSLAM_OK;
}

int vslam_pnp_ransac(vslam_ctx* ctx, const float* xyz_w, const float* uv, int n, double T_c_w[7], int max_iters, double reproj_err,
                     double confidence, int lm_iters, uint8_t* inlier, int* n_inliers, int* iters_run) {
    return pnp_ransac_impl(ctx, xyz_w, uv, n, T_c_w, max_iters, reproj_err, confidence, lm_iters, inlier, n_inliers, iters_run, nullptr, nullptr);
}

int vslam_pnp_ransac_dev(vslam_ctx* ctx, const float* d_xyz_w, const float* d_uv, const int32_t* d_n, int capacity, int B, double* d_T_c_w, int max_iters,
                         double reproj_err, double confidence, uint8_t* d_inlier, int32_t* d_n_inliers, int32_t* d_iters_run) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !d_xyz_w || !d_uv || !d_n || !d_T_c_w || capacity <= 0 || B < 0 || max_iters <= 0 || max_iters > 4096 || !(reproj_err > 0)) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    if (B == 0) return VSLAM_OK;
    const size_t need = pnp_ransac_scratch_bytes(B, max_iters);
    if (c->ransac_bytes < need) {
        VS_HIP(hipStreamSynchronize(c->stream));
        if (c->d_ransac) { (void)hipFree(c->d_ransac); c->dev_bytes -= c->ransac_bytes; }
        c->d_ransac = nullptr; c->ransac_bytes = 0;
        if (hipMalloc((void**)&c->d_ransac, need) != hipSuccess) { c->d_ransac = nullptr; set_error("RANSAC scratch hipMalloc(%zu) failed", need); return VSLAM_ERR_HIP; }
        c->ransac_bytes = need; c->dev_bytes += need;
    }
    double K[4];
    fill_K(c, K);
    return launch_pnp_ransac_batch(d_xyz_w, d_uv, d_n, capacity, B, max_iters, K, reproj_err, confidence, c->d_ransac, d_T_c_w, d_inlier, d_n_inliers, d_iters_run, c->stream);
}

// diagnostic form: additionally returns every hypothesis model ([R row-major | t], 12 doubles each) and its inlier count (-1: degenerate)
int vslam_pnp_ransac_models(vslam_ctx* ctx, const float* xyz_w, const float* uv, int n, double T_c_w[7], int max_iters, double reproj_err,
                            double confidence, int lm_iters, uint8_t* inlier, int* n_inliers, int* iters_run, double* models_Rt, int32_t* models_count) {
    if (!models_Rt || !models_count) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    return pnp_ransac_impl(ctx, xyz_w, uv, n, T_c_w, max_iters, reproj_err, confidence, lm_iters, inlier, n_inliers, iters_run, models_Rt, models_count);
}

// ---------------------------------------------------------------------------------------------- window optimisation
// Shared host wrapper of optimize_map / optimize_pose_only: stable-sorts the caller's edges by landmark (the
// kernel's CSR contract), runs one pass on the GPU, un-permutes chi2 and applies the chi2 classification of
// optimization.cpp:224-266 on the host with the caller's flag_lm (quirk Q1 lives in the caller's edge list).
static int window_host(vslam_ctx* ctx, int mode, int n_kf, double* T_c_w, int n_lm, float* xyz, int n_edge, const int32_t* kf_idx,
                       const int32_t* lm_idx, const float* uv, const double* K4, const int32_t* flag_lm, int iters, int update_poses, int update_lms,
                       uint8_t* lm_inlier, double* chi2_out, double* thr_out, vslam_lm_stats* stats) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || n_kf <= 0 || n_kf > VSLAM_MAX_KF || !T_c_w || n_lm <= 0 || !xyz || n_edge <= 0 || !kf_idx || !lm_idx || !uv || iters < 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    for (int e = 0; e < n_edge; ++e)
        if (kf_idx[e] < 0 || kf_idx[e] >= n_kf || lm_idx[e] < 0 || lm_idx[e] >= n_lm || (flag_lm && (flag_lm[e] < -1 || flag_lm[e] >= n_lm))) { set_error("edge %d: index out of range", e); return VSLAM_ERR_ARG; }
    // counting sort by landmark (stable)
    std::vector<int32_t> ptr(n_lm + 1, 0), perm(n_edge), skf(n_edge), slm(n_edge);
    std::vector<float> suv(2 * (size_t)n_edge);
    for (int e = 0; e < n_edge; ++e) ptr[lm_idx[e] + 1]++;
    for (int l = 0; l < n_lm; ++l) ptr[l + 1] += ptr[l];
    { std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1); for (int e = 0; e < n_edge; ++e) perm[fill[lm_idx[e]]++] = e; }
    for (int j = 0; j < n_edge; ++j) { const int e = perm[j]; skf[j] = kf_idx[e]; slm[j] = lm_idx[e]; suv[2 * j] = uv[2 * e]; suv[2 * j + 1] = uv[2 * e + 1]; }
    VS_ENTER(c);
    int rc;
    const size_t need = al256(56 * (size_t)n_kf) + al256(12 * (size_t)n_lm) + al256(n_lm) + 3 * al256(4 * (size_t)n_edge) + al256(8 * (size_t)n_edge) * 2 +
                        al256(sizeof(vslam_lm_stats)) + 4096;
    if ((rc = arena_reserve(c, need))) return rc;
    Arena ar(c);
    double* d_T = arena_take<double>(ar, 7 * (size_t)n_kf);
    float* d_xyz = arena_take<float>(ar, 3 * (size_t)n_lm);
    uint8_t* d_inl = arena_take<uint8_t>(ar, n_lm);
    int32_t* d_kf = arena_take<int32_t>(ar, n_edge);
    int32_t* d_lm = arena_take<int32_t>(ar, n_edge);
    float* d_uv = arena_take<float>(ar, 2 * (size_t)n_edge);
    double* d_chi = arena_take<double>(ar, n_edge);
    int32_t* d_off = arena_take<int32_t>(ar, 4);
    vslam_lm_stats* d_st = arena_take<vslam_lm_stats>(ar, 1);
    double* d_thr = arena_take<double>(ar, 1);
    const int32_t off[4] = {0, n_lm, 0, n_edge};
    VS_HIP(hipMemcpyAsync(d_T, T_c_w, 56 * (size_t)n_kf, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_xyz, xyz, 12 * (size_t)n_lm, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemsetAsync(d_inl, 1, n_lm, c->stream)); // the caller already filtered the graph (optimization.cpp:160 / :334)
    VS_HIP(hipMemcpyAsync(d_kf, skf.data(), 4 * (size_t)n_edge, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_lm, slm.data(), 4 * (size_t)n_edge, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_uv, suv.data(), 8 * (size_t)n_edge, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_off, off, sizeof(off), hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemsetAsync(d_st, 0, sizeof(vslam_lm_stats), c->stream));
    VS_HIP(hipMemsetAsync(d_chi, 0, 8 * (size_t)n_edge, c->stream));
    LmWindowArgs a;
    memset(&a, 0, sizeof(a));
    a.n_windows = 1; a.n_kf = n_kf; a.lm_off = d_off; a.edge_off = d_off + 2; a.T = d_T; a.xyz = d_xyz; a.reliable = nullptr;
    a.lm_inlier = d_inl; a.kf_idx = d_kf; a.lm_idx = d_lm; a.uv = d_uv; a.chi2 = d_chi; a.stats = d_st; a.chi2_thr = d_thr;
    fill_K(c, a.K); a.huber_delta = c->p.huber_delta; a.total_lm = n_lm; a.total_edge = n_edge;
    if (K4) memcpy(a.K, K4, sizeof(a.K)); // the caller's `const cv::Mat& K` (optimization.hpp:137-139)
    if ((rc = launch_lm_windows(a, 0, mode, iters, update_poses, update_lms, &c->lm, c->stream))) return rc;
    int32_t status = 0;
    if ((rc = lm_fetch_status(&c->lm, 1, &status, c->stream))) return rc;
    if (status != VSLAM_OK) { set_error("window optimisation rejected the graph (duplicate (keyframe, landmark) edge or bad index)"); return status; }
    std::vector<double> chi(n_edge), chi_sorted(n_edge);
    VS_HIP(hipMemcpy(chi_sorted.data(), d_chi, 8 * (size_t)n_edge, hipMemcpyDeviceToHost));
    for (int j = 0; j < n_edge; ++j) chi[perm[j]] = chi_sorted[j];
    if (update_poses) VS_HIP(hipMemcpy(T_c_w, d_T, 56 * (size_t)n_kf, hipMemcpyDeviceToHost));
    if (mode == 0 && update_lms) VS_HIP(hipMemcpy(xyz, d_xyz, 12 * (size_t)n_lm, hipMemcpyDeviceToHost));
    if (stats) VS_HIP(hipMemcpy(stats, d_st, sizeof(vslam_lm_stats), hipMemcpyDeviceToHost));
    if (chi2_out) memcpy(chi2_out, chi.data(), 8 * (size_t)n_edge);
    // adaptive threshold + flags (optimization.cpp:224-266), caller's edge order
    double th = c->p.huber_delta; // optimization.cpp:154: one variable (chi2_th) is both the Huber delta and the initial chi2 threshold
    for (int iteration = 0; iteration < 5; ++iteration) {
        int out = 0, in = 0;
        for (int e = 0; e < n_edge; ++e) { if (chi[e] > th) ++out; else ++in; }
        if (in / double(in + out) > 0.5) break;
        th *= 2;
    }
    if (lm_inlier)
        for (int e = 0; e < n_edge; ++e) {
            const int l = flag_lm ? flag_lm[e] : lm_idx[e];
            if (l >= 0) lm_inlier[l] = !(chi[e] > th);
        }
    if (thr_out) *thr_out = th;
    return VSLAM_OK;
}

int vslam_local_ba(vslam_ctx* ctx, int n_kf, double* T_c_w, int n_lm, float* xyz, int n_edge, const int32_t* kf_idx, const int32_t* lm_idx,
                   const float* uv, const double* K4, const int32_t* flag_lm, int iters, int update_poses, int update_lms, uint8_t* lm_inlier, double* chi2_out,
                   double* chi2_threshold_out, vslam_lm_stats* stats) {
    return window_host(ctx, 0, n_kf, T_c_w, n_lm, xyz, n_edge, kf_idx, lm_idx, uv, K4, flag_lm, iters, update_poses, update_lms, lm_inlier, chi2_out,
                       chi2_threshold_out, stats);
}

int vslam_pose_only_window(vslam_ctx* ctx, int n_kf, double* T_c_w, int n_lm, const float* xyz, int n_edge, const int32_t* kf_idx,
                           const int32_t* lm_idx, const float* uv, const double* K4, const int32_t* flag_lm, int iters, int update_poses, uint8_t* lm_inlier,
                           double* chi2_out, double* chi2_threshold_out, vslam_lm_stats* stats) {
    return window_host(ctx, 1, n_kf, T_c_w, n_lm, const_cast<float*>(xyz), n_edge, kf_idx, lm_idx, uv, K4, flag_lm, iters, update_poses, 0, lm_inlier,
                       chi2_out, chi2_threshold_out, stats);
}

int vslam_ba_batch_dev(vslam_ctx* ctx, const vslam_ba_batch* b, int schedule, int mode, int iters, int update_poses, int update_lms) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !b || b->n_windows <= 0 || !b->d_lm_off || !b->d_edge_off || !b->d_T_c_w || !b->d_xyz || !b->d_lm_inlier || !b->d_kf_idx ||
        !b->d_lm_idx || !b->d_uv || b->total_lm <= 0 || b->total_edge <= 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    LmWindowArgs a;
    memset(&a, 0, sizeof(a));
    a.n_windows = b->n_windows; a.n_kf = b->n_kf; a.n_kf_w = b->d_n_kf; a.lm_off = b->d_lm_off; a.edge_off = b->d_edge_off; a.T = b->d_T_c_w; a.xyz = b->d_xyz;
    a.reliable = b->d_reliable; a.lm_inlier = b->d_lm_inlier; a.kf_idx = b->d_kf_idx; a.lm_idx = b->d_lm_idx; a.uv = b->d_uv;
    a.chi2 = b->d_chi2; a.stats = b->d_stats; a.chi2_thr = nullptr;
    fill_K(c, a.K); a.huber_delta = c->p.huber_delta; a.total_lm = b->total_lm; a.total_edge = b->total_edge;
    if (b->K4) memcpy(a.K, b->K4, sizeof(a.K));
    return launch_lm_windows(a, schedule, mode, iters, update_poses, update_lms, &c->lm, c->stream);
}

int vslam_build_windows_dev(vslam_ctx* ctx, const vslam_tracks_in* in, int n_kf, int lm_capacity, int edge_capacity, vslam_ba_batch* out, int32_t* d_status) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !in || !out || !d_status || in->n_frames <= 0 || n_kf <= 0 || n_kf > VSLAM_MAX_KF || lm_capacity <= 0 || edge_capacity <= 0 ||
        in->kp_capacity <= 0 || in->lr_capacity <= 0 || in->match_capacity <= 0 || in->pnp_capacity <= 0 || !in->d_kps || !in->d_lr || !in->d_nlr ||
        !in->d_xyz || !in->d_valid || !in->d_reliable || (in->n_frames > 1 && (!in->d_f2f || !in->d_nf2f || !in->d_pose_inlier || !in->d_T_rel)) ||
        !out->d_lm_off || !out->d_edge_off || !out->d_T_c_w || !out->d_xyz || !out->d_reliable || !out->d_lm_inlier || !out->d_kf_idx || !out->d_lm_idx ||
        !out->d_uv || !out->d_n_kf) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    if ((long long)in->n_frames * in->kp_capacity > 0x7FFFFFFFll) { set_error("n_frames x kp_capacity exceeds the 31-bit node keys"); return VSLAM_ERR_ARG; }
    if ((in->d_carry_out != nullptr) != (in->carry_out_frame > 0) || in->carry_out_frame < 0 || in->carry_out_frame >= in->n_frames + (in->n_frames == 1 ? 1 : 0)) {
        if (in->d_carry_out || in->carry_out_frame != 0) { set_error("carry_out_frame %d needs d_carry_out and 1 <= carry_out_frame < n_frames (%d)", in->carry_out_frame, in->n_frames); return VSLAM_ERR_ARG; }
    }
    if (in->kp_capacity > 65536) { set_error("kp_capacity %d exceeds 65536 (the window builder packs a keypoint index into 16 bits)", in->kp_capacity); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    const size_t need = track_scratch_bytes(in->n_frames, in->kp_capacity, lm_capacity);
    if (c->track_bytes < need) {
        VS_HIP(hipStreamSynchronize(c->stream));
        if (c->d_track) { (void)hipFree(c->d_track); c->dev_bytes -= c->track_bytes; }
        c->d_track = nullptr; c->track_bytes = 0;
        if (hipMalloc((void**)&c->d_track, need) != hipSuccess) { c->d_track = nullptr; set_error("track scratch hipMalloc(%zu) failed", need); return VSLAM_ERR_HIP; }
        c->track_bytes = need; c->dev_bytes += need;
    }
    out->n_windows = in->n_frames; out->n_kf = n_kf; out->total_lm = lm_capacity; out->total_edge = edge_capacity;
    double K4[4];
    fill_K(c, K4);
    const int track_rule = c->tune.track_rule >= 0 ? c->tune.track_rule : 1;
    return launch_build_windows(*in, n_kf, lm_capacity, edge_capacity, K4, c->p.pnp_reproj_thr, track_rule, c->d_track, const_cast<int32_t*>(out->d_lm_off), const_cast<int32_t*>(out->d_edge_off),
                                const_cast<int32_t*>(out->d_n_kf), out->d_T_c_w, out->d_xyz, const_cast<uint8_t*>(out->d_reliable), out->d_lm_inlier,
                                const_cast<int32_t*>(out->d_kf_idx), const_cast<int32_t*>(out->d_lm_idx), const_cast<float*>(out->d_uv), d_status, c->stream);
}

int vslam_edge_jacobians(vslam_ctx* ctx, int n, const float* xyz_w, const float* uv, const double T_c_w[7], const double* K4, double* err, double* J_pose,
                         double* J_point, double* chi2, double* huber_w) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || n <= 0 || !xyz_w || !uv || !T_c_w) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    int rc;
    if ((rc = arena_reserve(c, al256(12 * (size_t)n) + al256(8 * (size_t)n) + 256 + al256(16 * (size_t)n) + al256(96 * (size_t)n) + al256(48 * (size_t)n) + 2 * al256(8 * (size_t)n) + 4096))) return rc;
    Arena ar(c);
    float* d_xyz = arena_take<float>(ar, 3 * (size_t)n);
    float* d_uv = arena_take<float>(ar, 2 * (size_t)n);
    double* d_T = arena_take<double>(ar, 7);
    double* d_err = arena_take<double>(ar, 2 * (size_t)n);
    double* d_Jp = arena_take<double>(ar, 12 * (size_t)n);
    double* d_Jl = arena_take<double>(ar, 6 * (size_t)n);
    double* d_chi = arena_take<double>(ar, n);
    double* d_hw = arena_take<double>(ar, n);
    VS_HIP(hipMemcpyAsync(d_xyz, xyz_w, 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_uv, uv, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipMemcpyAsync(d_T, T_c_w, 56, hipMemcpyHostToDevice, c->stream));
    double K[4];
    fill_K(c, K);
    if (K4) memcpy(K, K4, sizeof(K));
    if ((rc = launch_edge_jacobians(n, d_xyz, d_uv, d_T, K, c->p.huber_delta, d_err, d_Jp, d_Jl, d_chi, d_hw, c->stream))) return rc;
    if (err) VS_HIP(hipMemcpyAsync(err, d_err, 16 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (J_pose) VS_HIP(hipMemcpyAsync(J_pose, d_Jp, 96 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (J_point) VS_HIP(hipMemcpyAsync(J_point, d_Jl, 48 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (chi2) VS_HIP(hipMemcpyAsync(chi2, d_chi, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (huber_w) VS_HIP(hipMemcpyAsync(huber_w, d_hw, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    return VSLAM_OK;
}

int vslam_ba_status_dev(vslam_ctx* ctx, int n_windows, int32_t* h_status) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !h_status || n_windows <= 0) return VSLAM_ERR_ARG;
    VS_ENTER(c);
    return lm_fetch_status(&c->lm, n_windows, h_status, c->stream);
}

int vslam_ba_schedule_passes_dev(vslam_ctx* ctx, int n_windows, int32_t* h_passes) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !h_passes || n_windows <= 0) return VSLAM_ERR_ARG;
    VS_ENTER(c);
    return lm_fetch_passes(&c->lm, n_windows, h_passes, c->stream);
}

int vslam_ba_deferred_dev(vslam_ctx* ctx, int n_windows, int32_t* h_deferred) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !h_deferred || n_windows <= 0) return VSLAM_ERR_ARG;
    VS_ENTER(c);
    return lm_fetch_deferred(&c->lm, n_windows, h_deferred, c->stream);
}

// ---------------------------------------------------------------------------------------------- profiling + glue
int vslam_profile_enable(vslam_ctx* ctx, int on) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return VSLAM_ERR_ARG;
    if (!c->prof) c->prof = new Prof();
    if (on) { // the common time origin of vslam_profile_intervals: recorded once per device, on the first context that profiles there
        if (c->device < 0 || c->device >= 16) { set_error("vslam_profile_enable: device %d beyond the profiler's table (16)", c->device); return VSLAM_ERR_ARG; }
        std::lock_guard<std::mutex> lock(g_prof_origin_mu);
        if (!g_prof_origin[c->device]) {
            hipEvent_t ev = nullptr;
            VS_HIP(hipSetDevice(c->device));
            VS_HIP(hipEventCreate(&ev));
            VS_HIP(hipEventRecord(ev, c->stream));
            VS_HIP(hipEventSynchronize(ev));
            g_prof_origin[c->device] = ev;
        }
    }
    c->prof->on = on != 0;
    prof_set_current(on ? c->prof : nullptr);
    return VSLAM_OK;
}

int vslam_profile_read(vslam_ctx* ctx, vslam_kernel_time* out, int cap, int* n_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !out || !n_out || cap <= 0) return VSLAM_ERR_ARG;
    *n_out = 0;
    Prof* p = c->prof;
    if (!p) return VSLAM_OK;
    VS_ENTER(c);
    VS_HIP(hipStreamSynchronize(c->stream));
    int n = 0;
    for (const Prof::Rec& r : p->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
        int k = 0;
        for (; k < n; ++k) if (strncmp(out[k].name, r.name, sizeof(out[k].name) - 1) == 0) break;
        if (k == n) {
            if (n == cap) { p->pool.push_back(r.e0); p->pool.push_back(r.e1); continue; } // (events go back to the pool either way)
            memset(&out[n], 0, sizeof(out[n]));
            strncpy(out[n].name, r.name, sizeof(out[n].name) - 1);
            ++n;
        }
        out[k].total_ms += ms; out[k].launches += r.launches; out[k].calls += 1;
        p->pool.push_back(r.e0); p->pool.push_back(r.e1);
    }
    p->recs.clear();
    *n_out = n;
    return VSLAM_OK;
}

int vslam_profile_intervals(vslam_ctx* ctx, vslam_stage_interval* out, int cap, int* n_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !out || !n_out || cap <= 0) return VSLAM_ERR_ARG;
    *n_out = 0;
    Prof* p = c->prof;
    hipEvent_t origin = (c->device >= 0 && c->device < 16) ? g_prof_origin[c->device] : nullptr;
    if (!p || !origin) return VSLAM_OK;
    VS_ENTER(c);
    VS_HIP(hipStreamSynchronize(c->stream));
    int n = 0;
    for (const Prof::Rec& r : p->recs) {
        if (n < cap) {
            float a0 = 0.f, a1 = 0.f;
            if (hipEventElapsedTime(&a0, origin, r.e0) == hipSuccess && hipEventElapsedTime(&a1, origin, r.e1) == hipSuccess) {
                memset(&out[n], 0, sizeof(out[n]));
                strncpy(out[n].name, r.name, sizeof(out[n].name) - 1);
                out[n].t0_ms = a0; out[n].t1_ms = a1;
                ++n;
            }
        }
        p->pool.push_back(r.e0); p->pool.push_back(r.e1);
    }
    p->recs.clear();
    *n_out = n;
    return VSLAM_OK;
}

int vslam_build_pnp_inputs_dev(vslam_ctx* ctx, const vslam_dmatch* d_f2f, const int32_t* d_nf2f, int match_capacity,
                               const vslam_dmatch* d_lr, const int32_t* d_nlr, int lr_capacity, const float* d_xyz_lr,
                               const uint8_t* d_valid_lr, const vslam_keypoint* d_kps_cur, int kp_capacity, int B, int32_t* d_kp2lr,
                               float* d_xyz_out, float* d_uv_out, int32_t* d_nout, int out_capacity) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !d_f2f || !d_nf2f || !d_lr || !d_nlr || !d_xyz_lr || !d_valid_lr || !d_kps_cur || !d_kp2lr || !d_xyz_out || !d_uv_out || !d_nout ||
        match_capacity <= 0 || lr_capacity <= 0 || kp_capacity <= 0 || out_capacity <= 0) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    return launch_build_pnp_inputs(d_f2f, d_nf2f, match_capacity, d_lr, d_nlr, lr_capacity, d_xyz_lr, d_valid_lr, d_kps_cur, kp_capacity, B,
                                   d_kp2lr, d_xyz_out, d_uv_out, d_nout, out_capacity, c->stream);
}

// float4 streaming copy of `bytes` (src -> dst, both allocated here), `reps` timed launches after one warm-up, hipEvents on the
// context stream: *gbs_out = (bytes read + bytes written) / time.  The achievable-HBM figure bench.py reports next to the spec peak.
int vslam_hbm_copy_probe_variants(void) { return hbm_copy_probe_variants(); }

int vslam_hbm_copy_probe_variant(vslam_ctx* ctx, size_t bytes, int reps, int variant, double* gbs_out, char* name_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !gbs_out || bytes < (1u << 20) || reps <= 0 || variant < 0 || variant >= hbm_copy_probe_variants()) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    VS_ENTER(c);
    bytes &= ~(size_t)15;
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = VSLAM_OK;
    float ms = 0.f;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        set_error("copy probe: allocation failed"); rc = VSLAM_ERR_HIP;
    } else {
        (void)hipMemsetAsync(a, 1, bytes, c->stream);
        rc = launch_hbm_copy_probe(a, b, bytes, variant, c->stream);
        (void)hipEventRecord(e0, c->stream);
        for (int r = 0; r < reps && rc == VSLAM_OK; ++r) rc = launch_hbm_copy_probe(a, b, bytes, variant, c->stream);
        (void)hipEventRecord(e1, c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { set_error("copy probe: timing failed"); rc = VSLAM_ERR_HIP; }
    }
    if (rc == VSLAM_OK) *gbs_out = 2.0 * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
    if (rc == VSLAM_OK && name_out) { strncpy(name_out, hbm_copy_probe_name(variant), 63); name_out[63] = 0; }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}

int vslam_hbm_copy_probe(vslam_ctx* ctx, size_t bytes, int reps, double* gbs_out) { // best of the variants
    if (!gbs_out) { set_error("bad argument"); return VSLAM_ERR_ARG; }
    double best = 0.0;
    for (int v = 0; v < hbm_copy_probe_variants(); ++v) {
        double g = 0.0;
        const int rc = vslam_hbm_copy_probe_variant(ctx, bytes, reps, v, &g, nullptr);
        if (rc != VSLAM_OK) return rc;
        if (g > best) best = g;
    }
    *gbs_out = best;
    return VSLAM_OK;
}

// ---------------------------------------------------------------------------------------------- raw device memory helpers
// (for hosts that do not bring their own allocator; bench.py uses torch tensors instead)
int vslam_dev_alloc(void** p, size_t bytes) { if (!p) return VSLAM_ERR_ARG; VS_HIP(hipMalloc(p, bytes)); return VSLAM_OK; }
int vslam_dev_free(void* p) { if (p) VS_HIP(hipFree(p)); return VSLAM_OK; }
int vslam_dev_upload(vslam_ctx* ctx, void* d, const void* h, size_t bytes) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return VSLAM_ERR_ARG;
    VS_ENTER(c);
    VS_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    return VSLAM_OK;
}
int vslam_dev_download(vslam_ctx* ctx, void* h, const void* d, size_t bytes) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return VSLAM_ERR_ARG;
    VS_ENTER(c);
    VS_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    VS_HIP(hipStreamSynchronize(c->stream));
    return VSLAM_OK;
}
int vslam_dev_memset(vslam_ctx* ctx, void* d, int value, size_t bytes) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return VSLAM_ERR_ARG;
    VS_ENTER(c);
    VS_HIP(hipMemsetAsync(d, value, bytes, c->stream));
    return VSLAM_OK;
}

} // extern "C"
