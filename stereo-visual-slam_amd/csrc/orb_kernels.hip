// orb_kernels.hip -- K1..K6: ORB detect (pyramid, FAST-9/16 + NMS, Harris, retainBest, IC angle), ANMS, rBRIEF.
//
// Replaces the arithmetic behind VO::feature_detection
// (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:70-94): cv::ORB::create(3000)->detect (:80),
// VO::adaptive_non_maximal_suppresion (:96-157) and cv::ORB::create()->compute (:85).  SURVEY.md 8a rows A1-A3.
//
// Design (gfx950, wave64, batched over B images; everything integer except Harris / angle / rotation in f32).  State of round 6:
//   orb_pyrblur_kernel<8>   the batched path (>= 96 images per call), one launch per level: the 256 x 64 level-l tile (+ halo) is staged in LDS once and yields
//                           level l + 1 (8-bit INTER_LINEAR, the library's fixed-point chain), the blurred level l (7 x 7 Gaussian on the matrix cores:
//                           two banded v_mfma_i32_16x16x32_i8 products, exact) and level l's FAST-9/16 corners (streaming compass pre-test on packed i16,
//                           lane-mask queue pushes, 64 candidates scored at a time, 3 x 3 NMS, one global atomic per tile).  Eight waves per tile.
//   orb_resize_kernel / orb_fast_kernel / orb_blur_kernel   the same three steps as separate kernels for small batches and detect-only calls
//   orb_select_kernel       one workgroup per (image, level): 256-bin histogram cut on the FAST score (retainBest(2n) with ties), Harris 7 x 7, 4-pass radix
//                           select on the f32 response (retainBest(n) with ties), raster-order bitonic sort in LDS
//   orb_anms_kernel         one workgroup per image: register-blocked bitonic sort by response, suppression radii by a grid-accelerated nearest-stronger search
//                           (exact f64 distances), second sort for the num-th radius, ordered compaction, cv::ORB::compute's border cull + regroup by octave
//   orb_orient_kernel       intensity-centroid angle of the keypoints the ANMS kept: a wave per keypoint, 8 rows x 32 bytes per load instruction, v_dot4 moments
//   orb_describe_kernel     a wave per keypoint, software-pipelined: the 39 x 40 patch of the BLURRED level staged in LDS, 256 rotated tests, 4 bits per lane
// Float expressions that must round like the CPU (no FMA contraction) use explicit __f*_rn intrinsics; the file is
// also compiled with -ffp-contract=off.
#include "vslam_internal.h"

#include <math.h>

#include <vector>

#include "orb_pattern.inc"

#ifndef VSLAM_BLUR_TILE_H
#define VSLAM_BLUR_TILE_H 64 // rows of a pyramid / blur tile (a wave owns a quarter: 6 halo rows of horizontal passes per wave)
#endif

namespace vslam {

__device__ __constant__ signed char c_pattern[256 * 4];

// tuning aid (VSLAM_ORB_PROFILE=1): per-phase cycle counters, thread 0 of every block adds its clock64 deltas
__device__ long long* g_orb_dbg = nullptr;
// (per-block rows, plain stores: atomics on shared counters would serialise the blocks and distort the timing)
constexpr int kDbgRows = 8192;
#define OPH_INIT() long long* dbg__ = g_orb_dbg ? g_orb_dbg + 64 * (size_t)((blockIdx.x + 977u * blockIdx.y) % kDbgRows) : nullptr; long long tph__ = dbg__ ? clock64() : 0
#define OPH_ON() (dbg__ != nullptr)
#define OPH(slot) do { if (dbg__ && threadIdx.x == 0) { const long long t1__ = clock64(); dbg__[(slot) & 63] += t1__ - tph__; tph__ = t1__; } } while (0)
static long long* g_orb_dbg_host = nullptr;
void orb_debug_enable() {
    if (g_orb_dbg_host) return;
    hipMalloc((void**)&g_orb_dbg_host, sizeof(long long) * 64 * kDbgRows);
    hipMemset(g_orb_dbg_host, 0, sizeof(long long) * 64 * kDbgRows);
    hipMemcpyToSymbol(HIP_SYMBOL(g_orb_dbg), &g_orb_dbg_host, sizeof(g_orb_dbg_host));
}
void orb_debug_dump(hipStream_t stream) {
    if (!g_orb_dbg_host) return;
    hipStreamSynchronize(stream);
    static long long hall[64 * kDbgRows];
    hipMemcpy(hall, g_orb_dbg_host, sizeof(hall), hipMemcpyDeviceToHost);
    hipMemset(g_orb_dbg_host, 0, sizeof(hall));
    long long h[64] = {0};
    for (int r = 0; r < kDbgRows; ++r) for (int i = 0; i < 64; ++i) h[i] += hall[64 * r + i];
    static const char* names[64] = {"sel: hist+cut", "sel: harris", "sel: radix select", "sel: compact+sort", "sel: angle+out", nullptr, nullptr, nullptr,
                                    "desc: patch load", "desc: row blur", "desc: col blur", "desc: sincos+tests", nullptr, nullptr, nullptr, nullptr,
                                    "fast: tile load", "fast: corner test", "fast: score", "fast: nms+append", nullptr, nullptr, nullptr, nullptr,
                                    "anms: gather+sort", "anms: radii", "anms: radius sort", "anms: compact+regroup+out", nullptr, nullptr, nullptr, nullptr,
                                    "pyrfast: tile load", "pyrfast: resize", "pyrfast: blur", "pyrfast: edge columns", "pyrfast: pre-test", "pyrfast: queue pushes", "pyrfast: scores",
                                    "pyrfast: flush + barrier", "pyrfast: nms", "pyrfast: output"};
    for (int i = 0; i < 64; ++i) if (names[i] && h[i]) fprintf(stderr, "  [orb profile] %-28s %14lld block-cycles (sum over blocks)\n", names[i], h[i]);
}
static bool g_pattern_uploaded[16] = {false};

// status bits
constexpr int kStCornerOverflow = 1, kStCandOverflow = 2, kStSelOverflow = 4, kStAnmsOverflow = 8, kStOutOverflow = 16;

__device__ inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

struct LevelView {
    const uint8_t* ptr;
    int w, h, pitch;
};

struct LevelTable {
    int w[kNLevels], h[kNLevels], pitch[kNLevels], pyr_off[kNLevels];
    float scale[kNLevels];
    int nfeat[kNLevels];
    int corner_cap[kNLevels], corner_off[kNLevels];
    int tiles_x[kNLevels], tile_off[kNLevels + 1];
};

__device__ inline LevelView level_view(const LevelTable& T, int l, const uint8_t* img, int pitch0, const uint8_t* pyr) {
    LevelView v;
    v.w = T.w[l]; v.h = T.h[l];
    if (l == 0) { v.ptr = img; v.pitch = pitch0; }
    else { v.ptr = pyr + T.pyr_off[l]; v.pitch = T.pitch[l]; }
    return v;
}

// Stage a tw x th (tw % 4 == 0) pixel tile with origin (x0, y0) into LDS.  Interior tiles take one unaligned dword load
// per 4 pixels (global memory tolerates unaligned dwords); tiles that cross the image edge fall back to per-byte loads
// with reflect-101 (REFLECT) or clamped (!REFLECT) coordinates.
// SKIP_FAR: dwords that start more than 3 columns / rows beyond the image (a 7-tap filter never reads them) are left unwritten.
template <bool REFLECT, int NTHREADS, bool SKIP_FAR = false>
__device__ inline void load_tile_u8(uint8_t* lds, int lds_pitch, const uint8_t* __restrict__ src, int spitch, int W, int H, int x0, int y0,
                                    int tw, int th) {
    const int tw4 = tw >> 2;
    if (x0 >= 0 && y0 >= 0 && x0 + tw <= W && y0 + th <= H) { // uniform
        for (int i = threadIdx.x; i < th * tw4; i += NTHREADS) {
            const int r = i / tw4, c4 = (i - r * tw4) << 2;
            uint32_t v;
            __builtin_memcpy(&v, src + (size_t)(y0 + r) * spitch + x0 + c4, 4);
            *reinterpret_cast<uint32_t*>(lds + r * lds_pitch + c4) = v;
        }
    } else {
        for (int i = threadIdx.x; i < th * tw4; i += NTHREADS) {
            const int r = i / tw4, c4 = (i - r * tw4) << 2;
            const int xa = x0 + c4;
            if (SKIP_FAR && (xa >= W + 4 || y0 + r >= H + 3)) continue;
            const int y = REFLECT ? reflect101(y0 + r, H) : min(max(y0 + r, 0), H - 1);
            const uint8_t* row = src + (size_t)y * spitch;
            uint32_t v = 0;
            if (xa >= 0 && xa + 4 <= W) __builtin_memcpy(&v, row + xa, 4); // only the dwords that straddle the edge go byte by byte
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = REFLECT ? reflect101(xa + k, W) : min(max(xa + k, 0), W - 1);
                    v |= (uint32_t)row[x] << (8 * k);
                }
            }
            *reinterpret_cast<uint32_t*>(lds + r * lds_pitch + c4) = v;
        }
    }
}

// Wide variant for tiles whose rows are CHUNKS x 16 bytes (lds_pitch = 16 * CHUNKS, 16-byte aligned): every thread first ISSUES
// all of its 16-byte loads (a tile load is otherwise a chain of dependent round trips, one dword each), then stores them.
// Chunks nobody reads (more than 3 columns / rows beyond the image) are skipped.  Rows beyond the image come from the reflected
// (REFLECT) or clamped (!REFLECT) row, as whole rows.  Columns: every chunk is ONE 16-byte load -- a chunk that starts left of the
// image or ends beyond the row's pitch is loaded from the nearest 16 bytes inside the row and shifted into place (the pitch bounds the
// address, not the width: bytes between W and the pitch are padding, loaded like pixels); after a workgroup barrier the <= 4 + 3
// columns per row that lie outside the image are filled from their reflect-101 sources INSIDE the staged tile (LDS to LDS: 7 bytes
// per row).  !REFLECT (the FAST tile): pixels outside the image are never read by a pixel that is emitted, they stay undefined.
// [r5] The first version assembled every chunk that straddles the image border byte by byte from global memory (16 dependent-address
// byte loads + reflect arithmetic, ~400 instructions, and a second memory round trip for the workgroup): edge tiles are 40 % of the
// tiles of level 0 and all tiles of the small levels -- 0.5 ms of orb_pyrblur_kernel's 2.0 ms per 1024 images.
__device__ inline uint4 shl_bytes_128(uint4 v, int s) { // out byte k = in byte k - s (s = 0..15, zero fill)
    for (int t = 0; t < (s >> 2); ++t) { v.w = v.z; v.z = v.y; v.y = v.x; v.x = 0; }
    const uint32_t b = (uint32_t)s & 3u;
    if (b) {
        v.w = __builtin_amdgcn_alignbyte(v.w, v.z, 4u - b); v.z = __builtin_amdgcn_alignbyte(v.z, v.y, 4u - b);
        v.y = __builtin_amdgcn_alignbyte(v.y, v.x, 4u - b); v.x <<= 8u * b;
    }
    return v;
}
__device__ inline uint4 shr_bytes_128(uint4 v, int s) { // out byte k = in byte k + s (s = 0..15, zero fill)
    for (int t = 0; t < (s >> 2); ++t) { v.x = v.y; v.y = v.z; v.z = v.w; v.w = 0; }
    const uint32_t b = (uint32_t)s & 3u;
    if (b) {
        v.x = __builtin_amdgcn_alignbyte(v.y, v.x, b); v.y = __builtin_amdgcn_alignbyte(v.z, v.y, b);
        v.z = __builtin_amdgcn_alignbyte(v.w, v.z, b); v.w >>= 8u * b;
    }
    return v;
}
template <int NTHREADS, int CHUNKS, int ROWS, bool REFLECT>
__device__ inline void load_tile_b128(uint8_t* lds, const uint8_t* __restrict__ src, int spitch, int W, int H, int x0, int y0) {
    constexpr int kTotal = CHUNKS * ROWS, kIter = (kTotal + NTHREADS - 1) / NTHREADS;
    uint4 v[kIter];
    int shift[kIter]; // bytes the loaded chunk has to move (0 for all but the <= 2 border chunks of a row); kSkip: nothing to do
    constexpr int kSkip = 1 << 20;
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int i = threadIdx.x + it * NTHREADS;
        const int r = i / CHUNKS, c = i - r * CHUNKS, xa = x0 + 16 * c, ya = y0 + r;
        shift[it] = kSkip;
        if (i < kTotal && xa < W + 4 && ya < H + 3) {
            const int ay = ya < 0 ? -ya : ya;
            const int y = REFLECT ? max(min(ay, 2 * (H - 1) - ay), 0) : min(max(ya, 0), H - 1);
            const int xl = min(max(xa, 0), spitch - 16); // (spitch >= 64)
            __builtin_memcpy(&v[it], src + (size_t)y * spitch + xl, 16);
            shift[it] = xl - xa;
        }
    }
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int i = threadIdx.x + it * NTHREADS;
        if (shift[it] == kSkip) continue;
        if (shift[it] > 0) v[it] = shl_bytes_128(v[it], shift[it]);
        else if (shift[it] < 0) v[it] = shr_bytes_128(v[it], -shift[it]);
        *reinterpret_cast<uint4*>(lds + 16 * i) = v[it];
    }
    if (REFLECT && (x0 < 0 || x0 + 16 * CHUNKS > W)) { // (uniform) columns outside the image: reflect-101, from the staged pixels
        __syncthreads();
        for (int t = threadIdx.x; t < ROWS * 7; t += NTHREADS) {
            const int r = t / 7, k = t - 7 * r;
            const int x = k < 4 ? x0 + k : W + (k - 4);            // k < 4: the (<= 4) columns left of the image; else W, W + 1, W + 2
            if (k < 4 ? x >= 0 : (x - x0 >= 16 * CHUNKS)) continue;
            const int xs = x < 0 ? -x : 2 * (W - 1) - x;           // (W >= 5: one reflection)
            const int js = min(max(xs - x0, 0), 16 * CHUNKS - 1);
            lds[r * (16 * CHUNKS) + (x - x0)] = lds[r * (16 * CHUNKS) + js];
        }
    }
}

// ------------------------------------------------------------------------------------------- host plan
#ifndef VSLAM_FAST_TILE_H
#define VSLAM_FAST_TILE_H 32
#endif
#ifndef VSLAM_FAST_TILE_W
#define VSLAM_FAST_TILE_W 64
#endif
constexpr int kTileW = VSLAM_FAST_TILE_W, kTileH = VSLAM_FAST_TILE_H; // FAST output tile per workgroup
static inline int cv_round_host(double v) { return (int)lrint(v); }

int orb_plan_init(OrbPlan* plan, int w, int h, int nfeatures, int kp_capacity) {
    if (w < 64 || h < 64 || w > 4095 || h > 4095) { set_error("image size %dx%d unsupported (64..4095)", w, h); return VSLAM_ERR_ARG; }
    memset(plan, 0, sizeof(*plan));
    plan->w = w; plan->h = h;
    const double scale_factor = (double)1.2f; // cv::ORB scaleFactor=1.2f held as double
    float factor = (float)(1.0 / scale_factor);
    float ndesired = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)kNLevels));
    int sum = 0, pyr = 0, corners = 0, tiles = 0, tab = 0;
    for (int l = 0; l < kNLevels; ++l) {
        OrbLevel& L = plan->lv[l];
        L.scale = (float)pow(scale_factor, (double)l);
        L.w = cv_round_host((double)((float)w / L.scale));
        L.h = cv_round_host((double)((float)h / L.scale));
        if (l < kNLevels - 1) { L.nfeat = cv_round_host((double)ndesired); sum += L.nfeat; ndesired *= factor; }
        else L.nfeat = nfeatures - sum > 0 ? nfeatures - sum : 0;
        L.pyr_off = pyr;
        if (l > 0) pyr += ((L.w + 63) & ~63) * L.h;
        L.corner_cap = (L.w * L.h / 16 + 255) & ~255;
        L.corner_off = corners; corners += L.corner_cap;
        L.tiles_x = L.w > 62 ? (L.w - 62 + kTileW - 1) / kTileW : 0;
        L.tiles_y = L.h > 62 ? (L.h - 62 + kTileH - 1) / kTileH : 0;
        L.tile_off = tiles; tiles += L.tiles_x * L.tiles_y;
        L.tab_off = tab; tab += L.w;
    }
    int blur = 0;
    for (int l = 0; l < kNLevels; ++l) { plan->blur_off[l] = blur; blur += ((plan->lv[l].w + 63) & ~63) * plan->lv[l].h; }
    plan->blur_bytes = (blur + 255) & ~255;
    plan->pyr_bytes = (pyr + 255) & ~255;
    plan->corner_total = corners;
    plan->total_tiles = tiles;
    plan->sel_cap = 1024;
    (void)kp_capacity;
    return VSLAM_OK;
}

static void fill_level_table(const OrbPlan& plan, LevelTable* T) {
    for (int l = 0; l < kNLevels; ++l) {
        const OrbLevel& L = plan.lv[l];
        T->w[l] = L.w; T->h[l] = L.h; T->pitch[l] = (L.w + 63) & ~63; T->pyr_off[l] = L.pyr_off; T->scale[l] = L.scale;
        T->nfeat[l] = L.nfeat; T->corner_cap[l] = L.corner_cap; T->corner_off[l] = L.corner_off;
        T->tiles_x[l] = L.tiles_x; T->tile_off[l] = L.tile_off;
    }
    T->tile_off[kNLevels] = plan.total_tiles;
}

// resize coefficient tables: cv::resize(INTER_LINEAR, 8U) -- fx = (float)((dx+0.5)*scale - 0.5); floor; 11-bit coeffs
int orb_tables_init(const OrbPlan* plan, OrbTables* t) {
    memset(t, 0, sizeof(*t));
    int nx = 0, ny = 0;
    for (int l = 1; l < kNLevels; ++l) { t->x_off[l] = nx; nx += (plan->lv[l].w + 7) & ~3; t->y_off[l] = ny; ny += plan->lv[l].h; } // x tables padded: the kernel reads 4 entries at a time
    int* xofs = new int[nx]; short* ialpha = new short[2 * nx]; int* yofs = new int[ny]; short* ibeta = new short[2 * ny];
    for (int i = 0; i < nx; ++i) { xofs[i] = 0; ialpha[2 * i] = 2048; ialpha[2 * i + 1] = 0; }
    for (int l = 1; l < kNLevels; ++l) {
        const int sw = plan->lv[l - 1].w, sh = plan->lv[l - 1].h, dw = plan->lv[l].w, dh = plan->lv[l].h;
        const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
        for (int dx = 0; dx < dw; ++dx) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= (float)sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            xofs[t->x_off[l] + dx] = sx;
            ialpha[2 * (t->x_off[l] + dx)] = (short)lrintf((1.f - fx) * 2048.f);
            ialpha[2 * (t->x_off[l] + dx) + 1] = (short)lrintf(fx * 2048.f);
        }
        for (int dy = 0; dy < dh; ++dy) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = (int)floorf(fy);
            fy -= (float)sy;
            yofs[t->y_off[l] + dy] = sy;
            ibeta[2 * (t->y_off[l] + dy)] = (short)lrintf((1.f - fy) * 2048.f);
            ibeta[2 * (t->y_off[l] + dy) + 1] = (short)lrintf(fy * 2048.f);
        }
    }
    // tile ownership of the fused pyramid + blur kernel (256 x VSLAM_BLUR_TILE_H source tiles)
    std::vector<int> tdx, tdy;
    for (int l = 0; l + 1 < kNLevels; ++l) {
        const int sw = plan->lv[l].w, sh = plan->lv[l].h, dw = plan->lv[l + 1].w, dh = plan->lv[l + 1].h;
        const int ntx = (sw + 255) / 256, nty = (sh + VSLAM_BLUR_TILE_H - 1) / VSLAM_BLUR_TILE_H;
        t->tdx_off[l] = (int)tdx.size(); t->tdy_off[l] = (int)tdy.size();
        for (int tx = 0, dx = 0; tx <= ntx; ++tx) { while (dx < dw && xofs[t->x_off[l + 1] + dx] < tx * 256) ++dx; tdx.push_back(tx == ntx ? dw : dx); }
        for (int ty = 0, dy = 0; ty <= nty; ++ty) { while (dy < dh && yofs[t->y_off[l + 1] + dy] < ty * VSLAM_BLUR_TILE_H) ++dy; tdy.push_back(ty == nty ? dh : dy); }
    }
    int rc = VSLAM_OK;
    do {
        if (hipMalloc(&t->d_tile_dx, sizeof(int) * tdx.size()) != hipSuccess || hipMalloc(&t->d_tile_dy, sizeof(int) * tdy.size()) != hipSuccess) {
            set_error("orb_tables_init: hipMalloc failed"); rc = VSLAM_ERR_HIP; break;
        }
        hipMemcpy(t->d_tile_dx, tdx.data(), sizeof(int) * tdx.size(), hipMemcpyHostToDevice);
        hipMemcpy(t->d_tile_dy, tdy.data(), sizeof(int) * tdy.size(), hipMemcpyHostToDevice);
        if (hipMalloc(&t->d_xofs, sizeof(int) * nx) != hipSuccess || hipMalloc(&t->d_ialpha, sizeof(short) * 2 * nx) != hipSuccess ||
            hipMalloc(&t->d_yofs, sizeof(int) * ny) != hipSuccess || hipMalloc(&t->d_ibeta, sizeof(short) * 2 * ny) != hipSuccess) {
            set_error("orb_tables_init: hipMalloc failed"); rc = VSLAM_ERR_HIP; break;
        }
        hipMemcpy(t->d_xofs, xofs, sizeof(int) * nx, hipMemcpyHostToDevice);
        hipMemcpy(t->d_ialpha, ialpha, sizeof(short) * 2 * nx, hipMemcpyHostToDevice);
        hipMemcpy(t->d_yofs, yofs, sizeof(int) * ny, hipMemcpyHostToDevice);
        hipMemcpy(t->d_ibeta, ibeta, sizeof(short) * 2 * ny, hipMemcpyHostToDevice);
    } while (0);
    delete[] xofs; delete[] ialpha; delete[] yofs; delete[] ibeta;
    int dev = 0;
    hipGetDevice(&dev);
    if (rc == VSLAM_OK && dev < 16 && !g_pattern_uploaded[dev]) {
        if (hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), k_orb_pattern, sizeof(k_orb_pattern)) != hipSuccess) {
            set_error("orb pattern upload failed"); return VSLAM_ERR_HIP;
        }
        g_pattern_uploaded[dev] = true;
    }
    return rc;
}

void orb_tables_free(OrbTables* t) {
    if (t->d_xofs) hipFree(t->d_xofs);
    if (t->d_ialpha) hipFree(t->d_ialpha);
    if (t->d_yofs) hipFree(t->d_yofs);
    if (t->d_ibeta) hipFree(t->d_ibeta);
    if (t->d_tile_dx) hipFree(t->d_tile_dx);
    if (t->d_tile_dy) hipFree(t->d_tile_dy);
    memset(t, 0, sizeof(*t));
}

// ------------------------------------------------------------------------------------------- K1 pyramid
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t udot2_u16(uint32_t a, uint32_t b) { // v_dot2_u32_u16: a.lo * b.lo + a.hi * b.hi
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), 0u, false);
}
// Vertical pass + rounding + packing of four outputs of cv::resize's 8U INTER_LINEAR: v = ((b0 * (h0 >> 4) >> 16) + (b1 * (h1 >> 4) >> 16) + 2) >> 2.
// The + 2 rides on the second product (+ 2 << 16, a v_mad), the two ">> 16" are one v_perm per PAIR of pixels (the high halves of two
// products side by side), the sum and the ">> 2" are packed 16-bit operations on such pairs (the sum is <= 1023), one more v_perm picks
// the four result bytes: 41 instead of ~60 instructions per four outputs (both resize kernels were at their instruction-issue bound).
typedef unsigned short rs_us2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t resize_vertical4(const uint32_t (&h0)[4], const uint32_t (&h1)[4], uint32_t b0, uint32_t b1) {
    uint32_t p0[4], p1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { p0[k] = __umul24(b0, h0[k] >> 4); p1[k] = __umul24(b1, h1[k] >> 4) + 0x20000u; } // (11-bit x 15-bit products)
    rs_us2 t[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const rs_us2 x = __builtin_bit_cast(rs_us2, __builtin_amdgcn_perm(p0[2 * j + 1], p0[2 * j], 0x07060302u));
        const rs_us2 y = __builtin_bit_cast(rs_us2, __builtin_amdgcn_perm(p1[2 * j + 1], p1[2 * j], 0x07060302u));
        t[j] = (rs_us2)(x + y) >> (rs_us2){2, 2};
    }
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, t[1]), __builtin_bit_cast(uint32_t, t[0]), 0x06040200u);
}
// Workgroup = 4 waves; a wave owns kResizeRows consecutive output rows (so the row tables and every row base address are scalar)
// and a lane four consecutive output pixels: the column setup -- source offsets, coefficient words, byte selectors -- is done
// once and reused for every row.  Per source row ONE unaligned 8-byte load covers the <= 6 source pixels the four outputs
// interpolate (scale 1.2); v_perm picks source bytes rel, rel + 1 of the window as a u16 pair; the coefficient word already is
// (a0, a1) as u16 (both in [0, 2048]), so a row's horizontal pass is one v_dot2_u32_u16; the result is one dword store.
constexpr int kResizeRows = 4;
__global__ __launch_bounds__(256) void orb_resize_kernel(const uint8_t* __restrict__ src_base, size_t src_img_stride, int spitch,
                                                        int sw, int sh, uint8_t* __restrict__ dst_base, size_t dst_img_stride,
                                                        int dpitch, int dw, int dh, const int* __restrict__ xofs,
                                                        const short* __restrict__ ialpha, const int* __restrict__ yofs,
                                                        const short* __restrict__ ibeta) {
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dx0 = (blockIdx.x * 64 + lane) << 2;
    const int dyb = (blockIdx.y * 4 + wave) * kResizeRows;
    if (dx0 >= dw || dyb >= dh) return;
    const uint8_t* src = src_base + (size_t)b * src_img_stride;
    uint8_t* dst = dst_base + (size_t)b * dst_img_stride;
    const int4 xo = *reinterpret_cast<const int4*>(xofs + dx0);
    const uint4 al = *reinterpret_cast<const uint4*>(ialpha + 2 * dx0); // a0,a1 of 4 pixels as 8 shorts
    // window start: xo.x, pulled back if an 8-byte read would leave the row's pitch
    const int wx = min(xo.x, spitch - 8);
    const int sxs[4] = {xo.x, xo.y, xo.z, xo.w};
    const uint32_t alw[4] = {al.x, al.y, al.z, al.w};
    uint32_t sel[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) // when sx is the last source column a1 == 0: whatever the selector rel + 1 (<= 8) yields does not matter
        sel[k] = 0x0c010c00u + __umul24((uint32_t)(sxs[k] - wx), 0x00010001u); // rel = 0..7
#pragma unroll
    for (int r = 0; r < kResizeRows; ++r) {
        const int dy = dyb + r; // wave-uniform
        if (dy >= dh) break;
        const int sy = yofs[dy];
        const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
        const uint32_t b0 = (uint32_t)(int)ibeta[2 * dy], b1 = (uint32_t)(int)ibeta[2 * dy + 1];
        unsigned long long r0, r1;
        __builtin_memcpy(&r0, src + (size_t)y0 * spitch + wx, 8);
        __builtin_memcpy(&r1, src + (size_t)y1 * spitch + wx, 8);
        const uint32_t r0l = (uint32_t)r0, r0h = (uint32_t)(r0 >> 32), r1l = (uint32_t)r1, r1h = (uint32_t)(r1 >> 32);
        uint32_t h0[4], h1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h0[k] = udot2_u16(__builtin_amdgcn_perm(r0h, r0l, sel[k]), alw[k]);
            h1[k] = udot2_u16(__builtin_amdgcn_perm(r1h, r1l, sel[k]), alw[k]);
        }
        const uint32_t packed = resize_vertical4(h0, h1, b0, b1);
        *reinterpret_cast<uint32_t*>(dst + (size_t)dy * dpitch + dx0) = packed; // dpitch is a multiple of 64; lanes past dw write padding
    }
}

int launch_orb_pyramid(const OrbPlan& plan, const OrbTables& tab, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B,
                       uint8_t* d_pyr, hipStream_t stream) {
    ProfScope prof__(stream, "orb_resize_kernel", kNLevels - 1);
    for (int l = 1; l < kNLevels; ++l) {
        const OrbLevel& S = plan.lv[l - 1];
        const OrbLevel& D = plan.lv[l];
        const uint8_t* src = l == 1 ? d_imgs : d_pyr + S.pyr_off;
        const size_t sstride = l == 1 ? img_bytes : (size_t)plan.pyr_bytes;
        const int spitch = l == 1 ? pitch : ((S.w + 63) & ~63);
        const int dpitch = (D.w + 63) & ~63;
        dim3 grid(((D.w + 3) / 4 + 63) / 64, (D.h + 4 * kResizeRows - 1) / (4 * kResizeRows), B);
        hipLaunchKernelGGL(orb_resize_kernel, grid, dim3(256), 0, stream, src, sstride, spitch, S.w, S.h, d_pyr + D.pyr_off,
                           (size_t)plan.pyr_bytes, dpitch, D.w, D.h, tab.d_xofs + tab.x_off[l], tab.d_ialpha + 2 * tab.x_off[l],
                           tab.d_yofs + tab.y_off[l], tab.d_ibeta + 2 * tab.y_off[l]);
    }
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

// ------------------------------------------------------------------------------------------- K2 FAST
// (kTileW, kTileH are defined next to the host plan, which counts the tiles)
constexpr int kPixW = kTileW + 8, kPixH = kTileH + 8;  // 72 x 24 pixel tile (halo 4)
constexpr int kScW = kTileW + 2, kScH = kTileH + 2;    // 66 x 18 score tile (halo 1)
constexpr int kPixChunks = (kPixW + 15) / 16, kPixPitch = 16 * kPixChunks; // LDS rows of 5 x 16 bytes

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ inline s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ inline s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ inline s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

// ring offsets in the order of cv::FAST (pattern 16): (dx, dy)
#define RING_LOAD(P, pitch)                                                                                   \
    {(P)[0 + 3 * (pitch)],  (P)[1 + 3 * (pitch)],  (P)[2 + 2 * (pitch)],  (P)[3 + 1 * (pitch)],             \
     (P)[3],                (P)[3 - 1 * (pitch)],  (P)[2 - 2 * (pitch)],  (P)[1 - 3 * (pitch)],             \
     (P)[0 - 3 * (pitch)],  (P)[-1 - 3 * (pitch)], (P)[-2 - 2 * (pitch)], (P)[-3 - 1 * (pitch)],            \
     (P)[-3],               (P)[-3 + 1 * (pitch)], (P)[-2 + 2 * (pitch)], (P)[-1 + 3 * (pitch)]}

// cornerScore<16> of cv::FAST: the largest threshold for which the pixel still is a 9-of-16 corner = max over the 16 arcs of
// min(d[k..k+8]) (bright side) and of -max(d[k..k+8]) (dark side), d[k] = v - ring[k].  |d| <= 255, so two ring positions share a
// register as packed i16.  [r6] Pair j = (d[j], d[j+8]): with pre[k] = min(pair 0..k) and suf[k] = min(pair k..7) the arcs k and k + 8 are
// the two halves of min(suf[k], swap(pre[k])) -- arc k = positions k..7 (low halves of pairs k..7) and 8..k+8 (high halves of pairs 0..k), arc
// k + 8 the mirror image -- so all 16 arc minima cost 7 + 7 + 8 + 8 packed operations, the maximum over them 7 more: ~95 instructions for both
// sides (the doubling scheme -- minima over 2, 4, 9 consecutive positions of adjacent pairs -- took 112, one position per register ~200).
// `p` points at the ring's top-left corner (centre - 3 rows - 3 columns): every LDS offset below is non-negative and rides in the load instruction.
template <int PITCH>
__device__ inline int fast_score(const uint8_t* p, int thr) {
#define RP(dx, dy) ((uint32_t)p[((dy) + 3) * PITCH + (dx) + 3])
    const uint32_t v = RP(0, 0);
    // ring positions in the order of cv::FAST (pattern 16): 0 = (0, 3), 1 = (1, 3), 2 = (2, 2), 3 = (3, 1), 4 = (3, 0), ... clockwise
    const uint32_t r0 = RP(0, 3), r1 = RP(1, 3), r2 = RP(2, 2), r3 = RP(3, 1), r4 = RP(3, 0), r5 = RP(3, -1), r6 = RP(2, -2), r7 = RP(1, -3);
    const uint32_t r8 = RP(0, -3), r9 = RP(-1, -3), r10 = RP(-2, -2), r11 = RP(-3, -1), r12 = RP(-3, 0), r13 = RP(-3, 1), r14 = RP(-2, 2), r15 = RP(-1, 3);
#undef RP
    const s16x2 vv = as_s16x2(v | v << 16);
    const s16x2 R[8] = {vv - as_s16x2(r0 | r8 << 16),  vv - as_s16x2(r1 | r9 << 16),  vv - as_s16x2(r2 | r10 << 16), vv - as_s16x2(r3 | r11 << 16),
                        vv - as_s16x2(r4 | r12 << 16), vv - as_s16x2(r5 | r13 << 16), vv - as_s16x2(r6 | r14 << 16), vv - as_s16x2(r7 | r15 << 16)};
    s16x2 pn[8], sn[8], px[8], sx[8];
    pn[0] = R[0]; px[0] = R[0]; sn[7] = R[7]; sx[7] = R[7];
#pragma unroll
    for (int k = 1; k < 8; ++k) { pn[k] = pk_min(pn[k - 1], R[k]); px[k] = pk_max(px[k - 1], R[k]); }
#pragma unroll
    for (int k = 6; k >= 0; --k) { sn[k] = pk_min(sn[k + 1], R[k]); sx[k] = pk_max(sx[k + 1], R[k]); }
    s16x2 bright = {(short)-32768, (short)-32768}, dark = {(short)32767, (short)32767};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t pnk = __builtin_bit_cast(uint32_t, pn[k]), pxk = __builtin_bit_cast(uint32_t, px[k]);
        bright = pk_max(bright, pk_min(sn[k], as_s16x2(__builtin_amdgcn_alignbit(pnk, pnk, 16))));
        dark = pk_min(dark, pk_max(sx[k], as_s16x2(__builtin_amdgcn_alignbit(pxk, pxk, 16))));
    }
    const int a0 = max(thr, max((int)bright.x, (int)bright.y));
    const int b0 = -min((int)dark.x, (int)dark.y);
    return max(a0, b0) - 1;
}

__global__ __launch_bounds__(256) void orb_fast_kernel(LevelTable T, const uint8_t* __restrict__ d_imgs, size_t img_bytes, int pitch0,
                                                      const uint8_t* __restrict__ d_pyr, size_t pyr_bytes, int total_tiles,
                                                      int corner_total, int thr, uint32_t* __restrict__ d_corners,
                                                      int32_t* __restrict__ d_corner_cnt, int32_t* __restrict__ d_status) {
    const int b = blockIdx.y;
    int tile = blockIdx.x, l = 0;
#pragma unroll
    for (int k = 1; k < kNLevels; ++k) if (tile >= T.tile_off[k]) l = k;
    tile -= T.tile_off[l];
    const LevelView V = level_view(T, l, d_imgs + (size_t)b * img_bytes, pitch0, d_pyr + (size_t)b * pyr_bytes);
    const int tx = tile % T.tiles_x[l], ty = tile / T.tiles_x[l];
    const int ox = kEdge + tx * kTileW, oy = kEdge + ty * kTileH; // first emitted pixel of this tile

    __shared__ __attribute__((aligned(16))) uint8_t pix[kPixH * kPixPitch];
    __shared__ __attribute__((aligned(4))) uint8_t sc[(kScH * kScW + 3) / 4 * 4];
    __shared__ uint16_t queue[kScH * kScW];   // positions that pass the compass pre-test
    __shared__ uint16_t cqueue[kScH * kScW];  // corners
    __shared__ uint32_t outq[kTileW * kTileH / 4];
    __shared__ int qcount, ccount, ocount, obase;

    OPH_INIT();
    if (threadIdx.x == 0) { qcount = 0; ccount = 0; ocount = 0; }
    load_tile_b128<256, kPixChunks, kPixH, false>(pix, V.ptr, V.pitch, V.w, V.h, ox - 4, oy - 4);
    for (int i = threadIdx.x; i < (kScH * kScW + 3) / 4; i += 256) reinterpret_cast<uint32_t*>(sc)[i] = 0;
    __syncthreads();
    OPH(16);
#ifndef VSLAM_FAST_DBG
#define VSLAM_FAST_DBG 0 // tuning aid (timing only): 1 = tile load only, 2 = + pre-test, 3 = + corner score
#endif
    if (VSLAM_FAST_DBG == 1) { if (pix[threadIdx.x * 7] == 0xA7 && pix[threadIdx.x] == 0x3C && pix[5] == 1) atomicOr(&d_status[b], 64); return; }

    // (a) compass pre-test on the score region (tile + halo 1): a 9-arc always contains two ADJACENT compass pixels (ring
    // positions 0, 4, 8, 12), so a corner needs two adjacent compass pixels all brighter or all darker.  Cheap, and it
    // thins the candidates before the full 16-pixel test runs with all lanes busy.  One lane tests the four pixels of an aligned
    // dword of the pixel tile: five dword LDS reads, the bytes widened to packed i16 pairs (v_perm), AND = packed min, OR = packed max,
    // the verdict the sign of a packed difference.
    {
        constexpr int kGroups = (kScW + 3 + 3) / 4; // pixel-tile columns 3 .. kScW + 2 in dwords of four
        const s16x2 vthr = {(short)thr, (short)thr};
        for (int i = threadIdx.x; i < kScH * kGroups; i += 256) {
            const int sy = i / kGroups, g = i - sy * kGroups;
            const int y = oy - 1 + sy;
            if (y >= V.h - 3) continue; // outside cv::FAST's own range (left/top are always >= 30)
            const uint32_t* rowc = reinterpret_cast<const uint32_t*>(&pix[(sy + 3) * kPixPitch]);
            const uint32_t D0 = g > 0 ? rowc[g - 1] : 0u, D1 = rowc[g], D2 = rowc[g + 1];
            const uint32_t U = reinterpret_cast<const uint32_t*>(&pix[sy * kPixPitch])[g];       // row - 3 (ring position 8)
            const uint32_t L = reinterpret_cast<const uint32_t*>(&pix[(sy + 6) * kPixPitch])[g]; // row + 3 (ring position 0)
            uint32_t mask = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) { // pixels (4 g + 2 h, 4 g + 2 h + 1) of the pixel tile
                const s16x2 v = as_s16x2(__builtin_amdgcn_perm(0u, D1, h ? 0x0c030c02u : 0x0c010c00u));
                const s16x2 c0 = as_s16x2(__builtin_amdgcn_perm(0u, L, h ? 0x0c030c02u : 0x0c010c00u));
                const s16x2 c8 = as_s16x2(__builtin_amdgcn_perm(0u, U, h ? 0x0c030c02u : 0x0c010c00u));
                const s16x2 c4 = as_s16x2(h ? __builtin_amdgcn_perm(0u, D2, 0x0c020c01u) : __builtin_amdgcn_perm(D2, D1, 0x0c040c03u));   // x + 3
                const s16x2 c12 = as_s16x2(h ? __builtin_amdgcn_perm(D1, D0, 0x0c040c03u) : __builtin_amdgcn_perm(0u, D0, 0x0c020c01u)); // x - 3
                // "two adjacent compass pixels both brighter than v + t": the largest of the four pairwise minima exceeds v + t;
                // "both darker than v - t": the smallest of the four pairwise maxima is below v - t
                // (every adjacent pair takes one pixel of {c0, c8} and one of {c4, c12}, and all four combinations are adjacent pairs:
                //  max over pairs of min(a, b) = min(max(c0, c8), max(c4, c12)) -- three operations instead of seven; likewise the dark side)
                const s16x2 brightest_pair = pk_min(pk_max(c0, c8), pk_max(c4, c12));
                const s16x2 darkest_pair = pk_max(pk_min(c0, c8), pk_min(c4, c12));
                const s16x2 m = pk_max(brightest_pair - (v + vthr), (v - vthr) - darkest_pair);
                mask |= (uint32_t)(m.x > 0) << (2 * h) | (uint32_t)(m.y > 0) << (2 * h + 1);
            }
            // valid score columns: sx = 4 g - 3 + k in [0, kScW), image column x = ox - 1 + sx < V.w - 3
            const int sx0 = 4 * g - 3;
            const int kmin = max(0, -sx0), kmax = min(4, min(kScW - sx0, V.w - 3 - (ox - 1 + sx0)));
            mask &= kmax > kmin ? ((1u << kmax) - 1u) & ~((1u << kmin) - 1u) : 0u;
            while (mask) {
                const int k = __builtin_ctz(mask);
                mask &= mask - 1;
                queue[atomicAdd(&qcount, 1)] = (uint16_t)(sy * kScW + sx0 + k);
            }
        }
    }
    __syncthreads();
    // (b) the survivors' corner score (largest threshold for which the pixel is still a FAST-9/16 corner).  A pixel is a corner at
    // `thr` exactly when that score is >= thr, so the score doubles as the full 16-pixel test: no separate ring-mask pass.
    const int nq = qcount;
    if (VSLAM_FAST_DBG == 2) { if (nq == 77777 && queue[threadIdx.x] == 9) atomicOr(&d_status[b], 64); return; }
    for (int q = threadIdx.x; q < nq; q += 256) {
        const int i = queue[q];
        const int sy = i / kScW, sx = i - sy * kScW;
        const int score = fast_score<kPixPitch>(&pix[sy * kPixPitch + sx], thr);
        if (score >= thr) { sc[i] = (uint8_t)score; cqueue[atomicAdd(&ccount, 1)] = (uint16_t)i; }
    }
    __syncthreads();
    OPH(17);
    const int nc = ccount;
    OPH(18);
    if (VSLAM_FAST_DBG == 3) { if (nc == 77777 && cqueue[threadIdx.x] == 9) atomicOr(&d_status[b], 64); return; }
    // (d) 3x3 non-max suppression + border cull (edgeThreshold 31): survivors collected in LDS, ONE global atomic per block
    for (int q = threadIdx.x; q < nc; q += 256) {
        const int i = cqueue[q];
        const int sy = i / kScW, sx = i - sy * kScW;
        if (sx < 1 || sx > kTileW || sy < 1 || sy > kTileH) continue; // halo positions only feed the NMS
        const int x = ox - 1 + sx, y = oy - 1 + sy;
        if (x >= V.w - kEdge || y >= V.h - kEdge) continue;
        const uint8_t* s = &sc[i];
        const int v = s[0];
        if (v > s[-1] && v > s[1] && v > s[-kScW - 1] && v > s[-kScW] && v > s[-kScW + 1] && v > s[kScW - 1] && v > s[kScW] && v > s[kScW + 1])
            outq[atomicAdd(&ocount, 1)] = (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)v << 24);
    }
    __syncthreads();
    const int no = ocount; // a strict 3x3 maximum: at most one survivor per 2x2 cell, so no <= 1024 / 4
    if (no > 0) {
        if (threadIdx.x == 0) obase = atomicAdd(d_corner_cnt + b * kNLevels + l, no); // (the round trip of this returning atomic: 0.10 of the kernel's 1.78 ms per 1024 images)
        __syncthreads();
        uint32_t* corners = d_corners + (size_t)b * corner_total + T.corner_off[l];
        const int cap = T.corner_cap[l], base = obase;
        for (int q = threadIdx.x; q < no; q += 256) {
            if (base + q < cap) corners[base + q] = outq[q];
            else atomicOr(&d_status[b], kStCornerOverflow);
        }
    }
    OPH(19);
}

int launch_orb_fast(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr, int fast_thr,
                    uint32_t* d_corners, int32_t* d_corner_cnt, int32_t* d_status, hipStream_t stream) {
    LevelTable T;
    fill_level_table(plan, &T);
    VS_HIP(hipMemsetAsync(d_corner_cnt, 0, sizeof(int32_t) * B * kNLevels, stream));
    VS_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t) * B, stream));
    ProfScope prof__(stream, "orb_fast_kernel");
    if (plan.total_tiles > 0)
        hipLaunchKernelGGL(orb_fast_kernel, dim3(plan.total_tiles, B), dim3(256), 0, stream, T, d_imgs, img_bytes, pitch, d_pyr,
                           (size_t)plan.pyr_bytes, plan.total_tiles, plan.corner_total, fast_thr, d_corners, d_corner_cnt, d_status);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

// ------------------------------------------------------------------------------------------- sorting helper
// In-LDS bitonic sort (ascending) of n (power of two) keys by the whole workgroup.  A compare-exchange stage with stride j only
// couples keys inside aligned blocks of 2j, so every wave owns a chunk of 64 E consecutive keys (E per lane, blocked) and runs
// all stages with j < 64 E on registers: strides below E inside the lane, strides E..32 E as lane exchanges (ds_bpermute, no
// barrier).  Only the strides >= 64 E of the last log2(n / 64 E) merge levels go through LDS with a workgroup barrier: 10 of the
// 78 stages of a 4096-key sort on 1024 lanes (a 16-wave barrier costs ~400 cycles, a plain stage also one exposed LDS round trip).
template <typename K>
__device__ inline K shfl_xor_key(K v, int d) {
    if constexpr (sizeof(K) == 8) {
        const int lo = __shfl_xor((int)(uint32_t)v, d), hi = __shfl_xor((int)(uint32_t)(v >> 32), d);
        return ((K)(uint32_t)hi << 32) | (K)(uint32_t)lo;
    } else {
        return (K)__shfl_xor((int)v, d);
    }
}
// stages j = jtop, jtop / 2, ..., 1 of merge level k on a wave's chunk; v[r] is the key with global index g0 + r
template <typename K, int E>
__device__ inline void bitonic_chunk_stages(K (&v)[E], int g0, int k, int jtop) {
    const int lane = threadIdx.x & 63;
    for (int j = jtop; j >= E; j >>= 1) {
        const int d = j / E;
        const bool lower = (lane & d) == 0;
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const K o = shfl_xor_key(v[r], d);
            const bool take_min = lower == (((g0 + r) & k) == 0);
            v[r] = ((v[r] < o) == take_min) ? v[r] : o; // (equal keys: either copy)
        }
    }
#pragma unroll
    for (int j = E / 2; j >= 1; j >>= 1) {
        if (j > jtop) continue;
#pragma unroll
        for (int r = 0; r < E; ++r)
            if ((r & j) == 0) {
                const bool up = ((g0 + r) & k) == 0;
                const K x = v[r], y = v[r | j];
                if ((x > y) == up) { v[r] = y; v[r | j] = x; }
            }
    }
}
template <typename K, int E>
__device__ inline void bitonic_sort_lds_e(K* a, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int C = 64 * E;
    const int g0 = wave * C + lane * E; // this lane's first key
    const bool mine = g0 < n;           // (n is a power of two: a lane's E keys are all inside or all outside when n >= E)
    K v[E];
    auto load = [&]() {
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = (g0 + r < n) ? a[g0 + r] : (K)~(K)0;
    };
    auto store = [&]() {
#pragma unroll
        for (int r = 0; r < E; ++r) if (g0 + r < n) a[g0 + r] = v[r];
    };
    __syncthreads();
    // merge levels that fit a chunk: registers only
    if (wave * C < n) { // wave-uniform: whole waves take part in the lane exchanges
        load();
        for (int k = 2; k <= min(n, C); k <<= 1) bitonic_chunk_stages<K, E>(v, g0, k, k >> 1);
        store();
    }
    for (int k = 2 * C; k <= n; k <<= 1) {
        for (int j = k >> 1; j >= C; j >>= 1) { // strides that cross chunks: through LDS
            __syncthreads();
            for (int t = threadIdx.x; t < n / 2; t += blockDim.x) {
                const int lo = ((t / j) * 2 * j) + (t % j);
                const int hi = lo + j;
                const bool up = (lo & k) == 0;
                const K x = a[lo], y = a[hi];
                if ((x > y) == up) { a[lo] = y; a[hi] = x; }
            }
        }
        __syncthreads();
        if (wave * C < n) {
            load();
            bitonic_chunk_stages<K, E>(v, g0, k, C >> 1);
            store();
        }
    }
    __syncthreads();
}
template <typename K>
__device__ __attribute__((always_inline)) inline void bitonic_sort_lds(K* a, int n) {
    const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x; // keys per lane so that the waves' chunks cover the array
    if (per <= 1) bitonic_sort_lds_e<K, 1>(a, n);
    else if (per <= 2) bitonic_sort_lds_e<K, 2>(a, n);
    else if (per <= 4) bitonic_sort_lds_e<K, 4>(a, n);
    else if (per <= 8) bitonic_sort_lds_e<K, 8>(a, n);
    else { // (not reached with the capacities of this library: plain network, one barrier per stage)
        for (int k = 2; k <= n; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                __syncthreads();
                for (int t = threadIdx.x; t < n / 2; t += blockDim.x) {
                    const int lo = ((t / j) * 2 * j) + (t % j);
                    const int hi = lo + j;
                    const bool up = (lo & k) == 0;
                    const K x = a[lo], y = a[hi];
                    if ((x > y) == up) { a[lo] = y; a[hi] = x; }
                }
            }
        __syncthreads();
    }
}

__device__ inline uint32_t float_order_key(float f) { // monotone: larger float -> larger key; -0 == +0
    f = f + 0.0f;
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float float_from_order_key(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

// ------------------------------------------------------------------------------------------- K3 select
__device__ inline float harris_response_dev(const LevelView& V, int x0, int y0) {
    const int step = V.pitch;
    // 9 x 9 pixel window (7x7 block + 1 px Sobel apron) as 9 rows x 3 unaligned dwords, all loads in flight at once
    const uint8_t* base = V.ptr + (size_t)(y0 - 4) * step + (x0 - 4);
    uint32_t wdw[9][3];
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) __builtin_memcpy(&wdw[r][q], base + (size_t)r * step + 4 * q, 4);
    auto px = [&](int r, int c) -> int { return (int)((wdw[r][c >> 2] >> (8 * (c & 3))) & 0xFFu); };
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int Ix = (px(i + 1, j + 2) - px(i + 1, j)) * 2 + (px(i, j + 2) - px(i, j)) + (px(i + 2, j + 2) - px(i + 2, j));
            const int Iy = (px(i + 2, j + 1) - px(i, j + 1)) * 2 + (px(i + 2, j) - px(i, j)) + (px(i + 2, j + 2) - px(i, j + 2));
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float s2 = __fmul_rn(scale, scale);
    const float s4 = __fmul_rn(__fmul_rn(s2, scale), scale);
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float ab = __fadd_rn(fa, fb);
    const float t = __fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(0.04f, ab), ab));
    return __fmul_rn(t, s4);
}

__device__ inline float fast_atan2_dev(float y, float x) {
    const float rad2deg = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * rad2deg, p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg, p7 = -0.04432655554792128f * rad2deg;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// Largest bin d (255..0) such that the count of entries in bins >= d reaches `rank` (clamped to bin 0); optionally the
// count strictly above d.  Wave 0 does a suffix scan with 4 bins per lane; every thread returns the same answer.
__device__ inline int find_rank_bin(const int* hist, int rank, int* above_out) {
    __shared__ int s_bin, s_above;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int b0 = 252 - 4 * lane; // this lane owns bins b0+3, b0+2, b0+1, b0 (descending); lane 0 owns 255..252
        const int h3 = hist[b0 + 3], h2 = hist[b0 + 2], h1 = hist[b0 + 1], h0 = hist[b0];
        const int mine = h3 + h2 + h1 + h0;
        int incl = mine; // inclusive prefix over lanes = entries in bins >= b0
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        const int before = incl - mine; // entries in bins > b0 + 3
        const bool here = before < rank && incl >= rank;
        const bool last = lane == 63 && incl < rank; // fewer than `rank` entries in total: bin 0
        if (here || last) {
            int acc = before, d = b0 + 3;
            if (acc + h3 >= rank) d = b0 + 3;
            else { acc += h3; if (acc + h2 >= rank) d = b0 + 2; else { acc += h2; if (acc + h1 >= rank) d = b0 + 1; else { acc += h1; d = b0; } } }
            if (last) { d = 0; acc = incl - h0; }
            s_bin = d; s_above = acc;
        }
    }
    __syncthreads();
    if (above_out) *above_out = s_above;
    return s_bin;
}

constexpr int kSelBlock = 512;
constexpr int kCandCap = 4096;

__global__ __launch_bounds__(kSelBlock) void orb_select_kernel(LevelTable T, const uint8_t* __restrict__ d_imgs, size_t img_bytes,
                                                              int pitch0, const uint8_t* __restrict__ d_pyr, size_t pyr_bytes,
                                                              int corner_total, const uint32_t* __restrict__ d_corners,
                                                              const int32_t* __restrict__ d_corner_cnt, int sel_cap,
                                                              vslam_keypoint* __restrict__ d_sel, int32_t* __restrict__ d_sel_cnt,
                                                              int32_t* __restrict__ d_status) {
    // level-major work numbering: with the level as the fast grid index every level-0 workgroup (the heaviest) lands on the
    // same XCD (workgroups are dealt to the 8 XCDs round-robin)
    const int b = blockIdx.x, l = blockIdx.y;
    const LevelView V = level_view(T, l, d_imgs + (size_t)b * img_bytes, pitch0, d_pyr + (size_t)b * pyr_bytes);
    const uint32_t* corners = d_corners + (size_t)b * corner_total + T.corner_off[l];
    const int n = min(d_corner_cnt[b * kNLevels + l], T.corner_cap[l]);
    const int nfeat = T.nfeat[l];

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem); // kCandCap x u64: (harris order key << 32) | packed xy
    unsigned long long* keep = cand;                                        // the sort buffer of the survivors: the SAME storage ([r5] the candidates pass through registers,
                                                                            // below: 33 KB instead of 65 KB of LDS per workgroup, three or four (level, image) pairs per CU instead of two)
    int* hist = reinterpret_cast<int*>(cand + kCandCap);                    // 256 bins
    int& s_cut = hist[256]; int& s_ncand = hist[257]; int& s_rank = hist[258]; int& s_keep = hist[259];
    uint32_t& s_prefix = reinterpret_cast<uint32_t*>(hist)[260];

    OPH_INIT();
    // ---- retainBest(2 * nfeat) on the FAST score (ties at the cut are kept)
    for (int i = threadIdx.x; i < 256; i += kSelBlock) hist[i] = 0;
    if (threadIdx.x == 0) { s_ncand = 0; s_cut = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kSelBlock) atomicAdd(&hist[corners[i] >> 24], 1);
    __syncthreads();
    if (n > 2 * nfeat && nfeat > 0) { // uniform
        const int d = find_rank_bin(hist, 2 * nfeat, nullptr);
        if (threadIdx.x == 0) s_cut = d;
    }
    __syncthreads();
    const int cut = (nfeat > 0) ? s_cut : 256;
    OPH(0);
    // ---- Harris response of every survivor
    for (int i = threadIdx.x; i < n; i += kSelBlock) {
        const uint32_t c = corners[i];
        if ((int)(c >> 24) < cut) continue;
        const int slot = atomicAdd(&s_ncand, 1);
        if (slot < kCandCap) {
            const float r = harris_response_dev(V, c & 0xFFF, (c >> 12) & 0xFFF);
            cand[slot] = ((unsigned long long)float_order_key(r) << 32) | (c & 0xFFFFFFu);
        }
    }
    __syncthreads();
    int m = s_ncand;
    if (m > kCandCap) { if (threadIdx.x == 0) atomicOr(&d_status[b], kStCandOverflow); m = kCandCap; }
    OPH(1);
    // ---- retainBest(nfeat) on the Harris response: radix select of the nfeat-th largest key
    uint32_t cutkey = 0;
    if (m > nfeat) {
        if (threadIdx.x == 0) { s_prefix = 0; s_rank = nfeat; }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int i = threadIdx.x; i < 256; i += kSelBlock) hist[i] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            const uint32_t pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int i = threadIdx.x; i < m; i += kSelBlock) {
                const uint32_t k = (uint32_t)(cand[i] >> 32);
                if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & 0xFF], 1);
            }
            __syncthreads();
            {
                int above = 0;
                const int rank = s_rank;
                const int d = find_rank_bin(hist, rank, &above);
                __syncthreads();
                if (threadIdx.x == 0) { s_rank = rank - above; s_prefix = prefix | ((uint32_t)d << shift); }
            }
            __syncthreads();
        }
        cutkey = s_prefix;
    }
    // ---- compact survivors (key >= cutkey), then raster sort by (y, x)
    __syncthreads();
    OPH(2);
    if (threadIdx.x == 0) s_keep = 0;
    constexpr int kPerThread = kCandCap / kSelBlock;
    unsigned long long ev[kPerThread]; // this thread's candidates leave the buffer before any survivor is written into it
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) { const int i = threadIdx.x + u * kSelBlock; ev[u] = i < m ? cand[i] : 0ull; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
        const int i = threadIdx.x + u * kSelBlock;
        const unsigned long long e = ev[u];
        if (i < m && (uint32_t)(e >> 32) >= cutkey) {
            const int slot = atomicAdd(&s_keep, 1);
            const uint32_t xy = (uint32_t)e & 0xFFFFFFu;
            const uint32_t raster = ((xy >> 12) << 12) | (xy & 0xFFF); // y major, x minor (already that layout)
            keep[slot] = ((unsigned long long)raster << 32) | (uint32_t)(e >> 32);
        }
    }
    __syncthreads();
    const int nk = s_keep;
    int np2 = 1;
    while (np2 < nk) np2 <<= 1;
    for (int i = nk + threadIdx.x; i < np2; i += kSelBlock) keep[i] = ~0ull;
    __syncthreads();
    if (np2 > 1) bitonic_sort_lds(keep, np2);
    int nout = nk;
    if (nout > sel_cap) { if (threadIdx.x == 0) atomicOr(&d_status[b], kStSelOverflow); nout = sel_cap; }
    OPH(3);
    // ---- output.  The orientation is computed AFTER the ANMS (orb_orient_kernel): nothing between here and there reads it, and
    // the ANMS discards half of these keypoints at N = 1500 (five sixths at the reference's 500).
    vslam_keypoint* out = d_sel + ((size_t)b * kNLevels + l) * sel_cap;
    const float scale = T.scale[l];
    for (int i = threadIdx.x; i < nout; i += kSelBlock) {
        const unsigned long long e = keep[i];
        const uint32_t raster = (uint32_t)(e >> 32);
        const int x = raster & 0xFFF, y = raster >> 12;
        vslam_keypoint kp;
        kp.x = __fmul_rn((float)x, scale);
        kp.y = __fmul_rn((float)y, scale);
        kp.size = __fmul_rn(31.f, scale);
        kp.angle = 0.f;
        kp.response = float_from_order_key((uint32_t)e);
        kp.octave = l;
        kp.class_id = -1;
        out[i] = kp;
    }
    __syncthreads();
    OPH(4);
    if (threadIdx.x == 0) d_sel_cnt[b * kNLevels + l] = nout;
}

int launch_orb_select(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr,
                      const uint32_t* d_corners, const int32_t* d_corner_cnt, vslam_keypoint* d_sel, int32_t* d_sel_cnt,
                      int32_t* d_status, hipStream_t stream) {
    LevelTable T;
    fill_level_table(plan, &T);
    const size_t smem = (size_t)kCandCap * 8 + 272 * 4;
    static bool attr_set[16] = {false}; // per device
    int dev = 0;
    VS_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(orb_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    ProfScope prof__(stream, "orb_select_kernel");
    hipLaunchKernelGGL(orb_select_kernel, dim3(B, kNLevels), dim3(kSelBlock), smem, stream, T, d_imgs, img_bytes, pitch, d_pyr,
                       (size_t)plan.pyr_bytes, plan.corner_total, d_corners, d_corner_cnt, plan.sel_cap, d_sel, d_sel_cnt, d_status);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

// ------------------------------------------------------------------------------------------- K5 ANMS
constexpr int kAnmsBlock = 1024;
constexpr int kAnmsBrute = 160; // at most this many stronger keypoints: scanning them beats walking the grid

// K3b orb_orient_kernel: intensity-centroid orientation of the keypoints that survived the ANMS, and the (cos, sin) of the rBRIEF
// rotation.  A 16-lane group per keypoint (four per wave); a fixed set of groups per image walks the list.  The level
// coordinates are recovered as cvRound(pt / scale), the same identity the descriptor stage uses (orb.cpp computeDescriptors).
// XCD-aware numbering of the (image, block) grids of the two patch-gathering kernels below: the hardware deals workgroups to the eight XCDs
// round-robin by their linear id, and every XCD has its own L2.  With a (blocks, images) grid the nb blocks of ONE image land on all eight
// XCDs, and every L2 fetches most of that image's pyramid for itself (rocprofv3, round 4: 5.0 MB fetched per image for a 1.9 MB blurred
// pyramid).  Here the linear id is cut so that all blocks of image b have id = b (mod 8): one L2 serves an image's patches.
__device__ inline bool xcd_image_block(int nb, int B, int& b, int& bx) {
    const int id = blockIdx.x, per = 8 * nb, g = id / per, r = id - g * per;
    b = 8 * g + (r & 7); bx = r >> 3;
    return b < B;
}
#ifndef VSLAM_ORIENT_BLOCKS
#define VSLAM_ORIENT_BLOCKS 24
#endif
constexpr int kOrientBlocks = VSLAM_ORIENT_BLOCKS, kOrientThreads = 256, kOrientWaves = kOrientThreads / 64;
constexpr int kOrientPerWave = 4096 / (kOrientBlocks * kOrientWaves) + 1; // keypoints a wave can be handed, one lane each (kp_capacity <= kOrientPerWave x waves per image, checked at launch)
static_assert(kOrientPerWave <= 64, "orb_orient_kernel: a lane fetches one keypoint record of its wave's walk");
// per-lane constants of the patch sums (lane = 8 ry + cx, row slot s): {weights u + 15 | mask} of the lane's four bytes, 0 outside the circular patch
struct IcLaneTable { uint32_t wt[64][4], wm[64][4]; };
constexpr IcLaneTable make_ic_lane_table() {
    IcLaneTable t{};
    const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 4; ++s) {
            const int ry = lane >> 3, cx = lane & 7, v = -15 + ry + 8 * s, av = v < 0 ? -v : v;
            const int dmax = av <= 15 ? umax[av] : -1; // (v = 16: beyond the patch)
            uint32_t a = 0, m = 0;
            for (int bb = 0; bb < 4; ++bb) {
                const int u = -15 + 4 * cx + bb;
                const bool in = u <= 15 && u >= -dmax && u <= dmax;
                a |= (in ? (uint32_t)(u + 15) : 0u) << (8 * bb);
                m |= (in ? 1u : 0u) << (8 * bb);
            }
            t.wt[lane][s] = a; t.wm[lane][s] = m;
        }
    return t;
}
__device__ const IcLaneTable g_ic_lane = make_ic_lane_table();
// [r6] One WAVE per keypoint, lane = (row group ry = lane >> 3, column quad cx = lane & 7): a load instruction fetches 8 rows x 32 contiguous bytes (8 cache
// lines; the 16-lane-group form of rounds 2-5 gave every lane its own row: 64 lines per instruction, and the kernel sat at a third of the VALU issue rate
// waiting for the vector memory path), four of them cover the 31 x 31 patch.  Byte b of row slot s holds u = -15 + 4 cx + b, v = -15 + ry + 8 s; the circular
// mask |u| <= umax[|v|] and the weights u + 15 are per-lane constants, so a keypoint is 8 v_dot4_u32_u8 + 4 multiply-adds per lane (exact integers).  Two
// keypoints per trip; their four moments are summed over the wave with a halving butterfly (7 exchanges instead of 24).  The angle, its store and the f64 cos /
// sin of the rBRIEF rotation are done after the walk, one lane per keypoint.
__global__ __launch_bounds__(kOrientThreads) void orb_orient_kernel(LevelTable T, const uint8_t* __restrict__ d_imgs, size_t img_bytes, int pitch0,
                                                                  const uint8_t* __restrict__ d_pyr, size_t pyr_bytes,
                                                                  vslam_keypoint* __restrict__ d_kps, float2* __restrict__ d_cs, const int32_t* __restrict__ d_order,
                                                                  int kp_capacity, const int32_t* __restrict__ d_count, int nb, int B) {
    int b, bx;
    if (!xcd_image_block(nb, B, b, bx)) return; // (uniform)
    const int n = min(d_count[b], kp_capacity);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = nb * kOrientWaves, wv = bx * kOrientWaves + wave;
    vslam_keypoint* kps = d_kps + (size_t)b * kp_capacity;
    const int32_t* order = d_order ? d_order + (size_t)b * kp_capacity : nullptr;
    __shared__ int s_m10[kOrientWaves * kOrientPerWave], s_m01[kOrientWaves * kOrientPerWave], s_j[kOrientWaves * kOrientPerWave];
    for (int t = threadIdx.x; t < kOrientWaves * kOrientPerWave; t += kOrientThreads) s_j[t] = -1;
    __syncthreads();
    // per-lane constants
    const int ry = lane >> 3, cx = lane & 7;
    uint32_t wt[4], wm[4]; int vs[4], rsel[4];
    {
        const uint4 a_ = *reinterpret_cast<const uint4*>(g_ic_lane.wt[lane]), m_ = *reinterpret_cast<const uint4*>(g_ic_lane.wm[lane]);
        wt[0] = a_.x; wt[1] = a_.y; wt[2] = a_.z; wt[3] = a_.w; wm[0] = m_.x; wm[1] = m_.y; wm[2] = m_.z; wm[3] = m_.w;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int v = -15 + ry + 8 * s; // row of slot s relative to the keypoint (16: beyond the patch, weight 0, the row above is read instead)
        vs[s] = v > 15 ? 0 : v; rsel[s] = min(ry + 8 * s, 30);
    }
    const uint8_t* img0 = d_imgs + (size_t)b * img_bytes;
    const uint8_t* pyr0 = d_pyr + (size_t)b * pyr_bytes;
    // The walk positions of this wave are wv, wv + nwaves, ...: lane k takes the k-th one.  Two round trips for the whole walk (its slot in the output
    // order, then the keypoint record) instead of two per keypoint; the patches then stream four keypoints at a time, the next four requested before the
    // current four are summed.  (The version that chained "order -> record -> patch" per trip spent its time in those dependent round trips: rewriting its
    // access pattern alone changed nothing, 0.54 ms per 1024 images either way.)
    const int cnt = wv < n ? min((n - wv + nwaves - 1) / nwaves, kOrientPerWave) : 0; // (uniform)
    int my_j = -1, my_x = 15, my_y = 15, my_l = 0, my_ok = 0;
    if (lane < cnt) {
        const int i = wv + lane * nwaves;
        my_j = min(max(order ? order[i] : i, 0), n - 1);
        const vslam_keypoint kp = kps[my_j];
        my_l = min(max(kp.octave, 0), kNLevels - 1);
        const float inv_scale = __fdiv_rn(1.f, T.scale[my_l]);
        my_x = __float2int_rn(__fmul_rn(kp.x, inv_scale)); my_y = __float2int_rn(__fmul_rn(kp.y, inv_scale));
        // (keypoints come from the detector: >= 31 px from the level border, so the 31 x 31 patch is inside the level)
        my_ok = my_x >= 15 && my_y >= 15 && my_x + 16 <= T.w[my_l] && my_y + 15 < T.h[my_l];
        if (!my_ok) { my_x = 15; my_y = 15; } // (a keypoint without a patch reads the level's first patch: its sums are dropped)
        s_j[wave * kOrientPerWave + lane] = my_j;
    }
    auto request = [&](int k0, uint32_t (&px)[4][4]) { // patches of keypoints k0 .. k0 + 3 of this wave's walk (beyond cnt: lane values of an idle lane = the level's first patch)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = __builtin_amdgcn_readlane(my_x, k0 + u), y = __builtin_amdgcn_readlane(my_y, k0 + u), l = __builtin_amdgcn_readlane(my_l, k0 + u);
            const uint8_t* base = l == 0 ? img0 : pyr0 + T.pyr_off[l];
            const int pitch = l == 0 ? pitch0 : T.pitch[l];
            const uint8_t* p0 = base + (size_t)(y - 15) * pitch + (x - 15); // (uniform)
#pragma unroll
            for (int s = 0; s < 4; ++s) __builtin_memcpy(&px[u][s], p0 + (uint32_t)(rsel[s] * pitch + 4 * cx), 4); // unaligned dword: 8 lanes = 32 contiguous bytes of one row
        }
    };
    auto reduce_store = [&](int k0, const uint32_t (&px)[4][4]) {
        int m[8]; // {m10, m01} of the four keypoints
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t tsum = 0, asum = 0; int m01 = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint32_t as = __builtin_amdgcn_udot4(px[u][s], wm[s], 0u, false);
                tsum = __builtin_amdgcn_udot4(px[u][s], wt[s], tsum, false);
                asum += as;
                m01 += vs[s] * (int)as;
            }
            const bool live = __builtin_amdgcn_readlane(my_ok, k0 + u) != 0;
            m[2 * u] = live ? (int)tsum - 15 * (int)asum : 0;
            m[2 * u + 1] = live ? m01 : 0;
        }
        // halving butterfly over the 8 sums: 4 + 2 + 1 exchanges, then 3 full ones; lane q < 8 ends with the total of m[bit-reversed(q)]
        const bool h0 = (lane & 1) != 0, h1 = (lane & 2) != 0, h2 = (lane & 4) != 0;
        int a4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a4[q] = (h0 ? m[4 + q] : m[q]) + __shfl_xor(h0 ? m[q] : m[4 + q], 1);
        int a2[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) a2[q] = (h1 ? a4[2 + q] : a4[q]) + __shfl_xor(h1 ? a4[q] : a4[2 + q], 2);
        int r = (h2 ? a2[1] : a2[0]) + __shfl_xor(h2 ? a2[0] : a2[1], 4);
        for (int o = 8; o < 64; o <<= 1) r += __shfl_xor(r, o);
        // lane q holds m[4 * (q & 1) + 2 * ((q >> 1) & 1) + ((q >> 2) & 1)]: keypoint u = 2 * (q & 1) + ((q >> 1) & 1), m01 if q & 4
        if (lane < 8) {
            const int u = 2 * (lane & 1) + ((lane >> 1) & 1), at = wave * kOrientPerWave + k0 + u;
            if (k0 + u < cnt) { if (lane & 4) s_m01[at] = r; else s_m10[at] = r; }
        }
    };
    if (cnt > 0) {
        uint32_t pa[4][4], pb[4][4];
        request(0, pa);
        for (int k0 = 0; k0 < cnt; k0 += 8) { // (uniform)
            if (k0 + 4 < cnt) request(k0 + 4, pb);
            reduce_store(k0, pa);
            if (k0 + 4 < cnt) {
                if (k0 + 8 < cnt) request(k0 + 8, pa);
                reduce_store(k0 + 4, pb);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kOrientWaves * kOrientPerWave; t += kOrientThreads) {
        const int j = s_j[t];
        if (j < 0) continue;
        const float ang = fast_atan2_dev((float)s_m01[t], (float)s_m10[t]);
        kps[j].angle = ang;
        if (d_cs) {
            // rotation of the rBRIEF pattern: a = (float)cos(angle * pi/180), b = (float)sin(...), evaluated in f64 like the CPU side
            const float rad = __fmul_rn(ang, (float)(3.1415926535897932384626433832795 / 180.f));
            d_cs[(size_t)b * kp_capacity + j] = make_float2((float)cos((double)rad), (float)sin((double)rad));
        }
    }
}

int launch_orb_orient(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr, vslam_keypoint* d_kps,
                      float2* d_cs, const int32_t* d_order, int kp_capacity, const int32_t* d_count, hipStream_t stream) {
    LevelTable T;
    fill_level_table(plan, &T);
    if (kp_capacity > kOrientPerWave * kOrientBlocks * kOrientWaves) { set_error("kp_capacity %d exceeds the orientation walk (%d)", kp_capacity, kOrientPerWave * kOrientBlocks * kOrientWaves); return VSLAM_ERR_ARG; }
    ProfScope prof__(stream, "orb_orient_kernel");
    hipLaunchKernelGGL(orb_orient_kernel, dim3(kOrientBlocks * ((B + 7) / 8 * 8)), dim3(kOrientThreads), 0, stream, T, d_imgs, img_bytes, pitch, d_pyr,
                       (size_t)plan.pyr_bytes, d_kps, d_cs, d_order, kp_capacity, d_count, kOrientBlocks, B);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

__device__ inline int block_rank_1024(bool flag, int* s_wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave_tot[wave] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < kAnmsBlock / 64; ++w) {
        const int c = s_wave_tot[w];
        if (w < wave) base += c;
        tot += c;
    }
    total = tot;
    return base + rank;
}

// Input: either 8 per-level lists (d_sel, d_sel_cnt; in_capacity = sel_cap) or one flat list (levels = 1).
// CAP = keypoints the workgroup can hold.  LDS per workgroup = 32 KB sort buffer (4096 u64, also the home of the grid lists) +
// CAP x (8 B position + 4 B response + 2 B source index): 77.6 KB at CAP = 3328, so TWO workgroups share a CU (and the register
// budget is held to 64 VGPRs for the same reason) -- the phases of this kernel are chains of barriers and LDS round trips that
// leave the CU idle most of the time, and a second image fills the gaps.  The f64 radii live in global memory (written once, read
// twice, coalesced); the output-order list reuses the response array, which is dead by then.
constexpr size_t anms_lds_bytes(int cap) { return (size_t)kMaxRows * 8 + (size_t)cap * (8 + 4 + 2) + 16 + 4 * (kAnmsBlock / 64 + kNLevels + 1 + 3); }
constexpr int kAnmsCapPipe = 3328; // the detector emits at most nfeatures = 3000 keypoints plus ties at the per-level cuts
static_assert(2 * anms_lds_bytes(kAnmsCapPipe) <= 160 * 1024, "two ANMS workgroups must fit one CU's LDS");

// [r5] SGPR budget: a wave of this device is allocated its SGPRs in blocks of 16 PLUS 16 (trap handler), out of 800 per SIMD: 8 waves per SIMD --
// two of these 16-wave workgroups on a CU -- need a count of <= 80, while the compiler's occupancy model says 8 waves up to 96
// (tools/scratch/occupancy_probe.hip).  The f64 cos / sin of `emit` is a real function CALL inside a kernel of this size (argument reduction
// not inlined): the callee's 77 SGPRs and 8 bytes of stack became the kernel's, and with 83 SGPRs the second workgroup of a CU never
// started -- ONE image per CU at a time, 1.04 ms per 1024 images, since round 2.  The batched pipeline never asks this kernel for the
// rotation (orb_orient_kernel writes it), so that code lives in its own instance (WITH_CS: vslam_orb_compute, one image).
template <int CAP, bool WITH_CS>
__global__ __launch_bounds__(kAnmsBlock, 8) void orb_anms_kernel(const vslam_keypoint* __restrict__ d_in, const int32_t* __restrict__ d_nin,
                                                             int nlists, int in_capacity, int anms_num, int regroup, int img_w,
                                                             int img_h, vslam_keypoint* __restrict__ d_kps, float2* __restrict__ d_cs, int32_t* __restrict__ d_order, int kp_capacity,
                                                             int32_t* __restrict__ d_count, int32_t* __restrict__ d_status, double* __restrict__ d_rad) {
    const int b = blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* skey = reinterpret_cast<unsigned long long*>(smem);                  // kMaxRows u64 (sort buffer; grid lists during the radius phase)
    float2* sxy = reinterpret_cast<float2*>(smem + (size_t)kMaxRows * 8);                    // CAP (x, y) pairs: one broadcast read per candidate
    float* sr = reinterpret_cast<float*>(sxy + CAP);                                         // CAP responses (radius phase only)
    uint16_t* sord = reinterpret_cast<uint16_t*>(sr);                                        // CAP : output order (after the radius phase)
    uint16_t* sidx = reinterpret_cast<uint16_t*>(sr + CAP);                                  // CAP : source index per rank
    double* srad = d_rad + (size_t)b * kMaxRows;                                             // radii, global
    unsigned char* tail = reinterpret_cast<unsigned char*>(sidx + CAP);
    unsigned long long& s_final = *reinterpret_cast<unsigned long long*>(tail);
    int* s_wave_tot = reinterpret_cast<int*>(tail + 16);                                     // kAnmsBlock / 64
    int* s_off = s_wave_tot + kAnmsBlock / 64;                                               // kNLevels + 1

    OPH_INIT();
    // ---- gather (lists in order): flat index g -> (list, i)
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 0; k < nlists; ++k) { s_off[k] = acc; acc += min(max(d_nin[b * nlists + k], 0), in_capacity); }
        s_off[nlists] = acc;
    }
    __syncthreads();
    int N = s_off[nlists];
    if (N > CAP) { if (threadIdx.x == 0) atomicOr(&d_status[b], kStAnmsOverflow); N = CAP; }
    const vslam_keypoint* in = d_in + (size_t)b * nlists * in_capacity;
    auto src_ptr = [&](int g) -> const vslam_keypoint* {
        int k = 0;
        for (int t = 1; t < nlists; ++t) if (g >= s_off[t]) k = t;
        return in + (size_t)k * in_capacity + (g - s_off[k]);
    };
    // rotation of the rBRIEF pattern: a = (float)cos(angle * pi/180), b = (float)sin(...), evaluated in f64 like the CPU side
    auto emit = [&](int r, const vslam_keypoint* kp) {
        d_kps[(size_t)b * kp_capacity + r] = *kp;
        if (WITH_CS && d_cs) {
            const float ang = __fmul_rn(kp->angle, (float)(3.1415926535897932384626433832795 / 180.f));
            d_cs[(size_t)b * kp_capacity + r] = make_float2((float)cos((double)ang), (float)sin((double)ang));
        }
    };
    // Walk order for the stages that follow (orientation, descriptors): the output slots sorted by their SOURCE index.  The source lists
    // are raster-sorted per level, so consecutive entries are neighbours in the image and the patches the next kernels fetch share cache
    // lines; in response order every 39 x 40 patch is ~5 KB of lines nobody else touches.  One mark per source index (the upper half of
    // the sort buffer is free by now), then a ranked compaction over the source indices.
    auto emit_order = [&](int cnt, auto src_of) {
        if (!d_order) return;
        uint16_t* mark = reinterpret_cast<uint16_t*>(smem + (size_t)kMaxRows * 4);
        __syncthreads();
        for (int g = threadIdx.x; g < N; g += kAnmsBlock) mark[g] = 0xFFFFu;
        __syncthreads();
        for (int r = threadIdx.x; r < cnt; r += kAnmsBlock) mark[src_of(r)] = (uint16_t)r;
        __syncthreads();
        int written = 0;
        for (int base = 0; base < N; base += kAnmsBlock) {
            const int g = base + threadIdx.x;
            const uint32_t m = g < N ? mark[g] : 0xFFFFu;
            int total;
            const int rk = block_rank_1024(m != 0xFFFFu, s_wave_tot, total);
            if (m != 0xFFFFu) d_order[(size_t)b * kp_capacity + written + rk] = (int32_t)m;
            written += total;
        }
    };
    const bool do_anms = anms_num > 0 && N >= anms_num; // visual_odometry.cpp:100
    int M = N; // count after ANMS
    if (do_anms) {
        // ---- sort by response, strongest first; ties by input index (stable)
        int np2 = 1;
        while (np2 < N) np2 <<= 1;
        for (int g = threadIdx.x; g < np2; g += kAnmsBlock) {
            unsigned long long k = ~0ull;
            if (g < N) k = ((unsigned long long)(~float_order_key(src_ptr(g)->response)) << 32) | (uint32_t)g;
            skey[g] = k;
        }
        __syncthreads();
        bitonic_sort_lds(skey, np2);
        for (int r = threadIdx.x; r < N; r += kAnmsBlock) {
            const int g = (int)(skey[r] & 0xFFFFFFFFu);
            const vslam_keypoint* kp = src_ptr(g);
            sidx[r] = (uint16_t)g; sxy[r] = make_float2(kp->x, kp->y); sr[r] = kp->response;
        }
        __syncthreads();
        OPH(24);
        // ---- suppression radius (visual_odometry.cpp:124-138): distance to the nearest keypoint whose response exceeds
        // 1.11 x mine, i.e. to the nearest of the ranks [0, lo).  The reference scans all of them (O(N^2)); here the keypoints
        // are binned into a uniform grid (cell lists sorted by rank) and a query walks outward ring by ring until the best
        // distance found is below the distance to the unvisited cells.  Only candidates that cannot be the minimum are
        // skipped and every distance is evaluated exactly as before, so the radii are bit-identical.
#ifndef VSLAM_ANMS_CELL_MIN
#define VSLAM_ANMS_CELL_MIN 24 // (32 -> 24: 0.75 -> 0.70 ms per 1024 KITTI-sized images; the grid stays below its 1023 cells: 52 x 16)
#endif
        const int csz = max(VSLAM_ANMS_CELL_MIN, (int)ceilf(sqrtf((float)img_w * (float)img_h * (1.f / 900.f))));
        const int gx = min(max((img_w + csz - 1) / csz, 1), 1023), gy = max(min((img_h + csz - 1) / csz, 1023 / gx), 1), ncell = gx * gy;
        int* ccnt = reinterpret_cast<int*>(skey);                 // the sort buffer is free until the radius sort
        int* coff = ccnt + 1024;
        uint16_t* tmpl = reinterpret_cast<uint16_t*>(coff + 1024); // ranks per cell, arrival order
        uint16_t* clist = tmpl + kMaxRows;                          // ranks per cell, ascending
        uint16_t* cof = clist + kMaxRows;                           // cell of every rank
        auto cell_of = [&](float x, float y) -> int {
            const int cx = min(max((int)(x / (float)csz), 0), gx - 1), cy = min(max((int)(y / (float)csz), 0), gy - 1);
            return cy * gx + cx;
        };
        for (int c = threadIdx.x; c <= ncell; c += kAnmsBlock) ccnt[c] = 0;
        __syncthreads();
        for (int r = threadIdx.x; r < N; r += kAnmsBlock) { const int c = cell_of(sxy[r].x, sxy[r].y); cof[r] = (uint16_t)c; atomicAdd(&ccnt[c], 1); }
        __syncthreads();
        if (threadIdx.x < 64) { // exclusive prefix over the cells by one wave (16 cells per lane)
            const int per = (ncell + 63) / 64, c0 = threadIdx.x * per, c1 = min(c0 + per, ncell);
            int mine = 0;
            for (int c = c0; c < c1; ++c) mine += ccnt[c];
            int incl = mine;
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += t; }
            int run = incl - mine;
            for (int c = c0; c < c1; ++c) { const int n = ccnt[c]; coff[c] = run; run += n; }
            if (threadIdx.x == 63) coff[ncell] = incl;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < ncell; c += kAnmsBlock) ccnt[c] = 0;
        __syncthreads();
        for (int r = threadIdx.x; r < N; r += kAnmsBlock) { const int c = cof[r]; tmpl[coff[c] + atomicAdd(&ccnt[c], 1)] = (uint16_t)r; }
        __syncthreads();
        for (int r = threadIdx.x; r < N; r += kAnmsBlock) { // rank-sorted position inside the cell (arrival order is not deterministic)
            const int c = cof[r], o0 = coff[c], o1 = coff[c + 1];
            int pos = 0;
            for (int t = o0; t < o1; ++t) pos += tmpl[t] < r;
            clist[o0 + pos] = (uint16_t)r;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += kAnmsBlock) {
            const float thr = __fmul_rn(sr[i], 1.11f);
            // first j in [0, i) with !(sr[j] > thr); sr is non-increasing
            int lo = 0, hi = i;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (sr[mid] > thr) lo = mid + 1; else hi = mid; }
            const float xi = sxy[i].x, yi = sxy[i].y;
            double best = 1.7976931348623157e308;
            auto visit = [&](int j) {
                const float2 pj = sxy[j];
                const float dx = __fsub_rn(xi, pj.x), dy = __fsub_rn(yi, pj.y);
                // dx^2 and dy^2 are exact in f64 (24-bit factors), so the fused form rounds once, exactly like mul + mul + add
                best = fmin(best, __fma_rn((double)dx, (double)dx, __dmul_rn((double)dy, (double)dy)));
            };
            // (both candidate loops fetch four candidates before the first distance: the compiler leaves a one-candidate loop
            // serial, one LDS round trip -- two in the cell lists -- per candidate)
            auto visit_p = [&](float2 pj) {
                const float dx = __fsub_rn(xi, pj.x), dy = __fsub_rn(yi, pj.y);
                best = fmin(best, __fma_rn((double)dx, (double)dx, __dmul_rn((double)dy, (double)dy)));
            };
            if (lo <= kAnmsBrute) {
                int j = 0;
                for (; j + 4 <= lo; j += 4) {
                    const float2 p0 = sxy[j], p1 = sxy[j + 1], p2 = sxy[j + 2], p3 = sxy[j + 3];
                    visit_p(p0); visit_p(p1); visit_p(p2); visit_p(p3);
                }
                for (; j < lo; ++j) visit(j);
            } else {
                const int ci = cof[i], cxi = ci % gx, cyi = ci / gx;
                const int kmax = max(max(cxi, gx - 1 - cxi), max(cyi, gy - 1 - cyi));
                for (int k = 0; k <= kmax; ++k) {
                    if (k > 0) { // every unvisited keypoint lies outside the square of rings < k
                        const double bx = fmin((double)xi - (double)((cxi - k + 1) * csz), (double)((cxi + k) * csz) - (double)xi);
                        const double by = fmin((double)yi - (double)((cyi - k + 1) * csz), (double)((cyi + k) * csz) - (double)yi);
                        const double bnd = fmin(bx, by) * (1.0 - 1e-6); // (margin: the reference's dx, dy are f32-rounded differences)
                        if (bnd > 0 && best <= bnd * bnd) break;
                    }
                    const int y0 = cyi - k, y1 = cyi + k, x0 = cxi - k, x1 = cxi + k;
                    for (int cy = max(y0, 0); cy <= min(y1, gy - 1); ++cy) {
                        const bool edge_row = cy == y0 || cy == y1;
                        const int step = (edge_row || k == 0) ? 1 : 2 * k; // interior rows of the ring: only the two end cells
                        for (int cx = x0; cx <= x1; cx += step) {
                            if (cx < 0 || cx >= gx) continue;
                            const int c = cy * gx + cx;
                            for (int t = coff[c], t1 = coff[c + 1]; t < t1; t += 4) { // the list ascends in rank: stop at the first j >= lo
                                int jj[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) jj[q] = t + q < t1 ? (int)clist[t + q] : 0x7FFFFFFF;
                                float2 pp[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) pp[q] = sxy[min(jj[q], N - 1)];
#pragma unroll
                                for (int q = 0; q < 4; ++q) if (jj[q] < lo) visit_p(pp[q]);
                                if (jj[3] >= lo) break;
                            }
                        }
                    }
                }
            }
            srad[i] = lo > 0 ? sqrt(best) : 1.7976931348623157e308;
        }
        __syncthreads();
        OPH(25);
        // ---- the num-th largest radius (:141-146): sort the radii descending
        for (int g = threadIdx.x; g < np2; g += kAnmsBlock) skey[g] = g < N ? ~(unsigned long long)__double_as_longlong(srad[g]) : ~0ull;
        __syncthreads();
        bitonic_sort_lds(skey, np2);
        if (threadIdx.x == 0) s_final = ~skey[anms_num - 1];
        __syncthreads();
        OPH(26);
        const double final_radius = __longlong_as_double((long long)s_final);
        // ---- keep rad >= final radius, in response order (:147-153)
        int written = 0;
        for (int base = 0; base < N; base += kAnmsBlock) {
            const int i = base + threadIdx.x;
            const bool keep = i < N && srad[i] >= final_radius;
            int total;
            const int r = block_rank_1024(keep, s_wave_tot, total);
            if (keep) sord[written + r] = sidx[i];
            written += total;
        }
        __syncthreads();
        M = written;
    } else {
        for (int g = threadIdx.x; g < N; g += kAnmsBlock) sord[g] = (uint16_t)g;
        __syncthreads();
    }
    // ---- cv::ORB::compute prologue: border cull in level-0 coordinates, stable regroup by octave
    if (regroup) {
        int np2 = 1;
        while (np2 < M) np2 <<= 1;
        uint32_t* k32 = reinterpret_cast<uint32_t*>(skey);
        for (int r = threadIdx.x; r < np2; r += kAnmsBlock) {
            uint32_t k = 0xFFFFFFFFu;
            if (r < M) {
                const vslam_keypoint* kp = src_ptr(sord[r]);
                const bool inside = kp->x >= (float)kEdge && kp->x < (float)(img_w - kEdge) && kp->y >= (float)kEdge && kp->y < (float)(img_h - kEdge);
                if (inside) k = ((uint32_t)min(max(kp->octave, 0), 15) << 16) | (uint32_t)r;
            }
            k32[r] = k;
        }
        __syncthreads();
        if (np2 > 1) bitonic_sort_lds(k32, np2);
        int cnt = 0;
        // count survivors (keys != ~0): they are a prefix after sorting
        int lo = 0, hi = M;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (k32[mid] != 0xFFFFFFFFu) lo = mid + 1; else hi = mid; }
        cnt = lo;
        if (cnt > kp_capacity) { if (threadIdx.x == 0) atomicOr(&d_status[b], kStOutOverflow); cnt = kp_capacity; }
        for (int r = threadIdx.x; r < cnt; r += kAnmsBlock) emit(r, src_ptr(sord[k32[r] & 0xFFFFu]));
        if (threadIdx.x == 0) d_count[b] = cnt;
        emit_order(cnt, [&](int r) -> int { return sord[k32[r] & 0xFFFFu]; });
        OPH(27);
    } else {
        int cnt = M;
        if (cnt > kp_capacity) { if (threadIdx.x == 0) atomicOr(&d_status[b], kStOutOverflow); cnt = kp_capacity; }
        for (int r = threadIdx.x; r < cnt; r += kAnmsBlock) emit(r, src_ptr(sord[r]));
        if (threadIdx.x == 0) d_count[b] = cnt;
        emit_order(cnt, [&](int r) -> int { return sord[r]; });
    }
}

template <int CAP>
static int launch_anms_t(int B, const vslam_keypoint* d_in, const int32_t* d_nin, int nlists, int in_capacity, int anms_num,
                         int regroup, int img_w, int img_h, vslam_keypoint* d_kps, float2* d_cs, int32_t* d_order, int kp_capacity, int32_t* d_count,
                         int32_t* d_status, double* d_rad, hipStream_t stream) {
    constexpr size_t smem = anms_lds_bytes(CAP);
    static bool attr_set[16] = {false}; // per device: the > 64 KB dynamic-LDS opt-in is a per-device function attribute
    int dev = 0;
    VS_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(orb_anms_kernel<CAP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(orb_anms_kernel<CAP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    ProfScope prof__(stream, "orb_anms_kernel");
    if (d_cs)
        hipLaunchKernelGGL((orb_anms_kernel<CAP, true>), dim3(B), dim3(kAnmsBlock), smem, stream, d_in, d_nin, nlists, in_capacity, anms_num, regroup,
                           img_w, img_h, d_kps, d_cs, d_order, kp_capacity, d_count, d_status, d_rad);
    else
        hipLaunchKernelGGL((orb_anms_kernel<CAP, false>), dim3(B), dim3(kAnmsBlock), smem, stream, d_in, d_nin, nlists, in_capacity, anms_num, regroup,
                           img_w, img_h, d_kps, d_cs, d_order, kp_capacity, d_count, d_status, d_rad);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

int launch_orb_anms(const OrbPlan& plan, int B, const vslam_keypoint* d_sel, const int32_t* d_sel_cnt, int sel_cap, int anms_num,
                    int regroup, vslam_keypoint* d_kps, float2* d_cs, int32_t* d_order, int kp_capacity, int32_t* d_count, int32_t* d_status, double* d_rad,
                    hipStream_t stream) {
    return launch_anms_t<kAnmsCapPipe>(B, d_sel, d_sel_cnt, kNLevels, sel_cap, anms_num, regroup, plan.w, plan.h, d_kps, d_cs, d_order, kp_capacity, d_count,
                                       d_status, d_rad, stream);
}

int launch_anms_flat(int B, const vslam_keypoint* d_in, const int32_t* d_nin, int in_capacity, int anms_num, int regroup, int img_w,
                     int img_h, vslam_keypoint* d_kps, float2* d_cs, int32_t* d_order, int kp_capacity, int32_t* d_count, int32_t* d_status, double* d_rad,
                     hipStream_t stream) {
    return launch_anms_t<kMaxRows>(B, d_in, d_nin, 1, in_capacity, anms_num, regroup, img_w, img_h, d_kps, d_cs, d_order, kp_capacity, d_count, d_status, d_rad,
                                   stream);
}

// ------------------------------------------------------------------------------------------- K6 blur + rBRIEF
// K6a orb_blur_kernel: GaussianBlur 7x7 sigma 2 (8-bit fixed point, BORDER_REFLECT_101) of every pyramid level into a
// second pyramid, one launch for all levels.  256 x 64 output tiles: (264 x 70) raw pixels staged in LDS -- the arithmetic of
// cv::GaussianBlur's 8U path: taps cvRound(k*256) = {18,34,49,55,49,34,18} per pass, (sum + 2^15) >> 16 after the column pass.
constexpr int kBlurTileW = 256, kBlurTileH = VSLAM_BLUR_TILE_H, kBlurWaveRows = kBlurTileH / 4; // workgroup tile; a wave owns a quarter of the rows, all 256 columns
constexpr int kBlurRawH = kBlurTileH + 6, kBlurRawChunks = (kBlurTileW + 8 + 15) / 16, kBlurRawPitch = 16 * kBlurRawChunks; // raw tile starts at (x0 - 4, y0 - 3); rows of 17 x 16 B

struct BlurTable {
    int w[kNLevels], h[kNLevels], pitch[kNLevels], pyr_off[kNLevels], blur_off[kNLevels];
    int tiles_x[kNLevels], tile_off[kNLevels + 1];
};

static void fill_blur_table(const OrbPlan& plan, BlurTable* T) {
    int tiles = 0;
    for (int l = 0; l < kNLevels; ++l) {
        const OrbLevel& L = plan.lv[l];
        T->w[l] = L.w; T->h[l] = L.h; T->pitch[l] = (L.w + 63) & ~63; T->pyr_off[l] = L.pyr_off; T->blur_off[l] = plan.blur_off[l];
        T->tiles_x[l] = (L.w + kBlurTileW - 1) / kBlurTileW;
        T->tile_off[l] = tiles;
        tiles += T->tiles_x[l] * ((L.h + kBlurTileH - 1) / kBlurTileH);
    }
    T->tile_off[kNLevels] = tiles;
}

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ inline uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) { // v_dot2_u32_u16: a.lo * b.lo + a.hi * b.hi + c
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b), c, false);
}

// The raw tile (reflect-101 at the image border) is staged in LDS once; after that everything stays in registers.  A wave streams
// down its 16 + 6 raw rows; a lane owns four consecutive columns: the row pass is two v_dot4_u32_u8 per output on the three aligned
// dwords of the lane's 12-byte window, its u16 results are paired with the previous row's (one v_lshl_or per column), and the
// column pass of output row t - 6 is four v_dot2_u32_u16 on the pairs (t-6,t-5), (t-4,t-3), (t-2,t-1), (t-1,t) -- the last with
// weights (0, 18) -- seeded with the rounding constant; the high halves of two sums are paired (v_perm), saturated at 255
// (v_pk_min_u16) and packed into the dword that is stored.  The loop over the 22 rows is unrolled: the six live row pairs rotate by name.
__global__ __launch_bounds__(256) void orb_blur_kernel(BlurTable T, const uint8_t* __restrict__ d_imgs, size_t img_bytes, int pitch0,
                                                      const uint8_t* __restrict__ d_pyr, size_t pyr_bytes, uint8_t* __restrict__ d_blur,
                                                      size_t blur_bytes) {
    const int b = blockIdx.y;
    // Workgroups are dealt to the 8 XCDs round-robin by linear id and every XCD has its own L2: the grid is padded to a multiple
    // of 8 and XCD x walks the contiguous tile range [x * chunk, (x + 1) * chunk) of every image, so tiles that share halo lines
    // meet in one L2 a few workgroups apart.
    const int chunk = gridDim.x >> 3;
    int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3), l = 0;
    if (tile >= T.tile_off[kNLevels]) return;
#pragma unroll
    for (int k = 1; k < kNLevels; ++k) if (tile >= T.tile_off[k]) l = k;
    tile -= T.tile_off[l];
    const int W = T.w[l], H = T.h[l];
    const uint8_t* src = l == 0 ? d_imgs + (size_t)b * img_bytes : d_pyr + (size_t)b * pyr_bytes + T.pyr_off[l];
    const int spitch = l == 0 ? pitch0 : T.pitch[l];
    uint8_t* dst = d_blur + (size_t)b * blur_bytes + T.blur_off[l];
    const int dpitch = T.pitch[l];
    const int ox = (tile % T.tiles_x[l]) * kBlurTileW, oy = (tile / T.tiles_x[l]) * kBlurTileH;

    __shared__ __attribute__((aligned(16))) uint8_t raw[kBlurRawH * kBlurRawPitch];
    load_tile_b128<256, kBlurRawChunks, kBlurRawH, true>(raw, src, spitch, W, H, ox - 4, oy - 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = wave * kBlurWaveRows;                // first output row of this wave inside the tile
    const int nrows = min(kBlurWaveRows, H - (oy + row0)); // wave-uniform
    const int x = ox + 4 * lane;
    if (nrows <= 0) return;
    // taps cvRound(k * 256) = {18, 34, 49, 55, 49, 34, 18}, packed for v_dot4_u32_u8 / v_dot2_u32_u16
    constexpr uint32_t W0 = 18u | 34u << 8 | 49u << 16 | 55u << 24, W1 = 49u | 34u << 8 | 18u << 16;
    constexpr uint32_t V0 = 18u | 34u << 16, V1 = 49u | 55u << 16, V2 = 49u | 34u << 16, V3 = 18u << 16;
    const uint32_t* rp = reinterpret_cast<const uint32_t*>(raw + row0 * kBlurRawPitch) + lane; // window dwords A, B, C of raw row row0 + t
    uint8_t* out = dst + (size_t)(oy + row0) * dpitch + x;
    uint32_t P[kBlurWaveRows + 6][4], hprev[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < kBlurWaveRows + 6; ++t) {
        if (t - 6 >= nrows) break; // uniform
        const uint32_t A = rp[t * (kBlurRawPitch / 4)], B = rp[t * (kBlurRawPitch / 4) + 1], C = rp[t * (kBlurRawPitch / 4) + 2];
#pragma unroll
        for (int o = 0; o < 4; ++o) { // output column x + o reads window bytes o + 1 .. o + 7
            const uint32_t lo = o == 3 ? B : __builtin_amdgcn_alignbyte(B, A, o + 1), hi = o == 3 ? C : __builtin_amdgcn_alignbyte(C, B, o + 1);
            const uint32_t h = __builtin_amdgcn_udot4(hi, W1, __builtin_amdgcn_udot4(lo, W0, 0u, false), false); // <= 255 * 257
            P[t][o] = hprev[o] | h << 16; // rows (t - 1, t)
            hprev[o] = h;
        }
        if (t >= 6) {
            uint32_t sum[4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                sum[o] = udot2(P[t][o], V3, udot2(P[t - 1][o], V2, udot2(P[t - 3][o], V1, udot2(P[t - 5][o], V0, 1u << 15))));
            // sum >> 16 can reach 257 (the taps add up to 257 per pass): saturate as cv::saturate_cast<uchar> does
            const us2_t lim = {255, 255};
            const us2_t p01 = __builtin_elementwise_min(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(sum[1], sum[0], 0x07060302u)), lim);
            const us2_t p23 = __builtin_elementwise_min(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(sum[3], sum[2], 0x07060302u)), lim);
            const uint32_t px = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p23), __builtin_bit_cast(uint32_t, p01), 0x06040200u);
            if (x < W) *reinterpret_cast<uint32_t*>(out + (size_t)(t - 6) * dpitch) = px; // (columns past W land in the row's padding)
        }
    }
}

int launch_orb_blur(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr, uint8_t* d_blur,
                    hipStream_t stream) {
    BlurTable T;
    fill_blur_table(plan, &T);
    ProfScope prof__(stream, "orb_blur_kernel");
    hipLaunchKernelGGL(orb_blur_kernel, dim3(8 * ((T.tile_off[kNLevels] + 7) / 8), B), dim3(256), 0, stream, T, d_imgs, img_bytes, pitch, d_pyr,
                       (size_t)plan.pyr_bytes, d_blur, (size_t)plan.blur_bytes);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

// K1+K6a fused: orb_pyrblur_kernel.  The separate resize and blur kernels each stream the whole pyramid (resize: level l in, level l + 1
// out; blur: level l in, blurred level l out): 1.07 ms and ~3.3 GB of HBM traffic per 512 images, both at the memory system's rate.
// One launch per level stages the 256 x 64 level-l tile (+ halo) in LDS ONCE and produces from it
//   (1) the pixels of level l + 1 whose top-left source pixel lies in the tile (cv::resize INTER_LINEAR 8U arithmetic as in
//       orb_resize_kernel: v_perm picks the source byte pair out of the window, v_dot2_u32_u16 is the horizontal pass; the window now
//       comes from LDS: three aligned dwords, v_alignbyte), and
//   (2) the blurred level l (the register-streaming pass of orb_blur_kernel, unchanged).
// The tile is staged with reflect-101 coordinates (what the blur needs); the resize clamps its row index itself and its column
// tables never weight a pixel beyond the image, so the reflected halo is never interpolated.
#ifndef VSLAM_ORB_BLUR_NT
#define VSLAM_ORB_BLUR_NT 1 // non-temporal stores of the blurred pyramid (pyramid + blur 1.04 -> 1.02 ms, FAST 0.90 -> 0.89 ms per 512 images)
#endif
// [r6] (3) FAST-9/16 + 3x3 NMS of level l run on the SAME staged tile (orb_fast_kernel staged every level a second time: 0.54 of its
// 1.82 ms per 1024 images, and 8 GB of re-reads per step).  The tile carries a halo of 4 rows above / below (the blur needs 3, the ring of a
// score-halo pixel 3 + 1) and 4 / 12 columns left / right, so every pixel the 256 x 64 tile emits finds its ring and its eight neighbours'
// rings in LDS.  Structure (per wave, no workgroup barrier until the NMS):
//   pre-test   a lane owns the 4 columns of an aligned dword and walks 6 centre rows at a time: the dwords of 12 raw rows are widened once to
//              packed i16 (even / odd pixels), the compass rule "two adjacent compass pixels both brighter than v + t or both darker than
//              v - t" costs 12 packed operations per pixel pair, the verdict is the sign of a packed difference shifted into a per-lane mask;
//   queue      set bits become 16-bit tile positions in a 128-entry per-wave LDS queue (ballot ranks, the count lives in an SGPR);
//              whenever 64 are queued they are scored with ALL lanes busy (fast_score: corner <=> score >= threshold);
//   corners    scores go to the (66 x 258) score tile, corners that can be emitted to a per-wave list;
//   NMS        after one barrier every wave checks its own list against the score tile; survivors are collected in the (now dead) pixel
//              tile and appended to the level's corner list with ONE global atomic per workgroup.
// The workgroup's corner list holds kPfCornerCap entries (7 % of the tile's pixels); a denser tile takes the slow path: every thread scans score dwords.
constexpr int kPfRawH = kBlurTileH + 8;                       // raw rows: tile rows -4 .. kBlurTileH + 3
constexpr int kPfScPitch = 264, kPfScH = kBlurTileH + 2;      // score tile: rows -1 .. kBlurTileH, byte column = raw column (tile column + 4)
#ifndef VSLAM_PF_WAVES
#define VSLAM_PF_WAVES 8
#endif
constexpr int kPfWaves = VSLAM_PF_WAVES;     // waves per tile (4 | 8): the LDS tile allows four workgroups per CU, so 8 waves per tile fill the CU's 32 wave slots
constexpr int kPfQueue = 128;                // per wave: < 64 left over + the <= 64 candidates of one ballot
constexpr int kPfRawBytes = kPfRawH * kBlurRawPitch, kPfScBytes = kPfScH * kPfScPitch;
constexpr int kPfCornerCap = (40960 - 32 - kPfRawBytes - kPfScBytes - 2 * kPfQueue * kPfWaves) / 2; // per workgroup: corners (score >= threshold) the tile may emit, before the NMS (what is left of a quarter of the CU's LDS)
static_assert(kPfCornerCap >= 900, "corner list of orb_pyrblur_kernel");
static_assert(kBlurTileW == 256 && kBlurRawPitch == 272, "the FAST phase of orb_pyrblur_kernel assumes 64 lanes x 4 columns and 272-byte raw rows");
static_assert(kPfRawBytes >= 4 * (kBlurTileW * kBlurTileH / 4), "the NMS survivors are collected in the pixel tile's storage");
// [r6] The 7 x 7 blur of the fused kernel on the MATRIX cores (NW = 8 only): the kernel is bound by the integer VALU rate, the matrix pipe is idle,
// and the separable blur is two banded integer matrix products.  v_mfma_i32_16x16x32_i8 (A, B: 8 bytes per lane, k = 8 (lane >> 4) + byte; D lane l, register v =
// D[4 (l >> 4) + v][l & 15]; tools/scratch/mfma16_layout.hip).  A wave owns a strip of 32 tile columns, two 16-column tiles, all rows:
//   row pass     h'(16 raw rows x 16 columns) = (raw ^ 0x80)(16 rows x 32 raw columns) . Bh(32 x 16): the band of the taps; the columns of the window that no output
//                column reads (k-group 3) carry the constant 1 against weights 16 x 8 = 128, which centres the result: h' = h - 257 * 128 + 128 = h - 32768 is a
//                signed 16-bit value.  Five row tiles (raw rows 0 .. 79: rows 72 .. 79 lie behind the pixel tile and only meet zero weights).
//   planes       the D registers of a lane are four consecutive ROWS of one column: their low bytes (^ 0x80: signed) and high bytes, one v_perm chain per
//                tile, are the k-slots of the column pass's A operand as they stand -- k-slot (q, v) of k-group g = raw row 16 (T + q) + 4 g + v --, no lane moves
//   column pass  out^T(16 columns x 16 rows) = planes(T, T + 1) . Bv: one product for the low, one for the high plane; D lane (n, g) = four consecutive COLUMNS of output
//                row 16 T + n: (W << 8) + V + const is the reference's 24-bit sum + 2^15, >> 16, saturated, one dword
//   store        through LDS (the score tile's storage, 272-byte rows): after a barrier every lane stores 16 contiguous bytes of a row -- the round-5 experiment
//                (32 x 32 tiles, tools/scratch/orb_blur_mfma.hip.txt) stored 16-byte row pieces straight from the accumulator layout and lost its gain there.
// ~230 VALU instructions per strip instead of ~410.
#ifndef VSLAM_ORB_BLUR_MFMA
#define VSLAM_ORB_BLUR_MFMA 1
#endif
#ifndef VSLAM_PYRBLUR_DBG
#define VSLAM_PYRBLUR_DBG 0 // tuning aid (timing only, outputs incomplete): 1 = no blur half, 2 = no resize half, 4 = no FAST, 8 = FAST pre-test without the scores
#endif
typedef int bl_v4i __attribute__((ext_vector_type(4)));
struct BlurMfmaLane { uint32_t bh[64][2], g0[64], g1[64]; };
constexpr uint32_t blur_tap(int d) { return d == 0 || d == 6 ? 18u : d == 1 || d == 5 ? 34u : d == 2 || d == 4 ? 49u : d == 3 ? 55u : 0u; }
constexpr BlurMfmaLane make_blur_mfma_lane() {
    BlurMfmaLane t{};
    for (int lane = 0; lane < 64; ++lane) {
        const int j = lane & 15, g = lane >> 4;
        for (int q = 0; q < 2; ++q) {
            uint32_t w = 0;
            for (int bb = 0; bb < 4; ++bb) { // row pass: k = 8 g + 4 q + bb is raw column c0 + k; output column j reads raw columns c0 + j + 1 .. c0 + j + 7
                const int k = 8 * g + 4 * q + bb;
                w |= (g == 3 ? 16u : blur_tap(k - j - 1)) << (8 * bb);
            }
            t.bh[lane][q] = w;
        }
        uint32_t a = 0, b = 0;
        for (int v = 0; v < 4; ++v) { // column pass: k-slot (q, v) of k-group g is raw row 16 (T + q) + 4 g + v; output row 16 T + n (n = j) reads raw rows 16 T + n + 1 .. + 7
            a |= blur_tap(4 * g + v - j - 1) << (8 * v);
            b |= blur_tap(16 + 4 * g + v - j - 1) << (8 * v);
        }
        t.g0[lane] = a; t.g1[lane] = b;
    }
    return t;
}
__device__ const BlurMfmaLane g_blur_mfma_lane = make_blur_mfma_lane();

struct PyrBlurArgs {
    const uint8_t* src_base; size_t src_img_stride; int spitch, sw, sh;      // level l (raw)
    uint8_t* blur_base; size_t blur_img_stride; int bpitch;                  // blurred level l
    uint8_t* dst_base; size_t dst_img_stride; int dpitch, dw, dh;            // level l + 1 (nullptr at the last level)
    const int* xofs; const short* ialpha; const int* yofs; const short* ibeta; // resize tables of level l + 1
    const int* tile_dx; const int* tile_dy;                                  // output ownership per tile column / row
    int tiles_x;
    // FAST of level l (corners == nullptr: descriptor-only call, no detection)
    uint32_t* corners; size_t corner_img_stride; int corner_cap; int32_t* corner_cnt; int32_t* status; int thr, level;
};
__device__ inline int mbcnt64(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

template <int NW>
__global__ __launch_bounds__(64 * NW, 8) void orb_pyrblur_kernel(PyrBlurArgs a) { // (four tiles per CU: 8 waves per SIMD at NW = 8)
    constexpr int NT = 64 * NW;
    const int b = blockIdx.y;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const int ox = tx * kBlurTileW, oy = ty * kBlurTileH;
    const int W = a.sw, H = a.sh;
    const uint8_t* src = a.src_base + (size_t)b * a.src_img_stride;
    // (one array: the pre-test of the last wave reads up to two pixel rows past the tile for centre rows it then skips -- they land in the score tile)
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[kPfRawBytes + kPfScBytes];
    uint8_t* const raw = s_tile;
    uint8_t* const sc = s_tile + kPfRawBytes;
    __shared__ uint16_t s_wq[NW * kPfQueue];
    __shared__ uint16_t s_cq[kPfCornerCap];
    __shared__ int s_ocount, s_obase, s_dense, s_ccount;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the emitted pixels of this tile (cv::ORB keeps corners >= edgeThreshold from the level border), in image coordinates, inclusive
    const int ex_lo = max(ox, kEdge), ex_hi = min(ox + kBlurTileW, W - kEdge) - 1, ey_lo = max(oy, kEdge), ey_hi = min(oy + kBlurTileH, H - kEdge) - 1;
    const bool do_fast = a.corners != nullptr && ex_lo <= ex_hi && ey_lo <= ey_hi; // uniform
    constexpr bool kMfmaBlur = NW == 8 && VSLAM_ORB_BLUR_MFMA && !(VSLAM_PYRBLUR_DBG & 1); // (the matrix-core blur stages its output in the score tile and zeroes it afterwards)
    if (do_fast) {
        if (!kMfmaBlur)
            for (int i = threadIdx.x; i < kPfScBytes / 16; i += NT) reinterpret_cast<uint4*>(sc)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x == 0) { s_ocount = 0; s_dense = 0; s_ccount = 0; }
    }
    OPH_INIT();
    load_tile_b128<NT, kBlurRawChunks, kPfRawH, true>(raw, src, a.spitch, W, H, ox - 4, oy - 4);
    __syncthreads();
    OPH(32);
    // ---- (1) level l + 1
    if (a.dst_base && !(VSLAM_PYRBLUR_DBG & 2)) {
        const int dx_lo = a.tile_dx[tx], dx_hi = a.tile_dx[tx + 1], dy_lo = a.tile_dy[ty], dy_hi = a.tile_dy[ty + 1]; // uniform
        const int dx0 = (dx_lo & ~3) + 4 * lane; // this lane's aligned quad of output columns
        // [r6] a wave owns a CONTIGUOUS run of output rows: consecutive output rows mostly share a source row (scale 1.2: the lower source row of
        // output row dy is the upper one of dy + 1 five times out of six), and its horizontal pass -- three LDS reads, six byte moves, four dot products
        // per quad -- is then taken over instead of repeated (~15 % of the resize half).
        // The row tables of ALL rows of this wave are fetched once, one row per lane (<= 14 rows per wave: 64 source rows / 1.2 / NW waves),
        // by every lane (v_readlane below reads lanes that own no output column), and handed out with v_readlane: a table load per row
        // would be a dependent memory round trip at the head of every row
        const int rows_per_wave = (dy_hi - dy_lo + NW - 1) / NW;
        const int dy_w0 = dy_lo + wave * rows_per_wave, dy_w1 = min(dy_w0 + rows_per_wave, dy_hi);
        const int my_dy = min(dy_w0 + lane, a.dh - 1);
        const int my_sy = a.yofs[my_dy];
        const uint32_t my_beta = *reinterpret_cast<const uint32_t*>(a.ibeta + 2 * my_dy); // (b0, b1) as two shorts
        // EVERY lane runs the loop below (lanes past the tile's last quad compute on clamped table entries and store nothing): the row
        // hand-out reads lanes 0..15 with v_readlane, and a lane that a divergent branch has switched off holds no defined value for it
        // (the compiler may sink its loads into the branch)
        if (dy_w0 < dy_w1) { // uniform
            uint8_t* dst = a.dst_base + (size_t)b * a.dst_img_stride;
            const int dxl = min(dx0, (dx_hi + 3) & ~3);                              // tables are padded to whole quads (+ one more)
            const int4 xo = *reinterpret_cast<const int4*>(a.xofs + dxl);
            const uint4 al = *reinterpret_cast<const uint4*>(a.ialpha + 2 * dxl);
            const int sxs[4] = {xo.x, xo.y, xo.z, xo.w};
            const uint32_t alw[4] = {al.x, al.y, al.z, al.w};
            // window = 8 bytes from tile column c0 on: the four outputs interpolate source columns xo.x .. xo.w + 1 <= xo.x + 5.  A quad that
            // straddles the left end of the tile's range starts at most 4 source columns left of the tile (3 outputs x 1.2), inside the halo.
            const int c0 = min(max(xo.x - (ox - 4), 0), kBlurRawPitch - 12);
            const int i0 = c0 >> 2;
            const uint32_t off = (uint32_t)(c0 & 3);
            uint32_t sel[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sel[k] = 0x0c010c00u + __umul24((uint32_t)(sxs[k] - (ox - 4) - c0) & 7u, 0x00010001u); // rel = 0..7 (garbage for unowned columns)
            const uint32_t* rawd = reinterpret_cast<const uint32_t*>(raw);
            auto hpass = [&](int r, uint32_t (&h)[4]) { // horizontal pass of raw row r for this lane's quad
                const uint32_t* p = rawd + r * (kBlurRawPitch / 4) + i0;
                const uint32_t a0 = p[0], a1 = p[1], a2 = p[2];
                const uint32_t rl = __builtin_amdgcn_alignbyte(a1, a0, off), rh = __builtin_amdgcn_alignbyte(a2, a1, off);
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = udot2_u16(__builtin_amdgcn_perm(rh, rl, sel[k]), alw[k]);
            };
            uint32_t h0[4], h1[4];
            int r1_prev = -1000;
            int it = 0;
            for (int dy = dy_w0; dy < dy_w1; ++dy, ++it) { // wave-uniform
                const int sy = __builtin_amdgcn_readlane(my_sy, it);
                const uint32_t beta = (uint32_t)__builtin_amdgcn_readlane((int)my_beta, it);
                const int r0 = min(max(sy, 0), H - 1) - (oy - 4), r1 = min(max(sy + 1, 0), H - 1) - (oy - 4);
                const uint32_t b0 = (uint32_t)(int)(short)(beta & 0xFFFFu), b1 = (uint32_t)(int)(short)(beta >> 16);
                if (r0 == r1_prev) { // (uniform) the upper source row is the previous output row's lower one
#pragma unroll
                    for (int k = 0; k < 4; ++k) h0[k] = h1[k];
                } else hpass(r0, h0);
                hpass(r1, h1);
                r1_prev = r1;
                const uint32_t packed = resize_vertical4(h0, h1, b0, b1);
                uint8_t* o = dst + (size_t)dy * a.dpitch + dx0;
                if (dx0 >= dx_lo && dx0 + 4 <= dx_hi) *reinterpret_cast<uint32_t*>(o) = packed;
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (dx0 + k >= dx_lo && dx0 + k < dx_hi) o[k] = (uint8_t)(packed >> (8 * k));
                }
            }
        }
    }
    OPH(33);
    // ---- (2) blurred level l (see orb_blur_kernel).  NW = 4: a wave owns 16 rows x 256 columns, a lane four pixels.  NW = 8: a wave owns
    // 16 rows x 128 columns, a lane two pixels (the row pass of the 6 halo rows is repeated per row band, so halving the bands' height would cost more)
    if (!(VSLAM_PYRBLUR_DBG & 1)) {
        uint8_t* dstb = a.blur_base + (size_t)b * a.blur_img_stride;
        constexpr uint32_t W0 = 18u | 34u << 8 | 49u << 16 | 55u << 24, W1 = 49u | 34u << 8 | 18u << 16;
        constexpr uint32_t V0 = 18u | 34u << 16, V1 = 49u | 55u << 16, V2 = 49u | 34u << 16, V3 = 18u << 16;
        const us2_t lim = {255, 255};
        if constexpr (NW == 4) {
            const int row0 = wave * kBlurWaveRows;
            const int nrows = min(kBlurWaveRows, H - (oy + row0)); // wave-uniform
            const int x = ox + 4 * lane;
            if (nrows > 0) {
                const uint32_t* rp = reinterpret_cast<const uint32_t*>(raw + (row0 + 1) * kBlurRawPitch) + lane; // tile row r reads raw rows r + 1 .. r + 7
                uint8_t* out = dstb + (size_t)(oy + row0) * a.bpitch + x;
                uint32_t P[kBlurWaveRows + 6][4], hprev[4] = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < kBlurWaveRows + 6; ++t) {
                    if (t - 6 >= nrows) break; // uniform
                    const uint32_t A = rp[t * (kBlurRawPitch / 4)], Bw = rp[t * (kBlurRawPitch / 4) + 1], C = rp[t * (kBlurRawPitch / 4) + 2];
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const uint32_t lo = o == 3 ? Bw : __builtin_amdgcn_alignbyte(Bw, A, o + 1), hi = o == 3 ? C : __builtin_amdgcn_alignbyte(C, Bw, o + 1);
                        const uint32_t h = __builtin_amdgcn_udot4(hi, W1, __builtin_amdgcn_udot4(lo, W0, 0u, false), false);
                        P[t][o] = hprev[o] | h << 16;
                        hprev[o] = h;
                    }
                    if (t >= 6) {
                        uint32_t sum[4];
#pragma unroll
                        for (int o = 0; o < 4; ++o)
                            sum[o] = udot2(P[t][o], V3, udot2(P[t - 1][o], V2, udot2(P[t - 3][o], V1, udot2(P[t - 5][o], V0, 1u << 15))));
                        const us2_t p01 = __builtin_elementwise_min(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(sum[1], sum[0], 0x07060302u)), lim);
                        const us2_t p23 = __builtin_elementwise_min(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(sum[3], sum[2], 0x07060302u)), lim);
                        const uint32_t px = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p23), __builtin_bit_cast(uint32_t, p01), 0x06040200u);
#if VSLAM_ORB_BLUR_NT
                        if (x < W) __builtin_nontemporal_store(px, reinterpret_cast<uint32_t*>(out + (size_t)(t - 6) * a.bpitch)); // read again only by the descriptor kernel, five kernels later
#else
                        if (x < W) *reinterpret_cast<uint32_t*>(out + (size_t)(t - 6) * a.bpitch) = px;
#endif
                    }
                }
            }
        } else if constexpr (!kMfmaBlur) {
            const int band = wave >> 1, half = wave & 1;
            const int row0 = band * kBlurWaveRows;
            const int nrows = min(kBlurWaveRows, H - (oy + row0)); // wave-uniform
            const int xt = 128 * half + 2 * lane;                  // tile column of this lane's first pixel
            const int x = ox + xt;
            if (nrows > 0) {
                // taps of pixel xt: raw columns xt + 1 .. xt + 7 (raw column = tile column + 4); of pixel xt + 1: xt + 2 .. xt + 8.  Three aligned dwords from
                // raw column xt & ~3 on hold them; sh = (xt & 3) + 1 = 1 | 3 bytes bring the window's first byte to the front
                const uint32_t* rp = reinterpret_cast<const uint32_t*>(raw + (row0 + 1) * kBlurRawPitch) + (xt >> 2);
                const uint32_t sh = (uint32_t)(xt & 3) + 1u;
                uint8_t* out = dstb + (size_t)(oy + row0) * a.bpitch + x;
                uint32_t P[kBlurWaveRows + 6][2], hprev[2] = {0, 0};
#pragma unroll
                for (int t = 0; t < kBlurWaveRows + 6; ++t) {
                    if (t - 6 >= nrows) break; // uniform
                    const uint32_t A = rp[t * (kBlurRawPitch / 4)], Bw = rp[t * (kBlurRawPitch / 4) + 1], C = rp[t * (kBlurRawPitch / 4) + 2];
                    const uint32_t wl = __builtin_amdgcn_alignbyte(Bw, A, sh), wh = __builtin_amdgcn_alignbyte(C, Bw, sh); // window bytes 0 .. 7
                    const uint32_t lo1 = __builtin_amdgcn_alignbyte(wh, wl, 1u), hi1 = wh >> 8;
                    const uint32_t hA = __builtin_amdgcn_udot4(wh, W1, __builtin_amdgcn_udot4(wl, W0, 0u, false), false); // (W1's top byte is 0: window byte 7 does not count)
                    const uint32_t hB = __builtin_amdgcn_udot4(hi1, W1, __builtin_amdgcn_udot4(lo1, W0, 0u, false), false);
                    P[t][0] = hprev[0] | hA << 16; hprev[0] = hA;
                    P[t][1] = hprev[1] | hB << 16; hprev[1] = hB;
                    if (t >= 6) {
                        uint32_t sum[2];
#pragma unroll
                        for (int o = 0; o < 2; ++o)
                            sum[o] = udot2(P[t][o], V3, udot2(P[t - 1][o], V2, udot2(P[t - 3][o], V1, udot2(P[t - 5][o], V0, 1u << 15))));
                        const us2_t p01 = __builtin_elementwise_min(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(sum[1], sum[0], 0x07060302u)), lim);
                        const uint16_t px = (uint16_t)__builtin_amdgcn_perm(0u, __builtin_bit_cast(uint32_t, p01), 0x0c0c0200u);
#if VSLAM_ORB_BLUR_NT
                        if (x < W) __builtin_nontemporal_store(px, reinterpret_cast<uint16_t*>(out + (size_t)(t - 6) * a.bpitch));
#else
                        if (x < W) *reinterpret_cast<uint16_t*>(out + (size_t)(t - 6) * a.bpitch) = px;
#endif
                    }
                }
            }
        } else {
            // matrix-core form (see the comment above g_blur_mfma_lane)
            const int g = lane >> 4, j = lane & 15;
            const uint2 bh2 = *reinterpret_cast<const uint2*>(g_blur_mfma_lane.bh[lane]);
            const long Bh = (long)(((unsigned long long)bh2.y << 32) | bh2.x);
            const long Bv = (long)(((unsigned long long)g_blur_mfma_lane.g1[lane] << 32) | g_blur_mfma_lane.g0[lane]);
            constexpr int kInit = 257 * (128 + 32768) + 32768; // the offsets of both byte planes (low byte - 128, h' = h - 32768) + the rounding constant
            const bl_v4i zero4 = {0, 0, 0, 0}, init4 = {kInit, kInit, kInit, kInit};
            uint8_t* stage = sc; // 64 rows x 272 bytes (kBlurRawPitch)
            static_assert(kBlurTileH * kBlurRawPitch <= kPfScBytes, "the blurred tile is staged in the score tile's storage");
            const int rows_left = H - oy; // (uniform) output rows of this tile inside the image
            if (ox + 32 * wave < W) {     // (uniform) a strip entirely beyond the image computes nothing; its staging chunks are never stored
#pragma unroll 1
                for (int ct = 0; ct < 2; ++ct) {
                    const int c0 = 32 * wave + 16 * ct; // first tile column of this 16-column tile = raw column of the window's first byte
                    if (ox + c0 >= W) break;            // (uniform)
                    // ---- row pass + byte planes: pl[t] = {low plane of row tile t, high plane}
                    uint32_t plo[5], phi[5];
#pragma unroll
                    for (int rt = 0; rt < 5; ++rt) {
                        unsigned long long araw = 0x0101010101010101ull; // (k-group 3: the centring constant)
                        if (g < 3) araw = *reinterpret_cast<const unsigned long long*>(raw + (16 * rt + j) * kBlurRawPitch + c0 + 8 * g) ^ 0x8080808080808080ull;
                        const bl_v4i d = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)araw, Bh, zero4, 0, 0, 0);
                        const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)d[1], (uint32_t)d[0], 0x05010400u); // lo0 lo1 hi0 hi1
                        const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)d[3], (uint32_t)d[2], 0x05010400u);
                        plo[rt] = __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u;
                        phi[rt] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
                    }
                    // ---- column pass, output row tile T: k-slots = row tiles T and T + 1
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        if (16 * T >= rows_left) break; // (uniform)
                        const long alo = (long)(((unsigned long long)plo[T + 1] << 32) | plo[T]), ahi = (long)(((unsigned long long)phi[T + 1] << 32) | phi[T]);
                        const bl_v4i v = __builtin_amdgcn_mfma_i32_16x16x32_i8(alo, Bv, init4, 0, 0, 0); // (the constant rides in as the accumulator's initial value)
                        const bl_v4i w = __builtin_amdgcn_mfma_i32_16x16x32_i8(ahi, Bv, zero4, 0, 0, 0);
                        uint32_t o[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) o[u] = ((uint32_t)w[u] << 8) + (uint32_t)v[u];
                        const us2_t p01 = __builtin_elementwise_min(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(o[1], o[0], 0x07060302u)), lim);
                        const us2_t p23 = __builtin_elementwise_min(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(o[3], o[2], 0x07060302u)), lim);
                        const uint32_t px = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p23), __builtin_bit_cast(uint32_t, p01), 0x06040200u);
                        // lane (n = j, g): output row 16 T + n, columns c0 + 4 g .. + 3
                        *reinterpret_cast<uint32_t*>(stage + (16 * T + j) * kBlurRawPitch + c0 + 4 * g) = px;
                    }
                }
            }
            __syncthreads();
            // ---- rows of 256 contiguous bytes out of the staged tile; the lane that read a chunk zeroes it (the FAST phase needs a zeroed score tile)
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int i = 0; i < (kBlurTileH * 16) / NT; ++i) {
                const int chunk = threadIdx.x + i * NT, r = chunk >> 4, cc = chunk & 15;
                uint4* sp = reinterpret_cast<uint4*>(stage + r * kBlurRawPitch + 16 * cc);
                const uint4 px = *sp;
                *sp = z4;
                const int x = ox + 16 * cc;
                if (r < rows_left && x < W) { // (a chunk that starts inside the row ends inside its pitch: the pitch is a multiple of 64)
                    bl_v4i* q = reinterpret_cast<bl_v4i*>(dstb + (size_t)(oy + r) * a.bpitch + x);
                    const bl_v4i pv = {(int)px.x, (int)px.y, (int)px.z, (int)px.w};
#if VSLAM_ORB_BLUR_NT
                    __builtin_nontemporal_store(pv, q);
#else
                    *q = pv;
#endif
                }
            }
            // the bytes of the score tile no staging chunk covers: the 16 padding bytes of each staged row and the tail
            if (threadIdx.x < kBlurTileH) *reinterpret_cast<uint4*>(stage + threadIdx.x * kBlurRawPitch + 256) = z4;
            for (int t = kBlurTileH * kBlurRawPitch + 16 * threadIdx.x; t < kPfScBytes; t += 16 * NT) *reinterpret_cast<uint4*>(sc + t) = z4;
            __syncthreads();
        }
    }
    OPH(34);
    // ---- (3) FAST-9/16 of level l
    if (!do_fast || (VSLAM_PYRBLUR_DBG & 4)) return; // uniform
    const int thr = a.thr;
    // score region = emitted pixels dilated by one, in TILE coordinates (column -1 .. 256, row -1 .. kBlurTileH), inclusive
    const int vx_lo = ex_lo - 1 - ox, vx_hi = ex_hi + 1 - ox, vy_lo = ey_lo - 1 - oy, vy_hi = ey_hi + 1 - oy;
    uint16_t* wq = s_wq + wave * kPfQueue;
    int qn = 0; // wave-uniform count (an SGPR)
    // position id = raw row << 9 | raw column
    auto score_batch = [&](int first, int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the queue entries were written by other lanes of this wave
        if (VSLAM_PYRBLUR_DBG & 8) { if (lane < cnt && wq[first + lane] == 0xFFFF) sc[lane] = 1; return; } // (timing aid: pre-test only)
        bool corner = false; int id = 0;
        if (lane < cnt) {
            id = wq[first + lane];
            const int R = id >> 9, c = id & 511;
            const int s = fast_score<kBlurRawPitch>(raw + (R - 3) * kBlurRawPitch + c - 3, thr);
            if (s >= thr) {
                sc[(R - 3) * kPfScPitch + c] = (uint8_t)s;
                const int tcx = c - 4, tcy = R - 4; // tile coordinates: only positions this tile emits go on to the NMS
                corner = tcx > vx_lo && tcx < vx_hi && tcy > vy_lo && tcy < vy_hi;
            }
        }
        const unsigned long long bal = __ballot(corner);
        if (bal) { // uniform
            const int n = __popcll(bal);
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_ccount, n); // one returning LDS atomic per 64 scored candidates
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + n <= kPfCornerCap) { if (corner) s_cq[base + mbcnt64(bal)] = (uint16_t)id; }
            else if (lane == 0) s_dense = 1; // (benign race: every writer stores 1)
        }
    };
    // queue the lanes of the lane mask `bal` (uniform), lane i at position id; 64 queued candidates are scored on the spot
    auto push = [&](unsigned long long bal, int id) {
        if (bal) { // uniform
            if (__builtin_amdgcn_inverse_ballot_w64(bal)) wq[qn + mbcnt64(bal)] = (uint16_t)id;
            qn += __popcll(bal);
            if (qn >= 64) { qn -= 64; score_batch(qn, 64); }
        }
    };
    // columns -1 and 256 of the score region belong to no lane's dword: one thread per position, scalar form of the same rule
    if (wave < (2 * kPfScH + 63) / 64) { // (whole waves: the pushes are wave-wide ballots) -- waves 0 .. 2 at kBlurTileH = 64
        const int t = threadIdx.x, side = t >= kPfScH ? 1 : 0, rr = t - side * kPfScH; // rr: score row
        const int tcx = side ? kBlurTileW : -1, tcy = rr - 1;
        bool cand = false;
        const int R = tcy + 4, c = tcx + 4;
        if (t < 2 * kPfScH && tcx >= vx_lo && tcx <= vx_hi && tcy >= vy_lo && tcy <= vy_hi) {
            const uint8_t* p = raw + R * kBlurRawPitch + c;
            const int v = p[0], c0 = p[3 * kBlurRawPitch], c8 = p[-3 * kBlurRawPitch], c4 = p[3], c12 = p[-3];
            const int bp = min(max(c0, c8), max(c4, c12)), dp = max(min(c0, c8), min(c4, c12));
            cand = bp - v > thr || v - dp > thr;
        }
        push(__ballot(cand), R << 9 | c);
    }
    OPH(35);
    {
        constexpr int kRows = kBlurTileH / NW;                      // centre rows per wave (+ 2 for the last wave)
        constexpr int kG = NW == 4 ? 6 : 4;                         // centre rows per group: their kG + 6 raw rows are widened once
        constexpr int kGroups = (kRows + 2 + kG - 1) / kG;
        static_assert(kRows * (NW - 1) + kGroups * kG + 6 <= kPfRawH + 8, "the pre-test reads a few rows past the staged tile (into the score tile), for centre rows it skips");
        const int cb0 = kRows * wave - 1;                                                  // first centre row (tile coordinates) of this wave
        const int c_last = wave == NW - 1 ? kBlurTileH : cb0 + kRows - 1;                  // last one it owns (the next wave starts one row above its band)
        const int o_lo = max(cb0, vy_lo), o_hi = min(c_last, vy_hi);
        // this lane's four columns 4 lane + k: inside the score region?
        unsigned long long colm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) colm[k] = __ballot(4 * lane + k >= vx_lo && 4 * lane + k <= vx_hi);
        const int thr_hi = thr << 16 | 0xFFFF; // high half of a packed pair > thr  <=>  the whole word (signed) > thr_hi
        const int idv = 4 * lane + 4;           // raw column of this lane's first pixel
#pragma unroll 1
        for (int g = 0; g < kGroups; ++g) {
            const int cb = cb0 + kG * g; // centre rows cb .. cb + kG - 1
            if (cb > o_hi || cb + kG - 1 < o_lo) continue; // uniform
            // raw rows cb + 1 .. cb + kG + 6 (centre row cb + j is raw row cb + j + 4); rows never loaded (beyond the image) or past the tile
            // only feed centre rows outside the score region, which are skipped
            const uint32_t* rawd = reinterpret_cast<const uint32_t*>(raw) + (cb + 1) * (kBlurRawPitch / 4) + lane;
            s16x2 E[kG + 6], O[kG + 6];
#pragma unroll
            for (int k = 0; k < kG + 6; ++k) {
                const uint32_t Bw = rawd[k * (kBlurRawPitch / 4) + 1];
                E[k] = as_s16x2(Bw & 0x00FF00FFu);                                   // pixels 0, 2
                O[k] = as_s16x2(__builtin_amdgcn_perm(0u, Bw, 0x0c030c01u));          // pixels 1, 3
            }
#pragma unroll
            for (int j = 0; j < kG; ++j) {
                if (cb + j < o_lo || cb + j > o_hi) continue; // uniform
                const uint32_t A = rawd[(j + 3) * (kBlurRawPitch / 4)], Bw = rawd[(j + 3) * (kBlurRawPitch / 4) + 1], C = rawd[(j + 3) * (kBlurRawPitch / 4) + 2];
                // ring positions 4 (x + 3) and 12 (x - 3) of the even and of the odd pixels
                const s16x2 c4E = as_s16x2(__builtin_amdgcn_perm(C, Bw, 0x0c050c03u)), c4O = as_s16x2(C & 0x00FF00FFu);
                const s16x2 c12E = as_s16x2(__builtin_amdgcn_perm(0u, A, 0x0c030c01u)), c12O = as_s16x2(__builtin_amdgcn_perm(Bw, A, 0x0c040c02u));
                // "two adjacent compass pixels both brighter than v + t": every adjacent pair takes one of {0, 8} and one of {4, 12}, so the
                // brightest pair's darker pixel is min(max(c0, c8), max(c4, c12)); likewise the dark side
                const s16x2 bpE = pk_min(pk_max(E[j], E[j + 6]), pk_max(c4E, c12E)), dpE = pk_max(pk_min(E[j], E[j + 6]), pk_min(c4E, c12E));
                const s16x2 bpO = pk_min(pk_max(O[j], O[j + 6]), pk_max(c4O, c12O)), dpO = pk_max(pk_min(O[j], O[j + 6]), pk_min(c4O, c12O));
                const s16x2 mE = pk_max(bpE - E[j + 3], E[j + 3] - dpE), mO = pk_max(bpO - O[j + 3], O[j + 3] - dpO); // > thr: candidate
                // one lane mask per pixel column (the queue then holds runs of neighbouring lanes of ONE row, whose ring loads fall into different LDS
                // banks); the four pushes are ONE site in a rolled loop: the score code is inlined once per row, not once per push
                const unsigned long long m0 = __ballot((int)mE.x > thr) & colm[0], m1 = __ballot((int)mO.x > thr) & colm[1];
                const unsigned long long m2 = __ballot(__builtin_bit_cast(int, mE) > thr_hi) & colm[2], m3 = __ballot(__builtin_bit_cast(int, mO) > thr_hi) & colm[3];
                const int idr = ((cb + 4 + j) << 9) + idv;
#pragma unroll 1
                for (int k = 0; k < 4; ++k) push(k == 0 ? m0 : k == 1 ? m1 : k == 2 ? m2 : m3, idr + k);
            }
        }
    }
    if (qn > 0) score_batch(0, qn);
    __syncthreads();
    OPH(39);
    // ---- 3x3 non-maximum suppression: a corner survives when its score is larger than all eight neighbours'
    uint32_t* outq = reinterpret_cast<uint32_t*>(raw); // (the pixel tile is dead)
    auto nms_emit = [&](bool is, int id) { // whole wave; `is`: this lane holds a corner at position id that the tile may emit
        bool keep = false; uint32_t rec = 0;
        if (is) {
            const int R = id >> 9, c = id & 511;
            const uint8_t* s = sc + (R - 3) * kPfScPitch + c;
            const int v = s[0];
            keep = v > s[-1] && v > s[1] && v > s[-kPfScPitch - 1] && v > s[-kPfScPitch] && v > s[-kPfScPitch + 1] && v > s[kPfScPitch - 1] &&
                   v > s[kPfScPitch] && v > s[kPfScPitch + 1];
            rec = (uint32_t)(ox + c - 4) | (uint32_t)(oy + R - 4) << 12 | (uint32_t)v << 24;
        }
        const unsigned long long bal = __ballot(keep);
        if (bal) { // uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_ocount, __popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);
            if (keep) outq[base + mbcnt64(bal)] = rec;
        }
    };
    if (!s_dense) { // uniform
        const int nc = s_ccount;
        for (int q = 64 * wave; q < nc; q += NT) nms_emit(q + lane < nc, q + lane < nc ? s_cq[q + lane] : 0);
    } else { // a tile with more corners than the list holds: every thread walks score dwords (rows 0 .. kBlurTileH - 1, columns 0 .. 255)
        for (int i = threadIdx.x; i < 64 * kBlurTileH; i += NT) { // (whole waves take every turn)
            const int r = i >> 6, d = i & 63;
            uint32_t m = reinterpret_cast<const uint32_t*>(sc + (r + 1) * kPfScPitch)[1 + d];
            if (r <= vy_lo || r >= vy_hi) m = 0;
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const int tcx = 4 * d + k;
                const bool is = ((m >> (8 * k)) & 0xFFu) != 0 && tcx > vx_lo && tcx < vx_hi;
                if (__ballot(is)) nms_emit(is, (r + 4) << 9 | (tcx + 4));
            }
        }
    }
    __syncthreads();
    OPH(40);
    const int no = s_ocount; // a strict 3x3 maximum: at most one survivor per 2x2 cell
    if (no > 0) {
        if (threadIdx.x == 0) s_obase = atomicAdd(a.corner_cnt + b * kNLevels, no);
        __syncthreads();
        uint32_t* corners = a.corners + (size_t)b * a.corner_img_stride;
        const int cap = a.corner_cap, base = s_obase;
        for (int q = threadIdx.x; q < no; q += NT) {
            if (base + q < cap) corners[base + q] = outq[q];
            else atomicOr(&a.status[b], kStCornerOverflow);
        }
    }
    OPH(41);
}

// d_corners == nullptr: pyramid + blur only (the descriptor-only entry point); else the level's FAST corners are appended to d_corners /
// d_corner_cnt (zeroed here, like d_status) exactly as launch_orb_fast would
int launch_orb_pyrblur(const OrbPlan& plan, const OrbTables& tab, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, uint8_t* d_pyr,
                       uint8_t* d_blur, int fast_thr, uint32_t* d_corners, int32_t* d_corner_cnt, int32_t* d_status, hipStream_t stream) {
    if (d_corners) {
        VS_HIP(hipMemsetAsync(d_corner_cnt, 0, sizeof(int32_t) * B * kNLevels, stream));
        VS_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t) * B, stream));
    }
    ProfScope prof__(stream, "orb_pyrblur_kernel", kNLevels);
    for (int l = 0; l < kNLevels; ++l) {
        const OrbLevel& S = plan.lv[l];
        PyrBlurArgs a;
        memset(&a, 0, sizeof(a));
        a.src_base = l == 0 ? d_imgs : d_pyr + S.pyr_off;
        a.src_img_stride = l == 0 ? img_bytes : (size_t)plan.pyr_bytes;
        a.spitch = l == 0 ? pitch : ((S.w + 63) & ~63);
        a.sw = S.w; a.sh = S.h;
        a.blur_base = d_blur + plan.blur_off[l]; a.blur_img_stride = (size_t)plan.blur_bytes; a.bpitch = (S.w + 63) & ~63;
        a.tiles_x = (S.w + kBlurTileW - 1) / kBlurTileW;
        const int tiles_y = (S.h + kBlurTileH - 1) / kBlurTileH;
        if (l + 1 < kNLevels) {
            const OrbLevel& D = plan.lv[l + 1];
            a.dst_base = d_pyr + D.pyr_off; a.dst_img_stride = (size_t)plan.pyr_bytes; a.dpitch = (D.w + 63) & ~63; a.dw = D.w; a.dh = D.h;
            a.xofs = tab.d_xofs + tab.x_off[l + 1]; a.ialpha = tab.d_ialpha + 2 * tab.x_off[l + 1];
            a.yofs = tab.d_yofs + tab.y_off[l + 1]; a.ibeta = tab.d_ibeta + 2 * tab.y_off[l + 1];
            a.tile_dx = tab.d_tile_dx + tab.tdx_off[l]; a.tile_dy = tab.d_tile_dy + tab.tdy_off[l];
        }
        if (d_corners) {
            a.corners = d_corners + S.corner_off; a.corner_img_stride = (size_t)plan.corner_total; a.corner_cap = S.corner_cap;
            a.corner_cnt = d_corner_cnt + l; a.status = d_status; a.thr = fast_thr; a.level = l;
        }
        hipLaunchKernelGGL(orb_pyrblur_kernel<kPfWaves>, dim3(a.tiles_x * tiles_y, B), dim3(64 * kPfWaves), 0, stream, a);
    }
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

// K6b orb_describe_kernel: one wave per keypoint, 256 rotated tests (4 per lane) read straight from the blurred
// pyramid (L2-resident gathers inside a 37x37 window); rotation cos/sin come precomputed per keypoint (d_cs, written
// by the ANMS kernel with one lane per keypoint, so the f64 sin/cos is not repeated by all 64 lanes of a wave).
constexpr int kDescWaves = 4;
constexpr int kDescR = 19, kDescRows = 2 * kDescR + 1, kDescPitch = 40; // |pattern| <= 13 -> rotated radius <= 18.4; rows of 39 (+1) bytes

// A wave walks keypoints j = wave id, wave id + #waves, ... of its image.  The chain "keypoint record -> 39 x 40 patch -> LDS ->
// tests" is two memory round trips per keypoint, so it is software-pipelined: while the tests of keypoint j run out of LDS the patch
// dwords of the next keypoint are already in flight to registers and the record of the one after that is being fetched; the
// lane's four test pairs stay in registers for the whole walk.
#ifndef VSLAM_DESC_BLOCKS
#define VSLAM_DESC_BLOCKS 48
#endif
constexpr int kDescBlocksPerImage = VSLAM_DESC_BLOCKS; // x 4 waves = 192 waves per image: eight keypoints per wave at N = 1500 (32 / 48 / 64 / 96 / 144 blocks: 0.373 / 0.365 / 0.373 / 0.390 / 0.409 ms per 512 images)
constexpr int kDescPatchIters = (kDescRows * (kDescPitch / 4) + 63) / 64;
__global__ __launch_bounds__(kDescWaves * 64) void orb_describe_kernel(BlurTable T, LevelTable LT, const uint8_t* __restrict__ d_imgs,
                                                                      size_t img_bytes, int pitch0, const uint8_t* __restrict__ d_pyr,
                                                                      size_t pyr_bytes, const uint8_t* __restrict__ d_blur, size_t blur_bytes,
                                                                      const vslam_keypoint* __restrict__ d_kps, const float2* __restrict__ d_cs,
                                                                      const int32_t* __restrict__ d_order, int kp_capacity,
                                                                      const int32_t* __restrict__ d_count, uint8_t* __restrict__ d_desc, int nb, int B) {
    int b, bx;
    if (!xcd_image_block(nb, B, b, bx)) return; // (uniform; no block-level barrier below)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nwaves = nb * kDescWaves;
    const int n = min(d_count[b], kp_capacity);
    // the wave visits walk positions i = wave id + k * #waves; lane k preloads the output slot of the k-th visit (d_order: the slots in
    // (octave, raster) order, so that waves running side by side fetch neighbouring patches)
    const int i0 = bx * kDescWaves + wave; // wave-uniform
    if (i0 >= n) return; // no block-level barrier below
    int slots;
    {
        const int i = i0 + lane * nwaves;
        slots = i < n ? (d_order ? min(max(d_order[(size_t)b * kp_capacity + i], 0), n - 1) : i) : -1;
    }
    const int nvisit = min((n - i0 + nwaves - 1) / nwaves, 64);
    int visit = 0;
    int j = __builtin_amdgcn_readlane(slots, 0);
    const vslam_keypoint* kps = d_kps + (size_t)b * kp_capacity;
    const float2* css = d_cs + (size_t)b * kp_capacity;
    __shared__ __attribute__((aligned(16))) uint8_t patch[kDescWaves][kDescRows * kDescPitch];
    uint8_t* pl = patch[wave];
    float px0[4], py0[4], px1[4], py1[4]; // this lane's four test pairs
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const signed char* q = &c_pattern[(4 * lane + k) * 4];
        px0[k] = (float)q[0]; py0[k] = (float)q[1]; px1[k] = (float)q[2]; py1[k] = (float)q[3];
    }
    // position of this lane's patch dwords (same for every keypoint)
    int prow[kDescPatchIters], pcol[kDescPatchIters];
#pragma unroll
    for (int it = 0; it < kDescPatchIters; ++it) {
        const int i = lane + 64 * it;
        prow[it] = i / (kDescPitch / 4); pcol[it] = 4 * (i - prow[it] * (kDescPitch / 4));
    }
    struct Geo { int cx, cy, l, W, H, bpitch; bool inside; const uint8_t* blur; };
    auto geometry = [&](const vslam_keypoint& kp) -> Geo {
        Geo g;
        // (every lane holds the same keypoint record: pinning level and centre in scalar registers makes the patch origin a scalar
        // base, so the patch loads are scalar base + 32-bit lane offset instead of seven 64-bit address computations per keypoint)
        g.l = __builtin_amdgcn_readfirstlane(min(max(kp.octave, 0), kNLevels - 1));
        g.W = T.w[g.l]; g.H = T.h[g.l]; g.bpitch = T.pitch[g.l];
        g.blur = d_blur + (size_t)b * blur_bytes + T.blur_off[g.l];
        const float inv_scale = __fdiv_rn(1.f, LT.scale[g.l]);
        g.cx = __builtin_amdgcn_readfirstlane(__float2int_rn(__fmul_rn(kp.x, inv_scale)));
        g.cy = __builtin_amdgcn_readfirstlane(__float2int_rn(__fmul_rn(kp.y, inv_scale)));
        // Fast path (every keypoint the detector emits: edgeThreshold 31 > the rotated pattern radius 19): the wave stages the
        // 39 x 40 byte neighbourhood of the blurred level in LDS with coalesced unaligned-dword row loads and gathers the 512
        // samples from there -- scattered byte loads from global memory are bound by the texture-address line rate.
        g.inside = g.cx - kDescR >= 0 && g.cx + kDescR + 1 < g.W && g.cy - kDescR >= 0 && g.cy + kDescR < g.H; // uniform per wave
        return g;
    };
    auto patch_fetch = [&](const Geo& g, uint32_t (&pv)[kDescPatchIters]) {
        if (!g.inside) return;
        const uint8_t* org = g.blur + (size_t)(g.cy - kDescR) * g.bpitch + (g.cx - kDescR); // wave-uniform
#pragma unroll
        for (int it = 0; it < kDescPatchIters; ++it) // (24-bit multiply: full rate; rows < 39, pitch < 4160)
            if (prow[it] < kDescRows) __builtin_memcpy(&pv[it], org + (__umul24((uint32_t)prow[it], (uint32_t)g.bpitch) + (uint32_t)pcol[it]), 4);
    };
    vslam_keypoint kp = kps[j];
    float2 cs = css[j];
    Geo g = geometry(kp);
    uint32_t pv[kDescPatchIters];
    patch_fetch(g, pv);
    auto slot_of = [&](int v) -> int { return v < nvisit ? __builtin_amdgcn_readlane(slots, v) : -1; }; // v is wave-uniform
    int j1 = slot_of(1);                           // next keypoint: record in flight
    vslam_keypoint kp1 = kps[max(j1, 0)];
    float2 cs1 = css[max(j1, 0)];
    for (;;) {
        // patch of keypoint j: registers -> LDS
        if (g.inside) {
#pragma unroll
            for (int it = 0; it < kDescPatchIters; ++it)
                if (prow[it] < kDescRows) *reinterpret_cast<uint32_t*>(pl + prow[it] * kDescPitch + pcol[it]) = pv[it];
        }
        // keypoint j1: its patch goes in flight now; keypoint j2: its record
        const Geo g1 = geometry(kp1);
        uint32_t pv1[kDescPatchIters];
        if (j1 >= 0) patch_fetch(g1, pv1);
        const int j2 = slot_of(visit + 2);
        const vslam_keypoint kp2 = kps[max(j2, 0)];
        const float2 cs2 = css[max(j2, 0)];
        __builtin_amdgcn_wave_barrier(); // LDS traffic of one wave is ordered; this only pins the compiler
        const float ca = cs.x, sa = cs.y;
        int nib = 0;
        if (g.inside) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ix0 = __float2int_rn(__fsub_rn(__fmul_rn(px0[k], ca), __fmul_rn(py0[k], sa)));
                const int iy0 = __float2int_rn(__fadd_rn(__fmul_rn(px0[k], sa), __fmul_rn(py0[k], ca)));
                const int ix1 = __float2int_rn(__fsub_rn(__fmul_rn(px1[k], ca), __fmul_rn(py1[k], sa)));
                const int iy1 = __float2int_rn(__fadd_rn(__fmul_rn(px1[k], sa), __fmul_rn(py1[k], ca)));
                const int t0 = pl[(iy0 + kDescR) * kDescPitch + ix0 + kDescR], t1 = pl[(iy1 + kDescR) * kDescPitch + ix1 + kDescR];
                nib |= (t0 < t1) << k;
            }
        } else {
            const uint8_t* rawp = g.l == 0 ? d_imgs + (size_t)b * img_bytes : d_pyr + (size_t)b * pyr_bytes + T.pyr_off[g.l];
            const int rpitch = g.l == 0 ? pitch0 : T.pitch[g.l];
            auto sample = [&](int ix, int iy) -> int {
                const int x = g.cx + ix, y = g.cy + iy;
                if (x >= 0 && x < g.W && y >= 0 && y < g.H) return g.blur[(size_t)y * g.bpitch + x];
                // outside the level the reference reads the UNBLURRED reflect-101 border of its pyramid buffer
                return rawp[(size_t)reflect101(y, g.H) * rpitch + reflect101(x, g.W)];
            };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ix0 = __float2int_rn(__fsub_rn(__fmul_rn(px0[k], ca), __fmul_rn(py0[k], sa)));
                const int iy0 = __float2int_rn(__fadd_rn(__fmul_rn(px0[k], sa), __fmul_rn(py0[k], ca)));
                const int ix1 = __float2int_rn(__fsub_rn(__fmul_rn(px1[k], ca), __fmul_rn(py1[k], sa)));
                const int iy1 = __float2int_rn(__fadd_rn(__fmul_rn(px1[k], sa), __fmul_rn(py1[k], ca)));
                nib |= (sample(ix0, iy0) < sample(ix1, iy1)) << k;
            }
        }
        const int hi = __shfl_down(nib, 1);
        if ((lane & 1) == 0) d_desc[((size_t)b * kp_capacity + j) * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
        if (j1 < 0) break;
        __builtin_amdgcn_wave_barrier(); // the tests above have read the patch before the next one overwrites it
        ++visit;
        j = j1; j1 = j2; g = g1; cs = cs1; kp1 = kp2; cs1 = cs2;
#pragma unroll
        for (int it = 0; it < kDescPatchIters; ++it) pv[it] = pv1[it];
    }
}

int launch_orb_describe(const OrbPlan& plan, const uint8_t* d_imgs, size_t img_bytes, int pitch, int B, const uint8_t* d_pyr,
                        const uint8_t* d_blur, const vslam_keypoint* d_kps, const float2* d_cs, const int32_t* d_order, int kp_capacity, const int32_t* d_count,
                        uint8_t* d_desc, hipStream_t stream) {
    BlurTable T;
    fill_blur_table(plan, &T);
    LevelTable LT;
    fill_level_table(plan, &LT);
    // a fixed number of waves per image walks the keypoints (waves beyond d_count[b] exit immediately)
    const int max_kp = min(kp_capacity, kMaxRows);
    const int blocks = min(kDescBlocksPerImage, (max_kp + kDescWaves - 1) / kDescWaves);
    ProfScope prof__(stream, "orb_describe_kernel");
    hipLaunchKernelGGL(orb_describe_kernel, dim3(blocks * ((B + 7) / 8 * 8)), dim3(kDescWaves * 64), 0, stream, T, LT, d_imgs,
                       img_bytes, pitch, d_pyr, (size_t)plan.pyr_bytes, d_blur, (size_t)plan.blur_bytes, d_kps, d_cs, d_order, kp_capacity, d_count, d_desc, blocks, B);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

} // namespace vslam
