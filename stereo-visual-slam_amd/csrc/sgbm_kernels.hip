// sgbm_kernels.hip -- stereo depth by semi-global block matching ("next #1" of 8f).
//
// Replaces cv::StereoSGBM::create(0, 96, 9, 8*9*9, 32*9*9, 1, 63, 10, 100, 32)->compute + convertTo(CV_32F, 1/16) inside
// VO::disparity_map (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:159-174).  The algorithm is OpenCV
// 3.2's single-pass MODE_SGBM (calcPixelCostBT + computeDisparitySGBM + medianBlur 3x3 + filterSpeckles), including its
// border behaviours (see oracle/sgbm.c for the list); all arithmetic is 8/16/32-bit integer, so the result is bit-exact
// against the CPU oracle.
//
// gfx950 mapping (batched over B stereo pairs; every stage integer; the cost volume is 82.7 MB per pair).  Which kernels run depends on the
// batch (launch_sgbm; thresholds overridable per call through VSLAM_SGBM_FUSE_MIN / VSLAM_SGBM_FWD_MIN / VSLAM_SGBM_FW_ROWS):
//   sgbm_prefilter_kernel     x-Sobel clipped to [0,126] + raw rows, half-pixel min/max envelopes of both views        elementwise
//   >= 8 pairs:  sgbm_down_kernel<WITH_PATH>   Birchfield-Tomasi cost + 9x9 box sums (+ the vertical path) in one top-down sweep per 24-column tile;
//                                              operand strips staged through LDS once per row, 9-tap sums as v_dot4
//   <  8 pairs:  sgbm_hsum_kernel, sgbm_vsum_kernel, sgbm_path_kernel<0,1>   the same three steps as massively parallel kernels
//   >= 16 pairs: sgbm_forward_kernel<ROWS>     paths (1,0), (1,1), (0,1), (-1,1) as one wavefront sweep over t = x + 2y (C read once, S1 written
//                                              once); slabs of 32 / 64 image rows chained through a boundary buffer
//   <  16 pairs: sgbm_path_kernel<dx,dy>       one kernel per path: a 16-lane DPP row walks one image line (6 disparities per lane)
//   sgbm_path_kernel<-1,0> MODE 4 / 3 + sgbm_wta_kernel   last path with winner-take-all + uniqueness folded in (S never stored) / separate
//   sgbm_lrcheck_kernel       sub-pixel parabola, disp2 bookkeeping, left-right check (workgroup per row)
//   sgbm_median3_kernel       3x3 median, replicated borders
//   sgbm_ccl_{rows,union,count,apply}_kernel   speckle filter as connected-component labelling (atomic union-find) + size threshold, /16 -> f32
#include "vslam_internal.h"

#include <stdlib.h>

namespace vslam {

struct SgbmDims {
    int w, h, D, minX1, width1, P1, P2, SW2, SH2, uniq, disp12, ftzero, pitch;
    size_t img_bytes;
};

constexpr int kSgbmMaxCost = 32767;
constexpr int kTOffset = 8192; // L1+L2+L3 lies in [-7776, 46k]: stored as u16 with this offset

__device__ inline int sat16_dev(int v) { return min(max(v, -32768), 32767); }

// ------------------------------------------------------------------------------------------- prefilter
// pre[(b*2+side)][y][6][w]: planes {val, lo, hi} of channel 0 (x-Sobel clipped to [0, 2*ftzero]) and channel 1 (raw intensity;
// both channels have their two end pixels replaced by ftzero, like the reference row buffers).  lo / hi are the min / max
// of the pixel and its two half-pixel interpolants: the Birchfield-Tomasi operands of calcPixelCostBT, shared by both views.
__device__ inline void prefilter_px(const uint8_t* row, int pitch, int w, int h, int x, int y, int ftzero, int& g, int& r) {
    g = ftzero; r = ftzero;
    if (x >= 1 && x < w - 1) {
        const int n = y > 0 ? -pitch : 0, s = y < h - 1 ? pitch : 0;
        const int v = (row[x + 1] - row[x - 1]) * 2 + row[x + n + 1] - row[x + n - 1] + row[x + s + 1] - row[x + s - 1];
        g = min(max(v, -ftzero), ftzero) + ftzero;
        r = row[x];
    }
}
__global__ __launch_bounds__(256) void sgbm_prefilter_kernel(SgbmDims dm, const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                            uint8_t* __restrict__ pre) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, ii = blockIdx.z;
    if (x >= dm.w) return;
    const int b = ii >> 1, side = ii & 1;
    const uint8_t* row = (side ? right : left) + (size_t)b * dm.img_bytes + (size_t)y * dm.pitch;
    uint8_t* out = pre + (((size_t)ii * dm.h + y) * 6) * dm.w;
    int g, r, gl = 0, rl = 0, gr = 0, rr = 0;
    prefilter_px(row, dm.pitch, dm.w, dm.h, x, y, dm.ftzero, g, r);
    if (x > 0) prefilter_px(row, dm.pitch, dm.w, dm.h, x - 1, y, dm.ftzero, gl, rl);
    if (x < dm.w - 1) prefilter_px(row, dm.pitch, dm.w, dm.h, x + 1, y, dm.ftzero, gr, rr);
    const int ga = x > 0 ? (g + gl) >> 1 : g, gb = x < dm.w - 1 ? (g + gr) >> 1 : g;
    const int ra = x > 0 ? (r + rl) >> 1 : r, rb = x < dm.w - 1 ? (r + rr) >> 1 : r;
    out[x] = (uint8_t)g; out[dm.w + x] = (uint8_t)min(min(ga, gb), g); out[2 * dm.w + x] = (uint8_t)max(max(ga, gb), g);
    out[3 * dm.w + x] = (uint8_t)r; out[4 * dm.w + x] = (uint8_t)min(min(ra, rb), r); out[5 * dm.w + x] = (uint8_t)max(max(ra, rb), r);
}

// ------------------------------------------------------------------------------------------- block cost, horizontal part
// One workgroup = one image row x kHsSeg pixels.  Phase 1 evaluates the Birchfield-Tomasi pixel cost (both channels)
// once per (pixel, disparity) of the segment plus a 4-pixel halo (columns clamped to the volume, like the reference's
// box sums) into an LDS tile, four disparities per work item through unaligned dword loads of the right-view planes.
// Phase 2 forms the 9-tap horizontal sums, 8 disparities per work item (ds_read_b64, byte lanes accumulated as two
// packed u16 pairs), and stores them as int16.
constexpr int kHsSeg = 64, kHsBlock = 256;
__device__ inline uint32_t ld_u32_unaligned(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

__global__ __launch_bounds__(kHsBlock) void sgbm_hsum_kernel(SgbmDims dm, const uint8_t* __restrict__ pre, int16_t* __restrict__ hsum) {
    const int b = blockIdx.z, y = blockIdx.y, j0 = blockIdx.x * kHsSeg;
    __shared__ alignas(16) uint8_t tile[(kHsSeg + 8) * 96];
    const uint8_t* L = pre + (((size_t)(2 * b) * dm.h + y) * 6) * dm.w;
    const uint8_t* R = pre + (((size_t)(2 * b + 1) * dm.h + y) * 6) * dm.w;
    const int W1 = dm.width1, w = dm.w;
    for (int it = threadIdx.x; it < (kHsSeg + 8) * 24; it += kHsBlock) {
        const int t = it / 24, q = it - t * 24, d = 4 * q;
        const int jj = min(max(j0 - 4 + t, 0), W1 - 1), x = dm.minX1 + jj;
        uint32_t out = 0;
        uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint8_t* lp = L + 3 * c * w; const uint8_t* rp = R + 3 * c * w;
            const int u = lp[x], u0 = lp[w + x], u1 = lp[2 * w + x];
            const uint32_t vv = ld_u32_unaligned(rp + x - d - 3), v0v = ld_u32_unaligned(rp + w + x - d - 3), v1v = ld_u32_unaligned(rp + 2 * w + x - d - 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) { // disparity d + i <-> byte 3 - i
                const int sh = 8 * (3 - i);
                const int v = (vv >> sh) & 255, v0 = (v0v >> sh) & 255, v1 = (v1v >> sh) & 255;
                const int c0 = max(max(0, u - v1), v0 - u), c1 = max(max(0, v - u1), u0 - v);
                acc[i] += (uint32_t)(min(c0, c1) >> (c == 0 ? 0 : 2));
            }
        }
        out = acc[0] | (acc[1] << 8) | (acc[2] << 16) | (acc[3] << 24); // each <= 126 + 63
        *(uint32_t*)(tile + t * 96 + d) = out;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < kHsSeg * 12; it += kHsBlock) {
        const int t = it / 12, g = it - t * 12;
        if (j0 + t >= W1) continue;
        uint32_t e0 = 0, o0 = 0, e1 = 0, o1 = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const uint2 v = *(const uint2*)(tile + (t + i) * 96 + 8 * g);
            e0 += v.x & 0x00FF00FFu; o0 += (v.x >> 8) & 0x00FF00FFu;
            e1 += v.y & 0x00FF00FFu; o1 += (v.y >> 8) & 0x00FF00FFu;
        }
        uint4 o;
        o.x = (e0 & 0xFFFFu) | (o0 << 16); o.y = (e0 >> 16) | (o0 & 0xFFFF0000u);
        o.z = (e1 & 0xFFFFu) | (o1 << 16); o.w = (e1 >> 16) | (o1 & 0xFFFF0000u);
        *(uint4*)(hsum + (((size_t)b * dm.h + y) * W1 + j0 + t) * 96 + 8 * g) = o;
    }
}

// ------------------------------------------------------------------------------------------- block cost, vertical part
// C(y) = sum of hsum over rows y-4..y+4 (rows above the image replicate row 0); the reference stops sliding SH2 rows
// above the bottom (the last rows repeat C(h-1-SH2)) and never updates column 0 after the first row.  Each work item owns
// (column, 8 disparities) and slides down a chunk of rows with packed int16 adds: 2 loads + 1 store per row.
typedef short short2v __attribute__((ext_vector_type(2)));
struct alignas(16) S8 { short2v a, b, c, d; };
__device__ inline S8 s8_add(S8 p, S8 q) { return S8{p.a + q.a, p.b + q.b, p.c + q.c, p.d + q.d}; }
__device__ inline S8 s8_sub(S8 p, S8 q) { return S8{p.a - q.a, p.b - q.b, p.c - q.c, p.d - q.d}; }
constexpr int kVsChunk = 47;

__global__ __launch_bounds__(256) void sgbm_vsum_kernel(SgbmDims dm, const int16_t* __restrict__ hsum, int16_t* __restrict__ C) {
    const int b = blockIdx.z, ya = blockIdx.y * kVsChunk, yb = min(ya + kVsChunk, dm.h);
    const int it = blockIdx.x * 256 + threadIdx.x; // (j, g): 8 disparities
    if (it >= dm.width1 * 12) return;
    const int j = it / 12;
    const size_t rs = (size_t)dm.width1 * 96; // row stride (elements)
    const int16_t* hp = hsum + (size_t)b * dm.h * rs + (size_t)it * 8;
    int16_t* cp = C + (size_t)b * dm.h * rs + (size_t)it * 8;
    const int ylast = j == 0 ? 0 : dm.h - 1 - dm.SH2; // last row whose window is evaluated
    const int yy = min(ya, ylast);
    S8 acc = *(const S8*)(hp + (size_t)max(yy - dm.SH2, 0) * rs);
    for (int k = yy - dm.SH2 + 1; k <= yy + dm.SH2; ++k) acc = s8_add(acc, *(const S8*)(hp + (size_t)max(k, 0) * rs));
#pragma unroll 4
    for (int y = ya; y < yb; ++y) {
        *(S8*)(cp + (size_t)y * rs) = acc;
        if (y + 1 <= ylast) acc = s8_sub(s8_add(acc, *(const S8*)(hp + (size_t)(y + 1 + dm.SH2) * rs)), *(const S8*)(hp + (size_t)max(y - dm.SH2, 0) * rs));
    }
}

// ------------------------------------------------------------------------------------------- path aggregation
// The five SGM paths of MODE_SGBM's single pass are independent 1-D recurrences along image lines:
//   L_r(p, d) = C(p, d) + min(L_r(p-r, d), L_r(p-r, d-1) + P1, L_r(p-r, d+1) + P1, min_k L_r(p-r, k) + P2) - (min_k L_r(p-r, k) + P2)
// with L_r = 0 (and its minimum 0) outside the cost volume.  One 16-lane DPP row owns one line and walks it with the
// previous L in registers (6 disparities per lane, D = 96): neighbour disparities come from row_shr/row_shl, the
// per-pixel minimum from a 4-step quad_perm/row_mirror reduction -- no LDS, no barriers, loads prefetched kPathPF steps ahead.
// Accumulation keeps the reference's saturation order S = sat16(sat16(L0 + L1 + L2 + L3) + L4):
//   MODE 0: T  = L + kTOffset           (first of the three paths from the previous row; u16, wrap-safe)
//   MODE 1: T += L                      (the other two)
//   MODE 2: S1 = sat16(L + T - kTOffset)     (left -> right, in place)
//   MODE 3: S  = sat16(S1 + L)               (right -> left, in place; winner-take-all in its own kernel: small batches,
//                                              where the extra instructions on the sequential chain cost more than the traffic)
//   MODE 4: same, but S is consumed on the spot by the winner-take-all and never stored (large batches, HBM-bound)
constexpr int kSent = 30000; // out-of-range disparity neighbour: any value with kSent + P1 > max(delta) behaves like SHRT_MAX
struct alignas(4) U3 { uint32_t a, b, c; };

template <int CTRL>
__device__ inline int dpp_mov(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false); }
__device__ inline int row16_min(int v) {
    v = min(v, dpp_mov<0xB1>(v, v));  // quad_perm [1,0,3,2]
    v = min(v, dpp_mov<0x4E>(v, v));  // quad_perm [2,3,0,1]
    v = min(v, dpp_mov<0x141>(v, v)); // row_half_mirror
    v = min(v, dpp_mov<0x140>(v, v)); // row_mirror
    return v;
}
__device__ inline int row16_max(int v) {
    v = max(v, dpp_mov<0xB1>(v, v));
    v = max(v, dpp_mov<0x4E>(v, v));
    v = max(v, dpp_mov<0x141>(v, v));
    v = max(v, dpp_mov<0x140>(v, v));
    return v;
}
__device__ inline int lo16s(uint32_t v) { return (int)(int16_t)(v & 0xFFFFu); }
__device__ inline int hi16s(uint32_t v) { return (int)v >> 16; }
__device__ inline uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }

// winner-take-all for one pixel held by a DPP row (6 disparities per lane): first minimum and the uniqueness test.
// The lane holding the winner writes rec = {(minS + 32768) << 8 | d  (or -1 when not unique), S[d-1] | S[d] << 16, S[d+1]};
// the parabola sub-pixel step (an integer division) is left to the left-right kernel, off this sequential chain.
__device__ inline void wta_row16(const SgbmDims& dm, int s0, int s1, int s2, int s3, int s4, int s5, int r, size_t out_index,
                                 int4* __restrict__ rec, bool active) {
    const int d0 = 6 * r;
    int best = ((s0 + 32768) << 8) | d0;
    best = min(best, ((s1 + 32768) << 8) | (d0 + 1)); best = min(best, ((s2 + 32768) << 8) | (d0 + 2));
    best = min(best, ((s3 + 32768) << 8) | (d0 + 3)); best = min(best, ((s4 + 32768) << 8) | (d0 + 4));
    best = min(best, ((s5 + 32768) << 8) | (d0 + 5));
    best = row16_min(best);
    const int minS = (best >> 8) - 32768, bd = best & 0xFF;
    const int u = 100 - dm.uniq, lim = __mul24(minS, 100); // |S| < 2^15: 24-bit multiplies are exact (and full rate)
    int bad = ((__mul24(s0, u) < lim && abs(bd - d0) > 1) || (__mul24(s1, u) < lim && abs(bd - d0 - 1) > 1) ||
               (__mul24(s2, u) < lim && abs(bd - d0 - 2) > 1) || (__mul24(s3, u) < lim && abs(bd - d0 - 3) > 1) ||
               (__mul24(s4, u) < lim && abs(bd - d0 - 4) > 1) || (__mul24(s5, u) < lim && abs(bd - d0 - 5) > 1)) ? 1 : 0;
    bad = row16_max(bad);
    const int sm_in = dpp_mov<0x111>(0, s5), sp_in = dpp_mov<0x101>(0, s0);
    const int j = bd - d0;
    if (active && j >= 0 && j < 6) { // the lane holding the winner
        const int a[8] = {sm_in, s0, s1, s2, s3, s4, s5, sp_in};
        int sm = a[0], sc = a[1], sp = a[2];
#pragma unroll
        for (int q = 1; q < 6; ++q) if (j == q) { sm = a[q]; sc = a[q + 1]; sp = a[q + 2]; }
        rec[out_index] = make_int4(bad ? -1 : best, (int)pack16(sm, sc), sp, 0);
    }
}

#ifndef VSLAM_SGBM_PATH_BLOCK
#define VSLAM_SGBM_PATH_BLOCK 64
#endif
constexpr int kPathBlock = VSLAM_SGBM_PATH_BLOCK, kPathLines = kPathBlock / 16; // lines per workgroup (adjacent lines: one contiguous run of the volume per step)
// Non-temporal loads / stores on the streamed volumes (every byte is touched once per kernel): the two diagonal paths gain 7-12 %
// (1.90 -> 1.67, 1.74 -> 1.61 ms per 32 pairs), the two horizontal ones lose 3 % -- so NT is a property of the direction.
#ifndef VSLAM_SGBM_LAST_NT
#define VSLAM_SGBM_LAST_NT 0
#endif
#ifndef VSLAM_SGBM_H_NTST
#define VSLAM_SGBM_H_NTST 1 // non-temporal STORES (not loads) on the horizontal paths: 1.69 -> 1.65 ms for the left-to-right path
#endif
template <bool NT>
__device__ inline U3 ld_u3(const void* p) {
    if constexpr (NT) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
        return U3{__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2)}; // merged into one dwordx3 nt
    } else return *reinterpret_cast<const U3*>(p);
}
template <bool NT>
__device__ inline void st_u3(void* p, U3 v) {
    if constexpr (NT) {
        uint32_t* q = reinterpret_cast<uint32_t*>(p);
        __builtin_nontemporal_store(v.a, q); __builtin_nontemporal_store(v.b, q + 1); __builtin_nontemporal_store(v.c, q + 2);
    } else *reinterpret_cast<U3*>(p) = v;
}
template <int DX, int DY, int MODE, int kPathPF>
__global__ __launch_bounds__(kPathBlock) void sgbm_path_kernel(SgbmDims dm, const int16_t* __restrict__ C, uint16_t* T, int nlines, int4* __restrict__ rec) {
    const int b = blockIdx.y;
    constexpr bool kNT = (DX != 0 && DY != 0) || (MODE == 4 && VSLAM_SGBM_LAST_NT != 0); // loads (and stores) of the diagonal paths; tuning macro for the last path
    constexpr bool kNTst = kNT || (VSLAM_SGBM_H_NTST != 0); // stores of the horizontal paths (tuning macro)
    const int line = blockIdx.x * kPathLines + (threadIdx.x >> 4), r = threadIdx.x & 15;
    if (line >= nlines) return; // whole DPP row leaves
    const int W1 = dm.width1, h = dm.h;
    int x0, y0, len;
    if (DY == 0) { y0 = line; x0 = DX > 0 ? 0 : W1 - 1; len = W1; }
    else if (DX == 0) { x0 = line; y0 = 0; len = h; }
    else {
        if (line < W1) { x0 = line; y0 = 0; } else { x0 = DX > 0 ? 0 : W1 - 1; y0 = line - W1 + 1; }
        len = min(DX > 0 ? W1 - x0 : x0 + 1, h - y0);
    }
    const ptrdiff_t step = ((ptrdiff_t)DY * W1 + DX) * 96;
    const size_t first = (((size_t)b * h + y0) * W1 + x0) * 96 + 6 * r;
    const int16_t* cp = C + first;
    uint16_t* tp = T + first;
    // prefetch queue: loads are unconditional (indices clamped to the line) so that no wait is forced at the load site
    U3 cq[kPathPF], tq[kPathPF];
#pragma unroll
    for (int k = 0; k < kPathPF; ++k) {
        const ptrdiff_t o = (ptrdiff_t)min(k, len - 1) * step;
        cq[k] = ld_u3<kNT>(cp + o);
        tq[k] = MODE != 0 ? ld_u3<kNT>(tp + o) : U3{0, 0, 0};
    }
    int l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, l5 = 0, minPrev = 0;
    const int P1 = dm.P1, P2 = dm.P2;
    for (int s = 0; s < len; s += kPathPF) {
#pragma unroll
        for (int k = 0; k < kPathPF; ++k) {
            {
                const U3 c = cq[k], t = tq[k];
                {
                    const ptrdiff_t o = (ptrdiff_t)min(s + k + kPathPF, len - 1) * step;
                    cq[k] = ld_u3<kNT>(cp + o);
                    if (MODE != 0) tq[k] = ld_u3<kNT>(tp + o);
                }
                const int lm = dpp_mov<0x111>(kSent, l5); // row_shr:1 -- lane r-1's last disparity (d0 - 1)
                const int lp = dpp_mov<0x101>(kSent, l0); // row_shl:1 -- lane r+1's first disparity (d5 + 1)
                const int delta = minPrev + P2;
                const int n0 = lo16s(c.a) - delta + min(min(l0, min(lm, l1) + P1), delta);
                const int n1 = hi16s(c.a) - delta + min(min(l1, min(l0, l2) + P1), delta);
                const int n2 = lo16s(c.b) - delta + min(min(l2, min(l1, l3) + P1), delta);
                const int n3 = hi16s(c.b) - delta + min(min(l3, min(l2, l4) + P1), delta);
                const int n4 = lo16s(c.c) - delta + min(min(l4, min(l3, l5) + P1), delta);
                const int n5 = hi16s(c.c) - delta + min(min(l5, min(l4, lp) + P1), delta);
                l0 = n0; l1 = n1; l2 = n2; l3 = n3; l4 = n4; l5 = n5;
                minPrev = row16_min(min(min(min(n0, n1), min(n2, n3)), min(n4, n5)));
                U3 o;
                if (MODE == 0) {
                    o.a = pack16(n0 + kTOffset, n1 + kTOffset); o.b = pack16(n2 + kTOffset, n3 + kTOffset); o.c = pack16(n4 + kTOffset, n5 + kTOffset);
                } else if (MODE == 1) { // u16 wrap-around add: the true sum (+offset) always fits
                    o.a = pack16((int)(t.a & 0xFFFFu) + n0, (int)(t.a >> 16) + n1);
                    o.b = pack16((int)(t.b & 0xFFFFu) + n2, (int)(t.b >> 16) + n3);
                    o.c = pack16((int)(t.c & 0xFFFFu) + n4, (int)(t.c >> 16) + n5);
                } else if (MODE == 2) {
                    o.a = pack16(sat16_dev((int)(t.a & 0xFFFFu) - kTOffset + n0), sat16_dev((int)(t.a >> 16) - kTOffset + n1));
                    o.b = pack16(sat16_dev((int)(t.b & 0xFFFFu) - kTOffset + n2), sat16_dev((int)(t.b >> 16) - kTOffset + n3));
                    o.c = pack16(sat16_dev((int)(t.c & 0xFFFFu) - kTOffset + n4), sat16_dev((int)(t.c >> 16) - kTOffset + n5));
                } else {
                    const int f0 = sat16_dev(lo16s(t.a) + n0), f1 = sat16_dev(hi16s(t.a) + n1), f2 = sat16_dev(lo16s(t.b) + n2);
                    const int f3 = sat16_dev(hi16s(t.b) + n3), f4 = sat16_dev(lo16s(t.c) + n4), f5 = sat16_dev(hi16s(t.c) + n5);
                    if (MODE == 3) { o.a = pack16(f0, f1); o.b = pack16(f2, f3); o.c = pack16(f4, f5); }
                    else // last path: S is complete -- pick the winner here instead of storing it
                        wta_row16(dm, f0, f1, f2, f3, f4, f5, r, ((size_t)b * h + y0) * dm.w + dm.minX1 + x0 + (s + k) * DX, rec, s + k < len);
                }
                if (MODE != 4 && s + k < len) st_u3<kNTst>(tp + (ptrdiff_t)(s + k) * step, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- fused forward pass (4 of the 5 paths)
// sgbm_forward_kernel: the four paths whose predecessor lies earlier in raster order -- (1,0), (1,1), (0,1), (-1,1) -- in ONE sweep that
// reads C once and writes S1 = sat16(L0 + L1 + L2 + L3) once.  The per-path kernels are HBM-bound (4.9 TB/s of a 5.8 TB/s copy ceiling)
// and each of them re-reads C and reads + writes the running sum: 26.5 GB per 32 pairs for these four paths against 5.3 GB here.
//
// Parallelism is the skewed wavefront t = x + 2 y: pixel (x, y) needs (x-1, y) [step t-1], (x+1, y-1) [t-1], (x, y-1) [t-2] and
// (x-1, y-1) [t-3], so all pixels of one t are independent.  A workgroup owns a slab of 64 image rows, one 16-lane DPP row per image
// row (6 disparities per lane, as in the line kernels): the (1,0) path lives in the row's own registers; after every step a row leaves
// its three downward L vectors (36 B per lane) in an LDS slot, the row below picks them up one step later and delays (0,1) by one and
// (1,1) by two more steps in registers.  One barrier per step, two LDS slots per row.
//
// Slabs of one pair are chained through memory: the last row of slab k writes its downward vectors per pixel into a boundary buffer and
// publishes its progress every 32 pixels (release fence + flag); slab k+1 stages the records 32 at a time into LDS, one chunk ahead,
// with device-scope loads, after its first lane has seen the flag pass the chunk.  A workgroup only ever waits for ONE workgroup with
// a lower LOGICAL index, and the logical index is a ticket drawn from a per-launch atomic counter when the workgroup starts running (not
// blockIdx: HIP does not promise dispatch order), so every lower ticket belongs to a workgroup that is resident or finished -- the lowest
// unfinished one never waits: no co-residency requirement, no deadlock, whatever shares the device.  A spin limit remains as a backstop
// against a wedged predecessor: it sets the launch's error word (the API reports VSLAM_ERR_HIP), releases the successors and returns.
#ifndef VSLAM_SGBM_FW_CHUNK
#define VSLAM_SGBM_FW_CHUNK 32
#endif
#ifndef VSLAM_SGBM_FW_PF
#define VSLAM_SGBM_FW_PF 4
#endif
constexpr int kFwChunk = VSLAM_SGBM_FW_CHUNK, kFwPF = VSLAM_SGBM_FW_PF;
constexpr int kFwRecDw = 16 * 9;                       // boundary record of one pixel: 16 lanes x 3 paths x 3 dwords
template <int kFwRows>
struct FwShared {
    uint32_t slot[2][kFwRows][16 * 9]; // [step parity][row][lane][path (-1,1), (0,1), (1,1)][3 dwords]
    uint32_t bnd[2][kFwChunk * kFwRecDw];
    int ctl[2]; // [0] this workgroup's ticket (logical index), [1] abort flag of the spin-limit backstop
};
#ifndef VSLAM_SGBM_FW_ACQ
#define VSLAM_SGBM_FW_ACQ 2
#endif
#ifndef VSLAM_SGBM_FW_SPIN_LIMIT
#define VSLAM_SGBM_FW_SPIN_LIMIT (1 << 26) // polls of ~0.5 us: ~30 s
#endif
// The recurrence in packed 16-bit arithmetic (two disparities per register: L in [-P2, Cmax], delta <= Cmax + P2, kSent + P1 < 2^15 --
// nothing leaves int16).  The vectors arrive packed from the volume and from the row above and leave packed: no unpacking at all.
struct FwVec { short2v p[3]; }; // disparities (6r, 6r+1), (6r+2, 6r+3), (6r+4, 6r+5)
__device__ inline short2v fw_s2(uint32_t v) { return __builtin_bit_cast(short2v, v); }
__device__ inline uint32_t fw_u(short2v v) { return __builtin_bit_cast(uint32_t, v); }
__device__ inline FwVec fw_vec(uint32_t a, uint32_t b, uint32_t c) { return FwVec{{fw_s2(a), fw_s2(b), fw_s2(c)}}; }
template <int CTRL>
__device__ inline int fw_min_dpp(int v) { // v = min(v, v of the DPP-selected lane) (every lane has a source: quad_perm / mirrors)
    return min(v, __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true));
}
// nb_lo / nb_hi: the registers the neighbour lanes' edge disparities are shifted into.  Lane 0 (15) has no source for row_shr (row_shl)
// and keeps what it held: the registers start as kSent pairs and are only ever written by these shifts, so the edge lanes read kSent
// for ever -- without re-materialising the constant before every shift.
__device__ inline FwVec fw_path(const FwVec& l, const FwVec& c, short2v P1v, int P2, uint32_t& nb_lo, uint32_t& nb_hi) {
    const short2v m2 = __builtin_elementwise_min(__builtin_elementwise_min(l.p[0], l.p[1]), l.p[2]);
    int m = (int)min(m2.x, m2.y);
    m = fw_min_dpp<0xB1>(m); m = fw_min_dpp<0x4E>(m); m = fw_min_dpp<0x141>(m); m = fw_min_dpp<0x140>(m);
    const short dl = (short)(m + P2);
    const short2v delta = {dl, dl};
    nb_lo = (uint32_t)dpp_mov<0x111>((int)nb_lo, (int)fw_u(l.p[2])); // lane r - 1's (d4, d5): its high half is disparity 6r - 1
    nb_hi = (uint32_t)dpp_mov<0x101>((int)nb_hi, (int)fw_u(l.p[0])); // lane r + 1's (d0, d1): its low half is disparity 6r + 6
    const short2v dn0 = fw_s2(__builtin_amdgcn_alignbit(fw_u(l.p[0]), nb_lo, 16));           // (d-1, d0)
    const short2v dn1 = fw_s2(__builtin_amdgcn_alignbit(fw_u(l.p[1]), fw_u(l.p[0]), 16));    // (d1, d2)
    const short2v dn2 = fw_s2(__builtin_amdgcn_alignbit(fw_u(l.p[2]), fw_u(l.p[1]), 16));    // (d3, d4)
    const short2v up2 = fw_s2(__builtin_amdgcn_alignbit(nb_hi, fw_u(l.p[2]), 16));           // (d5, d6)
    FwVec n;
    n.p[0] = (c.p[0] - delta) + __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_min(dn0, dn1) + P1v, l.p[0]), delta);
    n.p[1] = (c.p[1] - delta) + __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_min(dn1, dn2) + P1v, l.p[1]), delta);
    n.p[2] = (c.p[2] - delta) + __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_min(dn2, up2) + P1v, l.p[2]), delta);
    return n;
}
template <int kFwRows> // image rows per slab = DPP rows per workgroup: 64 for throughput, 32 when the batch alone cannot fill the chip (shorter steps, twice the workgroups)
__global__ __launch_bounds__(kFwRows * 16) void sgbm_forward_kernel(SgbmDims dm, const int16_t* __restrict__ C, int16_t* __restrict__ S1,
                                                                     uint32_t* bndg, int* flags, int* ctl, int nslab) {
    constexpr int kFwThreads = kFwRows * 16;
    constexpr int kFwStage = (kFwChunk * kFwRecDw + kFwThreads - 1) / kFwThreads; // dwords per thread of one staged chunk
    __shared__ FwShared<kFwRows> sm;
    // slab-major order: all first slabs of the batch, then all second slabs ... -- a slab still only waits for a lower index, and in a batch
    // that does not fit the chip at once a slab is dispatched when its predecessor is long under way (pair-major order made every workgroup
    // sit out the ~190 steps until the slab above has its first boundary pixels)
    const int tid = threadIdx.x, row_l = tid >> 4, r = tid & 15;
    // Ticket = logical index, drawn when the workgroup STARTS RUNNING.  Eight ticket pools, one per XCD (pool p hands out the indices
    // = p mod 8, ascending): a slab's predecessor is index - nb, and with nb a multiple of 8 it comes out of the same pool -- claimed
    // earlier, hence resident or finished (the deadlock-freedom argument, per pool), and, when the workgroup draws from its own XCD's pool,
    // running on the same XCD: the boundary records and the flag then travel through one L2 (drawing from a single global counter put
    // chained slabs on unrelated XCDs: 2.9 -> 4.3 ms per 32 pairs).  The XCC id only picks the pool tried first -- speed, not correctness:
    // a workgroup whose pool is exhausted takes an index from the next one (there are exactly as many indices as workgroups).
    const int nb = gridDim.x / nslab;
#if defined(VSLAM_SGBM_FW_NO_TICKET) // A/B aid only: logical index = blockIdx (relies on in-order dispatch)
    if (tid == 0) { sm.ctl[0] = (int)blockIdx.x; sm.ctl[1] = 0; }
#else
    if (tid == 0) {
        const int npool = (nb % 8 == 0) ? 8 : 1, ntot = gridDim.x;
        const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) % npool; // HW_REG_XCC_ID[3:0]
        // A courtesy wait, bounded and irrelevant to correctness: when this workgroup's blockIdx says that t other workgroups of its XCD come
        // first (the dispatch order observed on this part), give them a few microseconds to draw before it does.  With in-order dispatch
        // every workgroup then gets ticket t == blockIdx / 8 of its pool -- early slabs on the chip first, late slabs doubling up with early ones on a CU --
        // the placement the sweep was tuned on (a free-for-all draw on a fully resident grid cost 3.0 -> 4.0 ms at 32 pairs).
        // (the XCC id is a permutation of blockIdx % 8 -- observed: block b on XCC (b + 7) % 8 -- so the turn is counted per pool, whatever its id)
        if (npool == 8) {
            const int th = (int)(blockIdx.x >> 3);
            for (int spin = 0; spin < 256 && __hip_atomic_load(&ctl[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < th; ++spin) __builtin_amdgcn_s_sleep(4);
        }
        int got = 0; // (never stays 0 unclaimed: ntot claims exist for ntot workgroups)
        for (int k = 0; k < npool; ++k) {
            const int p = (xcc + k) % npool;
            const int cand = atomicAdd(&ctl[p], 1) * npool + p;
            if (cand < ntot) { got = cand; break; }
        }
        sm.ctl[0] = got; sm.ctl[1] = 0;
#ifdef VSLAM_SGBM_FW_DEBUG
        printf("fw blk %d xcc %d lid %d t %lld\n", (int)blockIdx.x, xcc, got, (long long)wall_clock64());
#endif
    }
#endif
    __syncthreads();
    const int lid = sm.ctl[0];
    const int slab = lid / nb, b = lid - slab * nb;
    const int W1 = dm.width1, h = dm.h;
    const int y = slab * kFwRows + row_l;
    const bool rowok = y < h;
    const int P1 = dm.P1, P2 = dm.P2;
    const size_t rowbase = (((size_t)b * h + min(y, h - 1)) * W1) * 96 + 6 * r;
    const int16_t* cp = C + rowbase;
    int16_t* sp = S1 + rowbase;
    const bool has_pred = slab > 0, has_succ = slab + 1 < nslab;
    // Boundary records: pixel x >= 1 of a slab's last row lives at index x - 1 of a run padded to whole chunks (pixel 0, which the row
    // below needs once, before its first step, sits in a chunk of its own behind the run), so that chunk c of the reader (pixels 32 c + 1 .. 32 c + 32) is a run of WHOLE cache lines (32 x 576 B =
    // 144 lines).  The reader's L2 may keep what it fetched: a line shared by two chunks would be fetched with the first, before the
    // second's records exist, and served stale afterwards (device-scope loads do not re-fetch lines another XCD has written since).
    const size_t bnd_rec0 = (size_t)((W1 + kFwChunk - 1) / kFwChunk) * kFwChunk * kFwRecDw, bnd_run = bnd_rec0 + (size_t)kFwChunk * kFwRecDw;
    const uint32_t* bnd_in = bndg + ((size_t)b * (nslab - 1) + max(slab - 1, 0)) * bnd_run;
    uint32_t* bnd_out = bndg + ((size_t)b * (nslab - 1) + max(min(slab, nslab - 2), 0)) * bnd_run + r * 9;
    int* flag_in = flags + max(lid - nb, 0);
    int* flag_out = flags + lid;
    const bool writer = has_succ && row_l == kFwRows - 1; // (only the last slab has fewer than 64 image rows, and it has no successor)
    // wait until the slab above has published `need` pixels of its last row (first lane only; the workgroup follows through a barrier)
    // Returns false (uniformly) when the backstop fired: the caller leaves the kernel through fw_abort.
    auto wait_pred = [&](int need) -> bool {
        if (has_pred && tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(flag_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > VSLAM_SGBM_FW_SPIN_LIMIT) { sm.ctl[1] = 1; break; }
            }
        }
#if VSLAM_SGBM_FW_ACQ == 2
        if (has_pred && tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // polling wave only: the L1 it invalidates is the CU's
#endif
        __syncthreads();
#if VSLAM_SGBM_FW_ACQ == 1
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the records published before the flag are visible to the loads below
#endif
        return sm.ctl[1] == 0;
    };
    auto fw_abort = [&]() { // error word for the host, successors released (their results are void with the word set)
        if (tid == 0) {
            atomicExch(&ctl[8], 1);
            __hip_atomic_store(flag_out, 0x7FFFFFFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // records first .. first + 31 of the slab above (device-scope loads: they were written by another workgroup, possibly through another L2)
    uint32_t stage[kFwStage];
    auto stage_load = [&](int first) {
#pragma unroll
        for (int q = 0; q < kFwStage; ++q) {
            const int dw = tid + q * kFwThreads;
            const int rec = first + dw / kFwRecDw;
            stage[q] = 0;
            if (has_pred && dw < kFwChunk * kFwRecDw && rec < W1)
                stage[q] = __hip_atomic_load(bnd_in + (size_t)(first - 1) * kFwRecDw + dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < kFwStage; ++q) { const int dw = tid + q * kFwThreads; if (dw < kFwChunk * kFwRecDw) sm.bnd[buf][dw] = stage[q]; }
    };
    for (int i = tid; i < 2 * kFwRows * 16 * 9; i += kFwThreads) (&sm.slot[0][0][0])[i] = 0;
    // chunk 0 = records 1 .. 32 (row 0 at step t reads the record of pixel t + 1)
    if (!wait_pred(min(W1, 1 + kFwChunk))) { fw_abort(); return; }
    stage_load(1);
    stage_store(0);
    __syncthreads();
    U3 cq[kFwPF];
#pragma unroll
    for (int k = 0; k < kFwPF; ++k) cq[k] = ld_u3<true>(cp + (size_t)min(max(k - 2 * row_l, 0), W1 - 1) * 96);
    FwVec l10 = {{short2v{0, 0}, short2v{0, 0}, short2v{0, 0}}};
    const short2v P1v = {(short)dm.P1, (short)dm.P1};
    uint32_t nb_lo = (uint32_t)kSent | ((uint32_t)kSent << 16), nb_hi = nb_lo;
    uint32_t h01[3] = {0, 0, 0}, h11a[3] = {0, 0, 0}, h11b[3] = {0, 0, 0};
    if (has_pred && row_l == 0) { // pixel 0 of the row above: what an inner row picks up in the two steps before its own first pixel
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            h01[q] = __hip_atomic_load(bnd_in + bnd_rec0 + r * 9 + 3 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h11a[q] = __hip_atomic_load(bnd_in + bnd_rec0 + r * 9 + 6 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int nsteps = W1 + 2 * (kFwRows - 1);
    for (int t0 = 0; t0 < nsteps; t0 += kFwChunk) {
        const int chunk = t0 / kFwChunk;
        if (!wait_pred(min(W1, 1 + (chunk + 2) * kFwChunk))) { fw_abort(); return; } // records of chunk + 1: pixels 32 (chunk + 1) + 1 .. + 32
        stage_load(1 + (chunk + 1) * kFwChunk);
        for (int u0 = 0; u0 < kFwChunk; u0 += kFwPF) {
#pragma unroll
            for (int k = 0; k < kFwPF; ++k) {
                const int t = t0 + u0 + k, x = t - 2 * row_l;
                const bool act = rowok && x >= 0 && x < W1;
                const U3 c3 = cq[k];
                cq[k] = ld_u3<true>(cp + (size_t)min(max(x + kFwPF, 0), W1 - 1) * 96);
                // what the row above left behind one step ago: its L of pixel x + 1
                const uint32_t* src = row_l == 0 ? &sm.bnd[chunk & 1][(t & (kFwChunk - 1)) * kFwRecDw + r * 9] : &sm.slot[(t + 1) & 1][row_l - 1][r * 9];
                uint32_t a[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) a[q] = src[q];
                const FwVec cv = fw_vec(c3.a, c3.b, c3.c);
                FwVec n10 = fw_path(l10, cv, P1v, P2, nb_lo, nb_hi);
                const FwVec nm = fw_path(fw_vec(a[0], a[1], a[2]), cv, P1v, P2, nb_lo, nb_hi);           // (-1, 1): from (x + 1, y - 1), one step old
                const FwVec n01 = fw_path(fw_vec(h01[0], h01[1], h01[2]), cv, P1v, P2, nb_lo, nb_hi);    // (0, 1):  from (x, y - 1), two steps old
                const FwVec n11 = fw_path(fw_vec(h11b[0], h11b[1], h11b[2]), cv, P1v, P2, nb_lo, nb_hi); // (1, 1):  from (x - 1, y - 1), three steps old
#pragma unroll
                for (int q = 0; q < 3; ++q) { h01[q] = a[3 + q]; h11b[q] = h11a[q]; h11a[q] = a[6 + q]; }
                uint32_t* dst = &sm.slot[t & 1][row_l][r * 9];
                if (act) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) { dst[q] = fw_u(nm.p[q]); dst[3 + q] = fw_u(n01.p[q]); dst[6 + q] = fw_u(n11.p[q]); }
                    l10 = n10;
                    // sat16 of the four-term sum: the two pair sums cannot overflow int16 (|L| < 2^14), one saturating add finishes
                    U3 s3;
                    s3.a = fw_u(__builtin_elementwise_add_sat(n10.p[0] + nm.p[0], n01.p[0] + n11.p[0]));
                    s3.b = fw_u(__builtin_elementwise_add_sat(n10.p[1] + nm.p[1], n01.p[1] + n11.p[1]));
                    s3.c = fw_u(__builtin_elementwise_add_sat(n10.p[2] + nm.p[2], n01.p[2] + n11.p[2]));
                    st_u3<true>(sp + (size_t)x * 96, s3);
                    if (writer) {
                        uint32_t* wp = bnd_out + (x > 0 ? (size_t)(x - 1) * kFwRecDw : bnd_rec0);
#pragma unroll
                        for (int q = 0; q < 3; ++q) { wp[q] = fw_u(nm.p[q]); wp[3 + q] = fw_u(n01.p[q]); wp[6 + q] = fw_u(n11.p[q]); }
                        if ((x & (kFwChunk - 1)) == 0 || x == W1 - 1) { // progress = x + 1 pixels: 32 k + 1 is exactly what the slab below waits for
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                            if (r == 0) __hip_atomic_store(flag_out, x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                } else { // outside the volume every path restarts from zero
#pragma unroll
                    for (int q = 0; q < 9; ++q) dst[q] = 0;
#pragma unroll
                    for (int q = 0; q < 3; ++q) l10.p[q] = short2v{0, 0};
                }
                __syncthreads();
            }
        }
        stage_store((chunk + 1) & 1);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------- fused top-down pass
// sgbm_down_kernel = pixel cost + horizontal box + vertical box + the vertical path (0, 1) in ONE sweep: a workgroup owns kDnCols
// columns of the cost volume and walks down the rows.  Per row: (1) the Birchfield-Tomasi cost of its columns plus the 4-column
// halo goes into a small LDS tile (one work item = one pixel x 6 disparities, operands prefetched a row ahead); (2) the lane that
// owns (column, 6 disparities) of the path sums its nine tile entries = hsum; a nine-deep REGISTER ring of hsum rows turns it into
// the sliding vertical sum C (top rows replicated, the last SH2 rows frozen, column 0 frozen after the first row -- the library's
// border rules, see sgbm_vsum_kernel); (3) C is stored for the other four paths and consumed on the spot by the vertical path's
// recurrence, whose L goes out as T.  hsum never exists in memory and C is not re-read: 2 volume writes instead of
// 1 write (hsum) + 2 reads + 1 write (vsum) + 1 read + 1 write (path 0,1).
#ifndef VSLAM_SGBM_DN_NT
#define VSLAM_SGBM_DN_NT 1 // non-temporal stores of C / T in the fused top-down kernel: 2.69 -> 2.38 ms per 32 pairs (the 165 MB per pair it writes are next read a kernel later)
#endif
#ifndef VSLAM_SGBM_DN_COLS
#define VSLAM_SGBM_DN_COLS 24
#endif
#ifdef VSLAM_SGBM_PROFILE // tuning aid (tools/build_variant.sh ... -DVSLAM_SGBM_PROFILE): cycles per phase of one workgroup's wave 0
__device__ long long g_sgbm_dbg[8];
#define DN_T(slot) do { if (dbg__) { const long long t1__ = clock64(); acc__[slot] += t1__ - t0__; t0__ = t1__; } } while (0)
#else
#define DN_T(slot) do {} while (0)
#endif
constexpr int kDnCols = VSLAM_SGBM_DN_COLS, kDnThreads = (kDnCols + 8) * 16; // one pixel-cost item per (tile column, 16-lane slot)
static_assert(kDnCols % 4 == 0 && 6 * 4 * ((kDnCols + 8 + 97 + 3) / 4 + 1) <= 2 * kDnThreads, "the strip loader issues at most two dword loads per thread");
constexpr int kStripDw = (kDnCols + 8 + 97 + 3) / 4 + 1;  // dwords per (plane, byte shift) copy of the right-view strip: window starts 0 .. kDnCols + 97, 8 bytes each
struct DnStage { uint32_t a, b; uint8_t l; };             // one row's share of the operand strips on its way from memory to LDS (raw loads: nothing is
                                                          // computed on them at fetch time, or the wave would wait for its own prefetch)

template <bool WITH_PATH> // false: C only (the vertical path runs inside sgbm_forward_kernel)
__global__ __launch_bounds__(kDnThreads) void sgbm_down_kernel(SgbmDims dm, const uint8_t* __restrict__ pre, int16_t* __restrict__ C, uint16_t* __restrict__ T) {
    const int b = blockIdx.y, j0 = blockIdx.x * kDnCols;
    const int W1 = dm.width1, w = dm.w, h = dm.h;
    // [row parity][disparity][tile column] bytes (pitch 40: the 16 lanes of a DPP row hit 16 different banks): a dword holds ONE disparity of four
    // adjacent columns, so the 9-tap horizontal sum is three v_dot4_u32_u8 with 0/1 byte masks instead of nine unpack-and-add steps
    constexpr int kTilePitch = kDnCols + 8 + 8;
    __shared__ alignas(16) uint8_t tile[2][96 * kTilePitch];
    const int t = threadIdx.x >> 4, r = threadIdx.x & 15, d = 6 * r;
    const int jj = min(max(j0 - 4 + t, 0), W1 - 1), x = dm.minX1 + jj; // pixel-cost column of this item (clamped to the volume)
    const uint8_t* Lb = pre + ((size_t)(2 * b) * h * 6) * w;
    const uint8_t* Rb = pre + ((size_t)(2 * b + 1) * h * 6) * w;
    // Operands through LDS.  A workgroup row needs 32 left pixels and a 130-byte strip of the right view per plane; fetched per lane (six byte
    // loads + six unaligned 8-byte windows) that was 24.6 KB of requests for 0.9 KB of data, and the kernel was bound by the texture addresser
    // (TA busy 85-90 % of the kernel, VALU 41 %; profiles/r03_sgbm_detail_summary.txt).  Now the strips are staged once per row: the right view as FOUR
    // copies shifted by 0..3 bytes (loaded at byte offsets s + 4 i -- global loads may be unaligned, LDS reads may not), so that a lane whose
    // window starts at byte o reads two aligned dwords of copy o & 3; the left view as 8-byte records {val, lo, hi} x 2 channels per column.
    // <= 2 dword loads + 1 byte load per thread and row instead of 12 loads.
    __shared__ uint32_t rstrip[2][6 * 4 * kStripDw];
    __shared__ alignas(8) uint8_t lstrip[2][(kDnCols + 8) * 8];
    const int X0 = dm.minX1 + j0 - 4 - 95; // first byte of the strip (may be negative at the left border: those bytes are never consumed)
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Lb), 0, h * 6 * w + 8, 0x00020000);
    const auto rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Rb), 0, h * 6 * w + 8, 0x00020000);
    constexpr int kStripAll = 6 * 4 * kStripDw;
    const int tid = threadIdx.x;
    const int i0s = tid, i1s = min(tid + kDnThreads, kStripAll - 1);
    const bool has_b = tid + kDnThreads < kStripAll, has_l = tid < (kDnCols + 8) * 6;
    auto strip_off = [&](int idx) { const int q = idx / (4 * kStripDw), rem = idx - q * 4 * kStripDw, sft = rem / kStripDw, ii = rem - sft * kStripDw; return q * w + X0 + sft + 4 * ii; };
    const int offA = strip_off(i0s), offB = strip_off(i1s);   // (a negative offset is a huge unsigned one: out of range, reads 0)
    const int lq = tid / (kDnCols + 8), lt = tid - lq * (kDnCols + 8);
    const int offLs = min(lq, 5) * w + dm.minX1 + min(max(j0 - 4 + lt, 0), W1 - 1);
    auto fetch = [&](int y, DnStage& in) {
        const int row = y * 6 * w; // uniform
        in.a = __builtin_amdgcn_raw_buffer_load_b32(rsR, offA, row, 0);
        in.b = __builtin_amdgcn_raw_buffer_load_b32(rsR, offB, row, 0);
        in.l = __builtin_amdgcn_raw_buffer_load_b8(rsL, offLs, row, 0);
    };
    auto stage = [&](int par, const DnStage& in) {
        rstrip[par][i0s] = in.a;
        if (has_b) rstrip[par][i1s] = in.b;
        if (has_l) lstrip[par][lt * 8 + lq] = in.l;
    };
    // this lane's windows: byte o = (column offset in the tile) + 90 - 6 r of the strip, in every plane
    const int wo = (jj - (j0 - 4)) + 90 - d;
    const int roff = (wo & 3) * kStripDw + (wo >> 2), loff = (jj - (j0 - 4)) * 8;
    // path lane: column j = j0 + t (t < kDnCols), disparities 6 r .. 6 r + 5
    const bool path_lane = t < kDnCols;
    const uint32_t hm0 = 0x01010101u << (8 * (t & 3)), hm2 = 0x01010101u >> (8 * (3 - (t & 3))); // 0/1 byte masks of the 9-column window
    const int j = j0 + t;
    const bool live = path_lane && j < W1;
    const bool frozen = j == 0; // the reference never updates column 0 after the first row
    const size_t vbase = ((size_t)b * h * W1 + j) * 96 + 6 * r; // (row 0, column j)
    const size_t rstride = (size_t)W1 * 96;
    short2v ring[9][3], acc[3] = {short2v{0, 0}, short2v{0, 0}, short2v{0, 0}};
#pragma unroll
    for (int q = 0; q < 9; ++q) { ring[q][0] = ring[q][1] = ring[q][2] = short2v{0, 0}; }
    int l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, l5 = 0, minPrev = 0;
    const int P1 = dm.P1, P2 = dm.P2;
    // The operands of row i + 1 are fetched while row i is processed.  The two operand sets alternate BY NAME (the body is unrolled 18 =
    // 2 x 9 times: set = step parity, ring slot = step mod 9): a register copy "cur = nxt" would make the compiler wait for every
    // outstanding memory operation -- the C / T stores of the row before included -- once per row.
    DnStage in[2];
    fetch(0, in[0]);
    stage(0, in[0]);
    fetch(min(1, h - 1), in[0]);
    __syncthreads();
#ifdef VSLAM_SGBM_PROFILE
    const bool dbg__ = blockIdx.x == 7 && blockIdx.y == 0 && threadIdx.x == 0;
    long long acc__[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0__ = clock64();
#endif
    for (int i0 = 0; i0 < h + dm.SH2; i0 += 18) {
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            constexpr int kDummy = 0; (void)kDummy;
            const int sidx = step % 9;
            DnStage& cur = in[step & 1];       // row i + 1, fetched a step ago: goes to LDS now
            DnStage& nxt = in[(step + 1) & 1]; // row i + 2: requested now
            const int i = i0 + step; // hsum row produced in this step (while i < h); emitted row y = i - SH2
            if (i >= h + dm.SH2) continue; // uniform (no break: the ring index must stay a compile-time constant)
            { // straight-line on purpose (row indices clamped instead of branches): with the fetch inside a conditional the compiler's
              // wait-count insertion falls back to vmcnt(0) right after issuing the prefetch
                stage((i + 1) & 1, cur);
                fetch(min(i + 2, h - 1), nxt);
                const uint2 lv = *reinterpret_cast<const uint2*>(&lstrip[i & 1][loff]);
                // (1) pixel cost of (column jj, disparities d .. d + 5), both channels, two disparities per packed-i16 operation:
                // byte 5 - k of a right-view window is disparity d + k, so v_perm pulls the pairs (d, d+1), (d+2, d+3), (d+4, d+5) out of
                // the two window dwords; cost = min(max(0, u - v1, v0 - u), max(0, v - u1, u0 - v)), channel 1 (raw / 4) shifted before the add
                short2v cost[3];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    // left {val, lo, hi} of channel c: byte 3c + k of the column's record, splat into both halves
                    auto splat = [&](int k) { return __builtin_bit_cast(short2v, __builtin_amdgcn_perm(lv.y, lv.x, 0x0c000c00u | (uint32_t)k | ((uint32_t)k << 16))); };
                    const short2v U = splat(3 * c), U0 = splat(3 * c + 1), U1 = splat(3 * c + 2);
                    const uint32_t* rp = &rstrip[i & 1][roff];
                    const uint32_t vl = rp[(3 * c) * 4 * kStripDw], vh = rp[(3 * c) * 4 * kStripDw + 1];
                    const uint32_t v0l = rp[(3 * c + 1) * 4 * kStripDw], v0h = rp[(3 * c + 1) * 4 * kStripDw + 1];
                    const uint32_t v1l = rp[(3 * c + 2) * 4 * kStripDw], v1h = rp[(3 * c + 2) * 4 * kStripDw + 1];
                    const short2v zero = {0, 0};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const uint32_t sel = k == 0 ? 0x0c040c05u : (k == 1 ? 0x0c020c03u : 0x0c000c01u);
                        const short2v V = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(vh, vl, sel));
                        const short2v V0 = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(v0h, v0l, sel));
                        const short2v V1 = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(v1h, v1l, sel));
                        const short2v c0 = __builtin_elementwise_max(__builtin_elementwise_max(U - V1, zero), V0 - U);
                        const short2v c1 = __builtin_elementwise_max(__builtin_elementwise_max(V - U1, zero), U0 - V);
                        const short2v m = __builtin_elementwise_min(c0, c1);
                        if (c == 0) cost[k] = m;
                        else cost[k] = cost[k] + __builtin_bit_cast(short2v, (__builtin_bit_cast(uint32_t, m) >> 2) & 0x3FFF3FFFu);
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    tile[i & 1][(d + 2 * k) * kTilePitch + t] = (uint8_t)cost[k].x;
                    tile[i & 1][(d + 2 * k + 1) * kTilePitch + t] = (uint8_t)cost[k].y;
                }
            }
            DN_T(0);
            __syncthreads();
            DN_T(1);
            if (path_lane) {
                if (i < h) {
                    // (2) hsum(i) of column j: nine tile columns t .. t + 8 (= volume columns j - 4 .. j + 4, clamped)
                    // window = tile columns t .. t + 8 = bytes sh .. sh + 8 of the three aligned dwords at column t & ~3 (sh = t & 3)
                    short2v hs[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        uint32_t sum[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&tile[i & 1][(d + 2 * k + e) * kTilePitch + (t & ~3)]);
                            sum[e] = __builtin_amdgcn_udot4(w3[0], hm0, __builtin_amdgcn_udot4(w3[1], 0x01010101u, __builtin_amdgcn_udot4(w3[2], hm2, 0u, false), false), false);
                        }
                        hs[k] = __builtin_bit_cast(short2v, sum[0] | (sum[1] << 16));
                    }
                    if (i == 0) { // rows above the image replicate row 0: C(0) = 5 hsum(0) + hsum(1..4); every ring slot starts as hsum(0)
#pragma unroll
                        for (int q = 0; q < 9; ++q) { ring[q][0] = hs[0]; ring[q][1] = hs[1]; ring[q][2] = hs[2]; }
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc[k] = hs[k] + hs[k] + hs[k] + hs[k] + hs[k];
                    } else if (i <= dm.SH2) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) { ring[sidx][k] = hs[k]; acc[k] = acc[k] + hs[k]; }
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const short2v old = ring[sidx][k]; // hsum(max(i - 9, 0))
                            ring[sidx][k] = hs[k];
                            if (!frozen) acc[k] = acc[k] + hs[k] - old;
                        }
                    }
                }
                DN_T(2);
                if (i >= dm.SH2) {
                    // (3) row y = i - SH2: C = acc (frozen over the last SH2 rows: no hsum arrives any more), vertical path, T
                    const int y = i - dm.SH2;
                    U3 c;
                    c.a = __builtin_bit_cast(uint32_t, acc[0]); c.b = __builtin_bit_cast(uint32_t, acc[1]); c.c = __builtin_bit_cast(uint32_t, acc[2]);
                    if constexpr (WITH_PATH) {
                    const int lm = dpp_mov<0x111>(kSent, l5);
                    const int lp = dpp_mov<0x101>(kSent, l0);
                    const int delta = minPrev + P2;
                    const int n0 = lo16s(c.a) - delta + min(min(l0, min(lm, l1) + P1), delta);
                    const int n1 = hi16s(c.a) - delta + min(min(l1, min(l0, l2) + P1), delta);
                    const int n2 = lo16s(c.b) - delta + min(min(l2, min(l1, l3) + P1), delta);
                    const int n3 = hi16s(c.b) - delta + min(min(l3, min(l2, l4) + P1), delta);
                    const int n4 = lo16s(c.c) - delta + min(min(l4, min(l3, l5) + P1), delta);
                    const int n5 = hi16s(c.c) - delta + min(min(l5, min(l4, lp) + P1), delta);
                    l0 = n0; l1 = n1; l2 = n2; l3 = n3; l4 = n4; l5 = n5;
                    minPrev = row16_min(min(min(min(n0, n1), min(n2, n3)), min(n4, n5)));
                    if (live) {
                        U3 o;
                        o.a = pack16(n0 + kTOffset, n1 + kTOffset); o.b = pack16(n2 + kTOffset, n3 + kTOffset); o.c = pack16(n4 + kTOffset, n5 + kTOffset);
                        st_u3<VSLAM_SGBM_DN_NT != 0>(C + vbase + (size_t)y * rstride, c);
                        st_u3<VSLAM_SGBM_DN_NT != 0>(T + vbase + (size_t)y * rstride, o);
                    }
                    } else if (live) st_u3<VSLAM_SGBM_DN_NT != 0>(C + vbase + (size_t)y * rstride, c);
                }
                DN_T(3);
            }
        }
    }
#ifdef VSLAM_SGBM_PROFILE
    if (dbg__) for (int q = 0; q < 4; ++q) g_sgbm_dbg[q] = acc__[q];
#endif
}

// stand-alone winner-take-all over a stored S volume (MODE 3): one DPP row per pixel
__global__ __launch_bounds__(256) void sgbm_wta_kernel(SgbmDims dm, const uint16_t* __restrict__ S, int4* __restrict__ rec) {
    const int b = blockIdx.y;
    const size_t pixel = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int r = threadIdx.x & 15;
    const size_t npx = (size_t)dm.h * dm.width1;
    if (pixel >= npx) return;
    const U3 v = *(const U3*)(S + ((size_t)b * npx + pixel) * 96 + 6 * r);
    const int y = (int)(pixel / dm.width1), x = (int)(pixel - (size_t)y * dm.width1);
    wta_row16(dm, lo16s(v.a), hi16s(v.a), lo16s(v.b), hi16s(v.b), lo16s(v.c), hi16s(v.c), r, ((size_t)b * dm.h + y) * dm.w + x + dm.minX1, rec, true);
}

// ------------------------------------------------------------------------------------------- left-right check
// disp2[x2] = disparity of the cheapest winner landing on right-image column x2 = x - d (the reference scans x from the
// right with a strict ">", so among equal costs the largest x wins): packed-key atomicMin in LDS, then the consistency test.
constexpr int kLrBlock = 256;
__global__ __launch_bounds__(kLrBlock) void sgbm_lrcheck_kernel(SgbmDims dm, const int4* __restrict__ rec, int16_t* __restrict__ disp) {
    const int y = blockIdx.x, b = blockIdx.y;
    extern __shared__ int keys[]; // w keys + w/2 words of disp1
    int16_t* disp1 = (int16_t*)(keys + dm.w);
    const size_t row = ((size_t)b * dm.h + y) * dm.w;
    const int INVALID = -16;
    for (int i = threadIdx.x; i < dm.w; i += kLrBlock) keys[i] = 0x7FFFFFFF;
    __syncthreads();
    for (int x = threadIdx.x; x < dm.w; x += kLrBlock) {
        int dd = INVALID;
        if (x >= dm.minX1) {
            const int4 rc = rec[row + x];
            if (rc.x >= 0) {
                const int bd = rc.x & 0xFF;
                atomicMin(&keys[x - bd], ((rc.x >> 8) << 12) | (4095 - x));
                if (0 < bd && bd < dm.D - 1) {
                    const int sm = lo16s((uint32_t)rc.y), sc = hi16s((uint32_t)rc.y), sp = rc.z;
                    const int denom2 = max(sm + sp - 2 * sc, 1);
                    dd = bd * 16 + ((sm - sp) * 16 + denom2) / (denom2 * 2);
                } else dd = bd * 16;
            }
        }
        disp1[x] = (int16_t)dd;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < dm.w; x += kLrBlock) {
        int d1 = disp1[x];
        if (d1 != INVALID) {
            const int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
            const int _x = x - _d, x_ = x - d_;
            bool f1 = false, f2 = false;
            if (0 <= _x && _x < dm.w && keys[_x] != 0x7FFFFFFF) { const int d2 = (4095 - (keys[_x] & 4095)) - _x; f1 = abs(d2 - _d) > dm.disp12; }
            if (0 <= x_ && x_ < dm.w && keys[x_] != 0x7FFFFFFF) { const int d2 = (4095 - (keys[x_] & 4095)) - x_; f2 = abs(d2 - d_) > dm.disp12; }
            if (f1 && f2) d1 = INVALID;
        }
        disp[row + x] = (int16_t)d1;
    }
}

// ------------------------------------------------------------------------------------------- median 3x3
__global__ __launch_bounds__(256) void sgbm_median3_kernel(int w, int h, const int16_t* __restrict__ src, int16_t* __restrict__ dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= w) return;
    const int16_t* s = src + (size_t)b * w * h;
    const int y0 = max(y - 1, 0), y2 = min(y + 1, h - 1), x0 = max(x - 1, 0), x2 = min(x + 1, w - 1);
    int p[9] = {s[(size_t)y0 * w + x0], s[(size_t)y0 * w + x], s[(size_t)y0 * w + x2], s[(size_t)y * w + x0], s[(size_t)y * w + x],
                s[(size_t)y * w + x2], s[(size_t)y2 * w + x0], s[(size_t)y2 * w + x], s[(size_t)y2 * w + x2]};
#define CSWAP(a, b) { const int lo__ = min(p[a], p[b]), hi__ = max(p[a], p[b]); p[a] = lo__; p[b] = hi__; }
    // 9-element median network (19 compare-exchanges)
    CSWAP(1, 2) CSWAP(4, 5) CSWAP(7, 8) CSWAP(0, 1) CSWAP(3, 4) CSWAP(6, 7) CSWAP(1, 2) CSWAP(4, 5) CSWAP(7, 8)
    CSWAP(0, 3) CSWAP(5, 8) CSWAP(4, 7) CSWAP(3, 6) CSWAP(1, 4) CSWAP(2, 5) CSWAP(4, 7) CSWAP(4, 2) CSWAP(6, 4) CSWAP(4, 2)
#undef CSWAP
    dst[(size_t)b * w * h + (size_t)y * w + x] = (int16_t)p[4];
}

// ------------------------------------------------------------------------------------------- speckle filter (CCL)
__device__ inline int ccl_find(const int* parent, int p) {
    int q = parent[p];
    while (q != p) { p = q; q = parent[p]; }
    return p;
}
__device__ inline void ccl_union(int* parent, int a, int b) {
    while (true) {
        a = ccl_find(parent, a); b = ccl_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[b], a);
        if (old == b) return;
        b = old;
    }
}

// parent indices are image-local (p = y*w + x) inside each image's slice.  Rows are pre-merged: every pixel starts out
// pointing at the first pixel of its horizontal run, so the atomic union-find only has to stitch runs across rows.
constexpr int kCclBlock = 256;
__global__ __launch_bounds__(kCclBlock) void sgbm_ccl_rows_kernel(int w, int h, int maxDiff, int newVal, const int16_t* __restrict__ disp,
                                                                int* __restrict__ parent, int* __restrict__ count) {
    const int y = blockIdx.x, b = blockIdx.y;
    extern __shared__ int runs[]; // 2 x w (ping-pong max-scan of run starts)
    const int16_t* row = disp + ((size_t)b * h + y) * w;
    for (int x = threadIdx.x; x < w; x += kCclBlock) {
        const int d = row[x];
        const bool joined = x > 0 && d != newVal && row[x - 1] != newVal && abs(d - row[x - 1]) <= maxDiff;
        runs[x] = joined ? 0 : x;
    }
    __syncthreads();
    int cur = 0;
    for (int o = 1; o < w; o <<= 1) {
        for (int x = threadIdx.x; x < w; x += kCclBlock) runs[(cur ^ 1) * w + x] = x >= o ? max(runs[cur * w + x], runs[cur * w + x - o]) : runs[cur * w + x];
        cur ^= 1;
        __syncthreads();
    }
    const size_t base = (size_t)b * w * h + (size_t)y * w;
    for (int x = threadIdx.x; x < w; x += kCclBlock) { parent[base + x] = y * w + runs[cur * w + x]; count[base + x] = 0; }
}
__global__ __launch_bounds__(256) void sgbm_ccl_union_kernel(int w, int h, int maxDiff, int newVal, const int16_t* __restrict__ disp,
                                                            int* __restrict__ parent) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= w * (h - 1)) return;
    const int16_t* dsp = disp + (size_t)b * w * h;
    const int dp = dsp[p], dq = dsp[p + w];
    if (dp == newVal || dq == newVal || abs(dp - dq) > maxDiff) return;
    const int x = p % w;
    if (x > 0) { // the same two runs were already joined one pixel to the left
        const int dpl = dsp[p - 1], dql = dsp[p + w - 1];
        if (dpl != newVal && dql != newVal && abs(dp - dpl) <= maxDiff && abs(dq - dql) <= maxDiff && abs(dpl - dql) <= maxDiff) return;
    }
    ccl_union(parent + (size_t)b * w * h, p, p + w);
}
// component sizes, one atomic per horizontal run (the last pixel of a run adds the run length to its root), and
// flattening of the run starts so that every pixel is at most two hops from its root
__global__ __launch_bounds__(256) void sgbm_ccl_count_kernel(int w, int h, int maxDiff, int newVal, const int16_t* __restrict__ disp,
                                                            int* __restrict__ parent, int* __restrict__ count) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= w * h) return;
    const int16_t* dsp = disp + (size_t)b * w * h;
    const int d = dsp[p];
    if (d == newVal) return;
    const int x = p % w;
    const bool run_ends = x == w - 1 || dsp[p + 1] == newVal || abs(d - dsp[p + 1]) > maxDiff;
    if (!run_ends) return;
    int* par = parent + (size_t)b * w * h;
    const bool joined = x > 0 && dsp[p - 1] != newVal && abs(d - dsp[p - 1]) <= maxDiff;
    const int start = joined ? par[p] : p; // non-start pixels keep pointing at their run start
    const int r = ccl_find(par, start);
    if (r != start) par[start] = r;
    atomicAdd(&count[(size_t)b * w * h + r], p - start + 1);
}
__global__ __launch_bounds__(256) void sgbm_ccl_apply_kernel(int w, int h, int newVal, int maxSize, const int* __restrict__ parent,
                                                            const int* __restrict__ count, const int16_t* __restrict__ disp,
                                                            float* __restrict__ out_f32, int16_t* __restrict__ out_i16) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= w * h) return;
    const size_t g = (size_t)b * w * h + p;
    int d = disp[g];
    if (d != newVal && count[(size_t)b * w * h + ccl_find(parent + (size_t)b * w * h, p)] <= maxSize) d = newVal;
    if (out_i16) out_i16[g] = (int16_t)d;
    if (out_f32) out_f32[g] = (float)d * 0.0625f; // convertTo(CV_32F, 1/16): exact
}

// ------------------------------------------------------------------------------------------- host driver
int launch_sgbm(const Tuning& tune, const uint8_t* d_left, const uint8_t* d_right, size_t img_bytes, int pitch, int w, int h, int B, float* d_disp_f32,
                int16_t* d_disp_i16, int16_t* d_disp_raw, uint8_t** scratch, size_t* scratch_bytes, size_t* dev_bytes, hipStream_t stream) {
    if (B <= 0) return VSLAM_OK;
    SgbmDims dm;
    dm.w = w; dm.h = h; dm.D = 96; dm.minX1 = 96; dm.width1 = w - 96; dm.P1 = 8 * 9 * 9; dm.P2 = 32 * 9 * 9; dm.SW2 = 4; dm.SH2 = 4; dm.uniq = 10;
    dm.disp12 = 1; dm.ftzero = 63; dm.pitch = pitch; dm.img_bytes = img_bytes; // visual_odometry.cpp:163-164
    // width1 <= SW2 is undefined in OpenCV 3.2 (unclamped read of pixel-cost columns 0..SW2), so it is an argument error here.
    if (dm.width1 <= dm.SW2 || h <= 2 * dm.SH2 + 1 || w > 4096) { set_error("image size unsupported (need 100 < w <= 4096, h > 9)"); return VSLAM_ERR_ARG; }
    const size_t vol = (size_t)h * dm.width1 * dm.D, npix = (size_t)w * h;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t need = 256; // header: int32 [0..7] ticket pools of the forward sweep, [8] its error word (vslam_sgbm_status_dev)
    const size_t o_pre = need; need += al((size_t)2 * B * h * 6 * w);
    const size_t o_hs = need; need += al((size_t)B * vol * 2);
    const size_t o_C = need; need += al((size_t)B * vol * 2);
    const size_t o_T = o_hs; // hsum is dead once C exists: T reuses its storage (the fused top-down pass never materialises hsum at all)
    const size_t o_rec = need; need += al((size_t)B * npix * 16);
    const size_t o_d0 = need; need += al((size_t)B * npix * 2);
    const size_t o_d1 = need; need += al((size_t)B * npix * 2);
    const size_t o_par = need; need += al((size_t)B * npix * 4);
    const size_t o_cnt = need; need += al((size_t)B * npix * 4);
    // forward sweep: 32-row slabs (workgroups of 512 threads, two per CU) up to 32 pairs, 64-row slabs above.  The slabs of a pair are a chain
    // of ~2 200 (64 rows) / ~2 600 (32 rows) steps and a step's time is mostly its latency (barrier, LDS mailbox, the dependent minimum ->
    // delta -> update chain), so shorter workgroups win until the chip is full: 16 / 24 / 32 / 40 pairs 2.33 / 2.60 / 2.93 / 3.45 ms with 32
    // rows, 3.01 / 3.06 / 3.19 / 3.35 ms with 64 (48-row slabs: 2.64 / 2.76 / 2.92 / 3.31 -- no better anywhere).  Tuning::sgbm_fw_rows overrides.
    const int fw_rows = tune.sgbm_fw_rows > 0 ? tune.sgbm_fw_rows : (B <= 32 ? 32 : 64);
    const int nslab = (h + fw_rows - 1) / fw_rows;
    const size_t o_bnd = need; need += al((size_t)B * (nslab > 1 ? nslab - 1 : 1) * ((dm.width1 + kFwChunk - 1) / kFwChunk + 1) * kFwChunk * kFwRecDw * 4);
    const size_t o_flag = need; need += al((size_t)B * nslab * 4);
    if (*scratch_bytes < need) {
        VS_HIP(hipStreamSynchronize(stream));
        if (*scratch) { (void)hipFree(*scratch); *dev_bytes -= *scratch_bytes; }
        *scratch = nullptr; *scratch_bytes = 0;
        if (hipMalloc((void**)scratch, need) != hipSuccess) { *scratch = nullptr; set_error("SGBM scratch hipMalloc(%zu) failed", need); return VSLAM_ERR_HIP; }
        *scratch_bytes = need; *dev_bytes += need;
    }
    uint8_t* base = *scratch;
    VS_HIP(hipMemsetAsync(base, 0, 64, stream)); // ticket pools + error word of this launch
    uint8_t* pre = base + o_pre; int16_t* hsum = (int16_t*)(base + o_hs); int16_t* C = (int16_t*)(base + o_C);
    uint16_t* T = (uint16_t*)(base + o_T); int4* rec = (int4*)(base + o_rec);
    int16_t* d0 = (int16_t*)(base + o_d0); int16_t* d1 = (int16_t*)(base + o_d1); int* par = (int*)(base + o_par); int* cnt = (int*)(base + o_cnt);
    const int vblocks = (dm.width1 * dm.D + 255) / 256;
    { ProfScope p(stream, "sgbm_prefilter_kernel"); hipLaunchKernelGGL(sgbm_prefilter_kernel, dim3((w + 255) / 256, h, 2 * B), dim3(256), 0, stream, dm, d_left, d_right, pre); }
    // The fused top-down kernel sweeps the rows sequentially with 48 workgroups per pair: it pays from 8 pairs per call on (2.65 vs 4.08 ms
    // at 32 pairs); below that the three massively parallel kernels it replaces are faster (0.99 vs 1.31 ms for one pair).
    // Tuning::sgbm_fuse_min overrides the threshold (tests run both paths).
    const bool unfused = B < (tune.sgbm_fuse_min >= 0 ? tune.sgbm_fuse_min : 8);
    // ... and from there on the four forward paths run as one wavefront sweep (sgbm_forward_kernel) instead of one kernel per path.
    // The slabs of a pair are a chain (about 2200 sequential steps): it pays from 16 pairs per call on.  Tuning::sgbm_fwd_min overrides.
    const bool fwd = !unfused && B >= (tune.sgbm_fwd_min >= 0 ? tune.sgbm_fwd_min : 16);
    if (!unfused && !fwd) { ProfScope p(stream, "sgbm_down_kernel"); hipLaunchKernelGGL(sgbm_down_kernel<true>, dim3((dm.width1 + kDnCols - 1) / kDnCols, B), dim3(kDnThreads), 0, stream, dm, pre, C, T); }
    if (fwd) {
        { ProfScope p(stream, "sgbm_down_kernel"); hipLaunchKernelGGL(sgbm_down_kernel<false>, dim3((dm.width1 + kDnCols - 1) / kDnCols, B), dim3(kDnThreads), 0, stream, dm, pre, C, T); }
        VS_HIP(hipMemsetAsync(base + o_flag, 0, (size_t)B * nslab * 4, stream));
        ProfScope p(stream, "sgbm_forward_kernel");
        if (fw_rows == 64) hipLaunchKernelGGL(sgbm_forward_kernel<64>, dim3(B * nslab), dim3(64 * 16), 0, stream, dm, C, (int16_t*)T, (uint32_t*)(base + o_bnd), (int*)(base + o_flag), (int*)base, nslab);
        else hipLaunchKernelGGL(sgbm_forward_kernel<32>, dim3(B * nslab), dim3(32 * 16), 0, stream, dm, C, (int16_t*)T, (uint32_t*)(base + o_bnd), (int*)(base + o_flag), (int*)base, nslab);
    }
    if (unfused) { ProfScope p(stream, "sgbm_hsum_kernel"); hipLaunchKernelGGL(sgbm_hsum_kernel, dim3((dm.width1 + kHsSeg - 1) / kHsSeg, h, B), dim3(kHsBlock), 0, stream, dm, pre, hsum); }
    if (unfused) { ProfScope p(stream, "sgbm_vsum_kernel"); hipLaunchKernelGGL(sgbm_vsum_kernel, dim3((dm.width1 * 12 + 255) / 256, (h + kVsChunk - 1) / kVsChunk, B), dim3(256), 0, stream, dm, hsum, C); }
    { const int nv = dm.width1, nd = dm.width1 + h - 1;
      if (unfused) { ProfScope p(stream, "sgbm_path_kernel<0,1>"); hipLaunchKernelGGL((sgbm_path_kernel<0, 1, 0, 8>), dim3((nv + kPathLines - 1) / kPathLines, B), dim3(kPathBlock), 0, stream, dm, C, T, nv, (int4*)nullptr); }
      if (!fwd) { ProfScope p(stream, "sgbm_path_kernel<1,1>"); hipLaunchKernelGGL((sgbm_path_kernel<1, 1, 1, 8>), dim3((nd + kPathLines - 1) / kPathLines, B), dim3(kPathBlock), 0, stream, dm, C, T, nd, (int4*)nullptr); }
      if (!fwd) { ProfScope p(stream, "sgbm_path_kernel<-1,1>"); hipLaunchKernelGGL((sgbm_path_kernel<-1, 1, 1, 8>), dim3((nd + kPathLines - 1) / kPathLines, B), dim3(kPathBlock), 0, stream, dm, C, T, nd, (int4*)nullptr); }
      if (!fwd) { ProfScope p(stream, "sgbm_path_kernel<1,0>"); hipLaunchKernelGGL((sgbm_path_kernel<1, 0, 2, 16>), dim3((h + kPathLines - 1) / kPathLines, B), dim3(kPathBlock), 0, stream, dm, C, T, h, (int4*)nullptr); }
      if (B >= 4) { ProfScope p(stream, "sgbm_path_kernel<-1,0>"); hipLaunchKernelGGL((sgbm_path_kernel<-1, 0, 4, 16>), dim3((h + kPathLines - 1) / kPathLines, B), dim3(kPathBlock), 0, stream, dm, C, T, h, rec); }
      else {
        { ProfScope p(stream, "sgbm_path_kernel<-1,0>"); hipLaunchKernelGGL((sgbm_path_kernel<-1, 0, 3, 16>), dim3((h + kPathLines - 1) / kPathLines, B), dim3(kPathBlock), 0, stream, dm, C, T, h, (int4*)nullptr); }
        { ProfScope p(stream, "sgbm_wta_kernel"); hipLaunchKernelGGL(sgbm_wta_kernel, dim3((unsigned)(((size_t)h * dm.width1 + 15) / 16), B), dim3(256), 0, stream, dm, T, rec); }
      } }
    { ProfScope p(stream, "sgbm_lrcheck_kernel");
      hipLaunchKernelGGL(sgbm_lrcheck_kernel, dim3(h, B), dim3(kLrBlock), (size_t)(w + (w + 1) / 2) * sizeof(int), stream, dm, rec, d0); }
    if (d_disp_raw) VS_HIP(hipMemcpyAsync(d_disp_raw, d0, (size_t)B * npix * 2, hipMemcpyDeviceToDevice, stream));
    { ProfScope p(stream, "sgbm_median3_kernel"); hipLaunchKernelGGL(sgbm_median3_kernel, dim3((w + 255) / 256, h, B), dim3(256), 0, stream, w, h, d0, d1); }
    const int pblocks = (int)((npix + 255) / 256);
    const int newVal = -16, maxDiff = 16 * 32, maxSize = 100; // speckleWindowSize 100, speckleRange 32
    { ProfScope p(stream, "sgbm_ccl_kernels", 4);
      hipLaunchKernelGGL(sgbm_ccl_rows_kernel, dim3(h, B), dim3(kCclBlock), (size_t)2 * w * sizeof(int), stream, w, h, maxDiff, newVal, d1, par, cnt);
      hipLaunchKernelGGL(sgbm_ccl_union_kernel, dim3(pblocks, B), dim3(256), 0, stream, w, h, maxDiff, newVal, d1, par);
      hipLaunchKernelGGL(sgbm_ccl_count_kernel, dim3(pblocks, B), dim3(256), 0, stream, w, h, maxDiff, newVal, d1, par, cnt);
      hipLaunchKernelGGL(sgbm_ccl_apply_kernel, dim3(pblocks, B), dim3(256), 0, stream, w, h, newVal, maxSize, par, cnt, d1, d_disp_f32, d_disp_i16); }
    VS_HIP(hipGetLastError());
#ifdef VSLAM_SGBM_PROFILE
    { long long hdbg[8]; (void)hipStreamSynchronize(stream); (void)hipMemcpyFromSymbol(hdbg, HIP_SYMBOL(g_sgbm_dbg), sizeof(hdbg));
      fprintf(stderr, "[sgbm_down profile] cycles of one wave over %d rows: pixel cost %lld, barrier %lld, hsum+ring %lld, path+stores %lld\n", h, hdbg[0], hdbg[1], hdbg[2], hdbg[3]); }
#endif
    return VSLAM_OK;
}

} // namespace vslam
