// sgbm_kernels.hip -- stereo depth by semi-global block matching ("next #1" of 8f).
//
// Replaces cv::StereoSGBM::create(0, 96, 9, 8*9*9, 32*9*9, 1, 63, 10, 100, 32)->compute + convertTo(CV_32F, 1/16) inside
// VO::disparity_map (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:159-174).  The algorithm is OpenCV
// 3.2's single-pass MODE_SGBM (calcPixelCostBT + computeDisparitySGBM + medianBlur 3x3 + filterSpeckles), including its
// border behaviours (see oracle/sgbm.c for the list); all arithmetic is 8/16/32-bit integer, so the result is bit-exact
// against the CPU oracle.
//
// gfx950 mapping (batched over B stereo pairs; every stage integer, HBM/L2-bound -- the cost volume is 82.7 MB per pair):
//   sgbm_prefilter_kernel   x-Sobel clipped to [0,126] + raw rows                              elementwise
//   sgbm_pixcost_kernel     Birchfield-Tomasi cost per (y, x, d) -> u8 volume                   lanes along d (coalesced)
//   sgbm_hsum_kernel        9-tap horizontal box with clamped columns -> i16                    "
//   sgbm_vsum_kernel        9-tap vertical box (top clamped, bottom frozen, column 0 frozen)    "
//   sgbm_vertical_kernel    paths from the previous row (3 directions): the only row-sequential stage; one workgroup per
//                           pair walks the rows, 32-lane groups own one pixel (3 disparities per lane), the per-path
//                           minima are 5-step xor reductions; previous-row costs ping-pong through L2
//   sgbm_horizontal_kernel  one workgroup per image row: left->right scan, S1 = sat16(L0 + L1..3), right->left scan with
//                           winner-take-all, uniqueness, parabola sub-pixel, disp2 bookkeeping and the left-right check
//   sgbm_median3_kernel     3x3 median, replicated borders
//   sgbm_ccl_*              speckle filter as connected-component labelling (atomic union-find) + size threshold
//   sgbm_to_float_kernel    int16 / 16 -> f32 (invalid = -1)
#include "vslam_internal.h"

namespace vslam {

struct SgbmDims {
    int w, h, D, minX1, width1, P1, P2, SW2, SH2, uniq, disp12, ftzero, pitch;
    size_t img_bytes;
};

constexpr int kSgbmMaxCost = 32767;
constexpr int kTOffset = 8192; // L1+L2+L3 lies in [-7776, 46k]: stored as u16 with this offset

__device__ inline int sat16_dev(int v) { return min(max(v, -32768), 32767); }

// ------------------------------------------------------------------------------------------- prefilter
// pre[(b*2+side)][y][ch][x]: ch 0 = clipped x-Sobel + ftzero, ch 1 = raw (ends replaced by ftzero, like the reference buffers)
__global__ __launch_bounds__(256) void sgbm_prefilter_kernel(SgbmDims dm, const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                            uint8_t* __restrict__ pre) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, ii = blockIdx.z;
    if (x >= dm.w) return;
    const int b = ii >> 1, side = ii & 1;
    const uint8_t* img = (side ? right : left) + (size_t)b * dm.img_bytes;
    const uint8_t* row = img + (size_t)y * dm.pitch;
    uint8_t* out = pre + (((size_t)ii * dm.h + y) * 2) * dm.w;
    int g = dm.ftzero, r = dm.ftzero;
    if (x >= 1 && x < dm.w - 1) {
        const int n = y > 0 ? -dm.pitch : 0, s = y < dm.h - 1 ? dm.pitch : 0;
        const int v = (row[x + 1] - row[x - 1]) * 2 + row[x + n + 1] - row[x + n - 1] + row[x + s + 1] - row[x + s - 1];
        g = min(max(v, -dm.ftzero), dm.ftzero) + dm.ftzero;
        r = row[x];
    }
    out[x] = (uint8_t)g;
    out[dm.w + x] = (uint8_t)r;
}

// ------------------------------------------------------------------------------------------- pixel cost
__device__ inline void halfpix_minmax(const uint8_t* p, int x, int w, int& v, int& lo, int& hi) {
    v = p[x];
    const int vl = x > 0 ? (v + p[x - 1]) >> 1 : v;
    const int vr = x < w - 1 ? (v + p[x + 1]) >> 1 : v;
    lo = min(min(vl, vr), v);
    hi = max(max(vl, vr), v);
}

__global__ __launch_bounds__(256) void sgbm_pixcost_kernel(SgbmDims dm, const uint8_t* __restrict__ pre, uint8_t* __restrict__ pix) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x; // (j, d), d fastest
    if (idx >= dm.width1 * dm.D) return;
    const int j = idx / dm.D, d = idx - j * dm.D;
    const int x = dm.minX1 + j, xr = x - d;
    const uint8_t* L = pre + (((size_t)(2 * b) * dm.h + y) * 2) * dm.w;
    const uint8_t* R = pre + (((size_t)(2 * b + 1) * dm.h + y) * 2) * dm.w;
    int cost = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int u, u0, u1, v, v0, v1;
        halfpix_minmax(L + c * dm.w, x, dm.w, u, u0, u1);
        halfpix_minmax(R + c * dm.w, xr, dm.w, v, v0, v1);
        const int c0 = max(max(0, u - v1), v0 - u);
        const int c1 = max(max(0, v - u1), u0 - v);
        cost += min(c0, c1) >> (c == 0 ? 0 : 2);
    }
    pix[((size_t)b * dm.h + y) * dm.width1 * dm.D + idx] = (uint8_t)cost;
}

__global__ __launch_bounds__(256) void sgbm_hsum_kernel(SgbmDims dm, const uint8_t* __restrict__ pix, int16_t* __restrict__ hsum) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= dm.width1 * dm.D) return;
    const int j = idx / dm.D, d = idx - j * dm.D;
    const uint8_t* row = pix + ((size_t)b * dm.h + y) * dm.width1 * dm.D;
    int s = 0;
    for (int i = -dm.SW2; i <= dm.SW2; ++i) s += row[(size_t)min(max(j + i, 0), dm.width1 - 1) * dm.D + d];
    hsum[((size_t)b * dm.h + y) * dm.width1 * dm.D + idx] = (int16_t)s;
}

__global__ __launch_bounds__(256) void sgbm_vsum_kernel(SgbmDims dm, const int16_t* __restrict__ hsum, int16_t* __restrict__ C) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= dm.width1 * dm.D) return;
    const int j = idx / dm.D;
    // the reference never updates column 0 after the first row and stops sliding SH2 rows above the bottom
    const int yy = j == 0 ? 0 : min(y, dm.h - 1 - dm.SH2);
    const int16_t* base = hsum + (size_t)b * dm.h * dm.width1 * dm.D + idx;
    int s = 0;
    for (int k = yy - dm.SH2; k <= yy + dm.SH2; ++k) s += base[(size_t)min(max(k, 0), dm.h - 1) * dm.width1 * dm.D];
    C[((size_t)b * dm.h + y) * dm.width1 * dm.D + idx] = (int16_t)s;
}

// ------------------------------------------------------------------------------------------- vertical paths
// Lrow[buf][dir][x + 1][d]  (x = -1 and x = width1 are the zero borders), mrow[buf][dir][x + 1]
constexpr int kVBlock = 1024;

__global__ __launch_bounds__(kVBlock) void sgbm_vertical_kernel(SgbmDims dm, const int16_t* __restrict__ C, uint16_t* __restrict__ T,
                                                              int16_t* Lrow, int16_t* mrow) {
    const int b = blockIdx.x;
    const int D = dm.D, W1 = dm.width1;
    const size_t lstride = (size_t)3 * (W1 + 2) * D; // one buffer
    int16_t* Lb = Lrow + (size_t)b * 2 * lstride;
    int16_t* mb = mrow + (size_t)b * 2 * 3 * (W1 + 2);
    // previous row of row 0 = zeros; borders of both buffers = zeros
    for (size_t i = threadIdx.x; i < 2 * lstride; i += kVBlock) Lb[i] = 0;
    for (int i = threadIdx.x; i < 2 * 3 * (W1 + 2); i += kVBlock) mb[i] = 0;
    __syncthreads();
    const int grp = threadIdx.x >> 5, gl = threadIdx.x & 31, ngrp = kVBlock >> 5; // 32-lane group per pixel, lane = 3 disparities
    const int d0 = 3 * gl;
    const bool live = d0 < D; // D <= 96
    for (int y = 0; y < dm.h; ++y) {
        const int cur = y & 1, prv = cur ^ 1;
        const int16_t* Lp = Lb + (size_t)prv * lstride;
        int16_t* Lc = Lb + (size_t)cur * lstride;
        const int16_t* mp = mb + (size_t)prv * 3 * (W1 + 2);
        int16_t* mc = mb + (size_t)cur * 3 * (W1 + 2);
        const int16_t* Crow = C + ((size_t)b * dm.h + y) * W1 * D;
        uint16_t* Trow = T + ((size_t)b * dm.h + y) * W1 * D;
        for (int x0 = 0; x0 < W1; x0 += ngrp) { // uniform trip count (group shuffles)
            const int x = x0 + grp;
            const bool on = x < W1 && live;
            int sum[3] = {0, 0, 0};
            int c[3] = {0, 0, 0};
            if (on) { c[0] = Crow[(size_t)x * D + d0]; c[1] = Crow[(size_t)x * D + d0 + 1]; c[2] = Crow[(size_t)x * D + d0 + 2]; }
#pragma unroll
            for (int dir = 0; dir < 3; ++dir) {
                const int xp = x + dir - 1; // dir 0: from (x-1, y-1), 1: (x, y-1), 2: (x+1, y-1)  [reference directions 1, 2, 3]
                int lm = kSgbmMaxCost, l0 = 0, l1 = 0, l2 = 0, lp = kSgbmMaxCost, delta = 0;
                if (on) {
                    const int16_t* q = Lp + ((size_t)dir * (W1 + 2) + (xp + 1)) * D;
                    delta = mp[dir * (W1 + 2) + xp + 1] + dm.P2;
                    l0 = q[d0]; l1 = q[d0 + 1]; l2 = q[d0 + 2];
                    lm = d0 > 0 ? (int)q[d0 - 1] : kSgbmMaxCost;
                    lp = d0 + 3 < D ? (int)q[d0 + 3] : kSgbmMaxCost;
                }
                const int L0 = c[0] + min(min(l0, lm + dm.P1), min(l1 + dm.P1, delta)) - delta;
                const int L1 = c[1] + min(min(l1, l0 + dm.P1), min(l2 + dm.P1, delta)) - delta;
                const int L2 = c[2] + min(min(l2, l1 + dm.P1), min(lp + dm.P1, delta)) - delta;
                int mn = on ? min(L0, min(L1, L2)) : kSgbmMaxCost;
                for (int o = 16; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
                if (on) {
                    int16_t* w = Lc + ((size_t)dir * (W1 + 2) + (x + 1)) * D;
                    w[d0] = (int16_t)L0; w[d0 + 1] = (int16_t)L1; w[d0 + 2] = (int16_t)L2;
                    if (gl == 0) mc[dir * (W1 + 2) + x + 1] = (int16_t)mn;
                    sum[0] += L0; sum[1] += L1; sum[2] += L2;
                }
            }
            if (on) {
                Trow[(size_t)x * D + d0] = (uint16_t)(sum[0] + kTOffset);
                Trow[(size_t)x * D + d0 + 1] = (uint16_t)(sum[1] + kTOffset);
                Trow[(size_t)x * D + d0 + 2] = (uint16_t)(sum[2] + kTOffset);
            }
        }
        __syncthreads(); // the row is complete (global writes of this workgroup) before the next row reads it
    }
}

// ------------------------------------------------------------------------------------------- horizontal paths + winner
constexpr int kHBlock = 128; // lanes 0..95 own one disparity each

__device__ inline int block_min2(int v, int* s_red) { // min over the 2 waves of the block, all threads get the result
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return min(s_red[0], s_red[1]);
}

__global__ __launch_bounds__(kHBlock) void sgbm_horizontal_kernel(SgbmDims dm, const int16_t* __restrict__ C, uint16_t* __restrict__ T,
                                                                 int16_t* __restrict__ disp) {
    const int y = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const int D = dm.D, W1 = dm.width1;
    const bool live = d < D;
    const int16_t* Crow = C + ((size_t)b * dm.h + y) * W1 * D;
    uint16_t* Trow = T + ((size_t)b * dm.h + y) * W1 * D; // in: L1+L2+L3 (+offset); out: S1 (as int16 bits)
    __shared__ int16_t Lp[kHBlock + 2]; // Lp[d + 1], sentinels at d = -1 and d = D
    __shared__ int16_t Srow[kHBlock];
    __shared__ int s_red[2];
    __shared__ int s_flag;
    extern __shared__ int16_t dsm[]; // disp2[w], disp2cost[w], disp1[w]
    int16_t* disp2 = dsm; int16_t* disp2cost = dsm + dm.w; int16_t* disp1 = dsm + 2 * dm.w;
    const int INVALID = -16;
    for (int i = threadIdx.x; i < dm.w; i += kHBlock) { disp2[i] = (int16_t)INVALID; disp2cost[i] = (int16_t)kSgbmMaxCost; disp1[i] = (int16_t)INVALID; }
    // ---- left -> right
    Lp[threadIdx.x + 1] = 0;
    if (threadIdx.x == 0) { Lp[0] = (int16_t)kSgbmMaxCost; Lp[D + 1] = (int16_t)kSgbmMaxCost; }
    int minPrev = 0; // border: min L = 0
    __syncthreads();
    for (int x = 0; x < W1; ++x) {
        const int delta = minPrev + dm.P2;
        int L = kSgbmMaxCost;
        if (live) {
            const int a = Lp[d + 1], bm = Lp[d] + dm.P1, bp = Lp[d + 2] + dm.P1;
            L = Crow[(size_t)x * D + d] + min(min(a, bm), min(bp, delta)) - delta;
        }
        minPrev = block_min2(L, s_red); // (barrier inside: every lane has read Lp)
        if (live) {
            Lp[d + 1] = (int16_t)L;
            const int t = (int)Trow[(size_t)x * D + d] - kTOffset;
            Trow[(size_t)x * D + d] = (uint16_t)(int16_t)sat16_dev(L + t); // S1
        }
        __syncthreads();
    }
    // ---- right -> left + winner-take-all
    if (live) Lp[d + 1] = 0;
    minPrev = 0;
    __syncthreads();
    for (int x = W1 - 1; x >= 0; --x) {
        const int delta = minPrev + dm.P2;
        int L = kSgbmMaxCost, S = kSgbmMaxCost;
        if (live) {
            const int a = Lp[d + 1], bm = Lp[d] + dm.P1, bp = Lp[d + 2] + dm.P1;
            L = Crow[(size_t)x * D + d] + min(min(a, bm), min(bp, delta)) - delta;
            S = sat16_dev((int)(int16_t)Trow[(size_t)x * D + d] + L);
        }
        minPrev = block_min2(L, s_red);
        if (live) { Lp[d + 1] = (int16_t)L; Srow[d] = (int16_t)S; }
        if (threadIdx.x == 0) s_flag = 0;
        // first minimum of S over d: key = (S + 32768) << 8 | d
        const int key = live ? (((S + 32768) << 8) | d) : 0x7FFFFFFF;
        const int best = block_min2(key, s_red); // (barriers inside also publish Lp, Srow, s_flag)
        const int minS = (best >> 8) - 32768, bestD = best & 0xFF;
        if (live && S * (100 - dm.uniq) < minS * 100 && abs(bestD - d) > 1) s_flag = 1;
        __syncthreads();
        if (threadIdx.x == 0 && !s_flag) {
            int dd = bestD;
            const int x2 = x + dm.minX1 - dd;
            if (disp2cost[x2] > minS) { disp2cost[x2] = (int16_t)minS; disp2[x2] = (int16_t)dd; }
            if (0 < dd && dd < D - 1) {
                const int sm = Srow[dd - 1], sp = Srow[dd + 1], s0 = Srow[dd];
                const int denom2 = max(sm + sp - 2 * s0, 1);
                dd = dd * 16 + ((sm - sp) * 16 + denom2) / (denom2 * 2);
            } else dd *= 16;
            disp1[x + dm.minX1] = (int16_t)dd;
        }
        __syncthreads();
    }
    // ---- left-right consistency
    int16_t* out = disp + ((size_t)b * dm.h + y) * dm.w;
    for (int x = threadIdx.x; x < dm.w; x += kHBlock) {
        int d1 = disp1[x];
        if (x >= dm.minX1 && d1 != INVALID) {
            const int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
            const int _x = x - _d, x_ = x - d_;
            if (0 <= _x && _x < dm.w && disp2[_x] >= 0 && abs(disp2[_x] - _d) > dm.disp12 && 0 <= x_ && x_ < dm.w && disp2[x_] >= 0 &&
                abs(disp2[x_] - d_) > dm.disp12)
                d1 = INVALID;
        }
        out[x] = (int16_t)d1;
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

__global__ __launch_bounds__(256) void sgbm_ccl_init_kernel(int n, int* __restrict__ parent, int* __restrict__ count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)n) { parent[i] = (int)(i % 0x7FFFFFFF); count[i] = 0; }
}
// parent indices are image-local (p = y*w + x) inside each image's slice
__global__ __launch_bounds__(256) void sgbm_ccl_local_init_kernel(int w, int h, int* __restrict__ parent, int* __restrict__ count) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= w * h) return;
    parent[(size_t)b * w * h + p] = p;
    count[(size_t)b * w * h + p] = 0;
}
__global__ __launch_bounds__(256) void sgbm_ccl_union_kernel(int w, int h, int maxDiff, int newVal, const int16_t* __restrict__ disp,
                                                            int* __restrict__ parent) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= w * h) return;
    const int16_t* dsp = disp + (size_t)b * w * h;
    int* par = parent + (size_t)b * w * h;
    const int dp = dsp[p];
    if (dp == newVal) return;
    const int x = p % w, y = p / w;
    if (x + 1 < w) { const int dq = dsp[p + 1]; if (dq != newVal && abs(dp - dq) <= maxDiff) ccl_union(par, p, p + 1); }
    if (y + 1 < h) { const int dq = dsp[p + w]; if (dq != newVal && abs(dp - dq) <= maxDiff) ccl_union(par, p, p + w); }
}
__global__ __launch_bounds__(256) void sgbm_ccl_count_kernel(int w, int h, int newVal, const int16_t* __restrict__ disp, int* __restrict__ parent,
                                                            int* __restrict__ count) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= w * h) return;
    if (disp[(size_t)b * w * h + p] == newVal) return;
    int* par = parent + (size_t)b * w * h;
    const int r = ccl_find(par, p);
    par[p] = r; // flatten (monotone: roots only ever decrease, r is final after the union kernel completed)
    atomicAdd(&count[(size_t)b * w * h + r], 1);
}
__global__ __launch_bounds__(256) void sgbm_ccl_apply_kernel(int w, int h, int newVal, int maxSize, const int* __restrict__ parent,
                                                            const int* __restrict__ count, const int16_t* __restrict__ disp,
                                                            float* __restrict__ out_f32, int16_t* __restrict__ out_i16) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= w * h) return;
    const size_t g = (size_t)b * w * h + p;
    int d = disp[g];
    if (d != newVal && count[(size_t)b * w * h + parent[g]] <= maxSize) d = newVal;
    if (out_i16) out_i16[g] = (int16_t)d;
    if (out_f32) out_f32[g] = (float)d * 0.0625f; // convertTo(CV_32F, 1/16): exact
}

// ------------------------------------------------------------------------------------------- host driver
int launch_sgbm(const uint8_t* d_left, const uint8_t* d_right, size_t img_bytes, int pitch, int w, int h, int B, float* d_disp_f32,
                int16_t* d_disp_i16, int16_t* d_disp_raw, uint8_t** scratch, size_t* scratch_bytes, size_t* dev_bytes, hipStream_t stream) {
    if (B <= 0) return VSLAM_OK;
    SgbmDims dm;
    dm.w = w; dm.h = h; dm.D = 96; dm.minX1 = 96; dm.width1 = w - 96; dm.P1 = 8 * 9 * 9; dm.P2 = 32 * 9 * 9; dm.SW2 = 4; dm.SH2 = 4; dm.uniq = 10;
    dm.disp12 = 1; dm.ftzero = 63; dm.pitch = pitch; dm.img_bytes = img_bytes; // visual_odometry.cpp:163-164
    if (dm.width1 <= 0 || h <= 2 * dm.SH2 + 1) { set_error("image too small for 96 disparities / 9x9 blocks"); return VSLAM_ERR_ARG; }
    const size_t vol = (size_t)h * dm.width1 * dm.D, npix = (size_t)w * h;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t need = 0;
    const size_t o_pre = need; need += al((size_t)2 * B * h * 2 * w);
    const size_t o_pix = need; need += al((size_t)B * vol);
    const size_t o_hs = need; need += al((size_t)B * vol * 2);
    const size_t o_C = need; need += al((size_t)B * vol * 2);
    const size_t o_T = o_hs; // hsum is dead once C exists: T reuses its storage
    const size_t o_L = need; need += al((size_t)B * 2 * 3 * (dm.width1 + 2) * dm.D * 2);
    const size_t o_m = need; need += al((size_t)B * 2 * 3 * (dm.width1 + 2) * 2);
    const size_t o_d0 = need; need += al((size_t)B * npix * 2);
    const size_t o_d1 = need; need += al((size_t)B * npix * 2);
    const size_t o_par = need; need += al((size_t)B * npix * 4);
    const size_t o_cnt = need; need += al((size_t)B * npix * 4);
    if (*scratch_bytes < need) {
        VS_HIP(hipStreamSynchronize(stream));
        if (*scratch) { (void)hipFree(*scratch); *dev_bytes -= *scratch_bytes; }
        *scratch = nullptr; *scratch_bytes = 0;
        if (hipMalloc((void**)scratch, need) != hipSuccess) { *scratch = nullptr; set_error("SGBM scratch hipMalloc(%zu) failed", need); return VSLAM_ERR_HIP; }
        *scratch_bytes = need; *dev_bytes += need;
    }
    uint8_t* base = *scratch;
    uint8_t* pre = base + o_pre; uint8_t* pix = base + o_pix; int16_t* hsum = (int16_t*)(base + o_hs); int16_t* C = (int16_t*)(base + o_C);
    uint16_t* T = (uint16_t*)(base + o_T); int16_t* Lrow = (int16_t*)(base + o_L); int16_t* mrow = (int16_t*)(base + o_m);
    int16_t* d0 = (int16_t*)(base + o_d0); int16_t* d1 = (int16_t*)(base + o_d1); int* par = (int*)(base + o_par); int* cnt = (int*)(base + o_cnt);
    const int vblocks = (dm.width1 * dm.D + 255) / 256;
    { ProfScope p(stream, "sgbm_prefilter_kernel"); hipLaunchKernelGGL(sgbm_prefilter_kernel, dim3((w + 255) / 256, h, 2 * B), dim3(256), 0, stream, dm, d_left, d_right, pre); }
    { ProfScope p(stream, "sgbm_pixcost_kernel"); hipLaunchKernelGGL(sgbm_pixcost_kernel, dim3(vblocks, h, B), dim3(256), 0, stream, dm, pre, pix); }
    { ProfScope p(stream, "sgbm_hsum_kernel"); hipLaunchKernelGGL(sgbm_hsum_kernel, dim3(vblocks, h, B), dim3(256), 0, stream, dm, pix, hsum); }
    { ProfScope p(stream, "sgbm_vsum_kernel"); hipLaunchKernelGGL(sgbm_vsum_kernel, dim3(vblocks, h, B), dim3(256), 0, stream, dm, hsum, C); }
    { ProfScope p(stream, "sgbm_vertical_kernel"); hipLaunchKernelGGL(sgbm_vertical_kernel, dim3(B), dim3(kVBlock), 0, stream, dm, C, T, Lrow, mrow); }
    { ProfScope p(stream, "sgbm_horizontal_kernel"); hipLaunchKernelGGL(sgbm_horizontal_kernel, dim3(h, B), dim3(kHBlock), (size_t)3 * w * sizeof(int16_t), stream, dm, C, T, d0); }
    if (d_disp_raw) VS_HIP(hipMemcpyAsync(d_disp_raw, d0, (size_t)B * npix * 2, hipMemcpyDeviceToDevice, stream));
    { ProfScope p(stream, "sgbm_median3_kernel"); hipLaunchKernelGGL(sgbm_median3_kernel, dim3((w + 255) / 256, h, B), dim3(256), 0, stream, w, h, d0, d1); }
    const int pblocks = (int)((npix + 255) / 256);
    const int newVal = -16, maxDiff = 16 * 32, maxSize = 100; // speckleWindowSize 100, speckleRange 32
    { ProfScope p(stream, "sgbm_ccl_kernels", 4);
      hipLaunchKernelGGL(sgbm_ccl_local_init_kernel, dim3(pblocks, B), dim3(256), 0, stream, w, h, par, cnt);
      hipLaunchKernelGGL(sgbm_ccl_union_kernel, dim3(pblocks, B), dim3(256), 0, stream, w, h, maxDiff, newVal, d1, par);
      hipLaunchKernelGGL(sgbm_ccl_count_kernel, dim3(pblocks, B), dim3(256), 0, stream, w, h, newVal, d1, par, cnt);
      hipLaunchKernelGGL(sgbm_ccl_apply_kernel, dim3(pblocks, B), dim3(256), 0, stream, w, h, newVal, maxSize, par, cnt, d1, d_disp_f32, d_disp_i16); }
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

} // namespace vslam
