/*
 * oracle/sgbm.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see vo_oracle.h) for row A6 of SURVEY.md section 8:
 * VO::disparity_map (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:159-174):
 *   cv::StereoSGBM::create(0, 96, 9, 8*9*9, 32*9*9, 1, 63, 10, 100, 32)->compute(left, right, disp16);
 *   disp16.convertTo(disparity, CV_32F, 1/16)
 *
 * [UPSTREAM, PARITY UNPINNED] restated from the published OpenCV 3.2 algorithm (modules/calib3d/src/stereosgbm.cpp:
 * calcPixelCostBT, computeDisparitySGBM in MODE_SGBM = single pass, 5 directions; StereoSGBMImpl::compute =
 * computeDisparitySGBM + medianBlur 3x3 + filterSpeckles).  OpenCV is not in the image; pinned only by the tests
 * (synthetic stereo pairs with known disparity, structural properties).
 *
 * Semantics reproduced on purpose (they matter for bit-level agreement between this oracle and the HIP path):
 *   - Birchfield-Tomasi cost on the x-Sobel image clipped to [-63, 63] (+63) plus the raw image (>> 2); the raw rows'
 *     first and last pixels are replaced by 63 (the buffers are initialised with tab[0] and only 1..w-2 are written);
 *   - 9x9 box aggregation with the window clamped left/right/top, while the bottom SH2 rows REUSE the last full sum and
 *     column 0 keeps its first-row value (the in-place row update starts at the second column);
 *   - path costs: left->right plus the three paths from the previous row, borders L = 0 and min L = 0, then a
 *     right->left path on the same row; S = sat16(sat16(L0+L1+L2+L3) + L4);
 *   - winner = first minimum of S; uniqueness 10 %; parabola sub-pixel in 1/16 px; disp2 keeps the lowest-S winner per
 *     right pixel (ties: the larger x); left-right check with both roundings; invalid = -16.
 */
#include "vo_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

typedef short cost_t;
#define SGBM_MAX_COST SHRT_MAX

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline cost_t sat16(int v) { return (cost_t)(v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v)); }

/* rows of the pre-filtered (Sobel-x, clipped) and raw images: g[x], r[x] for one image row */
static void prefilter_row(const uint8_t* img, int w, int h, int stride, int y, int ftzero, uint8_t* g, uint8_t* r) {
    const uint8_t* row = img + (size_t)y * stride;
    const int n = y > 0 ? -stride : 0, s = y < h - 1 ? stride : 0;
    g[0] = g[w - 1] = (uint8_t)ftzero; /* tab[0] */
    r[0] = r[w - 1] = (uint8_t)ftzero; /* the raw channel's ends also keep tab[0] */
    for (int x = 1; x < w - 1; ++x) {
        int v = (row[x + 1] - row[x - 1]) * 2 + row[x + n + 1] - row[x + n - 1] + row[x + s + 1] - row[x + s - 1];
        g[x] = (uint8_t)(imin(imax(v, -ftzero), ftzero) + ftzero);
        r[x] = row[x];
    }
}

/* Birchfield-Tomasi pixel cost of one row: cost[(x - minX1) * D + d], x in [minX1, w), d in [0, D) */
static void pixel_cost_row(const uint8_t* g1, const uint8_t* r1, const uint8_t* g2, const uint8_t* r2, int w, int D, int minX1, cost_t* cost) {
    const int width1 = w - minX1;
    memset(cost, 0, sizeof(cost_t) * (size_t)width1 * D);
    uint8_t* lo = (uint8_t*)malloc((size_t)w);
    uint8_t* hi = (uint8_t*)malloc((size_t)w);
    for (int c = 0; c < 2; ++c) {
        const uint8_t* p1 = c == 0 ? g1 : r1;
        const uint8_t* p2 = c == 0 ? g2 : r2;
        const int diff_scale = c == 0 ? 0 : 2;
        /* half-pixel interpolated min / max of the right row (OpenCV stores the right row reversed; min/max of the two
         * half-pixel neighbours is symmetric, so plain indexing gives the same values) */
        for (int x = 0; x < w; ++x) {
            int v = p2[x];
            int vl = x > 0 ? (v + p2[x - 1]) / 2 : v;
            int vr = x < w - 1 ? (v + p2[x + 1]) / 2 : v;
            lo[x] = (uint8_t)imin(imin(vl, vr), v);
            hi[x] = (uint8_t)imax(imax(vl, vr), v);
        }
        for (int x = minX1; x < w; ++x) {
            int u = p1[x];
            int ul = x > 0 ? (u + p1[x - 1]) / 2 : u;
            int ur = x < w - 1 ? (u + p1[x + 1]) / 2 : u;
            int u0 = imin(imin(ul, ur), u), u1 = imax(imax(ul, ur), u);
            for (int d = 0; d < D; ++d) {
                int v = p2[x - d], v0 = lo[x - d], v1 = hi[x - d];
                int c0 = imax(imax(0, u - v1), v0 - u);
                int c1 = imax(imax(0, v - u1), u0 - v);
                cost_t* q = &cost[(size_t)(x - minX1) * D + d];
                *q = (cost_t)(*q + (imin(c0, c1) >> diff_scale));
            }
        }
    }
    free(lo); free(hi);
}

int vo_sgbm_compute(const uint8_t* left, const uint8_t* right, int w, int h, int stride, int num_disp, int block, int P1, int P2,
                    int disp12_max_diff, int pre_filter_cap, int uniqueness, int speckle_window, int speckle_range, int16_t* disp,
                    int16_t* disp_raw /* optional: before median/speckle */) {
    const int minD = 0, D = num_disp, maxD = minD + D;
    const int DISP_SHIFT = 4, DISP_SCALE = 16;
    const int ftzero = imax(pre_filter_cap, 15) | 1;
    const int SW2 = block / 2, SH2 = block / 2;
    const int minX1 = imax(maxD, 0), maxX1 = w + imin(minD, 0), width1 = maxX1 - minX1;
    const int INVALID = (minD - 1) * DISP_SCALE;
    if (D % 16 != 0 || w <= 0 || h <= 0) return -1;
    for (size_t i = 0; i < (size_t)w * h; ++i) disp[i] = (int16_t)INVALID;
    if (width1 <= 0) return 0;
    /* OpenCV 3.2's horizontal-sum initialisation reads pixel-cost columns 0..SW2 without clamping; with width1 <= SW2 that
     * is a read past its pixDiff buffer (undefined output).  Such widths are rejected here and by the HIP path. */
    if (width1 <= SW2) return -2;
    const size_t rowsz = (size_t)width1 * D;

    /* hsum ring of 2*SH2+2 rows, C (one row, updated in place), S, Lr (2 rows x 4 directions), minLr */
    const int NROWS = SH2 * 2 + 2;
    cost_t* hsum = (cost_t*)calloc(rowsz * NROWS, sizeof(cost_t));
    cost_t* pix = (cost_t*)malloc(rowsz * sizeof(cost_t));
    cost_t* C = (cost_t*)calloc(rowsz, sizeof(cost_t));
    cost_t* S = (cost_t*)calloc(rowsz, sizeof(cost_t));
    /* Lr[row][x+1][dir][d+1] with x in -1..width1 and d in -1..D */
    const int D2 = D + 2;
    const size_t lrrow = (size_t)(width1 + 2) * 4 * D2;
    cost_t* Lr[2] = {(cost_t*)calloc(lrrow, sizeof(cost_t)), (cost_t*)calloc(lrrow, sizeof(cost_t))};
    cost_t* mLr[2] = {(cost_t*)calloc((size_t)(width1 + 2) * 4, sizeof(cost_t)), (cost_t*)calloc((size_t)(width1 + 2) * 4, sizeof(cost_t))};
    uint8_t* g1 = (uint8_t*)malloc((size_t)w * 4); uint8_t *r1 = g1 + w, *g2 = r1 + w, *r2 = g2 + w;
    int16_t* disp2 = (int16_t*)malloc(sizeof(int16_t) * (size_t)w);
    cost_t* disp2cost = (cost_t*)malloc(sizeof(cost_t) * (size_t)w);
#define LR(r, x, dir, d) Lr[r][(((size_t)((x) + 1) * 4 + (dir)) * D2) + (d) + 1]
#define MLR(r, x, dir) mLr[r][(size_t)((x) + 1) * 4 + (dir)]

    for (int y = 0; y < h; ++y) {
        int16_t* dptr = disp + (size_t)y * w;
        /* ---- C(y): add the box row(s) entering the window */
        const int dy1 = y == 0 ? 0 : y + SH2, dy2 = y == 0 ? SH2 : dy1;
        for (int k = dy1; k <= dy2; ++k) {
            cost_t* hadd = hsum + (size_t)(imin(k, h - 1) % NROWS) * rowsz;
            if (k < h) {
                prefilter_row(left, w, h, stride, k, ftzero, g1, r1);
                prefilter_row(right, w, h, stride, k, ftzero, g2, r2);
                pixel_cost_row(g1, r1, g2, r2, w, D, minX1, pix);
                memset(hadd, 0, sizeof(cost_t) * D);
                for (int x = 0; x <= SW2; ++x) {
                    int scale = x == 0 ? SW2 + 1 : 1;
                    for (int d = 0; d < D; ++d) hadd[d] = (cost_t)(hadd[d] + pix[(size_t)x * D + d] * scale);
                }
                const cost_t* hsub = hsum + (size_t)(imax(y - SH2 - 1, 0) % NROWS) * rowsz;
                for (int x = 1; x < width1; ++x) {
                    const cost_t* padd = pix + (size_t)imin(x + SW2, width1 - 1) * D;
                    const cost_t* psub = pix + (size_t)imax(x - SW2 - 1, 0) * D;
                    for (int d = 0; d < D; ++d) {
                        int hv = hadd[(size_t)x * D + d] = (cost_t)(hadd[(size_t)(x - 1) * D + d] + padd[d] - psub[d]);
                        if (y > 0) C[(size_t)x * D + d] = (cost_t)(C[(size_t)x * D + d] + hv - hsub[(size_t)x * D + d]);
                    }
                }
                /* OpenCV 3.2 starts this loop at the second column: for y > 0 column 0 of C is never updated and keeps its
                 * y = 0 value (single-pass mode updates C in place).  Reproduced. */
            }
            if (y == 0) {
                int scale = k == 0 ? SH2 + 1 : 1;
                for (size_t i = 0; i < rowsz; ++i) C[i] = (cost_t)(C[i] + hadd[i] * scale);
            }
        }
        memset(S, 0, sizeof(cost_t) * rowsz);
        /* ---- clear the left / right borders of the current Lr row */
        for (int dir = 0; dir < 4; ++dir) {
            for (int d = -1; d <= D; ++d) { LR(0, -1, dir, d) = 0; LR(0, width1, dir, d) = 0; }
            MLR(0, -1, dir) = 0; MLR(0, width1, dir) = 0;
        }
        /* ---- four paths: 0 from (x-1, y), 1 from (x-1, y-1), 2 from (x, y-1), 3 from (x+1, y-1) */
        for (int x = 0; x < width1; ++x) {
            const int px[4] = {x - 1, x - 1, x, x + 1}, pr[4] = {0, 1, 1, 1};
            int delta[4], minL[4] = {SGBM_MAX_COST, SGBM_MAX_COST, SGBM_MAX_COST, SGBM_MAX_COST};
            for (int dir = 0; dir < 4; ++dir) {
                delta[dir] = MLR(pr[dir], px[dir], dir) + P2;
                LR(pr[dir], px[dir], dir, -1) = SGBM_MAX_COST; LR(pr[dir], px[dir], dir, D) = SGBM_MAX_COST;
            }
            for (int d = 0; d < D; ++d) {
                const int Cpd = C[(size_t)x * D + d];
                int sum = 0;
                for (int dir = 0; dir < 4; ++dir) {
                    const int a = LR(pr[dir], px[dir], dir, d), b = LR(pr[dir], px[dir], dir, d - 1) + P1, c = LR(pr[dir], px[dir], dir, d + 1) + P1;
                    const int L = Cpd + imin(a, imin(b, imin(c, delta[dir]))) - delta[dir];
                    LR(0, x, dir, d) = (cost_t)L;
                    minL[dir] = imin(minL[dir], L);
                    sum += L;
                }
                S[(size_t)x * D + d] = sat16(S[(size_t)x * D + d] + sum);
            }
            for (int dir = 0; dir < 4; ++dir) MLR(0, x, dir) = (cost_t)minL[dir];
        }
        /* ---- right->left path, winner, uniqueness, sub-pixel, disp2 */
        for (int x = 0; x < w; ++x) { dptr[x] = (int16_t)INVALID; disp2[x] = (int16_t)INVALID; disp2cost[x] = SGBM_MAX_COST; }
        for (int x = width1 - 1; x >= 0; --x) {
            cost_t* Sp = S + (size_t)x * D;
            int minS = SGBM_MAX_COST, bestDisp = -1, minL0 = SGBM_MAX_COST;
            const int delta0 = MLR(0, x + 1, 0) + P2;
            LR(0, x + 1, 0, -1) = SGBM_MAX_COST; LR(0, x + 1, 0, D) = SGBM_MAX_COST;
            for (int d = 0; d < D; ++d) {
                const int a = LR(0, x + 1, 0, d), b = LR(0, x + 1, 0, d - 1) + P1, c = LR(0, x + 1, 0, d + 1) + P1;
                const int L0 = C[(size_t)x * D + d] + imin(a, imin(b, imin(c, delta0))) - delta0;
                LR(0, x, 0, d) = (cost_t)L0;
                minL0 = imin(minL0, L0);
                const int Sval = Sp[d] = sat16(Sp[d] + L0);
                if (Sval < minS) { minS = Sval; bestDisp = d; }
            }
            MLR(0, x, 0) = (cost_t)minL0;
            int d;
            for (d = 0; d < D; ++d)
                if (Sp[d] * (100 - uniqueness) < minS * 100 && abs(bestDisp - d) > 1) break;
            if (d < D) continue;
            d = bestDisp;
            const int x2 = x + minX1 - d - minD;
            if (disp2cost[x2] > minS) { disp2cost[x2] = (cost_t)minS; disp2[x2] = (int16_t)(d + minD); }
            if (0 < d && d < D - 1) {
                const int denom2 = imax(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
                d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
            } else d *= DISP_SCALE;
            dptr[x + minX1] = (int16_t)(d + minD * DISP_SCALE);
        }
        /* ---- left-right consistency */
        for (int x = minX1; x < maxX1; ++x) {
            const int d1 = dptr[x];
            if (d1 == INVALID) continue;
            const int _d = d1 >> DISP_SHIFT, d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
            const int _x = x - _d, x_ = x - d_;
            if (0 <= _x && _x < w && disp2[_x] >= minD && abs(disp2[_x] - _d) > disp12_max_diff && 0 <= x_ && x_ < w && disp2[x_] >= minD &&
                abs(disp2[x_] - d_) > disp12_max_diff)
                dptr[x] = (int16_t)INVALID;
        }
        /* ---- rotate the Lr rows */
        { cost_t* t = Lr[0]; Lr[0] = Lr[1]; Lr[1] = t; t = mLr[0]; mLr[0] = mLr[1]; mLr[1] = t; }
    }
    if (disp_raw) memcpy(disp_raw, disp, sizeof(int16_t) * (size_t)w * h);

    /* ---- medianBlur(disp, disp, 3): 3x3 median, replicated borders */
    {
        int16_t* src = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);
        memcpy(src, disp, sizeof(int16_t) * (size_t)w * h);
        for (int i = 0; i < h; ++i) {
            const int16_t* row[3] = {src + (size_t)imax(i - 1, 0) * w, src + (size_t)i * w, src + (size_t)imin(i + 1, h - 1) * w};
            for (int j = 0; j < w; ++j) {
                const int j0 = j >= 1 ? j - 1 : j, j2 = j < w - 1 ? j + 1 : j;
                int16_t p[9] = {row[0][j0], row[0][j], row[0][j2], row[1][j0], row[1][j], row[1][j2], row[2][j0], row[2][j], row[2][j2]};
                for (int a = 1; a < 9; ++a) { int16_t v = p[a]; int b = a - 1; while (b >= 0 && p[b] > v) { p[b + 1] = p[b]; --b; } p[b + 1] = v; }
                disp[(size_t)i * w + j] = p[4];
            }
        }
        free(src);
    }
    /* ---- filterSpeckles(disp, INVALID, speckle_window, 16 * speckle_range): 4-connected regions of |diff| <= maxDiff
     * with at most speckle_window pixels become invalid */
    if (speckle_window > 0) {
        const int maxDiff = DISP_SCALE * speckle_range, newVal = INVALID;
        int* labels = (int*)calloc((size_t)w * h, sizeof(int));
        int* stack = (int*)malloc(sizeof(int) * (size_t)w * h);
        uint8_t* small = (uint8_t*)calloc((size_t)w * h + 1, 1);
        int cur = 0;
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                const size_t p0 = (size_t)i * w + j;
                if (disp[p0] == newVal) continue;
                if (labels[p0]) { if (small[labels[p0]]) disp[p0] = (int16_t)newVal; continue; }
                int top = 0, count = 0;
                ++cur;
                labels[p0] = cur; stack[top++] = (int)p0;
                while (top) {
                    const int p = stack[--top];
                    const int py = p / w, pxx = p % w;
                    const int dp = disp[p];
                    ++count;
                    const int nb[4] = {pxx < w - 1 ? p + 1 : -1, pxx > 0 ? p - 1 : -1, py < h - 1 ? p + w : -1, py > 0 ? p - w : -1};
                    for (int k = 0; k < 4; ++k) {
                        const int q = nb[k];
                        if (q < 0 || labels[q] || disp[q] == newVal) continue;
                        if (abs(dp - disp[q]) <= maxDiff) { labels[q] = cur; stack[top++] = q; }
                    }
                }
                if (count <= speckle_window) { small[cur] = 1; disp[p0] = (int16_t)newVal; }
            }
        free(labels); free(stack); free(small);
    }
#undef LR
#undef MLR
    free(hsum); free(pix); free(C); free(S); free(Lr[0]); free(Lr[1]); free(mLr[0]); free(mLr[1]); free(g1); free(disp2); free(disp2cost);
    return 0;
}

/* VO::disparity_map: the reference's fixed parameters + convertTo(CV_32F, 1/16) */
int vo_disparity_map(const uint8_t* left, const uint8_t* right, int w, int h, int stride, float* disparity) {
    int16_t* d16 = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);
    int rc = vo_sgbm_compute(left, right, w, h, stride, 96, 9, 8 * 9 * 9, 32 * 9 * 9, 1, 63, 10, 100, 32, d16, NULL);
    if (rc == 0)
        for (size_t i = 0; i < (size_t)w * h; ++i) disparity[i] = (float)((double)d16[i] * (1.0 / 16.0));
    free(d16);
    return rc;
}
