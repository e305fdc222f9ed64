/*
 * oracle/orb.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see vo_oracle.h) for rows A1, A2, A3 of SURVEY.md
 * section 8: cv::ORB::detect, VO::adaptive_non_maximal_suppresion, cv::ORB::compute as called from
 * VO::feature_detection (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:70-157).
 *
 * PARITY UNPINNED for everything marked [UPSTREAM]: OpenCV is not in this image; these are restatements of
 * the published OpenCV 3.2 algorithms (modules/features2d/src/{orb,fast,fast_score,keypoint}.cpp,
 * modules/imgproc/src/{imgwarp,smooth,filter}.cpp, modules/core/src/mathfuncs_core.cpp), pinned only by the
 * known-answer tests in tests/.  One piece IS pinned against independent third-party code: the FAST-9/16 corner set equals
 * scikit-image 0.18.3's corner_fast(n=9, threshold=20) on three seeded images (tests/golden/skimage_fast9.npz,
 * generator tests/golden/make_skimage_fast9.py, check tests/test_oracle_orb.py), and the intensity-centroid angle agrees with
 * scikit-image's corner_orientations on the same 749-pixel circular patch within 0.01 degree at 200 seeded points
 * (tests/golden/skimage_ic_angle.npz, generator tests/golden/make_skimage_ic_angle.py).
 *
 * Deviation (documented in DESIGN.md): OpenCV's retainBest leaves keypoints in std::nth_element order, which
 * is implementation-defined.  This oracle defines the total order "level ascending, raster (y,x) within a
 * level" for detect output and a stable sort on response ties in ANMS.
 *
 * Compile with -ffp-contract=off: float expressions must round after every operation like the x86-64 SSE2
 * build of OpenCV does.
 */
#include "vo_oracle.h"
#include "orb_pattern.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cvRound: round half to even (lrint under the default rounding mode) */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) {
    int i = (int)v;
    return i - (v < (float)i);
}
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* ------------------------------------------------------------------ layout -------------------------- */

void vo_orb_layout_init(int w, int h, int nfeatures, vo_orb_layout* L) {
    /* ORB_Impl stores scaleFactor as double(1.2f) [UPSTREAM orb.cpp:  ORB::create(..., float scaleFactor = 1.2f)] */
    const double scale_factor = (double)1.2f;
    for (int l = 0; l < VO_ORB_NLEVELS; ++l) {
        float s = (float)pow(scale_factor, (double)l); /* getScale(level, firstLevel=0, scaleFactor) */
        L->scale[l] = s;
        L->w[l] = cv_round_d((double)((float)w / s));
        L->h[l] = cv_round_d((double)((float)h / s));
    }
    /* computeKeyPoints: per-level budgets */
    float factor = (float)(1.0 / scale_factor);
    float ndesired = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)VO_ORB_NLEVELS));
    int sum = 0;
    for (int l = 0; l < VO_ORB_NLEVELS - 1; ++l) {
        L->nfeat[l] = cv_round_d((double)ndesired);
        sum += L->nfeat[l];
        ndesired *= factor;
    }
    L->nfeat[VO_ORB_NLEVELS - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

/* ------------------------------------------------------------------ resize -------------------------- */

void vo_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    /* resize(): inv_scale = dsize/ssize; scale = 1./inv_scale (double) */
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
    int* rows[2];
    rows[0] = (int*)malloc(sizeof(int) * (size_t)dw);
    rows[1] = (int*)malloc(sizeof(int) * (size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = (short)cv_round_f((1.f - fx) * 2048.f); /* saturate_cast<short>(float) */
        ialpha[2 * dx + 1] = (short)cv_round_f(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= (float)sy;
        short b0 = (short)cv_round_f((1.f - fy) * 2048.f);
        short b1 = (short)cv_round_f(fy * 2048.f);
        for (int k = 0; k < 2; ++k) {
            int yy = sy + k;
            yy = yy < 0 ? 0 : (yy >= sh ? sh - 1 : yy);
            const uint8_t* S = src + (size_t)yy * sstride;
            for (int dx = 0; dx < dw; ++dx) {
                int sx = xofs[dx];
                int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
                rows[k][dx] = S[sx] * ialpha[2 * dx] + S[sx1] * ialpha[2 * dx + 1];
            }
        }
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx) {
            int v = (((b0 * (rows[0][dx] >> 4)) >> 16) + ((b1 * (rows[1][dx] >> 4)) >> 16) + 2) >> 2;
            if (vo_variant_flags & VO_VAR_RESIZE_ROUND) /* one rounding of the full 22-bit product instead of the library's >>4, >>16, +2 >>2 chain */
                v = (b0 * rows[0][dx] + b1 * rows[1][dx] + (1 << 21)) >> 22;
            D[dx] = (uint8_t)v; /* uchar(...) truncation; value is always within 0..255 */
        }
    }
    free(xofs); free(ialpha); free(rows[0]); free(rows[1]);
}

void vo_orb_build_pyramid(const uint8_t* img, int stride, const vo_orb_layout* L, int nlevels, uint8_t* const* dst) {
    for (int y = 0; y < L->h[0]; ++y) memcpy(dst[0] + (size_t)y * L->w[0], img + (size_t)y * stride, (size_t)L->w[0]);
    for (int l = 1; l < nlevels; ++l)
        vo_resize_linear_u8(dst[l - 1], L->w[l - 1], L->h[l - 1], L->w[l - 1], dst[l], L->w[l], L->h[l], L->w[l]);
}

/* ------------------------------------------------------------------ blur ---------------------------- */

void vo_gaussian_kernel7_fixed(int k[7]) {
    /* getGaussianKernel(7, 2, CV_32F): cf[i] = (float)exp(-x^2/(2 sigma^2)); sum in double; cf *= 1/sum;
     * then createSeparableLinearFilter: convertTo(CV_32S, 256) = cvRound(k*256). */
    const int n = 7;
    const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = cv_round_d((double)cf[i] * 256.0);
    }
}

void vo_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    int k[7];
    vo_gaussian_kernel7_fixed(k);
    int* tmp = (int*)malloc(sizeof(int) * (size_t)w * (size_t)h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int i = -3; i <= 3; ++i) s += k[i + 3] * S[reflect101(x + i, w)];
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; ++y) {
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int i = -3; i <= 3; ++i) s += k[i + 3] * tmp[(size_t)reflect101(y + i, h) * w + x];
            int v = (s + (1 << 15)) >> 16; /* FixedPtCastEx<int,uchar>(bits=16) */
            D[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------ FAST ---------------------------- */

static const int k_ring[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static void make_offsets(int pixel[25], int stride) {
    for (int k = 0; k < 16; ++k) pixel[k] = k_ring[k][0] + k_ring[k][1] * stride;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
}

static int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
    const int N = 25;
    int v = ptr[0];
    int d[25];
    for (int k = 0; k < N; ++k) d[k] = v - ptr[pixel[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int i = 4; i <= 8; ++i) if (d[k + i] < a) a = d[k + i];
        int m = a < d[k] ? a : d[k];
        if (m > a0) a0 = m;
        m = a < d[k + 9] ? a : d[k + 9];
        if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int i = 3; i <= 5; ++i) if (d[k + i] > b) b = d[k + i];
        if (b >= b0) continue;
        for (int i = 6; i <= 8; ++i) if (d[k + i] > b) b = d[k + i];
        int m = b > d[k] ? b : d[k];
        if (m < b0) b0 = m;
        m = b > d[k + 9] ? b : d[k + 9];
        if (m < b0) b0 = m;
    }
    return -b0 - 1;
}

int vo_fast_corner_score(const uint8_t* p, int stride, int threshold) {
    int pixel[25];
    make_offsets(pixel, stride);
    return corner_score16(p, pixel, threshold);
}

static int is_fast_corner(const uint8_t* ptr, const int pixel[25], int threshold) {
    const int K = 8, N = 25;
    int v = ptr[0];
    int vt = v - threshold, count = 0;
    for (int k = 0; k < N; ++k) {
        if (ptr[pixel[k]] < vt) { if (++count > K) return 1; }
        else count = 0;
    }
    vt = v + threshold; count = 0;
    for (int k = 0; k < N; ++k) {
        if (ptr[pixel[k]] > vt) { if (++count > K) return 1; }
        else count = 0;
    }
    return 0;
}

int vo_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax, vo_keypoint* out, int cap) {
    int pixel[25];
    make_offsets(pixel, stride);
    threshold = threshold < 0 ? 0 : (threshold > 255 ? 255 : threshold);
    /* full score map (0 where not a corner); OpenCV keeps a 3-row ring, the result is the same */
    uint8_t* score = (uint8_t*)calloc((size_t)w * (size_t)h, 1);
    uint8_t* corner = (uint8_t*)calloc((size_t)w * (size_t)h, 1);
    for (int i = 3; i < h - 3; ++i)
        for (int j = 3; j < w - 3; ++j) {
            const uint8_t* p = img + (size_t)i * stride + j;
            if (is_fast_corner(p, pixel, threshold)) {
                corner[(size_t)i * w + j] = 1;
                if (nonmax) score[(size_t)i * w + j] = (uint8_t)corner_score16(p, pixel, threshold);
            }
        }
    int n = 0, overflow = 0;
    /* OpenCV emits row i-1 while scanning row i, for i = 4 .. h-3  =>  rows 3 .. h-4 */
    for (int i = 3; i < h - 3; ++i)
        for (int j = 3; j < w - 3; ++j) {
            if (!corner[(size_t)i * w + j]) continue;
            int s = score[(size_t)i * w + j];
            if (nonmax) {
                const uint8_t* c = score + (size_t)i * w + j;
                if (!(s > c[1] && s > c[-1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] && s > c[w - 1] &&
                      s > c[w] && s > c[w + 1]))
                    continue;
            }
            if (n >= cap) { overflow = 1; continue; }
            vo_keypoint kp = {(float)j, (float)i, 7.f, -1.f, (float)s, 0, -1};
            out[n++] = kp;
        }
    free(score); free(corner);
    return overflow ? -1 : n;
}

/* ------------------------------------------------------------------ Harris / angle ------------------ */

float vo_harris_response(const uint8_t* img, int stride, int x0, int y0) {
    const int blockSize = 7, r = blockSize / 2, step = stride;
    const float harris_k = 0.04f;
    float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    const uint8_t* ptr0 = img + (size_t)(y0 - r) * step + (x0 - r);
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < blockSize; ++i)
        for (int j = 0; j < blockSize; ++j) {
            const uint8_t* ptr = ptr0 + i * step + j;
            int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
            int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    return ((float)a * (float)b - (float)c * (float)c - harris_k * ((float)a + (float)b) * ((float)a + (float)b)) * scale_sq_sq;
}

/* ------------------------------------------------------------------ [UPSTREAM] ambiguity switches ----
 * tests/oracle_sensitivity.py flips ONE reading of an OpenCV / g2o detail at a time and counts what changes downstream.
 * 0 (the default, and the only value the parity tests and the HIP path ever see) = the readings documented in DESIGN.md. */
unsigned vo_variant_flags = 0;
void vo_set_variant(unsigned flags) { vo_variant_flags = flags; }
unsigned vo_get_variant(void) { return vo_variant_flags; }

float vo_fast_atan2(float y, float x) {
    if (vo_variant_flags & VO_VAR_ATAN2F) { /* libm atan2f in degrees instead of cv::fastAtan2's 7th-order polynomial */
        float a = atan2f(y, x) * (float)(180.0 / 3.1415926535897932384626433832795);
        return a < 0 ? a + 360.f : a;
    }
    static const float rad2deg = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * rad2deg, p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg, p7 = -0.04432655554792128f * rad2deg;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static const int k_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

float vo_ic_angle(const uint8_t* img, int stride, int x, int y) {
    const int half_k = 15, step = stride;
    const uint8_t* center = img + (size_t)y * step + x;
    int m_01 = 0, m_10 = 0;
    for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
    for (int v = 1; v <= half_k; ++v) {
        int v_sum = 0, d = k_umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return vo_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------ retainBest ---------------------- */

static int cmp_float_desc(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x < y) - (x > y);
}

int vo_retain_best(vo_keypoint* kps, int n, int npoints) {
    if (npoints < 0 || n <= npoints) return n;
    if (npoints == 0) return 0;
    float* r = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) r[i] = kps[i].response;
    qsort(r, (size_t)n, sizeof(float), cmp_float_desc);
    float cut = r[npoints - 1];
    free(r);
    int m = 0;
    if (vo_variant_flags & VO_VAR_RETAIN_EXACT) { /* exactly npoints: ties at the cut are dropped in raster order (one possible nth_element outcome) */
        int above = 0;
        for (int i = 0; i < n; ++i) above += kps[i].response > cut;
        int ties_left = npoints - above;
        for (int i = 0; i < n; ++i) {
            if (kps[i].response > cut) kps[m++] = kps[i];
            else if (kps[i].response == cut && ties_left > 0) { kps[m++] = kps[i]; --ties_left; }
        }
        return m;
    }
    for (int i = 0; i < n; ++i)
        if (kps[i].response >= cut) kps[m++] = kps[i];
    return m;
}

/* ------------------------------------------------------------------ detect -------------------------- */

static uint8_t* alloc_levels(const vo_orb_layout* L, int nlevels, uint8_t** lv) {
    size_t total = 0;
    for (int l = 0; l < nlevels; ++l) total += (size_t)L->w[l] * L->h[l];
    uint8_t* buf = (uint8_t*)malloc(total);
    size_t off = 0;
    for (int l = 0; l < nlevels; ++l) { lv[l] = buf + off; off += (size_t)L->w[l] * L->h[l]; }
    return buf;
}

int vo_orb_detect(const uint8_t* img, int w, int h, int stride, int nfeatures, vo_keypoint* out, int cap) {
    const int edge = 31, fast_thr = 20;
    vo_orb_layout L;
    vo_orb_layout_init(w, h, nfeatures, &L);
    uint8_t* lv[VO_ORB_NLEVELS];
    uint8_t* buf = alloc_levels(&L, VO_ORB_NLEVELS, lv);
    vo_orb_build_pyramid(img, stride, &L, VO_ORB_NLEVELS, (uint8_t* const*)lv);
    int total = 0, overflow = 0;
    for (int l = 0; l < VO_ORB_NLEVELS; ++l) {
        int lw = L.w[l], lh = L.h[l];
        int kcap = lw * lh / 4 + 16;
        vo_keypoint* kps = (vo_keypoint*)malloc(sizeof(vo_keypoint) * (size_t)kcap);
        int n = vo_fast9_16(lv[l], lw, lh, lw, fast_thr, 1, kps, kcap);
        if (n < 0) { free(kps); free(buf); return -2; }
        /* KeyPointsFilter::runByImageBorder(keypoints, img.size(), edgeThreshold): keep edge <= x < w-edge */
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (kps[i].x >= (float)edge && kps[i].x < (float)(lw - edge) && kps[i].y >= (float)edge &&
                kps[i].y < (float)(lh - edge))
                kps[m++] = kps[i];
        n = m;
        /* retainBest(2 * featuresNum) on the FAST score, then Harris, then retainBest(featuresNum) */
        n = vo_retain_best(kps, n, 2 * L.nfeat[l]);
        for (int i = 0; i < n; ++i) {
            kps[i].octave = l;
            kps[i].size = 31.f * L.scale[l];
            kps[i].response = vo_harris_response(lv[l], lw, cv_round_f(kps[i].x), cv_round_f(kps[i].y));
        }
        n = vo_retain_best(kps, n, L.nfeat[l]);
        for (int i = 0; i < n; ++i) {
            kps[i].angle = vo_ic_angle(lv[l], lw, cv_round_f(kps[i].x), cv_round_f(kps[i].y));
            kps[i].x *= L.scale[l];
            kps[i].y *= L.scale[l];
            if (total < cap) out[total++] = kps[i];
            else overflow = 1;
        }
        free(kps);
    }
    free(buf);
    return overflow ? -1 : total;
}

/* ------------------------------------------------------------------ ANMS (reference's own code) ----- */

typedef struct { vo_keypoint kp; int idx; } kp_idx;
static int cmp_kp_resp_desc_stable(const void* a, const void* b) {
    const kp_idx* x = (const kp_idx*)a; const kp_idx* y = (const kp_idx*)b;
    if (x->kp.response > y->kp.response) return -1;
    if (x->kp.response < y->kp.response) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
static int cmp_double_desc(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x < y) - (x > y);
}

int vo_anms(vo_keypoint* kps, int n, int num) {
    /* visual_odometry.cpp:100 */
    if (n < num) return n;
    /* :106 sort by response, strongest first (stable on ties: defined order) */
    kp_idx* s = (kp_idx*)malloc(sizeof(kp_idx) * (size_t)n);
    for (int i = 0; i < n; ++i) { s[i].kp = kps[i]; s[i].idx = i; }
    qsort(s, (size_t)n, sizeof(kp_idx), cmp_kp_resp_desc_stable);
    double* rad = (double*)malloc(sizeof(double) * (size_t)n);
    double* rad_sorted = (double*)malloc(sizeof(double) * (size_t)n);
    const float c_robust = 1.11f; /* :120 `const float c_robust = 1.11;` */
    for (int i = 0; i < n; ++i) {
        const float response = s[i].kp.response * c_robust; /* :126 float product */
        double radius = DBL_MAX;                             /* :129 */
        for (int j = 0; j < i && s[j].kp.response > response; ++j) { /* :131 */
            float dx = s[i].kp.x - s[j].kp.x, dy = s[i].kp.y - s[j].kp.y; /* Point2f difference */
            double d = sqrt((double)dx * dx + (double)dy * dy);          /* cv::norm(Point2f) */
            if (d < radius) radius = d;
        }
        rad[i] = radius;
        rad_sorted[i] = radius;
    }
    qsort(rad_sorted, (size_t)n, sizeof(double), cmp_double_desc); /* :141 */
    const double final_radius = rad_sorted[num - 1];               /* :146 */
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (rad[i] >= final_radius) kps[m++] = s[i].kp; /* :147-153 */
    free(s); free(rad); free(rad_sorted);
    return m;
}

/* ------------------------------------------------------------------ compute (rBRIEF) ---------------- */

int vo_orb_compute(const uint8_t* img, int w, int h, int stride, vo_keypoint* kps, int n, uint8_t* desc) {
    const int edge = 31, B = VO_ORB_BORDER;
    if (n == 0) return 0;
    /* nLevels = max octave + 1; sortedByLevel check */
    int nlevels = 0, sorted = 1;
    for (int i = 0; i < n; ++i) {
        if (i > 0 && kps[i].octave < kps[i - 1].octave) sorted = 0;
        if (kps[i].octave > nlevels) nlevels = kps[i].octave;
    }
    nlevels++;
    if (nlevels > VO_ORB_NLEVELS) return -1;
    /* runByImageBorder(keypoints, image.size(), edgeThreshold) in level-0 coordinates */
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (kps[i].x >= (float)edge && kps[i].x < (float)(w - edge) && kps[i].y >= (float)edge && kps[i].y < (float)(h - edge))
            kps[m++] = kps[i];
    n = m;
    if (n == 0) return 0;
    if (!sorted) { /* regroup by level, stable */
        vo_keypoint* tmp = (vo_keypoint*)malloc(sizeof(vo_keypoint) * (size_t)n);
        int k = 0;
        for (int l = 0; l < nlevels; ++l)
            for (int i = 0; i < n; ++i)
                if (kps[i].octave == l) tmp[k++] = kps[i];
        memcpy(kps, tmp, sizeof(vo_keypoint) * (size_t)n);
        free(tmp);
    }
    vo_orb_layout L;
    vo_orb_layout_init(w, h, 500, &L);
    uint8_t* lv[VO_ORB_NLEVELS];
    uint8_t* buf = alloc_levels(&L, nlevels, lv);
    vo_orb_build_pyramid(img, stride, &L, nlevels, (uint8_t* const*)lv);
    /* per level: blurred image with a 32 px reflect-101 border (copyMakeBorder, then GaussianBlur on the ROI:
     * out-of-ROI taps read the border pixels, which equal reflect-101 of the unblurred level) */
    uint8_t* ext[VO_ORB_NLEVELS];
    int estride[VO_ORB_NLEVELS];
    for (int l = 0; l < nlevels; ++l) {
        int lw = L.w[l], lh = L.h[l], ew = lw + 2 * B, eh = lh + 2 * B;
        uint8_t* blur = (uint8_t*)malloc((size_t)lw * lh);
        vo_gaussian_blur7_u8(lv[l], lw, lh, lw, blur, lw);
        ext[l] = (uint8_t*)malloc((size_t)ew * eh);
        estride[l] = ew;
        /* inside: blurred; border: the UNBLURRED reflect-101 copy that copyMakeBorder wrote before the blur */
        for (int y = 0; y < eh; ++y)
            for (int x = 0; x < ew; ++x) {
                int sx = x - B, sy = y - B;
                if (sx >= 0 && sx < lw && sy >= 0 && sy < lh) ext[l][(size_t)y * ew + x] = blur[(size_t)sy * lw + sx];
                else ext[l][(size_t)y * ew + x] = lv[l][(size_t)reflect101(sy, lh) * lw + reflect101(sx, lw)];
            }
        free(blur);
    }
    for (int j = 0; j < n; ++j) {
        const vo_keypoint* kpt = &kps[j];
        int l = kpt->octave;
        float scale = 1.f / L.scale[l];
        float angle = kpt->angle;
        angle *= (float)(3.1415926535897932384626433832795 / 180.f);
        float a = (float)cos((double)angle), b = (float)sin((double)angle);
        if (vo_variant_flags & VO_VAR_COSF) { a = cosf(angle); b = sinf(angle); } /* the float overloads instead of the double ones */
        int cy = cv_round_f(kpt->y * scale) + B, cx = cv_round_f(kpt->x * scale) + B;
        const uint8_t* center = ext[l] + (size_t)cy * estride[l] + cx;
        const int step = estride[l];
        uint8_t* d = desc + (size_t)j * 32;
        const signed char* pat = vo_orb_pattern;
        for (int i = 0; i < 32; ++i, pat += 32) { /* 16 points = 8 tests x {x0,y0,x1,y1} per byte */
            int val = 0;
            for (int k = 0; k < 8; ++k) {
                const signed char* q = pat + 4 * k;
                float x0 = (float)q[0] * a - (float)q[1] * b, y0 = (float)q[0] * b + (float)q[1] * a;
                float x1 = (float)q[2] * a - (float)q[3] * b, y1 = (float)q[2] * b + (float)q[3] * a;
                int t0 = center[cv_round_f(y0) * step + cv_round_f(x0)];
                int t1 = center[cv_round_f(y1) * step + cv_round_f(x1)];
                val |= (t0 < t1) << k;
            }
            d[i] = (uint8_t)val;
        }
    }
    for (int l = 0; l < nlevels; ++l) free(ext[l]);
    free(buf);
    return n;
}

int vo_feature_detection(const uint8_t* img, int w, int h, int stride, int nfeatures, int anms_num,
                         vo_keypoint* kps, int cap, uint8_t* desc) {
    int n = vo_orb_detect(img, w, h, stride, nfeatures, kps, cap); /* visual_odometry.cpp:80 */
    if (n < 0) return n;
    n = vo_anms(kps, n, anms_num);                                 /* :82 */
    return vo_orb_compute(img, w, h, stride, kps, n, desc);        /* :85 */
}
