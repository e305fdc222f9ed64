/*
 * oracle/match.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see vo_oracle.h) for row A5 of SURVEY.md section 8:
 * VO::feature_matching (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:219-251).
 *
 * PARITY UNPINNED for the [UPSTREAM] part: cv::BFMatcher(NORM_HAMMING, crossCheck=true)::match restated from
 * OpenCV 3.2 modules/features2d/src/matchers.cpp (BFMatcher::knnMatchImpl) and modules/core/src/
 * batch_distance.cpp (batchDistance with crosscheck).  Independent check: tests/test_oracle_match.py
 * re-derives the same rule in numpy.
 */
#include "vo_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

static inline int hamming256(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

int vo_bf_match_hamming_xcheck(const uint8_t* q, int nq, const uint8_t* t, int nt, vo_dmatch* out) {
    /* knnMatchImpl: empty query or empty train collection -> no matches */
    if (nq <= 0 || nt <= 0) return 0;
    int* dist = (int*)malloc(sizeof(int) * (size_t)nq);
    int* nidx = (int*)malloc(sizeof(int) * (size_t)nq);
    for (int i = 0; i < nq; ++i) { dist[i] = INT_MAX; nidx[i] = -1; }
    /* crosscheck: batchDistance(src2=train, src1=query, K=1): for every TRAIN row j the nearest QUERY row,
     * first minimum (strict '<', ascending index) ... */
    for (int j = 0; j < nt; ++j) {
        int best = INT_MAX, bi = -1;
        for (int i = 0; i < nq; ++i) {
            int d = hamming256(t + (size_t)j * 32, q + (size_t)i * 32);
            if (d < best) { best = d; bi = i; }
        }
        /* ... then, j ascending: if d < dist[idx] the query row idx takes train j */
        if (best < dist[bi]) { dist[bi] = best; nidx[bi] = j; }
    }
    int n = 0;
    for (int i = 0; i < nq; ++i)
        if (nidx[i] >= 0) {
            out[n].queryIdx = i; out[n].trainIdx = nidx[i]; out[n].imgIdx = 0; out[n].distance = (float)dist[i];
            ++n;
        }
    free(dist); free(nidx);
    return n;
}

int vo_feature_matching(const uint8_t* q, int nq, const uint8_t* t, int nt, double frame_gap, vo_dmatch* out) {
    int cap = nq > 0 ? nq : 1;
    vo_dmatch* m = (vo_dmatch*)malloc(sizeof(vo_dmatch) * (size_t)cap);
    int n = vo_bf_match_hamming_xcheck(q, nq, t, nt, m); /* visual_odometry.cpp:225 */
    /* :229-234 min distance.  The reference dereferences min_element of an empty vector (UB, quirk Q7);
     * here an empty match set simply yields no matches. */
    int k = 0;
    if (n > 0) {
        float dmin = m[0].distance;
        for (int i = 1; i < n; ++i) if (m[i].distance < dmin) dmin = m[i].distance;
        double a = 2.0 * (double)dmin, b = 30.0 * frame_gap; /* :242 */
        double thr = a > b ? a : b;
        for (int i = 0; i < n; ++i)
            if ((double)m[i].distance <= thr) out[k++] = m[i];
    }
    free(m);
    return k;
}
