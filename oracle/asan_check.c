/* asan_check.c -- TEST INFRASTRUCTURE: a small workload through every file of the CPU oracle, built with
 * -fsanitize=address,undefined (oracle/Makefile target `asan_check`, run by tests/test_oracle_sanitizers.py).
 * Inputs are generated here (LCG noise with blocky structure, a shifted copy as the right view, a synthetic PnP / BA problem);
 * the program prints a few checksums and exits 0 -- the sanitizers abort with a report on any out-of-bounds access, use of
 * uninitialised padding in arithmetic they track, signed overflow or misaligned access. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vo_oracle.h"

static unsigned lcg(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void) {
    const int w = 320, h = 160;
    unsigned seed = 12345u;
    uint8_t* L = (uint8_t*)malloc((size_t)w * h); uint8_t* R = (uint8_t*)malloc((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            unsigned s1 = (unsigned)((y / 11) * 131 + (x / 11) * 7919), s2 = (unsigned)((y / 4) * 977 + (x / 4) * 313);
            L[y * w + x] = (uint8_t)(60 + lcg(&s1) % 90 + lcg(&s2) % 50 + lcg(&seed) % 6);
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) R[y * w + x] = L[y * w + (x + 7 < w ? x + 7 : w - 1)];
    /* ORB detect + ANMS + rBRIEF on both views, matcher */
    const int cap = 8192;
    vo_keypoint* k0 = (vo_keypoint*)malloc(sizeof(vo_keypoint) * cap); vo_keypoint* k1 = (vo_keypoint*)malloc(sizeof(vo_keypoint) * cap);
    uint8_t* d0 = (uint8_t*)malloc((size_t)cap * 32); uint8_t* d1 = (uint8_t*)malloc((size_t)cap * 32);
    const int n0 = vo_feature_detection(L, w, h, w, 1000, 200, k0, cap, d0), n1 = vo_feature_detection(R, w, h, w, 1000, 200, k1, cap, d1);
    vo_dmatch* m = (vo_dmatch*)malloc(sizeof(vo_dmatch) * (size_t)(n0 > 0 ? n0 : 1));
    const int nm = n0 > 0 && n1 > 0 ? vo_feature_matching(d0, n0, d1, n1, 1.0, m) : 0;
    printf("orb %d %d keypoints, %d matches\n", n0, n1, nm);
    /* SGBM */
    float* disp = (float*)malloc(sizeof(float) * (size_t)w * h);
    const int rc = vo_disparity_map(L, R, w, h, w, disp);
    double acc = 0; int valid = 0;
    for (int i = 0; i < w * h; ++i) if (disp[i] >= 0) { acc += disp[i]; ++valid; }
    printf("sgbm rc %d valid %d mean %.3f\n", rc, valid, valid ? acc / valid : 0.0);
    /* PnP: motion-only + RANSAC on a synthetic problem */
    const double K[4] = {718.856, 718.856, 607.1928, 185.2157};
    const int M = 120;
    float* xyz = (float*)malloc(sizeof(float) * 3 * M); float* uv = (float*)malloc(sizeof(float) * 2 * M);
    for (int i = 0; i < M; ++i) {
        const double X = (double)(lcg(&seed) % 2000) / 100.0 - 10.0, Y = (double)(lcg(&seed) % 600) / 100.0 - 3.0, Z = 10.0 + (double)(lcg(&seed) % 3000) / 100.0;
        xyz[3 * i] = (float)X; xyz[3 * i + 1] = (float)Y; xyz[3 * i + 2] = (float)Z;
        const double Zc = Z - 0.8; /* true pose: translation (0.1, 0, -0.8) */
        uv[2 * i] = (float)(K[0] * (X + 0.1) / Zc + K[2] + (i % 9 == 0 ? 35.0 : 0.0)); uv[2 * i + 1] = (float)(K[1] * Y / Zc + K[3]);
    }
    double T[7] = {0, 0, 0, 1, 0, 0, 0};
    uint8_t* inl = (uint8_t*)malloc(M); int iters = 0;
    const int ni = vo_pnp_ransac(xyz, uv, M, K, T, 100, 4.0, 0.99, 10, inl, &iters);
    printf("ransac inliers %d iters %d t = %.4f %.4f %.4f\n", ni, iters, T[4], T[5], T[6]);
    free(L); free(R); free(k0); free(k1); free(d0); free(d1); free(m); free(disp); free(xyz); free(uv); free(inl);
    return 0;
}
