// latency probes for gfx950 (single wave unless noted): cycles per dependent step, s_memtime units
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__global__ void probe(double* out, long long* cyc, int n, double a, double b) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double v = a + lane * 1e-12;
    long long t0, t1;
    // (0) dependent v_fma_f64
    t0 = clock64();
    for (int i = 0; i < n; ++i) v = fma(v, b, a);
    t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0;
    // (1) mul -> readlane -> fma chain (the backward substitution step)
    t0 = clock64();
    for (int i = 0; i < n; ++i) { const double x = readlane_f64(v * b, i & 63); v = lane == (i & 63) ? x : fma(-a, x, v); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[1] = t1 - t0;
    // (2) dependent LDS read (pointer chase through values)
    int idx = lane;
    t0 = clock64();
    for (int i = 0; i < n; ++i) { const double x = lds[idx & 1023]; idx = (int)(x * 3.0) + idx + 1; }
    t1 = clock64(); if (threadIdx.x == 0) cyc[2] = t1 - t0;
    v += idx;
    // (3) rsq + 2 newton steps, dependent
    t0 = clock64();
    for (int i = 0; i < n; ++i) { double r = __builtin_amdgcn_rsq(v); r = fma(r * 0.5, fma(-v * r, r, 1.0), r); r = fma(r * 0.5, fma(-v * r, r, 1.0), r); v = r + 2.0; }
    t1 = clock64(); if (threadIdx.x == 0) cyc[3] = t1 - t0;
    // (4) IEEE sqrt + division, dependent
    t0 = clock64();
    for (int i = 0; i < n; ++i) { v = 1.0 / sqrt(v) + 2.0; }
    t1 = clock64(); if (threadIdx.x == 0) cyc[4] = t1 - t0;
    // (5) __syncthreads round trips (all waves)
    t0 = clock64();
    for (int i = 0; i < n; ++i) { __syncthreads(); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // (6) LDS write -> barrier -> read by another wave -> dependent
    t0 = clock64();
    for (int i = 0; i < n; ++i) { lds[threadIdx.x] = v; __syncthreads(); v = lds[(threadIdx.x + 64) % blockDim.x] + 1.0; __syncthreads(); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[6] = t1 - t0;
    // (7) dependent v_add_f64
    t0 = clock64();
    for (int i = 0; i < n; ++i) v = v + b;
    t1 = clock64(); if (threadIdx.x == 0) cyc[7] = t1 - t0;
    // (8) independent fma x4 per step (throughput, one wave)
    double w0 = v, w1 = v + 1, w2 = v + 2, w3 = v + 3;
    t0 = clock64();
    for (int i = 0; i < n; ++i) { w0 = fma(w0, b, a); w1 = fma(w1, b, a); w2 = fma(w2, b, a); w3 = fma(w3, b, a); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[8] = t1 - t0;
    v = w0 + w1 + w2 + w3;
    // (9) wall clock vs clock64: wall_clock64 ticks over the same region
    long long w_0 = wall_clock64();
    t0 = clock64();
    for (int i = 0; i < n; ++i) v = fma(v, b, a);
    t1 = clock64();
    long long w_1 = wall_clock64();
    if (threadIdx.x == 0) { cyc[9] = t1 - t0; cyc[10] = w_1 - w_0; }
    out[threadIdx.x] = v;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 512); hipMalloc(&cyc, 8 * 16);
    const int n = 2000;
    for (int threads : {64, 512}) {
        hipMemset(cyc, 0, 128);
        hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, out, cyc, n, 1.0000001, 0.9999999);
        hipDeviceSynchronize();
        long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
        const char* names[] = {"dep fma_f64", "mul+readlane+fma step", "dep LDS read", "rsq+2 newton", "IEEE 1/sqrt", "__syncthreads", "LDS write+2 barriers+read", "dep add_f64", "4 indep fma"};
        printf("threads=%d\n", threads);
        for (int i = 0; i < 9; ++i) printf("  %-28s %8.1f ticks/step\n", names[i], (double)h[i] / n);
        printf("  clock64 %lld ticks vs wall_clock64 %lld ticks (100 MHz) -> clock64 = %.1f MHz\n", h[9], h[10], (double)h[9] / h[10] * 100.0);
    }
    return 0;
}
