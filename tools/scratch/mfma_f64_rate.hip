// FP64 MFMA rate on gfx950: v_mfma_f64_16x16x4_f64 alone, v_fma_f64 alone, and the two CO-ISSUED from the two
// waves of every SIMD (a 512-thread workgroup per CU: waves 0-3 = one per SIMD run the MFMA stream, waves 4-7 the VALU stream).
// Question: does the pair sustain more f64 work per SIMD-cycle than the VALU alone (> 1.3x would justify moving the off-diagonal Schur
// accumulation of lm_window_kernel onto MFMA)?
//   hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

// mode bit 0: waves with (wave & 4) == 0 run MFMA; bit 1: the other waves run v_fma_f64.  A wave whose stream is off leaves at once.
__global__ __launch_bounds__(512) void mix(double* out, long long* cyc, int n, int mode, int all_same) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = all_same ? (mode & 1) : ((wave & 4) == 0);
    double r = 0.0;
    const long long t0 = clock64();
    if (mfma_wave && (mode & 1)) {
        v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001;
        for (int i = 0; i < n; ++i) { // four independent accumulator chains
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (!mfma_wave && (mode & 2)) {
        double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        const double b = 1.0000001, c = 1e-9;
        for (int i = 0; i < n; ++i) {
            a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
            a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

static float run(int mode, int all_same, int n, double* out, long long* cyc, long long* h) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<<<256, 512>>>(out, cyc, n, mode, all_same); hipDeviceSynchronize();
    hipEventRecord(e0); mix<<<256, 512>>>(out, cyc, n, mode, all_same); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, sizeof(long long) * 8, hipMemcpyDeviceToHost);
    return ms;
}

int main() {
    double* out; long long* cyc; long long h[8];
    hipMalloc(&out, sizeof(double) * 256 * 512); hipMalloc(&cyc, sizeof(long long) * 256 * 8);
    const int n = 20000;
    const double mfma_flop = 2048.0 * 4 * n, fma_flop = 128.0 * 8 * n; // per wave
    float ms;
    ms = run(1, 0, n, out, cyc, h);
    printf("MFMA f64 16x16x4, 1 wave/SIMD:            %.3f ms  %.1f cycles/MFMA (clock64)  %.1f TFLOP/s chip\n", ms, (double)h[0] / (4.0 * n), 256 * 4 * mfma_flop / (ms * 1e-3) / 1e12);
    ms = run(1, 1, n, out, cyc, h);
    printf("MFMA f64 16x16x4, 2 waves/SIMD:           %.3f ms  %.1f cycles/MFMA per SIMD        %.1f TFLOP/s chip\n", ms, (double)h[0] / (8.0 * n), 256 * 8 * mfma_flop / (ms * 1e-3) / 1e12);
    ms = run(2, 0, n, out, cyc, h);
    printf("v_fma_f64, 1 wave/SIMD:                   %.3f ms  %.1f cycles/FMA  (clock64)  %.1f TFLOP/s chip\n", ms, (double)h[4] / (8.0 * n), 256 * 4 * fma_flop / (ms * 1e-3) / 1e12);
    ms = run(2, 1, n, out, cyc, h);
    printf("v_fma_f64, 2 waves/SIMD:                  %.3f ms  %.1f cycles/FMA per SIMD         %.1f TFLOP/s chip\n", ms, (double)h[0] / (16.0 * n), 256 * 8 * fma_flop / (ms * 1e-3) / 1e12);
    ms = run(3, 0, n, out, cyc, h);
    printf("co-issue: MFMA wave + FMA wave per SIMD:  %.3f ms  MFMA wave %.1f cycles/MFMA, FMA wave %.1f cycles/FMA  %.1f TFLOP/s chip (%.1f MFMA + %.1f VALU)\n", ms,
           (double)h[0] / (4.0 * n), (double)h[4] / (8.0 * n), 256 * 4 * (mfma_flop + fma_flop) / (ms * 1e-3) / 1e12,
           256 * 4 * mfma_flop / (ms * 1e-3) / 1e12, 256 * 4 * fma_flop / (ms * 1e-3) / 1e12);
    printf("useful-work view for the Schur accumulation: one 6x6 += (6x2)(2x2)(2x6) hit needs ~185 f64 VALU ops per lane-hit (64 hits per wave-row: 185 x 4 = 740 cycles per 64 hits\n"
           "= 11.6 cycles per hit per SIMD); a 16x16x4 MFMA tile holds at most two 6x6 blocks with K = 2 each (12 x 12 x 4 of 16 x 16 x 4 = 56 %% of its MACs, and only the\n"
           "outer product, not the Jacobian front-end) -- compare its cycles per MFMA above with 2 x 11.6.\n");
    return 0;
}
