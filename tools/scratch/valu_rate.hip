// VALU issue rates on gfx950: wave-instructions per SIMD-cycle for independent v_fma_f64 / v_fma_f32 / v_add_u32 / v_mov streams at
// 1, 2, 4 and 8 waves per SIMD (one workgroup per CU; wall time from hipEvents, cycles from clock64 and from the 2.4 GHz nominal clock).
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void rate(double* out, long long* cyc, int n) {
    double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const double b = 1.0000001, c = 1e-9;
    const float bf = 1.0000001f, cf = 1e-9f;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) { a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c); }
        if (KIND == 1) { f0 = fmaf(f0, bf, cf); f1 = fmaf(f1, bf, cf); f2 = fmaf(f2, bf, cf); f3 = fmaf(f3, bf, cf); f4 = fmaf(f4, bf, cf); f5 = fmaf(f5, bf, cf); f6 = fmaf(f6, bf, cf); f7 = fmaf(f7, bf, cf); }
        if (KIND == 2) { u0 = u0 * 3u + 1u; u1 = u1 * 3u + 1u; u2 = u2 * 3u + 1u; u3 = u3 * 3u + 1u; u4 = u4 * 3u + 1u; u5 = u5 * 3u + 1u; u6 = u6 * 3u + 1u; u7 = u7 * 3u + 1u; } // v_mad_u32_u24 / mul_lo
        if (KIND == 3) { a0 = a0 + b; a1 = a1 + b; a2 = a2 + b; a3 = a3 + b; a4 = a4 + b; a5 = a5 + b; a6 = a6 + b; a7 = a7 + b; }
        if (KIND == 4) { u0 = (u0 ^ u1) + 1u; u1 = (u1 ^ u2) + 1u; u2 = (u2 ^ u3) + 1u; u3 = (u3 ^ u4) + 1u; u4 = (u4 ^ u5) + 1u; u5 = (u5 ^ u6) + 1u; u6 = (u6 ^ u7) + 1u; u7 = (u7 ^ u0) + 1u; } // 2 ops each
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7;
}
template <int KIND>
static void run(const char* name, int ops_per_iter) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * 256 * 2048); hipMalloc(&cyc, sizeof(long long) * 256);
    const int n = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int threads = 64 * 4 * wps; // waves per SIMD x 4 SIMDs
        if (threads > 1024 && wps > 4) { // 8 waves per SIMD: two workgroups of 1024 per CU
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            rate<KIND><<<512, 1024>>>(out, cyc, n); hipDeviceSynchronize();
            hipEventRecord(e0); rate<KIND><<<512, 1024>>>(out, cyc, n); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double winst = (double)n * ops_per_iter * 8.0; // wave-instructions per SIMD
            printf("%-10s %d waves/SIMD: %.3f ms  -> %.2f SIMD-cycles per wave-instruction at 2.4 GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / winst);
            continue;
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        rate<KIND><<<256, threads>>>(out, cyc, n); hipDeviceSynchronize();
        hipEventRecord(e0); rate<KIND><<<256, threads>>>(out, cyc, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const double winst = (double)n * ops_per_iter * wps;
        printf("%-10s %d waves/SIMD: %.3f ms  -> %.2f SIMD-cycles per wave-instruction at 2.4 GHz; clock64: %.2f ticks per wave-instruction per SIMD\n", name, wps, ms,
               ms * 1e-3 * 2.4e9 / winst, (double)h[0] / winst);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("fma_f64", 8); run<3>("add_f64", 8); run<1>("fma_f32", 8); run<2>("mad_u32", 8); run<4>("xor+add", 16);
    return 0;
}
