#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k(int iters, int* out) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    v16i acc[CH];
    for (int c = 0; c < CH; ++c) for (int v = 0; v < 16; ++v) acc[c][v] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
    }
    int r = 0;
    for (int c = 0; c < CH; ++c) for (int v = 0; v < 16; ++v) r += acc[c][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int CH>
void run(int blocks, int threads, const char* name) {
    int* d; hipMalloc(&d, blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, 10, d);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, iters, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * (threads / 64) * iters * 8 * CH;
    printf("%s chains %d blocks %d threads %d: %.3f ms  %.2f POPS  (%.1f ns per MFMA per wave)\n", name, CH, blocks, threads, ms, mf * 32768 * 2 / ms / 1e12,
           ms * 1e6 / (iters * 8.0 * CH));
    hipFree(d);
}
int main() {
    run<1>(256, 256, "1 wave/SIMD"); run<2>(256, 256, "1 wave/SIMD"); run<4>(256, 256, "1 wave/SIMD");
    run<2>(512, 256, "2 waves/SIMD"); run<2>(1024, 256, "4 waves/SIMD");
    return 0;
}
