// occupancy_probe: how many 1024-thread workgroups with a given dynamic LDS size / VGPR budget does one CU of the MI355X hold at a time?
// 512 workgroups that each spin ~1 ms: total time T (two per CU) or 2 T (one per CU).   hipcc --offload-arch=gfx950 -O3 occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int NT, int MINW>
__global__ __launch_bounds__(NT, MINW) void spin(long long cycles, int* sink) {
    extern __shared__ unsigned char smem[];
    const long long t0 = clock64();
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    while (clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
    if (smem[(threadIdx.x * 7) % NT] == 255 && cycles == 3) sink[0] = 1;
}
template <int NT, int MINW>
__global__ __launch_bounds__(NT, MINW) void spin_scratch(long long cycles, int* sink, int k) {
    extern __shared__ unsigned char smem[];
    volatile int priv[2]; // forces a private (scratch) segment, like a kernel with spills
    priv[0] = k; priv[1] = k + 1;
    const long long t0 = clock64();
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    while (clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
    if (smem[(threadIdx.x * 7) % NT] == 255 && cycles == 3) sink[0] = priv[k & 1];
}
template <int NT, int MINW, int TOPV>
__global__ __launch_bounds__(NT, MINW) void spin_vgpr(long long cycles, int* sink) {
    extern __shared__ unsigned char smem[];
    if (TOPV == 62) asm volatile("v_mov_b32 v62, 1" ::: "v62");
    if (TOPV == 55) asm volatile("v_mov_b32 v55, 1" ::: "v55");
    if (TOPV == 47) asm volatile("v_mov_b32 v47, 1" ::: "v47");
    const long long t0 = clock64();
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    while (clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
    if (smem[(threadIdx.x * 7) % NT] == 255 && cycles == 3) sink[0] = 1;
}
template <int NT, int MINW, int TOPV>
int run_vgpr(int wgs, size_t lds, int* sink) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_vgpr<NT, MINW, TOPV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((spin_vgpr<NT, MINW, TOPV>), dim3(wgs), dim3(NT), lds, 0, 1000, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((spin_vgpr<NT, MINW, TOPV>), dim3(wgs), dim3(NT), lds, 0, 200000, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("VGPR top v%d: threads %4d  workgroups %4d  LDS %6zu B : %.3f ms\n", TOPV, NT, wgs, lds, ms);
    return 0;
}
template <int NT, int MINW, int MODE>
__global__ __launch_bounds__(NT, MINW) void spin_all(long long cycles, int* sink, int k) {
    extern __shared__ unsigned char smem[];
    volatile int priv[2];
    if (MODE & 1) { priv[0] = k; priv[1] = k + 1; }
    if (MODE & 2) asm volatile("v_mov_b32 v62, 1" ::: "v62");
    if (MODE & 4) asm volatile("s_mov_b32 s80, 0" ::: "s80");
    if (MODE & 8) asm volatile("s_mov_b32 s95, 0" ::: "s95");
    if (MODE == 16) asm volatile("s_mov_b32 s63, 0" ::: "s63");
    if (MODE == 32) asm volatile("s_mov_b32 s71, 0" ::: "s71");
    if (MODE == 48) asm volatile("s_mov_b32 s73, 0" ::: "s73");
    if (MODE == 64) asm volatile("s_mov_b32 s75, 0" ::: "s75");
    if (MODE == 80) asm volatile("s_mov_b32 s77, 0" ::: "s77");
    const long long t0 = clock64();
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    while (clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(8); }
    if (smem[(threadIdx.x * 7) % NT] == 255 && cycles == 3) sink[0] = (MODE & 1) ? priv[k & 1] : 1;
}
template <int NT, int MINW, int MODE>
int run_all(int wgs, size_t lds, int* sink) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_all<NT, MINW, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((spin_all<NT, MINW, MODE>), dim3(wgs), dim3(NT), lds, 0, 1000, sink, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((spin_all<NT, MINW, MODE>), dim3(wgs), dim3(NT), lds, 0, 200000, sink, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("MODE %2d (1 scratch, 2 v62, 4 s80, 8 s95): threads %4d  workgroups %4d  LDS %6zu B : %.3f ms\n", MODE, NT, wgs, lds, ms);
    return 0;
}
template <int NT, int MINW>
int run_scratch(int wgs, size_t lds, int* sink) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_scratch<NT, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((spin_scratch<NT, MINW>), dim3(wgs), dim3(NT), lds, 0, 1000, sink, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((spin_scratch<NT, MINW>), dim3(wgs), dim3(NT), lds, 0, 200000, sink, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("WITH SCRATCH: threads %4d  workgroups %4d  LDS %6zu B : %.3f ms\n", NT, wgs, lds, ms);
    return 0;
}
template <int NT, int MINW>
int run(int wgs, size_t lds, int* sink) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin<NT, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((spin<NT, MINW>), dim3(wgs), dim3(NT), lds, 0, 1000, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((spin<NT, MINW>), dim3(wgs), dim3(NT), lds, 0, 200000, sink); // 200 k ticks of clock64
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("threads %4d  min waves/SIMD %d  workgroups %4d  LDS %6zu B : %.3f ms\n", NT, MINW, wgs, lds, ms);
    return 0;
}
int main() {
    int* sink; CK(hipMalloc(&sink, 64));
    run<1024, 8>(256, 79488, sink);
    for (size_t lds : {32768ul, 65536ul, 70000ul, 75000ul, 79488ul, 81920ul}) run<1024, 8>(512, lds, sink);
    run<1024, 8>(1024, 79488, sink);
    run<512, 8>(1024, 79488, sink);
    run<512, 8>(1024, 40000, sink);
    run<256, 8>(2048, 19040, sink);
    run_vgpr<1024, 8, 62>(512, 79488, sink);
    run_vgpr<1024, 8, 55>(512, 79488, sink);
    run_vgpr<1024, 8, 47>(512, 79488, sink);
    run_vgpr<256, 8, 62>(2048, 1024, sink);
    run_vgpr<256, 8, 55>(2048, 1024, sink);
    run_all<1024, 8, 16>(512, 79488, sink);
    run_all<1024, 8, 32>(512, 79488, sink);
    run_all<1024, 8, 48>(512, 79488, sink);
    run_all<1024, 8, 64>(512, 79488, sink);
    run_all<1024, 8, 80>(512, 79488, sink);
    run_all<256, 8, 4>(2048, 1024, sink);
    run_all<256, 8, 32>(2048, 1024, sink);
    run_all<1024, 8, 4>(512, 79488, sink);
    run_all<1024, 8, 8>(512, 79488, sink);
    run_all<1024, 8, 6>(512, 79488, sink);
    run_all<1024, 8, 7>(512, 79488, sink);
    run_all<1024, 8, 15>(512, 79488, sink);
    run_all<1024, 8, 7>(512, 1024, sink);
    run_scratch<1024, 8>(256, 79488, sink);
    run_scratch<1024, 8>(512, 79488, sink);
    run_scratch<1024, 8>(512, 32768, sink);
    run_scratch<256, 8>(2048, 19040, sink);
    return 0;
}
