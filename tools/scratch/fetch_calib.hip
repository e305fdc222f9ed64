// FETCH_SIZE / WRITE_SIZE calibration on gfx950 for the access widths the BA kernel uses (MI355X_MICROARCH.md: only the wide
// coalesced 16 B/lane read is calibrated there: FETCH_SIZE reports 1/2 of its bytes).  Every kernel touches `bytes` bytes of a
// buffer larger than the Infinity Cache exactly once; run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE) and compare.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void read8(const double* __restrict__ p, size_t n, double* out) { // 8 B per lane, coalesced
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456) out[0] = s;
}
__global__ void read16(const double2* __restrict__ p, size_t n, double* out) { // 16 B per lane, coalesced
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
__global__ void read32(const double4* __restrict__ p, size_t n, double* out) { // 32-B records, coalesced (two 16-B loads per lane)
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456) out[0] = s;
}
__global__ void gather32(const double4* __restrict__ p, size_t n, double* out) { // 32-B records, permuted (every record once)
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = (i * 2654435761ull) % n; // n is a power of two, the multiplier is odd: a permutation
        const double4 v = p[j]; s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456) out[0] = s;
}
__global__ void write8(double* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void write32(double4* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double4(i, 1, 2, 3);
}
int main() {
    const size_t bytes = (size_t)1 << 31; // 2 GiB: 8x the Infinity Cache
    void* buf; double* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 8);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const dim3 g(256 * 16), b(256);
    hipLaunchKernelGGL(read8, g, b, 0, 0, (const double*)buf, bytes / 8, out);
    hipLaunchKernelGGL(read16, g, b, 0, 0, (const double2*)buf, bytes / 16, out);
    hipLaunchKernelGGL(read32, g, b, 0, 0, (const double4*)buf, bytes / 32, out);
    hipLaunchKernelGGL(gather32, g, b, 0, 0, (const double4*)buf, bytes / 32, out);
    hipLaunchKernelGGL(write8, g, b, 0, 0, (double*)buf, bytes / 8);
    hipLaunchKernelGGL(write32, g, b, 0, 0, (double4*)buf, bytes / 32);
    hipDeviceSynchronize();
    printf("each kernel touches %zu bytes (%.1f KiB)\n", bytes, bytes / 1024.0);
    return 0;
}
