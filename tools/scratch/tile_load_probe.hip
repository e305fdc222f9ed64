// tile_load_probe: how fast can workgroups stage (CHUNKS x 16 B) x ROWS tiles of a batch of pitch-1280 images into LDS, by tile shape?
// (machine probe for the ORB pyramid / blur kernel's load phase: 0.73 ms per 1024 images = 2.2 TB/s with 272 B x 70 row tiles)
// build: hipcc --offload-arch=gfx950 -O3 -o tile_load_probe tile_load_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CHUNKS, int ROWS, int HALO_X, int HALO_Y, int NT>
__global__ __launch_bounds__(NT) void probe(const uint8_t* __restrict__ imgs, size_t img_bytes, int pitch, int W, int H, int tiles_x, uint32_t* __restrict__ sink) {
    constexpr int TW = CHUNKS * 16 - HALO_X, TH = ROWS - HALO_Y;
    const int b = blockIdx.y, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    __shared__ __attribute__((aligned(16))) uint8_t lds[CHUNKS * ROWS * 16];
    const uint8_t* src = imgs + (size_t)b * img_bytes;
    constexpr int kTotal = CHUNKS * ROWS, kIter = (kTotal + NT - 1) / NT;
    uint4 v[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int i = threadIdx.x + it * NT;
        const int r = i / CHUNKS, c = i - r * CHUNKS;
        const int y = min(y0 + r, H - 1), x = min(x0 + 16 * c, pitch - 16);
        v[it] = make_uint4(0, 0, 0, 0);
        if (i < kTotal) v[it] = *reinterpret_cast<const uint4*>(src + (size_t)y * pitch + x);
    }
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int i = threadIdx.x + it * NT;
        if (i < kTotal) *reinterpret_cast<uint4*>(lds + 16 * i) = v[it];
    }
    __syncthreads();
    if (threadIdx.x == 0 && lds[(b * 7 + tx) % (CHUNKS * ROWS * 16)] == 0xA7 && lds[3] == 0x11) sink[0] = 1; // (keeps the loads alive)
}

template <int CHUNKS, int ROWS, int HALO_X, int HALO_Y, int NT>
int run(const char* name, const uint8_t* d, size_t img_bytes, int pitch, int W, int H, int B, uint32_t* sink) {
    constexpr int TW = CHUNKS * 16 - HALO_X, TH = ROWS - HALO_Y;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<CHUNKS, ROWS, HALO_X, HALO_Y, NT>), dim3(tiles_x * tiles_y, B), dim3(NT), 0, 0, d, img_bytes, pitch, W, H, tiles_x, sink);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<CHUNKS, ROWS, HALO_X, HALO_Y, NT>), dim3(tiles_x * tiles_y, B), dim3(NT), 0, 0, d, img_bytes, pitch, W, H, tiles_x, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double useful = (double)B * W * H, staged = (double)B * tiles_x * tiles_y * CHUNKS * ROWS * 16;
    printf("%-28s tiles %3d x %3d  LDS %6d B  %.3f ms  image bytes %.2f TB/s  staged bytes %.2f TB/s\n", name, tiles_x, tiles_y, CHUNKS * ROWS * 16, ms, useful / ms * 1e-9, staged / ms * 1e-9);
    return 0;
}

int main() {
    const int W = 1241, H = 376, pitch = 1280, B = 1024;
    const size_t img_bytes = (size_t)pitch * H;
    uint8_t* d; uint32_t* sink;
    CK(hipMalloc(&d, img_bytes * B + 4096)); CK(hipMemset(d, 1, img_bytes * B + 4096)); CK(hipMalloc(&sink, 64));
    run<17, 70, 16, 6, 256>("272 B x 70 rows (current)", d, img_bytes, pitch, W, H, B, sink);
    run<17, 38, 16, 6, 256>("272 B x 38 rows", d, img_bytes, pitch, W, H, B, sink);
    run<33, 38, 16, 6, 256>("528 B x 38 rows", d, img_bytes, pitch, W, H, B, sink);
    run<65, 22, 16, 6, 256>("1040 B x 22 rows", d, img_bytes, pitch, W, H, B, sink);
    run<80, 22, 0, 6, 256>("1280 B x 22 rows (full rows)", d, img_bytes, pitch, W, H, B, sink);
    run<80, 14, 0, 6, 256>("1280 B x 14 rows (full rows)", d, img_bytes, pitch, W, H, B, sink);
    run<80, 38, 0, 6, 256>("1280 B x 38 rows (full rows)", d, img_bytes, pitch, W, H, B, sink);
    run<80, 38, 0, 6, 512>("1280 B x 38 rows, 512 thr", d, img_bytes, pitch, W, H, B, sink);
    run<80, 22, 0, 6, 512>("1280 B x 22 rows, 512 thr", d, img_bytes, pitch, W, H, B, sink);
    run<17, 70, 16, 6, 512>("272 B x 70 rows, 512 thr", d, img_bytes, pitch, W, H, B, sink);
    run<5, 40, 16, 8, 256>("80 B x 40 rows (FAST tile)", d, img_bytes, pitch, W, H, B, sink);
    return 0;
}
