// Operand layout check of v_mfma_i32_16x16x32_i8 on gfx950 (8 bytes of A / B per lane): assumed A[i = l & 15][k = 8 (l >> 4) + byte], B[k = 8 (l >> 4) + byte][j = l & 15],
// D lane l, register v = D[i = 4 (l >> 4) + v][j = l & 15].   hipcc --offload-arch=gfx950 -O3 mfma16_layout.hip -o mfma16_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t* A /* 16 x 32 row-major */, const int8_t* Bt /* Bt[j][k] 16 x 32 */, int* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    long a, b;
    memcpy(&a, A + r * 32 + g * 8, 8);
    memcpy(&b, Bt + r * 32 + g * 8, 8);
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = acc[v];
}
int main() {
    int8_t hA[512], hB[512]; int hD[256];
    int8_t *dA, *dB; int* dD;
    (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 1024);
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 32; ++kk) { hA[i * 32 + kk] = (int8_t)((i * 7 + kk * 3) % 11 - 5); hB[i * 32 + kk] = (int8_t)((i * 5 + kk * 2 + 1) % 13 - 6); }
    (void)hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const int j = l & 15, i = 4 * (l >> 4) + v;
        int ref = 0; for (int kk = 0; kk < 32; ++kk) ref += hA[i * 32 + kk] * hB[j * 32 + kk];
        bad += ref != hD[l * 4 + v];
    }
    printf("v_mfma_i32_16x16x32_i8: mismatches with the assumed layout (D lane l reg v = D[4 (l >> 4) + v][l & 15]): %d\n", bad);
    return bad != 0;
}
