#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// A: 32 x 32 int8 row-major (row i, k), B: 32 x 32 int8 stored column-major as Bt[j][k]
__global__ void k(const int8_t* A, const int8_t* Bt, int* D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a, b;
    memcpy(&a, A + r * 32 + h * 16, 16);
    memcpy(&b, Bt + r * 32 + h * 16, 16);
    v16i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D[l * 16 + v] = acc[v];
}
int main() {
    int8_t hA[1024], hB[1024]; int hD[1024];
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    for (int test = 0; test < 3; ++test) {
        for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 32; ++kk) {
            if (test == 0) { hA[i * 32 + kk] = kk == 0 ? i + 1 : 0; hB[i * 32 + kk] = kk == 0 ? 1 : 0; }
            if (test == 1) { hA[i * 32 + kk] = kk == 0 ? 1 : 0; hB[i * 32 + kk] = kk == 0 ? i + 1 : 0; }
            if (test == 2) { hA[i * 32 + kk] = (int8_t)((i * 7 + kk * 3) % 5 - 2); hB[i * 32 + kk] = (int8_t)((i * 5 + kk) % 7 - 3); }
        }
        hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        if (test < 2) {
            printf("test %d lane0:", test); for (int v = 0; v < 16; ++v) printf(" %d", hD[v]);
            printf("\n lane1:"); for (int v = 0; v < 16; ++v) printf(" %d", hD[16 + v]);
            printf("\n lane32:"); for (int v = 0; v < 16; ++v) printf(" %d", hD[32 * 16 + v]);
            printf("\n lane33:"); for (int v = 0; v < 16; ++v) printf(" %d", hD[33 * 16 + v]); printf("\n");
        } else {
            int bad = 0;
            for (int l = 0; l < 64; ++l) for (int v = 0; v < 16; ++v) {
                const int j = l & 31, i = 8 * (v / 4) + 4 * (l >> 5) + (v & 3);
                int ref = 0; for (int kk = 0; kk < 32; ++kk) ref += hA[i * 32 + kk] * hB[j * 32 + kk];
                bad += ref != hD[l * 16 + v];
            }
            printf("test 2 mismatches with assumed layout (col = l%%32, row = 8*(v/4) + 4*(l/32) + v%%4): %d\n", bad);
        }
    }
    return 0;
}
