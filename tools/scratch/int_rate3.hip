// Does the VALU rate depend on how many VGPR operands an instruction reads?  (gfx950; 8 chains per lane, compiler-scheduled)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void rate(unsigned* out, int n, unsigned ks, unsigned ks2) {
    unsigned u[8]; for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7 + i;
    const unsigned kv = ks + threadIdx.x, kv2 = ks2 ^ threadIdx.x; // VGPR operands; ks, ks2 stay SGPRs
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = __builtin_bit_cast(float, u[j]);
            if (KIND == 0) u[j] = u[j] + ks;                                   // add v, s
            if (KIND == 1) u[j] = u[j] + kv;                                   // add v, v
            if (KIND == 2) u[j] = __builtin_bit_cast(unsigned, f * __builtin_bit_cast(float, ks) + __builtin_bit_cast(float, ks2));   // fma v, s, s
            if (KIND == 3) u[j] = __builtin_bit_cast(unsigned, f * __builtin_bit_cast(float, kv) + __builtin_bit_cast(float, ks2));   // fma v, v, s
            if (KIND == 4) u[j] = __builtin_bit_cast(unsigned, f * __builtin_bit_cast(float, kv) + __builtin_bit_cast(float, kv2));   // fma v, v, v
            if (KIND == 5) u[j] = __builtin_amdgcn_perm(u[j], ks, ks2);        // perm v, s, s
            if (KIND == 6) u[j] = __builtin_amdgcn_perm(u[j], kv, ks2);        // perm v, v, s
            if (KIND == 7) u[j] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(s2, u[j]), __builtin_bit_cast(s2, ks)));  // pk_min v, s
            if (KIND == 8) u[j] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(s2, u[j]), __builtin_bit_cast(s2, kv)));  // pk_min v, v
            if (KIND == 9) u[j] = __builtin_amdgcn_udot4(u[j], ks, ks2, false); // dot4 v, s, s
            if (KIND == 10) u[j] = __builtin_amdgcn_udot4(u[j], kv, ks2, false); // dot4 v, v, s
            if (KIND == 11) u[j] = __builtin_amdgcn_udot4(kv, ks, u[j], false); // dot4 v, s, v(acc)
            if (KIND == 12) u[j] = (u[j] << 3) + ks;                            // lshl_add v, 3, s
            if (KIND == 13) u[j] = max((int)u[j], (int)ks);                     // max v, s
            if (KIND == 14) u[j] = (u[j] & ks) | ks2;                           // and_or v, s, s
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s += u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
static void run(const char* name) {
    unsigned* out; (void)hipMalloc(&out, 4 * 1024 * 1024);
    const int n = 20000;
    printf("%-18s", name);
    for (int wps = 4; wps <= 8; wps *= 2) {
        const int blocks = wps == 8 ? 512 : 256;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        rate<KIND><<<blocks, 1024>>>(out, n, 3, 5); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); rate<KIND><<<blocks, 1024>>>(out, n, 3, 5); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %d waves/SIMD: %.2f cycles/instr", wps, ms * 1e-3 * 2.4e9 / ((double)n * 8 * wps));
    }
    printf("\n"); (void)hipFree(out);
}
int main() {
    run<0>("add v,s"); run<1>("add v,v"); run<2>("fma v,s,s"); run<3>("fma v,v,s"); run<4>("fma v,v,v"); run<5>("perm v,s,s"); run<6>("perm v,v,s");
    run<7>("pk_min v,s"); run<8>("pk_min v,v"); run<9>("dot4 v,s,s"); run<10>("dot4 v,v,s"); run<11>("dot4 v,s,v"); run<12>("lshl_add v,3,s");
    run<13>("max v,s"); run<14>("and_or v,s,s");
    return 0;
}
