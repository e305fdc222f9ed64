// Same question as int_rate.hip with compiler-scheduled code (no inline asm): cycles per wave-instruction per SIMD, 8 chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void rate(unsigned* out, int n, unsigned k, unsigned k2) {
    unsigned u[8]; for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7 + i;
    k += threadIdx.x; k2 ^= threadIdx.x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) u[j] = u[j] + k;
            if (KIND == 1) u[j] = u[j] ^ k;
            if (KIND == 2) u[j] = (u[j] << 3) + k;                        // v_lshl_add_u32
            if (KIND == 3) u[j] = __builtin_amdgcn_perm(u[j], k, k2);
            if (KIND == 4) u[j] = __builtin_amdgcn_alignbyte(u[j], k, 1u);
            if (KIND == 5) u[j] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(s2, u[j]), __builtin_bit_cast(s2, k)));
            if (KIND == 6) u[j] = __builtin_amdgcn_udot4(u[j], k, k2, false);
            if (KIND == 7) u[j] = max((int)u[j], (int)k);
            if (KIND == 8) u[j] = __umul24(u[j], k) + k2;                 // v_mad_u32_u24
            if (KIND == 9) u[j] = (u[j] & k) | k2;                        // v_and_or_b32
            if (KIND == 10) u[j] = __builtin_bit_cast(unsigned, __builtin_bit_cast(s2, u[j]) - __builtin_bit_cast(s2, k));
            if (KIND == 11) u[j] = (u[j] >> 3) & 0x1FFu;                   // v_bfe_u32
            if (KIND == 12) u[j] = u[j] & k;
            if (KIND == 13) u[j] = u[j] << (k & 7);                         // v_lshlrev_b32
            if (KIND == 14) u[j] = min(u[j], k);
            if (KIND == 15) { float f = __builtin_bit_cast(float, u[j]); f = f * 1.0001f + 0.5f; u[j] = __builtin_bit_cast(unsigned, f); }
            if (KIND == 16) { float f = __builtin_bit_cast(float, u[j]); f = f + __builtin_bit_cast(float, k); u[j] = __builtin_bit_cast(unsigned, f); }
            if (KIND == 17) u[j] = (int)u[j] > (int)k ? k2 : u[j];         // v_cmp + v_cndmask
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s += u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
static void run(const char* name, int ops) {
    unsigned* out; (void)hipMalloc(&out, 4 * 1024 * 1024);
    const int n = 20000;
    printf("%-18s", name);
    for (int wps = 4; wps <= 8; wps *= 2) {
        const int blocks = wps == 8 ? 512 : 256;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        rate<KIND><<<blocks, 1024>>>(out, n, 3, 5); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); rate<KIND><<<blocks, 1024>>>(out, n, 3, 5); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %d waves/SIMD: %.2f cycles/instr", wps, ms * 1e-3 * 2.4e9 / ((double)n * 8 * ops * wps));
    }
    printf("\n"); (void)hipFree(out);
}
int main() {
    run<0>("add", 1); run<1>("xor", 1); run<2>("lshl_add", 1); run<3>("perm", 1); run<4>("alignbyte", 1); run<5>("pk_min_i16", 1); run<6>("udot4", 1);
    run<7>("max_i32", 1); run<8>("mad_u32_u24", 1); run<9>("and_or", 1); run<10>("pk_sub_i16", 1); run<11>("bfe", 1); run<12>("and", 1);
    run<13>("lshlrev", 1); run<14>("min_u32", 1); run<15>("fma_f32", 1); run<16>("add_f32", 1); run<17>("cmp+cndmask", 2);
    return 0;
}
