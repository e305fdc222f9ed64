// Issue cost of the integer / packed / media VALU instructions the ORB kernels are made of (gfx950): SIMD-cycles per wave-instruction for
// eight independent chains per lane at 4 and 8 waves per SIMD, relative to v_add_u32.  Build: hipcc --offload-arch=gfx950 -O3 int_rate.hip -o int_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN8(INSN)                                                                                                      \
    asm volatile(INSN(%0) "\n" INSN(%1) "\n" INSN(%2) "\n" INSN(%3) "\n" INSN(%4) "\n" INSN(%5) "\n" INSN(%6) "\n" INSN(%7) \
                 : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(k), "v"(k2))
#define I_ADD(x) "v_add_u32 " #x ", " #x ", %8"
#define I_DOT4(x) "v_dot4_u32_u8 " #x ", " #x ", %8, %9"
#define I_DOT2(x) "v_dot2_u32_u16 " #x ", " #x ", %8, %9"
#define I_PKMIN(x) "v_pk_min_i16 " #x ", " #x ", %8"
#define I_PKSUB(x) "v_pk_sub_i16 " #x ", " #x ", %8"
#define I_PERM(x) "v_perm_b32 " #x ", " #x ", %8, %9"
#define I_ALIGN(x) "v_alignbyte_b32 " #x ", " #x ", %8, 1"
#define I_MAD24(x) "v_mad_u32_u24 " #x ", " #x ", %8, %9"
#define I_MUL24(x) "v_mul_u32_u24 " #x ", " #x ", %8"
#define I_LSHLADD(x) "v_lshl_add_u32 " #x ", " #x ", 1, %8"
#define I_ANDOR(x) "v_and_or_b32 " #x ", " #x ", %8, %9"
#define I_MBCNT(x) "v_mbcnt_lo_u32_b32 " #x ", %8, " #x
#define I_MAX3(x) "v_max3_i32 " #x ", " #x ", %8, %9"
#define I_PKLSHR(x) "v_pk_lshrrev_b16 " #x ", 15, " #x
#define I_BFE(x) "v_bfe_u32 " #x ", " #x ", 3, 9"
#define I_MULLO(x) "v_mul_lo_u32 " #x ", " #x ", %8"
#define I_SAD(x) "v_sad_u8 " #x ", " #x ", %8, %9"
#define I_CMP(x) "v_cmp_gt_i32 vcc, " #x ", %8"
template <int KIND>
__global__ void rate(unsigned* out, int n) {
    unsigned u[8]; for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
    unsigned k = threadIdx.x * 3 + 1, k2 = threadIdx.x + 77;
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) CHAIN8(I_ADD); if (KIND == 1) CHAIN8(I_DOT4); if (KIND == 2) CHAIN8(I_DOT2); if (KIND == 3) CHAIN8(I_PKMIN);
        if (KIND == 4) CHAIN8(I_PKSUB); if (KIND == 5) CHAIN8(I_PERM); if (KIND == 6) CHAIN8(I_ALIGN); if (KIND == 7) CHAIN8(I_MAD24);
        if (KIND == 8) CHAIN8(I_MUL24); if (KIND == 9) CHAIN8(I_LSHLADD); if (KIND == 10) CHAIN8(I_ANDOR); if (KIND == 11) CHAIN8(I_MBCNT);
        if (KIND == 12) CHAIN8(I_MAX3); if (KIND == 13) CHAIN8(I_PKLSHR); if (KIND == 14) CHAIN8(I_BFE); if (KIND == 15) CHAIN8(I_MULLO);
        if (KIND == 16) CHAIN8(I_SAD);
        if (KIND == 17) asm volatile(I_CMP(%0) "\n" I_CMP(%1) "\n" I_CMP(%2) "\n" I_CMP(%3) "\n" I_CMP(%4) "\n" I_CMP(%5) "\n" I_CMP(%6) "\n" I_CMP(%7)
                 : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(k), "v"(k2) : "vcc");
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s += u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
static void run(const char* name) {
    unsigned* out; hipMalloc(&out, 4 * 1024 * 1024);
    const int n = 20000;
    printf("%-18s", name);
    for (int wps = 4; wps <= 8; wps *= 2) {
        const int blocks = wps == 8 ? 512 : 256; // 1024 threads = 4 waves per SIMD; two such workgroups per CU = 8
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        rate<KIND><<<blocks, 1024>>>(out, n); hipDeviceSynchronize();
        hipEventRecord(e0); rate<KIND><<<blocks, 1024>>>(out, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %d waves/SIMD: %.2f cycles/instr (2.4 GHz)", wps, ms * 1e-3 * 2.4e9 / ((double)n * 8 * wps));
    }
    printf("\n"); hipFree(out);
}
int main() {
    run<0>("v_add_u32"); run<1>("v_dot4_u32_u8"); run<2>("v_dot2_u32_u16"); run<3>("v_pk_min_i16"); run<4>("v_pk_sub_i16"); run<5>("v_perm_b32");
    run<6>("v_alignbyte_b32"); run<7>("v_mad_u32_u24"); run<8>("v_mul_u32_u24"); run<9>("v_lshl_add_u32"); run<10>("v_and_or_b32"); run<11>("v_mbcnt_lo");
    run<12>("v_max3_i32"); run<13>("v_pk_lshrrev_b16"); run<14>("v_bfe_u32"); run<15>("v_mul_lo_u32"); run<16>("v_sad_u8"); run<17>("v_cmp_gt_i32");
    return 0;
}
