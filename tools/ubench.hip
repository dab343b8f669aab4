// Tuning only: VALU/SALU issue-rate microbenchmark on gfx950 (8 waves per SIMD, every CU busy).
// Prints cycles per wave-instruction per SIMD for the instructions the scan kernel is built from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, unsigned long long *cyc)
{
    uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x9e3779b9u, c = a + 77, d = b + 13, e = a ^ b, f = c ^ d, g = 5, h = 9;
    uint64_t q = ((uint64_t)a << 32) | b, r = ((uint64_t)c << 32) | d, q2 = q ^ r;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 1) { REP64(asm volatile("v_alignbyte_b32 %0, %0, %1, 1\n v_alignbyte_b32 %2, %2, %3, 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 2) { REP64(asm volatile("v_min3_u32 %0, %0, %1, %2\n v_min3_u32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 3) { REP64(asm volatile("v_cmp_eq_u32 s[20:21], %0, %1\n v_cmp_eq_u32 s[22:23], %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s20", "s21", "s22", "s23");) }
        if (OP == 4) { REP64(asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0\n v_qsad_pk_u16_u8 %3, %1, %4, %3" : "+v"(q), "+v"(r), "+v"(a), "+v"(q2), "+v"(e));) }
        if (OP == 5) { REP64(asm volatile("v_pk_min_u16 %0, %0, %1\n v_pk_min_u16 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 6) { REP64(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xc8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 7) { REP64(asm volatile("s_or_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[24:25], s[24:25], s[22:23]" ::: "s20", "s21", "s22", "s23", "s24", "s25", "scc");) }
        if (OP == 8) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n s_or_b64 s[20:21], s[20:21], s[22:23]" : "+v"(a), "+v"(b) :: "s20", "s21", "s22", "s23", "scc");) }
        if (OP == 20) { REP64(asm volatile("v_cmp_eq_u32 s[20:21], %0, %1\n s_or_b64 s[22:23], s[22:23], s[24:25]" : "+v"(a), "+v"(b) :: "s20", "s21", "s22", "s23", "s24", "s25", "scc");) }
        if (OP == 21) { REP64(asm volatile("s_add_u32 s20, s20, s21\n s_add_u32 s22, s22, s23" ::: "s20", "s21", "s22", "s23", "scc");) }
        if (OP == 22) { REP64(asm volatile("s_nop 4\n s_nop 4" :::);) }
        if (OP == 23) { REP64(asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1" : "+v"(a), "+v"(b) :: "s20", "s21");) }
        if (OP == 24) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n s_add_u32 s20, s20, s21\n s_add_u32 s22, s22, s23\n s_add_u32 s24, s24, s23" : "+v"(a), "+v"(b) :: "s20", "s21", "s22", "s23", "s24", "scc");) }
        if (OP == 25) { REP64(asm volatile("s_cmp_eq_u32 s20, 77\n s_cbranch_scc1 1f\n s_nop 0\n1:\n s_cmp_lg_u32 s20, 77\n s_cbranch_scc1 2f\n s_nop 0\n2:" ::: "s20", "scc");) }
        if (OP == 26) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: );) }
        if (OP == 27) { REP64(asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_or3_b32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 28) { REP64(asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 29) { REP64(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
        if (OP == 9) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 10) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 12) { REP64(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 13) { REP64(asm volatile("v_sad_u8 %0, %0, %1, %2\n v_sad_u8 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 14) { REP64(asm volatile("v_msad_u8 %0, %0, %1, %2\n v_msad_u8 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 15) { REP64(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
        if (OP == 16) { REP64(asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 17) { REP64(asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_sub_u16 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 18) { REP64(asm volatile("v_cmp_eq_u16 s[20:21], %0, %1\n v_cmp_eq_u16 s[22:23], %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s20", "s21", "s22", "s23");) }
        if (OP == 19) { REP64(asm volatile("v_lshrrev_b64 %0, 8, %0\n v_lshrrev_b64 %1, 8, %1" : "+v"(q), "+v"(r));) }
        if (OP == 30) { REP64(asm volatile("v_dot4_u32_u8 %0, %1, %2, %0\n v_dot4_u32_u8 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 31) { REP64(asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n v_lshrrev_b32_sdwa %3, %4, %5 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 32) { REP64(asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_lshrrev_b32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 33) { REP64(asm volatile("v_alignbit_b32 %0, %1, %0, 1\n v_alignbit_b32 %2, %3, %2, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 34) { REP64(asm volatile("v_lshlrev_b32 %0, 3, %1\n v_and_b32 %2, 0xf8f8f8f8, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 35) { REP64(asm volatile("v_lshl_or_b32 %0, %1, 8, %0\n v_lshl_or_b32 %2, %3, 8, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 36) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 37) { REP64(asm volatile("v_mad_u32_u24 %0, %1, %2, %0\n v_mad_u32_u24 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 38) { REP64(asm volatile("v_bfe_u32 %0, %1, 5, 10\n v_bfe_u32 %2, %3, 15, 10" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 39) { REP64(asm volatile("v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %2, %3, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 40) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n s_nop 0" : "+v"(a), "+v"(b));) }
        if (OP == 41) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));) }
        if (OP == 42) { REP64(asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %3, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (uint32_t)q + (uint32_t)r + (uint32_t)q2;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP> void run(const char *name, uint32_t *out, unsigned long long *cyc, int blocks)
{
    const int iters = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 64 * 2;                 // instructions per wave
    const int waves_per_simd = blocks / 256;                   // 1 wave of each block per SIMD, blocks/256 blocks per CU
    printf("%-28s %6.2f cyc/instr/wave (s_memtime)  -> %5.2f cyc/instr/SIMD at %d waves/SIMD ; wall %.3f ms -> %.2f ns/instr/SIMD\n", name,
           c / n, c / n / waves_per_simd, waves_per_simd, ms, ms * 1e6 / (n * waves_per_simd));
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    int bpc = argc > 1 ? atoi(argv[1]) : 8;
    int op = argc > 2 ? atoi(argv[2]) : -1;
    int blocks = 256 * bpc;
    uint32_t *out; unsigned long long *cyc;
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    printf("op %d start\n", op);
#define RUN(N, NAME) if (op == N || op < 0) run<N>(NAME, out, cyc, blocks);
    RUN(0, "v_xor_b32") RUN(9, "v_add_u32") RUN(10, "v_fma_f32") RUN(1, "v_alignbyte_b32") RUN(2, "v_min3_u32")
    RUN(3, "v_cmp_eq_u32 -> sgpr") RUN(18, "v_cmp_eq_u16 -> sgpr") RUN(15, "v_cmp vcc + v_addc") RUN(4, "v_qsad_pk_u16_u8")
    RUN(5, "v_pk_min_u16") RUN(17, "v_pk_add/sub_u16") RUN(6, "v_bitop3_b32") RUN(12, "v_perm_b32") RUN(13, "v_sad_u8")
    RUN(14, "v_msad_u8") RUN(16, "v_mov_b32_dpp wave_shl") RUN(19, "v_lshrrev_b64") RUN(7, "s_or_b64") RUN(8, "v_xor + s_or interleaved") RUN(20, "v_cmp->sgpr + s_or") RUN(21, "s_add_u32") RUN(22, "s_nop 4") RUN(23, "v_readfirstlane") RUN(24, "v_xor + 3 s_add") RUN(25, "s_cmp+branch (1 taken,1 not)") RUN(26, "v_cndmask vcc") RUN(27, "v_and_or/v_or3") RUN(28, "v_min_u32") RUN(29, "v_cmp_eq_u32 vcc (VOPC)")
    RUN(30, "v_dot4_u32_u8") RUN(31, "v_lshrrev_sdwa byte->byte") RUN(32, "v_lshrrev_sdwa byte->dword") RUN(33, "v_alignbit_b32") RUN(34, "v_lshlrev/v_and (e32)") RUN(35, "v_lshl_or_b32") RUN(36, "v_mul_u32_u24") RUN(37, "v_mad_u32_u24") RUN(38, "v_bfe_u32") RUN(39, "v_lshrrev_b32 (vgpr shift)") RUN(40, "v_xor + s_nop 0") RUN(41, "v_xor + s_waitcnt") RUN(42, "v_add_u32_sdwa")
    return 0;
}
