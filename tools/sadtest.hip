// Tuning only: semantics and issue cost of the byte-SAD instructions on gfx950 (v_qsad_pk_u16_u8, v_mqsad_pk_u16_u8,
// v_mqsad_u32_u8, v_msad_u8): which operand's zero bytes mask a lane of the sum, how the four results are packed,
// and ns per instruction per SIMD at 8 waves/SIMD (same harness as tools/ubench.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void sem(const uint64_t *s0, const uint32_t *s1, uint64_t *d_q, uint64_t *d_mq, uint4 *d_mq32, uint32_t *d_msad, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d_q[i]  = __builtin_amdgcn_qsad_pk_u16_u8(s0[i], s1[i], 0x0001000200030004ull);
    d_mq[i] = __builtin_amdgcn_mqsad_pk_u16_u8(s0[i], s1[i], 0x0001000200030004ull);
    u4 z = {1, 2, 3, 4};
    u4 r = __builtin_amdgcn_mqsad_u32_u8(s0[i], s1[i], z);
    d_mq32[i] = make_uint4(r.x, r.y, r.z, r.w);
    d_msad[i] = __builtin_amdgcn_msad_u8((uint32_t)s0[i], s1[i], 7u);
}

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void __launch_bounds__(256) rate(uint32_t *out, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x9e3779b9u;
    uint64_t q = ((uint64_t)a << 32) | b, r = q * 3, q2 = q ^ r, r2 = r + 5;
    u4 w = {a, b, a + 1, b + 1}, w2 = {b, a, b + 2, a + 2};
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP64(asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0\n v_qsad_pk_u16_u8 %3, %4, %2, %3" : "+v"(q), "+v"(r), "+v"(a), "+v"(q2), "+v"(r2));) }
        if (OP == 1) { REP64(asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0\n v_mqsad_pk_u16_u8 %3, %4, %2, %3" : "+v"(q), "+v"(r), "+v"(a), "+v"(q2), "+v"(r2));) }
        if (OP == 2) { REP64(asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0\n v_mqsad_u32_u8 %3, %4, %2, %3" : "+v"(w), "+v"(r), "+v"(a), "+v"(w2), "+v"(r2));) }
        if (OP == 3) { REP64(asm volatile("v_qsad_pk_u16_u8 %0, %1, s20, %0\n v_qsad_pk_u16_u8 %2, %3, s21, %2" : "+v"(q), "+v"(r), "+v"(q2), "+v"(r2) :: "s20", "s21");) }
        if (OP == 4) { REP64(asm volatile("v_pk_sub_u16 %0, %0, %1 clamp\n v_pk_sub_u16 %2, %2, %3 clamp" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
        if (OP == 5) { REP64(asm volatile("v_sad_u8 %0, %1, s20, %0\n v_sad_u8 %2, %3, s21, %2" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y):: "s20", "s21");) }
        if (OP == 6) { REP64(asm volatile("v_lshl_or_b32 %0, %1, 3, %0\n v_lshl_or_b32 %2, %3, 5, %2" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
        if (OP == 7) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
        if (OP == 8) { REP64(asm volatile("v_mul_hi_u32_u24 %0, %0, %1\n v_mul_hi_u32_u24 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
        if (OP == 9) { REP64(asm volatile("v_bfe_u32 %0, %0, %1, 1\n v_bfe_u32 %2, %2, %3, 1" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
        if (OP == 10) { REP64(asm volatile("v_lshrrev_b32 %0, 3, %1\n v_lshrrev_b32 %2, 5, %3" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
        if (OP == 11) { REP64(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
        if (OP == 12) { REP64(asm volatile("v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %2, %3, %2" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
        if (OP == 13) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %3, %0" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
        if (OP == 14) { REP64(asm volatile("v_cmp_eq_u16_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:WORD_0\n v_cmp_eq_u16_sdwa vcc, %2, %3 src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y) :: "vcc");) }
        if (OP == 15) { REP64(asm volatile("v_and_or_b32 %0, %1, 4, %0\n v_and_or_b32 %2, %3, 8, %2" : "+v"(a), "+v"(b), "+v"(w.x), "+v"(w.y));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + (uint32_t)q + (uint32_t)r + (uint32_t)q2 + (uint32_t)r2 + w.x + w.y + w.z + w2.x + w2.w;
}
template <int OP> void run(const char *name, uint32_t *out, int blocks)
{
    const int iters = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 64 * 2;
    const int waves_per_simd = blocks / 256;
    printf("%-28s wall %.3f ms -> %.2f ns/instr/SIMD at %d waves/SIMD\n", name, ms, ms * 1e6 / (n * waves_per_simd), waves_per_simd);
}

static uint32_t absd(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n = 4096;
    std::vector<uint64_t> s0(n); std::vector<uint32_t> s1(n);
    srand(7);
    for (int i = 0; i < n; i++) {
        uint64_t x = 0; for (int b = 0; b < 8; b++) x |= (uint64_t)((rand() % 4 == 0) ? 0 : (rand() & 0xFF)) << (8 * b);
        uint32_t y = 0; for (int b = 0; b < 4; b++) y |= (uint32_t)((rand() % 3 == 0) ? 0 : (rand() & 0xFF)) << (8 * b);
        s0[i] = x; s1[i] = y;
    }
    uint64_t *d0, *dq, *dmq; uint32_t *d1, *dms; uint4 *d32;
    hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 4); hipMalloc(&dq, n * 8); hipMalloc(&dmq, n * 8); hipMalloc(&d32, n * 16); hipMalloc(&dms, n * 4);
    hipMemcpy(d0, s0.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(d1, s1.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem, dim3(n / 256), dim3(256), 0, 0, d0, d1, dq, dmq, d32, dms, n);
    std::vector<uint64_t> q(n), mq(n); std::vector<uint4> m32(n); std::vector<uint32_t> ms(n);
    hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost); hipMemcpy(mq.data(), dmq, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(m32.data(), d32, n * 16, hipMemcpyDeviceToHost); hipMemcpy(ms.data(), dms, n * 4, hipMemcpyDeviceToHost);
    // models: result k (k = 0..3) = acc_k + sum_b f(text byte (k + b), ref byte b); mask variants: none / ref byte zero / text byte zero
    long bad_q = 0, bad_mq_ref = 0, bad_mq_txt = 0, bad_32_ref = 0, bad_32_txt = 0, bad_ms_ref = 0, bad_ms_txt = 0;
    const uint32_t acc16[4] = {4, 3, 2, 1};        // 0x0001000200030004: low half first
    const uint32_t acc32[4] = {1, 2, 3, 4};
    for (int i = 0; i < n; i++) {
        uint32_t full[4], mref[4], mtxt[4];
        for (int k = 0; k < 4; k++) {
            full[k] = mref[k] = mtxt[k] = 0;
            for (int b = 0; b < 4; b++) {
                uint32_t t = (s0[i] >> (8 * (k + b))) & 0xFF, r = (s1[i] >> (8 * b)) & 0xFF;
                full[k] += absd(t, r);
                if (r) mref[k] += absd(t, r);
                if (t) mtxt[k] += absd(t, r);
            }
        }
        for (int k = 0; k < 4; k++) {
            uint32_t gq = (q[i] >> (16 * k)) & 0xFFFF, gm = (mq[i] >> (16 * k)) & 0xFFFF;
            uint32_t g32 = k == 0 ? m32[i].x : k == 1 ? m32[i].y : k == 2 ? m32[i].z : m32[i].w;
            bad_q += gq != ((full[k] + acc16[k]) & 0xFFFF);
            bad_mq_ref += gm != ((mref[k] + acc16[k]) & 0xFFFF);
            bad_mq_txt += gm != ((mtxt[k] + acc16[k]) & 0xFFFF);
            bad_32_ref += g32 != mref[k] + acc32[k];
            bad_32_txt += g32 != mtxt[k] + acc32[k];
        }
        bad_ms_ref += ms[i] != mref[0] + 7;
        bad_ms_txt += ms[i] != mtxt[0] + 7;
    }
    printf("qsad_pk (no mask, result k in bits [16k,16k+16), text byte k+b vs ref byte b): mismatches %ld\n", bad_q);
    printf("mqsad_pk: mask = ref byte zero: %ld mismatches; mask = text byte zero: %ld\n", bad_mq_ref, bad_mq_txt);
    printf("mqsad_u32: mask = ref byte zero: %ld mismatches; mask = text byte zero: %ld\n", bad_32_ref, bad_32_txt);
    printf("msad_u8: mask = ref(S1) byte zero: %ld mismatches; mask = S0 byte zero: %ld\n", bad_ms_ref, bad_ms_txt);
    for (int i = 0; i < 3; i++) printf("  s0=%016llx s1=%08x q=%016llx mq=%016llx mq32=%x,%x,%x,%x msad=%x\n", (unsigned long long)s0[i], s1[i], (unsigned long long)q[i],
                                       (unsigned long long)mq[i], m32[i].x, m32[i].y, m32[i].z, m32[i].w, ms[i]);
    uint32_t *out; hipMalloc(&out, (size_t)2048 * 256 * 4);
    run<0>("v_qsad_pk_u16_u8", out, 2048); run<1>("v_mqsad_pk_u16_u8", out, 2048); run<2>("v_mqsad_u32_u8", out, 2048); run<3>("v_qsad_pk (sgpr ref)", out, 2048);
    run<4>("v_pk_sub_u16 clamp", out, 2048); run<5>("v_sad_u8 (sgpr ref)", out, 2048); run<6>("v_lshl_or_b32", out, 2048); run<7>("v_mul_u32_u24", out, 2048);
    run<8>("v_mul_hi_u32_u24", out, 2048); run<9>("v_bfe_u32", out, 2048); run<10>("v_lshrrev_b32", out, 2048); run<11>("v_and_b32", out, 2048);
    run<12>("v_bcnt_u32_b32", out, 2048); run<13>("v_mad_u32_u24", out, 2048); run<14>("v_cmp_eq_u16_sdwa", out, 2048); run<15>("v_and_or_b32", out, 2048);
    return 0;
}
