/*
 * kmp_dev_common.h -- device-side building blocks shared by the scan kernels (gfx950 only):
 * SWAR byte tests, DPP wave shift, the register-ring wait, the KMP transition, the buffer-load
 * ring issue, the VALU-only 4-byte filter, match-offset emission and the rare-path confirmation.
 * Included by kmp_scan_stream.hip, kmp_scan_multi.hip and kmp_scan_general.hip.
 */
#ifndef KMP_DEV_COMMON_H
#define KMP_DEV_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmp_device.h"

namespace {

constexpr uint32_t ONES  = 0x01010101u;
constexpr uint32_t HIGHS = 0x80808080u;

/* Non-zero iff x has a 0x00 byte; the lowest set bit marks the first one exactly. */
__device__ __forceinline__ uint32_t zero_byte_mask(uint32_t x) { return (x - ONES) & ~x & HIGHS; }

/* lane i <- lane i+1 ; lane 63 <- fill (wave-uniform).  DPP wave_shl:1 (gfx9 family). */
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xF, 0xF, false);
}

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

template <bool MASKED>
__device__ __forceinline__ bool is_cand(uint32_t d, uint32_t first, uint32_t mask)
{
    return MASKED ? (((d ^ first) & mask) == 0u) : (d == first);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));


/* Wait until at most N ring loads are outstanding; a and b are the registers about to be read.
 *
 * Rule for every ring in these kernels: a slot is named by a COMPILE-TIME index (the loops over the slots are
 * fully unrolled).  The loads are issued from inline asm, so the compiler does not know that a slot's value
 * arrives late; it only sees that this wait "rewrites" the slot.  Reached through a run-time index (a loop
 * that was not unrolled), a slot would be copied into a temporary BEFORE the wait -- i.e. before the data has
 * landed (seen once, with a ring period of 12 steps: counts went wrong on the GPU, not in any compile step).
 * Keep ring periods short enough to unroll and check the ISA for one s_waitcnt per slot after changing a ring. */
template <int N>
__device__ __forceinline__ void ring_wait(u32x4 &a, u32x4 &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

/* One KMP transition on text byte ch (serial.c:199-212), pattern/failure table in LDS. */
__device__ __forceinline__ void kmp_step(uint32_t ch, uint32_t &j, uint32_t m, const uint8_t *pat,
                                         const uint8_t *fail, uint32_t &cnt)
{
    while (j > 0u && pat[j] != ch) j = fail[j - 1u];          /* serial.c:207-209 */
    if (pat[j] == ch) ++j;                                     /* serial.c:199-202 */
    if (j == m) { ++cnt; j = fail[m - 1u]; }                   /* serial.c:203-206: overlapping matches count */
}

/* ballot of a lane predicate: the compare's SGPR pair itself, no VALU select */
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

/* 128-bit buffer resource (raw buffer, stride 0) over [base, base + bytes): loads past the end
 * return zeros, so the tail chunk needs neither address clamping nor lane masks -- a zero lane can
 * never be a candidate (patterns are NUL-free) and only ends a packet that ends there anyway. */
__device__ __forceinline__ i32x4 make_rsrc(const uint8_t *base, uint32_t bytes)
{
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    i32x4 r;
    r.x = (int32_t)sgpr((uint32_t)p);
    r.y = (int32_t)sgpr((uint32_t)(p >> 32) & 0xFFFFu);
    r.z = (int32_t)sgpr(bytes);
    r.w = 0x00020000;
    return r;
}

/* buffer_load_dwordx4 vdst, voffset, srsrc, soffset offen: per-lane offset is a loop constant,
 * the chunk offset lives in an SGPR -- no vector arithmetic per load.  Same asm rules as
 * ring_issue (destination only read after a ring_wait naming it).  FRESH: the resource's words have just been written by
 * v_readfirstlane (make_rsrc), which a VMEM instruction may only read five wait states later: the first loads of a range carry an
 * s_nop 4; the loads inside the loop do not (their chunk offset comes from the scalar ALU, which needs none) -- the pass is bound by
 * instruction issue, and an s_nop is an instruction. */
template <bool NT, bool FRESH = false, bool SAME_REGS = false>
__device__ __forceinline__ void flat_issue(u32x4 &dst, i32x4 rsrc, uint32_t vo, uint32_t so)
{
    if (SAME_REGS) {
        /* a second place that starts a ring (the fused pass after an empty work unit): into the registers the ring is in, not into
         * new ones that then have to be copied over (twelve registers the kernel does not have) */
        if (NT) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "+v"(dst) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
        else    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
    } else if (FRESH) {
        if (NT) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(dst) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
        else    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
    } else {
        if (NT) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(dst) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
        else    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
    }
}

/* Per-launch constants of one pattern as the streaming kernels use them: its length and its first 16 bytes as
 * four dwords, 0x00 beyond the pattern's end (the masked SAD skips a 0x00 reference byte, so patterns shorter than a
 * dword, or ending inside one, need no mask registers and no kernel variant of their own). */
struct PatConst {
    uint32_t m;
    uint32_t p[4];
};

__device__ __forceinline__ PatConst load_pat_const(const kmp_pattern_dev *gp)
{
    PatConst pc;
    pc.m = gp->m;
#pragma unroll
    for (int d = 0; d < 4; ++d) pc.p[d] = reinterpret_cast<const uint32_t *>(gp->pat)[d];
    /* Take the scalar loads' results HERE.  Left alone the compiler postpones the s_waitcnt lgkmcnt(0) to the
     * first use of each constant, i.e. to several places inside the streaming loop, where such a wait also
     * covers the bitmap words the loop has just asked for (scalar loads return out of order: only lgkmcnt(0)
     * exists for them). */
    asm volatile("" : "+s"(pc.m), "+s"(pc.p[0]), "+s"(pc.p[1]), "+s"(pc.p[2]), "+s"(pc.p[3]));
    return pc;
}

/* Match-offset emission (kmpgpu_scan_offsets): {packet, offset, pattern} appended to a device
 * buffer.  The lanes that found a match are compacted with a ballot: one atomic add per wavefront
 * reserves their slots, each lane's rank inside the ballot (mbcnt) is its slot. */
struct Emitter {
    uint4              *out;        /* kmpgpu_match[cap] viewed as 16-byte records */
    unsigned long long *counter;    /* matches found so far (may exceed cap)       */
    unsigned long long  cap;
    uint32_t            pattern;
};

template <bool EMIT>
__device__ __forceinline__ void emit_match_as(bool ok, uint64_t pkt, uint32_t offset, uint32_t pattern, const Emitter &e)
{
    if (!EMIT) return;
    const uint64_t b = ballot64(ok);                 /* among the lanes that are active here */
    if (b == 0ull) return;
    const uint32_t leader = (uint32_t)__builtin_ctzll(b);
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long base = 0ull;
    if (lane == leader) base = atomicAdd(e.counter, (unsigned long long)__builtin_popcountll(b));
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, (int)leader);
    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), (int)leader);
    if (ok) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
        const unsigned long long slot = (((unsigned long long)bhi << 32) | blo) + rank;
        if (slot < e.cap) e.out[slot] = make_uint4((uint32_t)pkt, (uint32_t)(pkt >> 32), offset, pattern);
    }
}

template <bool EMIT>
__device__ __forceinline__ void emit_match(bool ok, uint64_t pkt, uint32_t offset, const Emitter &e)
{
    emit_match_as<EMIT>(ok, pkt, offset, e.pattern, e);
}

/* Rare path, part 1: cut a lane's largest valid start index down by the strlen() rule -- no start
 * behind a 0x00 of the same packet (earlier lanes since the packet's start lane, or an earlier
 * chunk: dead_in), and none behind the first 0x00 of the lane's own 16 bytes. */
__device__ __forceinline__ int32_t nul_limit(int32_t maxi, const uint32_t (&w)[5], uint64_t zl, uint64_t st, bool dead_in, uint32_t lane)
{
    const uint64_t below = (1ull << lane) - 1ull;
    const uint64_t st_le = st & (below | (1ull << lane));
    bool nul_before;
    if (st_le == 0ull) nul_before = dead_in || ((zl & below) != 0ull);
    else {
        const uint32_t sl = 63u - (uint32_t)__builtin_clzll(st_le);
        nul_before = (zl & below & ~((1ull << sl) - 1ull)) != 0ull;
    }
    const uint32_t m0 = zero_byte_mask(w[0]), m1 = zero_byte_mask(w[1]), m2 = zero_byte_mask(w[2]), m3 = zero_byte_mask(w[3]);
    uint32_t zi = 16u;                               /* first 0x00 inside the lane's own 16 bytes (16 = none) */
    if (m3) zi = 12u + ((uint32_t)__builtin_ctz(m3) >> 3);
    if (m2) zi = 8u + ((uint32_t)__builtin_ctz(m2) >> 3);
    if (m1) zi = 4u + ((uint32_t)__builtin_ctz(m1) >> 3);
    if (m0) zi = (uint32_t)__builtin_ctz(m0) >> 3;
    return nul_before ? -1 : min(maxi, (int32_t)zi - 1);
}

/* packed 16-bit helpers (VOP3P), written as asm so that the instruction selection is what the cost model assumes;
 * the second operand may sit in an SGPR (constants) */
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
/* per half: max(0, k - b), k a wave-uniform constant */
__device__ __forceinline__ uint32_t pk_rsub_sat_u16(uint32_t k, uint32_t b)
{
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "s"(k), "v"(b));
    return r;
}
/* non-zero iff one of the two 16-bit halves of x is zero */
__device__ __forceinline__ uint32_t zero_half_mask(uint32_t x) { return (x - 0x00010001u) & ~x & 0x80008000u; }

/* v_mqsad_pk_u16_u8: four masked byte-SADs in one instruction.  Result k (bits [16k, 16k+16)) = acc_k + the sum over
 * the reference's four bytes b of |text byte (k + b) - ref byte b|, a 0x00 reference byte being skipped; text = the 8
 * bytes {lo, hi}.  So ONE instruction compares one pattern dword at four consecutive start offsets, alignment included,
 * and accumulates: a start offset matches the pattern's bytes seen so far iff its sum is still 0.  Semantics and cost
 * measured on MI355X (tools/sadtest.hip, profiles/r02_sadtest.txt): 7.3-8.2 ns per instruction per SIMD = 4
 * three-operand VALU instructions, against 3 v_alignbyte + 4 v_xor + 4 v_or (filter) or + 4 v_sad_u8 (comparison) for
 * the same four results. */
__device__ __forceinline__ uint64_t mqsad(uint32_t lo, uint32_t hi, uint32_t ref, uint64_t acc)
{
    return __builtin_amdgcn_mqsad_pk_u16_u8(((uint64_t)hi << 32) | lo, ref, acc);
}

/* Candidate test for the 16 start offsets of a lane: S[q] = the four sums of the pattern's first dword (its first
 * min(m, 4) bytes) against the start offsets 4q .. 4q+3; t[q] = their minimum per 16-bit half; the return value has a
 * zero half iff some start offset of this lane shows the pattern's first bytes.  4 v_mqsad + 7 v_pk_min_u16.  The sums
 * stay in registers: the rare path continues to accumulate the pattern's other dwords into them. */
__device__ __forceinline__ uint32_t filter_sad(const uint32_t (&w)[5], uint32_t first, uint64_t (&S)[4], uint32_t (&t)[4])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        S[q] = mqsad(w[q], w[q + 1], first, 0ull);
        t[q] = pk_min_u16((uint32_t)S[q], (uint32_t)(S[q] >> 32));
    }
    return pk_min_u16(pk_min_u16(t[0], t[1]), pk_min_u16(t[2], t[3]));
}

/* != 0 iff a 16-bit half of x is zero (sums stay below 2^15): the product of the halves, one SDWA instruction */
__device__ __forceinline__ uint32_t halves_product(uint32_t x)
{
    uint32_t r;
    asm("v_mul_u32_u24_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(x));
    return r;
}

/* Rare path, part 2: count (and optionally emit) the matches that start at index <= maxi of each lane.
 *
 * Every start offset of every lane is compared in full, with no per-candidate branching, so that text made of
 * candidates (one letter, two letters, a pattern that is everywhere) costs a small multiple of the filter instead of
 * a byte-serial automaton per lane:
 *   S[q]   the filter's sums (first pattern dword) for the start offsets 4q .. 4q+3;
 *   block 0 adds the pattern dwords 1..3 (bytes 4..15) against the lane's own 16 bytes and the 16 that follow (DPP
 *          wave_shl:1, lane 63 continuing in the next chunk); block k >= 1 (patterns of more than 16 bytes) compares
 *          the bytes 16k .. 16k+15 against the data of lane + k and lane + k + 1.  Zero pattern bytes past the
 *          pattern's end are skipped by the masked SAD.  Between blocks the wavefront leaves as soon as no lane has a
 *          zero sum left (false candidates of a long pattern die in block 0);
 *   a start offset counts iff its sum is still 0 after the last block and its index is <= maxi.
 * Only the groups of four offsets in which some lane passed the filter are worked on: with sparse candidates that is
 * one group, with dense ones all four.  ND = pattern dwords compared in block 0 (1: the filter was exact, 4: all).
 * barred (wave-uniform): some candidate lane may not count all of its 16 start offsets (payload end or a 0x00 nearby):
 * the sums of the offsets above maxi are raised first (per half: max(sum, max(0, index - maxi))). */
template <bool EMIT>
__device__ __forceinline__ uint32_t tally_group(int q, uint64_t Sq, bool barred, uint32_t nv2, uint32_t found, uint32_t p0, uint32_t &cnt, uint64_t pkt,
                                                const Emitter &em)
{
    uint32_t lo = (uint32_t)Sq, hi = (uint32_t)(Sq >> 32);
    if (barred) {
        asm volatile("" ::: "memory");
        lo = pk_max_u16(lo, pk_rsub_sat_u16((uint32_t)(4 * q + 1) | ((uint32_t)(4 * q + 2) << 16), nv2));
        hi = pk_max_u16(hi, pk_rsub_sat_u16((uint32_t)(4 * q + 3) | ((uint32_t)(4 * q + 4) << 16), nv2));
    }
    if (!EMIT) return pk_add_u16(found, pk_add_u16(pk_rsub_sat_u16(0x00010001u, lo), pk_rsub_sat_u16(0x00010001u, hi)));     /* 1 - min(x, 1) per half */
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const bool ok = (((a < 2 ? lo : hi) >> (16 * (a & 1))) & 0xFFFFu) == 0u;
        cnt += ok ? 1u : 0u;
        emit_match<EMIT>(ok, pkt, p0 + (uint32_t)(4 * q + a), em);
    }
    return found;
}

template <int ND, bool EMIT>
__device__ __forceinline__ void confirm_block0(uint64_t (&S)[4], const uint32_t (&t)[4], const uint32_t (&c)[8], bool barred, uint32_t nv2, uint32_t p0,
                                               const PatConst &pc, bool more, uint32_t &cnt, uint64_t pkt, const Emitter &em, bool (&on)[4])
{
    uint32_t found = 0u;                                            /* packed: matches at the even / odd offsets */
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        on[q] = ballot64(halves_product(t[q]) == 0u) != 0ull;
        if (!on[q]) continue;
        asm volatile("" ::: "memory");                              /* keep the wave-uniform branch: no select-both-ways */
        if (ND > 1) {
            /* all three further dwords, whatever the length: a dword past the pattern's end is 0x00000000 and adds
             * nothing (two idle instructions per group for patterns of 5..8 bytes, against a branch per dword and group) */
            S[q] = mqsad(c[q + 1], c[q + 2], pc.p[1], S[q]);
            S[q] = mqsad(c[q + 2], c[q + 3], pc.p[2], S[q]);
            S[q] = mqsad(c[q + 3], c[q + 4], pc.p[3], S[q]);
        }
        if (more) continue;                                         /* a pattern of more than 16 bytes: not decided yet */
        found = tally_group<EMIT>(q, S[q], barred, nv2, found, p0, cnt, pkt, em);
    }
    if (!EMIT) cnt += (found & 0xFFFFu) + (found >> 16);
}

/* fz != 0 marks the lanes with a candidate; maxi < 0 those that may count nothing */
template <bool EMIT>
__device__ __forceinline__ void confirm_sad(uint64_t (&S)[4], const uint32_t (&t)[4], const uint32_t (&w)[5], uint4 v, u32x4 bn, uint32_t fz, int32_t maxi,
                                            uint32_t p0, const PatConst &pc, const kmp_pattern_dev *gp, uint32_t &cnt, uint64_t pkt,
                                            const Emitter &em)
{
    const uint32_t m = pc.m;
    /* the common case of the rare path: every candidate lane may count all 16 of its start offsets */
    const bool barred = ballot64(maxi < 15 && fz != 0u) != 0ull;
    uint32_t nv2 = 0u;
    if (barred) {
        if (ballot64(maxi >= 0) == 0ull) return;
        const uint32_t nv = (uint32_t)min(max(maxi + 1, 0), 16);
        nv2 = nv | (nv << 16);
    }
    uint32_t c[8] = {w[0], w[1], w[2], w[3], w[4], 0u, 0u, 0u};
    bool on[4];
    if (m <= 4u) { confirm_block0<1, EMIT>(S, t, c, barred, nv2, p0, pc, false, cnt, pkt, em, on); return; }
    /* the 16 bytes that follow the lane's own (the first dword of them is w[4]) */
    c[5] = wave_shl1(v.y, sgpr(bn.y)); c[6] = wave_shl1(v.z, sgpr(bn.z)); c[7] = wave_shl1(v.w, sgpr(bn.w));
    const bool more = m > 16u;
    confirm_block0<4, EMIT>(S, t, c, barred, nv2, p0, pc, more, cnt, pkt, em, on);
    if (!more) return;

    for (uint32_t k = 1u; 16u * k < m; ++k) {
        /* does any lane still have a start offset whose sum is 0? */
        uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (on[q]) mn = pk_min_u16(mn, pk_min_u16((uint32_t)S[q], (uint32_t)(S[q] >> 32)));
        if (ballot64(halves_product(mn) == 0u) == 0ull) return;
        /* 16 bytes further on; lane 63 continues with lane k of the next chunk */
        const uint4 pk = reinterpret_cast<const uint4 *>(gp->pat)[k];
        const uint32_t left = m - 16u * k;
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = c[j + 4];
        c[4] = wave_shl1(c[4], (uint32_t)__builtin_amdgcn_readlane((int)bn.x, (int)k));
        c[5] = wave_shl1(c[5], (uint32_t)__builtin_amdgcn_readlane((int)bn.y, (int)k));
        c[6] = wave_shl1(c[6], (uint32_t)__builtin_amdgcn_readlane((int)bn.z, (int)k));
        c[7] = wave_shl1(c[7], (uint32_t)__builtin_amdgcn_readlane((int)bn.w, (int)k));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!on[q]) continue;
            asm volatile("" ::: "memory");
            S[q] = mqsad(c[q], c[q + 1], pk.x, S[q]);
            if (left > 4u)  { asm volatile("" ::: "memory"); S[q] = mqsad(c[q + 1], c[q + 2], pk.y, S[q]); }
            if (left > 8u)  { asm volatile("" ::: "memory"); S[q] = mqsad(c[q + 2], c[q + 3], pk.z, S[q]); }
            if (left > 12u) { asm volatile("" ::: "memory"); S[q] = mqsad(c[q + 3], c[q + 4], pk.w, S[q]); }
        }
    }
    /* matches = the sums that are still 0, at start offsets 0 .. maxi */
    uint32_t found = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!on[q]) continue;
        asm volatile("" ::: "memory");
        found = tally_group<EMIT>(q, S[q], barred, nv2, found, p0, cnt, pkt, em);
    }
    if (!EMIT) cnt += (found & 0xFFFFu) + (found >> 16);
}


/* Per wavefront: first packet index and byte offset of its range (kmp_plan_kernel). */
struct kmp_plan_entry { uint64_t k; uint64_t off; };

}  // namespace

#endif
