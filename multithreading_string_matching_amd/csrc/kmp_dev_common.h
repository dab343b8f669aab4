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

/* v_min3_u32: written as asm because hipcc re-associates min(a, min(b, c)) chains into extra v_min_u32 */
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

/* Candidate test for the 16 start offsets of a lane, VALU only: min over (dword ^ first), kept per group of
 * four offsets (g[q] covers offsets 4q..4q+3) so that the rare path only looks at the groups that hold a
 * candidate; 12 v_alignbyte + 16 v_xor + 4 v_min + 6 v_min3 and no scalar work. */
template <bool MASKED>
__device__ __forceinline__ uint32_t filter_min(const uint32_t (&w)[5], uint32_t first, uint32_t mask, uint32_t (&g)[4])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t lo = w[q], hi = w[q + 1];
        uint32_t x0 = lo ^ first;
        uint32_t x1 = __builtin_amdgcn_alignbyte(hi, lo, 1) ^ first;
        uint32_t x2 = __builtin_amdgcn_alignbyte(hi, lo, 2) ^ first;
        uint32_t x3 = __builtin_amdgcn_alignbyte(hi, lo, 3) ^ first;
        if (MASKED) { x0 &= mask; x1 &= mask; x2 &= mask; x3 &= mask; }
        g[q] = min3u(min(x0, x1), x2, x3);
    }
    return min3u(min(g[0], g[1]), g[2], g[3]);      /* 0 iff some start offset of this lane shows the pattern's first bytes */
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
 * ring_issue (destination only read after a ring_wait naming it). */
template <bool NT>
__device__ __forceinline__ void flat_issue(u32x4 &dst, i32x4 rsrc, uint32_t vo, uint32_t so)
{
    if (NT)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(dst) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
    else
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
}

/* Per-launch constants of one pattern as the streaming kernels use them. */
struct PatConst {
    uint32_t m, first, mask;
    uint32_t pd[4], pm[4];          /* pattern bytes 4..19 and their byte masks (direct confirmation, m <= 20) */
};

__device__ __forceinline__ PatConst load_pat_const(const kmp_pattern_dev *gp)
{
    PatConst pc;
    pc.m = gp->m; pc.first = gp->first; pc.mask = gp->mask;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t lo = 4u * (uint32_t)(d + 1);
        pc.pd[d] = reinterpret_cast<const uint32_t *>(gp->pat)[d + 1];
        pc.pm[d] = (pc.m >= lo + 4u) ? 0xFFFFFFFFu : (pc.m <= lo) ? 0u : ((1u << (8u * (pc.m - lo))) - 1u);
    }
    /* Take the scalar loads' results HERE.  Left alone the compiler postpones the s_waitcnt lgkmcnt(0) to the
     * first use of each constant, i.e. to several places inside the streaming loop, where such a wait also
     * covers the bitmap words the loop has just asked for (scalar loads return out of order: only lgkmcnt(0)
     * exists for them). */
    asm volatile("" : "+s"(pc.m), "+s"(pc.first), "+s"(pc.mask));
    asm volatile("" : "+s"(pc.pd[0]), "+s"(pc.pd[1]), "+s"(pc.pd[2]), "+s"(pc.pd[3]));
    asm volatile("" : "+s"(pc.pm[0]), "+s"(pc.pm[1]), "+s"(pc.pm[2]), "+s"(pc.pm[3]));
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

/* KMP automaton for the streaming kernels: the lane scans the text from its own first byte; it
 * stops at a 0x00 (serial.c:191) or at the payload end; matches found all start inside the lane's
 * 16 bytes because at most 15 + m bytes are consumed. */
template <bool EMIT>
__device__ __forceinline__ void automaton_flat(uint4 cur, uint4 nxt, bool act, uint32_t p0, uint32_t L, uint32_t m,
                                               const kmp_pattern_dev &sp, uint32_t &cnt, uint64_t pkt, const Emitter &em)
{
    uint32_t j = 0u;
    const uint32_t nsteps = 15u + m;
    uint32_t c0 = cur.x, c1 = cur.y, c2 = cur.z, c3 = cur.w;
    uint32_t n0 = nxt.x, n1 = nxt.y, n2 = nxt.z, n3 = nxt.w;
    uint32_t t0 = p0;
    for (uint32_t done = 0u; done < nsteps; done += 16u) {
        const uint32_t lim = min(16u, nsteps - done);
#pragma unroll 1
        for (uint32_t s = 0u; s < lim; ++s) {
            const uint32_t w  = (s & 8u) ? ((s & 4u) ? c3 : c2) : ((s & 4u) ? c1 : c0);
            const uint32_t ch = (w >> (8u * (s & 3u))) & 0xFFu;
            act = act && (ch != 0u) && (t0 + s < L);
            if (act) {
                const uint32_t before = cnt;
                kmp_step(ch, j, m, sp.pat, sp.fail, cnt);
                emit_match<EMIT>(cnt != before, pkt, t0 + s + 1u - m, em);
            }
        }
        t0 += 16u;
        if (ballot64(act) == 0ull) break;
        const uint32_t f0 = sgpr(n0), f1 = sgpr(n1), f2 = sgpr(n2), f3 = sgpr(n3);
        c0 = wave_shl1(c0, f0); c1 = wave_shl1(c1, f1); c2 = wave_shl1(c2, f2); c3 = wave_shl1(c3, f3);
        n0 = wave_shl1(n0, 0u); n1 = wave_shl1(n1, 0u); n2 = wave_shl1(n2, 0u); n3 = wave_shl1(n3, 0u);
    }
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

/* Rare path, part 2: count (and optionally emit) the matches that start at index <= maxi of each
 * lane.  Patterns of <= 4 bytes are exact after the filter; 5..20 bytes with few candidate lanes are
 * compared dword-wise straight from registers (W = the lane's 16 bytes + the next 20 of the stream);
 * everything else runs the KMP automaton. */
template <bool MASKED, bool EMIT>
__device__ __forceinline__ void confirm_lanes(const uint32_t (&w)[5], const uint32_t (&g)[4], uint4 v, u32x4 bn, int32_t maxi, uint32_t p0, uint32_t L,
                                              const PatConst &pc, const kmp_pattern_dev &sp, uint32_t &cnt, uint64_t pkt,
                                              const Emitter &em)
{
    const uint32_t m = pc.m;
    const uint64_t ba = ballot64(maxi >= 0);
    if (m <= 4u) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ballot64(g[q] == 0u) == 0ull) continue;           /* no lane has a candidate among offsets 4q..4q+3 */
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const uint32_t d0 = a ? __builtin_amdgcn_alignbyte(w[q + 1], w[q], a) : w[q];
                const bool ok = is_cand<MASKED>(d0, pc.first, pc.mask) && (4 * q + a) <= maxi;
                cnt += ok ? 1u : 0u;
                emit_match<EMIT>(ok, pkt, p0 + (uint32_t)(4 * q + a), em);
            }
        }
    } else if (ba != 0ull) {
        if (m <= 20u && __builtin_popcountll(ba) <= 16) {
            const uint32_t W[10] = {w[0], w[1], w[2], w[3], w[4], wave_shl1(v.y, sgpr(bn.y)), wave_shl1(v.z, sgpr(bn.z)),
                                    wave_shl1(v.w, sgpr(bn.w)), wave_shl1(w[4], (uint32_t)__builtin_amdgcn_readlane((int)bn.x, 1)), 0u};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ballot64(g[q] == 0u) == 0ull) continue;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const uint32_t d0 = a ? __builtin_amdgcn_alignbyte(W[q + 1], W[q], a) : W[q];
                    bool ok = (d0 == pc.first) && (4 * q + a) <= maxi;
                    if (ballot64(ok) != 0ull) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const uint32_t t = a ? __builtin_amdgcn_alignbyte(W[q + d + 2], W[q + d + 1], a) : W[q + d + 1];
                            ok = ok && (((t ^ pc.pd[d]) & pc.pm[d]) == 0u);
                        }
                        cnt += ok ? 1u : 0u;
                        emit_match<EMIT>(ok, pkt, p0 + (uint32_t)(4 * q + a), em);
                    }
                }
            }
        } else {
            automaton_flat<EMIT>(v, make_uint4(bn.x, bn.y, bn.z, bn.w), maxi >= 0, p0, L, m, sp, cnt, pkt, em);
        }
    }
}


/* Per wavefront: first packet index and byte offset of its range (kmp_plan_kernel). */
struct kmp_plan_entry { uint64_t k; uint64_t off; };

}  // namespace

#endif
