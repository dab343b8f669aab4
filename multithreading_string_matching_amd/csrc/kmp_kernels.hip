/*
 * kmp_kernels.hip -- the hot path on gfx950 (MI355X / CDNA4): per-pattern match counts over a
 * payload arena.  Replaces the reference loop serial.c:153-155 (= openmp_data.c:157-175) and its
 * kernel function kmp_matcher (serial.c:190-215).  Hand-written for 64-lane wavefronts; no MFMA
 * (byte scan, HBM-bound).
 *
 * Shape of the scan kernel
 *   - one packet per wavefront at a time; a persistent grid strides over the packets;
 *   - a packet is read in 1 KiB chunks: one global_load_dwordx4 per lane, perfectly coalesced,
 *     DEPTH chunk loads in flight per wavefront (register ring), packets pipelined back to back;
 *   - the pattern and its KMP failure table are staged in LDS once per block;
 *   - per chunk, every lane tests its 16 start offsets with a 4-byte SWAR compare against the
 *     pattern's first dword (halo dword from the next lane by DPP wave_shl:1) and looks for a
 *     0x00 byte with the has-zero trick; both results are wave-reduced with ballots;
 *   - the common case (no candidate in the chunk) ends there.  Otherwise the candidates are
 *     confirmed: patterns of <= 4 bytes are already exact; longer ones run the KMP automaton
 *     (LDS pattern + failure table) over the lane's 16 + m - 1 bytes, all in registers, the halo
 *     arriving by repeated wave_shl:1 shifts;
 *   - the reference's strlen() rule (serial.c:191): a start offset s counts only if
 *     s + m <= E, E = min(len, first 0x00).  A chunk that holds the first NUL ends the packet.
 *   - counts: per-lane -> wave -> block, one partial per (block, pattern), summed by
 *     kmp_reduce_kernel (no atomics, deterministic).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kmp_device.h"
#include "kmp_launch.h"

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

/* Iterator over the (packet, chunk) sequence of one wavefront.  Everything is wave-uniform
 * (SGPRs).  The index entry of the wavefront's NEXT packet is requested one packet ahead so the
 * scalar-load latency never sits between two chunk loads. */
struct ChunkIter {
    uint64_t k;        /* packet index                              */
    uint64_t off;      /* arena offset of the packet                */
    uint32_t L;        /* payload length                            */
    uint32_t c;        /* chunk index inside the packet             */
    uint32_t nch;      /* chunks in the packet (>= 1)               */
    uint32_t seq;      /* running packet number of this wavefront   */
    uint64_t off_n;    /* prefetched index entry of packet k+stride */
    uint32_t L_n;
    bool     valid;
};

__device__ __forceinline__ uint32_t chunks_of(uint32_t L) { return L ? (L + KMP_CHUNK - 1u) / KMP_CHUNK : 1u; }

__device__ __forceinline__ void iter_prefetch(ChunkIter &it, const uint64_t *__restrict__ pkt_off,
                                              const uint32_t *__restrict__ pkt_len, uint64_t n, uint64_t stride)
{
    const uint64_t kn = it.k + stride;
    it.off_n = 0; it.L_n = 0;
    if (kn < n) { it.off_n = pkt_off[kn]; it.L_n = pkt_len[kn]; }
}

__device__ __forceinline__ void iter_init(ChunkIter &it, uint64_t k0, const uint64_t *__restrict__ pkt_off,
                                          const uint32_t *__restrict__ pkt_len, uint64_t n, uint64_t stride)
{
    it.k = k0; it.c = 0; it.seq = 0; it.valid = (k0 < n); it.off = 0; it.L = 0;
    if (it.valid) { it.off = pkt_off[k0]; it.L = pkt_len[k0]; }
    it.nch = chunks_of(it.L);
    iter_prefetch(it, pkt_off, pkt_len, n, stride);
}

__device__ __forceinline__ void iter_next(ChunkIter &it, const uint64_t *__restrict__ pkt_off,
                                          const uint32_t *__restrict__ pkt_len, uint64_t n, uint64_t stride)
{
    if (!it.valid) return;           /* past the end: stays on a harmless (offset 0, length 0) chunk */
    if (++it.c >= it.nch) {
        it.k += stride; it.c = 0; ++it.seq;
        it.valid = (it.k < n);
        it.off = it.off_n; it.L = it.L_n; it.nch = chunks_of(it.L);
        iter_prefetch(it, pkt_off, pkt_len, n, stride);
    }
}

/* What the consumer needs to know about a chunk that sits in the register ring (3 SGPRs):
 * cw = chunk's byte offset inside the packet | has-next-chunk << 30 | valid << 31. */
struct Slot {
    uint32_t L, cw, seq;
};
constexpr uint32_t SLOT_NEXT  = 1u << 30;
constexpr uint32_t SLOT_VALID = 1u << 31;
constexpr uint32_t SLOT_BASE  = SLOT_NEXT - 1u;

__device__ __forceinline__ Slot slot_of(const ChunkIter &it)
{
    Slot s;
    s.L = it.L; s.seq = it.seq;
    s.cw = (it.c * KMP_CHUNK) | ((it.c + 1u < it.nch) ? SLOT_NEXT : 0u) | (it.valid ? SLOT_VALID : 0u);
    return s;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

/*
 * Issue the 16 B/lane load of the iterator's chunk: global_load_dwordx4 vdst, voffset, s[base].
 * Lanes past the slot's end re-read the chunk's first 16 bytes (same cache line, no extra traffic)
 * so the load is unconditional.
 *
 * The ring's loads and waits are inline asm ON PURPOSE: hipcc's own s_waitcnt insertion degrades
 * to vmcnt(0) in this loop (conditional per-chunk work between the loads), which serialises the
 * ring.  Rules kept (guide section 5.7): the destination is "=v" and is only read after a
 * ring_wait() statement that names it "+v"; every ring load is waited for before its register is
 * refilled or the kernel leaves the loop (ring_drain), so no in-flight load ever targets a
 * register the compiler considers free; s_nop 4 covers an SGPR base fresh from a VALU write.
 */
template <bool NT>
__device__ __forceinline__ void ring_issue(u32x4 &dst, const uint8_t *__restrict__ arena, const ChunkIter &it, uint32_t lane)
{
    const uint32_t cb  = it.c * KMP_CHUNK;
    const uint32_t bo  = cb + lane * KMP_LANE_BYTES;
    const uint32_t L16 = (it.L + 15u) & ~15u;
    const uint32_t vo  = (bo < L16) ? bo : cb;
    const uint8_t *base = arena + it.off;                  /* wave-uniform -> SGPR pair */
    if (NT)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(vo), "s"(base) : "memory");
    else
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(vo), "s"(base) : "memory");
}

/* Wait until at most N ring loads are outstanding; a and b are the registers about to be read. */
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

/* KMP automaton over the text bytes p0 .. p0+15+m-1 of every active lane (its own 16 start
 * offsets plus the halo), bytes at positions >= Eloc excluded.  cur = this chunk's registers,
 * nxt = the next chunk's (zeros when the packet ends here).  Runs in wave-uniform control flow. */
__device__ __forceinline__ void automaton_chunk(uint4 cur, uint4 nxt, bool act, uint32_t p0, uint32_t Eloc,
                                                uint32_t m, const kmp_pattern_dev &sp, uint32_t &cnt)
{
    uint32_t j = 0u;
    const uint32_t nsteps = 15u + m;
    uint32_t c0 = cur.x, c1 = cur.y, c2 = cur.z, c3 = cur.w;
    uint32_t n0 = nxt.x, n1 = nxt.y, n2 = nxt.z, n3 = nxt.w;
    uint32_t t0 = p0;
    for (uint32_t done = 0u; done < nsteps; done += 16u) {
        const uint32_t lim = min(16u, nsteps - done);
        /* not unrolled on purpose: this is the rare path, keep it small (s is wave-uniform) */
#pragma unroll 1
        for (uint32_t s = 0u; s < lim; ++s) {
            const uint32_t w  = (s & 8u) ? ((s & 4u) ? c3 : c2) : ((s & 4u) ? c1 : c0);
            const uint32_t ch = (w >> (8u * (s & 3u))) & 0xFFu;
            if (act && (t0 + s < Eloc)) kmp_step(ch, j, m, sp.pat, sp.fail, cnt);
        }
        t0 += 16u;
        if (__ballot(act && t0 < Eloc) == 0ull) break;
        /* bring the next 16 bytes of the stream into every lane */
        const uint32_t f0 = sgpr(n0), f1 = sgpr(n1), f2 = sgpr(n2), f3 = sgpr(n3);
        c0 = wave_shl1(c0, f0); c1 = wave_shl1(c1, f1); c2 = wave_shl1(c2, f2); c3 = wave_shl1(c3, f3);
        n0 = wave_shl1(n0, 0u); n1 = wave_shl1(n1, 0u); n2 = wave_shl1(n2, 0u); n3 = wave_shl1(n3, 0u);
    }
}

/* Exact index (inside the packet) of the first 0x00 of this chunk; bz = ballot(lane has a zero byte) != 0. */
__device__ __forceinline__ uint32_t first_nul_pos(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint64_t bz,
                                                  uint32_t chunk_base)
{
    const uint32_t m0 = zero_byte_mask(w0), m1 = zero_byte_mask(w1), m2 = zero_byte_mask(w2), m3 = zero_byte_mask(w3);
    uint32_t zi = 12u + ((uint32_t)__builtin_ctz(m3 | 0x80000000u) >> 3);
    if (m2) zi = 8u + ((uint32_t)__builtin_ctz(m2) >> 3);
    if (m1) zi = 4u + ((uint32_t)__builtin_ctz(m1) >> 3);
    if (m0) zi = (uint32_t)__builtin_ctz(m0) >> 3;
    const uint32_t fl = (uint32_t)__builtin_ctzll(bz);
    const uint32_t zl = (uint32_t)__builtin_amdgcn_readlane((int)zi, (int)fl);
    return chunk_base + fl * KMP_LANE_BYTES + zl;
}

/*
 * MODE 0: filter + confirm (default).  MODE 1: KMP automaton on every chunk (no filter) -- an
 * independent second implementation used by the parity tests and for pathological inputs.
 *
 * Register ring: DEPTH chunk loads per wavefront, statically assigned (the loop body is unrolled
 * DEPTH times), so each s_waitcnt only waits for the chunk it consumes plus its successor (halo).
 */
template <int DEPTH, bool MASKED, int MODE, bool NT>
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_kernel(const uint8_t *__restrict__ arena, const uint64_t *__restrict__ pkt_off,
                const uint32_t *__restrict__ pkt_len, uint64_t n_pkts,
                const kmp_pattern_dev *__restrict__ patterns, const uint32_t *__restrict__ pat_ids,
                unsigned long long *__restrict__ partials)
{
    __shared__ kmp_pattern_dev s_pat;
    __shared__ unsigned long long s_wave_cnt[KMP_BLOCK_WAVES];

    const uint32_t pid = pat_ids[blockIdx.y];
    const kmp_pattern_dev *gp = patterns + pid;
    if (threadIdx.x < sizeof(kmp_pattern_dev) / 4u)
        reinterpret_cast<uint32_t *>(&s_pat)[threadIdx.x] = reinterpret_cast<const uint32_t *>(gp)[threadIdx.x];
    __syncthreads();

    const uint32_t m = gp->m, first = gp->first, mask = gp->mask;     /* wave-uniform scalar loads */
    const uint32_t lane = threadIdx.x & (KMP_WAVE - 1u);
    const uint32_t wave = sgpr(threadIdx.x >> 6);
    const uint64_t stride = (uint64_t)gridDim.x * KMP_BLOCK_WAVES;

    ChunkIter iss;                      /* next chunk whose load gets issued */
    iter_init(iss, (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + wave, pkt_off, pkt_len, n_pkts, stride);

    /* Register ring: buf[s] is refilled with the chunk DEPTH positions ahead right after it has
     * been consumed.  Loads return in order, so before step s "vmcnt(DEPTH-2)" guarantees that
     * buf[s] and buf[s+1] (the halo source) have landed while DEPTH-2 younger loads stay in flight.
     * An exhausted iterator keeps issuing a harmless re-read of arena[0:16). */
    u32x4 buf[DEPTH];
    Slot  meta[DEPTH];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
        meta[s] = slot_of(iss);
        ring_issue<NT>(buf[s], arena, iss, lane);
        iter_next(iss, pkt_off, pkt_len, n_pkts, stride);
    }

    uint32_t cnt = 0u;                  /* per-lane match count */
    uint32_t dead_seq = ~0u;            /* packet whose remaining chunks lie behind its first NUL */

    while (meta[0].cw & SLOT_VALID) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            ring_wait<DEPTH - 2>(buf[s], buf[(s + 1) % DEPTH]);
            const Slot me = meta[s];
            if ((me.cw & SLOT_VALID) && me.seq != dead_seq) {
                const uint4    v          = make_uint4(buf[s].x, buf[s].y, buf[s].z, buf[s].w);
                const bool     has_next   = (me.cw & SLOT_NEXT) != 0u;
                const u32x4    bn         = buf[(s + 1) % DEPTH];
                const uint4    vn         = has_next ? make_uint4(bn.x, bn.y, bn.z, bn.w) : make_uint4(0u, 0u, 0u, 0u);
                const uint32_t chunk_base = me.cw & SLOT_BASE;
                const uint32_t p0         = chunk_base + lane * KMP_LANE_BYTES;
                const uint32_t L          = me.L;
                const bool     inb        = p0 < ((L + 15u) & ~15u);     /* lane holds real slot bytes */
                const uint32_t w[5] = {v.x, v.y, v.z, v.w, wave_shl1(v.x, sgpr(vn.x))};

                const uint32_t zm = zero_byte_mask(w[0]) | zero_byte_mask(w[1]) | zero_byte_mask(w[2]) | zero_byte_mask(w[3]);
                const uint64_t bz = __ballot(inb && zm != 0u);
                if (bz != 0ull) dead_seq = me.seq;     /* the first NUL (or the slot padding) is in this chunk */

                if (MODE == 1) {
                    uint32_t Eloc = L;
                    if (bz != 0ull) Eloc = min(L, first_nul_pos(w[0], w[1], w[2], w[3], bz, chunk_base));
                    automaton_chunk(v, vn, p0 + m <= Eloc, p0, Eloc, m, s_pat, cnt);
                } else {
                    bool any = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t lo = w[q], hi = w[q + 1];
                        any |= is_cand<MASKED>(lo, first, mask);
                        any |= is_cand<MASKED>(__builtin_amdgcn_alignbyte(hi, lo, 1), first, mask);
                        any |= is_cand<MASKED>(__builtin_amdgcn_alignbyte(hi, lo, 2), first, mask);
                        any |= is_cand<MASKED>(__builtin_amdgcn_alignbyte(hi, lo, 3), first, mask);
                    }
                    any = any && inb;
                    if (__ballot(any) != 0ull) {
                        /* rare path: some lane saw the pattern's first bytes */
                        uint32_t Eloc = L;
                        if (bz != 0ull) Eloc = min(L, first_nul_pos(w[0], w[1], w[2], w[3], bz, chunk_base));
                        if (m <= 4u) {
                            /* the filter compared all m bytes: count the starts inside text[0:E) */
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const uint32_t lo = w[q], hi = w[q + 1];
#pragma unroll
                                for (int a = 0; a < 4; ++a) {
                                    const uint32_t d = a ? __builtin_amdgcn_alignbyte(hi, lo, a) : lo;
                                    if (is_cand<MASKED>(d, first, mask) && (p0 + (uint32_t)(4 * q + a) + m <= Eloc)) ++cnt;
                                }
                            }
                        } else {
                            const bool act = any && (p0 + m <= Eloc);
                            if (__ballot(act) != 0ull) automaton_chunk(v, vn, act, p0, Eloc, m, s_pat, cnt);
                        }
                    }
                }
            }
            /* refill this ring slot with the chunk DEPTH positions ahead */
            __builtin_amdgcn_sched_barrier(0);
            meta[s] = slot_of(iss);
            ring_issue<NT>(buf[s], arena, iss, lane);
            iter_next(iss, pkt_off, pkt_len, n_pkts, stride);
        }
    }
    /* drain: every ring register must be idle before the compiler reuses it */
#pragma unroll
    for (int s = 0; s < DEPTH; s += 2) ring_wait<0>(buf[s], buf[(s + 1) % DEPTH]);

    /* lanes -> wave -> block -> one partial per (pattern, block) */
    unsigned long long c64 = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c64 += __shfl_xor(c64, o);
    if (lane == 0u) s_wave_cnt[wave] = c64;
    __syncthreads();
    if (threadIdx.x == 0u) {
        unsigned long long t = 0ull;
#pragma unroll
        for (uint32_t i = 0; i < KMP_BLOCK_WAVES; ++i) t += s_wave_cnt[i];
        partials[(uint64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

/* ================================================================================================
 * Uniform-stride arenas (every payload the same length, slots back to back): flat streaming.
 *
 * Each wavefront owns a CONTIGUOUS run of packets, i.e. one contiguous byte range of the arena,
 * and streams it in 1 KiB chunks irrespective of packet boundaries: every lane always holds 16
 * useful bytes, consecutive chunk loads are consecutive addresses, and there is no per-packet
 * scalar work at all.  Because slots are 16-byte aligned a lane's 16 bytes belong to exactly one
 * packet; the lane tracks p0 = offset of its first byte inside that packet's slot with one
 * add + min per chunk.  Still one packet per wavefront at a time: the packets of a range are
 * scanned in order by the same wavefront, so the "first 0x00 ends the text" rule (serial.c:191)
 * is wave-local state (dead: the packet entering the chunk already had a NUL).
 *
 * A start offset s (lane position i, s = p0 + i) counts iff
 *     s + m <= L                       window inside the payload                  (serial.c:193,198)
 *     no 0x00 in the packet before s   strlen() stopped earlier otherwise         (serial.c:191)
 *     text[s : s+m] == pattern         (a NUL inside the window fails here: patterns are NUL-free)
 * ============================================================================================== */

/* v_min3_u32: written as asm because hipcc re-associates min(a, min(b, c)) chains into extra v_min_u32 */
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

/* Candidate test for the 16 start offsets of a lane, VALU only: min over (dword ^ first);
 * 12 v_alignbyte + 16 v_xor + 8 v_min3 and no scalar work. */
template <bool MASKED>
__device__ __forceinline__ uint32_t filter_min(const uint32_t (&w)[5], uint32_t first, uint32_t mask)
{
    uint32_t acc = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t lo = w[q], hi = w[q + 1];
        uint32_t x0 = lo ^ first;
        uint32_t x1 = __builtin_amdgcn_alignbyte(hi, lo, 1) ^ first;
        uint32_t x2 = __builtin_amdgcn_alignbyte(hi, lo, 2) ^ first;
        uint32_t x3 = __builtin_amdgcn_alignbyte(hi, lo, 3) ^ first;
        if (MASKED) { x0 &= mask; x1 &= mask; x2 &= mask; x3 &= mask; }
        acc = min3u(acc, x0, x1);
        acc = min3u(acc, x2, x3);
    }
    return acc;                 /* 0 iff some start offset of this lane shows the pattern's first bytes */
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
__device__ __forceinline__ void emit_match(bool ok, uint64_t pkt, uint32_t offset, const Emitter &e)
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
        if (slot < e.cap) e.out[slot] = make_uint4((uint32_t)pkt, (uint32_t)(pkt >> 32), offset, e.pattern);
    }
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
__device__ __forceinline__ void confirm_lanes(const uint32_t (&w)[5], uint4 v, u32x4 bn, int32_t maxi, uint32_t p0, uint32_t L,
                                              const PatConst &pc, const kmp_pattern_dev &sp, uint32_t &cnt, uint64_t pkt,
                                              const Emitter &em)
{
    const uint32_t m = pc.m;
    const uint64_t ba = ballot64(maxi >= 0);
    if (m <= 4u) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
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

template <int DEPTH, bool MASKED, bool NT, bool EMIT = false>
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_flat_kernel(const uint8_t *__restrict__ arena, uint64_t n_pkts, uint32_t stride, uint32_t L,
                     uint32_t pkts_per_wave, const kmp_pattern_dev *__restrict__ patterns,
                     const uint32_t *__restrict__ pat_ids, unsigned long long *__restrict__ partials, Emitter em)
{
    __shared__ kmp_pattern_dev s_pat;
    __shared__ unsigned long long s_wave_cnt[KMP_BLOCK_WAVES];

    const uint32_t pid = pat_ids[blockIdx.y];
    const kmp_pattern_dev *gp = patterns + pid;
    if (threadIdx.x < sizeof(kmp_pattern_dev) / 4u)
        reinterpret_cast<uint32_t *>(&s_pat)[threadIdx.x] = reinterpret_cast<const uint32_t *>(gp)[threadIdx.x];
    __syncthreads();

    const PatConst pc = load_pat_const(gp);
    const uint32_t m = pc.m, first = pc.first, mask = pc.mask;
    const uint32_t lane = threadIdx.x & (KMP_WAVE - 1u);
    const uint32_t wave = sgpr(threadIdx.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + wave;
    if (EMIT) em.pattern = pid;
    /* this wavefront's packets [k0, k1) = bytes [0, range) behind base */
    const uint64_t k0 = gw * pkts_per_wave;
    const uint64_t k1 = min(n_pkts, k0 + pkts_per_wave);
    const uint32_t range = (k0 < n_pkts) ? (uint32_t)(k1 - k0) * stride : 0u;      /* host guarantees < 2^31 */
    const uint8_t *base = arena + ((k0 < n_pkts) ? k0 * (uint64_t)stride : 0ull);
    const uint32_t step_mod = KMP_CHUNK % stride;                                    /* p0 advance per chunk (mod stride) */

    uint32_t cnt = 0u;
    if (range) {
        const i32x4    rsrc = make_rsrc(base, range);
        const uint32_t vo0 = lane * KMP_LANE_BYTES;
        u32x4 buf[DEPTH];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) flat_issue<NT>(buf[s], rsrc, vo0, (uint32_t)s * KMP_CHUNK);

        uint32_t p0 = vo0 % stride;          /* offset of this lane's first byte inside its packet's slot */
        bool     dead = false;               /* the packet that enters the chunk already had a 0x00      */
        uint32_t cb = 0u;                    /* byte offset of the chunk being consumed                   */

        while (cb < range) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                ring_wait<DEPTH - 2>(buf[s], buf[(s + 1) % DEPTH]);
                if (cb < range) {
                    const uint4    v   = make_uint4(buf[s].x, buf[s].y, buf[s].z, buf[s].w);
                    const u32x4    bn  = buf[(s + 1) % DEPTH];                /* next chunk (zeros past the range) */
                    const uint32_t w[5] = {v.x, v.y, v.z, v.w, wave_shl1(v.x, sgpr(bn.x))};

                    const uint32_t zm = zero_byte_mask(w[0]) | zero_byte_mask(w[1]) | zero_byte_mask(w[2]) | zero_byte_mask(w[3]);
                    const uint32_t fm = filter_min<MASKED>(w, first, mask);
                    const uint64_t zl = ballot64(zm != 0u);                   /* lanes holding a 0x00           */
                    const uint64_t st = ballot64(p0 == 0u);                   /* lanes where a packet starts    */
                    const uint64_t cl = ballot64(fm == 0u);                   /* lanes with a candidate         */
                    const bool dead_in = dead;
                    /* carry for the next chunk: is there a 0x00 at or after the last packet start of this chunk? */
                    if (zl == 0ull) { if (st != 0ull) dead = false; }
                    else            dead = (st == 0ull) ? true : ((zl >> (63u - (uint32_t)__builtin_clzll(st))) != 0ull);

                    if (cl != 0ull) {
                        /* rare path.  maxi = largest start index (0..15) of this lane that still counts:
                         * window inside the payload, no 0x00 before it, lane has a candidate at all. */
                        int32_t maxi = (int32_t)L - (int32_t)m - (int32_t)p0;
                        if (fm != 0u) maxi = -1;
                        if (zl != 0ull || dead_in) maxi = nul_limit(maxi, w, zl, st, dead_in, lane);
                        const uint64_t pkt = EMIT ? (k0 + (uint64_t)((cb + vo0 - p0) / stride)) : 0ull;
                        confirm_lanes<MASKED, EMIT>(w, v, bn, maxi, p0, L, pc, s_pat, cnt, pkt, em);
                    }
                    /* this lane's position inside its packet, one chunk further */
                    p0 += step_mod;
                    p0 = min(p0, p0 - stride);               /* unsigned: subtracts stride iff p0 >= stride */
                }
                __builtin_amdgcn_sched_barrier(0);
                flat_issue<NT>(buf[s], rsrc, vo0, cb + (uint32_t)DEPTH * KMP_CHUNK);
                cb += KMP_CHUNK;
            }
        }
#pragma unroll
        for (int s = 0; s < DEPTH; s += 2) ring_wait<0>(buf[s], buf[(s + 1) % DEPTH]);
    }

    unsigned long long c64 = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c64 += __shfl_xor(c64, o);
    if (lane == 0u) s_wave_cnt[wave] = c64;
    __syncthreads();
    if (threadIdx.x == 0u) {
        unsigned long long t = 0ull;
#pragma unroll
        for (uint32_t i = 0; i < KMP_BLOCK_WAVES; ++i) t += s_wave_cnt[i];
        partials[(uint64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

/* ================================================================================================
 * Packed arenas of arbitrary payload lengths (real captures, mixed-length traffic): the same flat
 * streaming as above, with the two things the uniform kernel gets from arithmetic taken from small
 * side tables built once when the arena is loaded:
 *   - bitmap: one bit per 16-byte slot of the arena, set where a payload starts (0.8 % of the arena
 *     size); the 64 bits of a chunk ARE the packet-start ballot, fetched by one scalar load per chunk;
 *   - plan: per wavefront the first packet index and byte offset of its range.  Ranges are cut at
 *     packet starts at equal BYTE distance, which is the load balancing for mixed lengths
 *     (BASELINE configs[4]): every wavefront streams the same number of bytes whatever the lengths.
 * Lanes learn their packet (index, offset, length) only on the rare path, from the start ballot:
 * packet index = packets started before this chunk + starts at lanes <= own lane.
 * ============================================================================================== */
struct kmp_plan_entry { uint64_t k; uint64_t off; };

__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_build_bitmap_kernel(const uint64_t *__restrict__ pkt_off, uint64_t n, unsigned long long *__restrict__ bitmap)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t slot = pkt_off[k] >> 4;
        atomicOr(bitmap + (slot >> 6), 1ull << (slot & 63ull));
    }
}

/* plan[w] = first packet whose offset is >= off[0] + w * bytes_per_wave (w = 0..nwaves); plan[nwaves] = {n, end}. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_plan_kernel(const uint64_t *__restrict__ pkt_off, const uint32_t *__restrict__ pkt_len, uint64_t n, uint64_t nwaves,
                uint64_t bytes_per_wave, kmp_plan_entry *__restrict__ plan)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwaves) return;
    const uint64_t l16 = ((uint64_t)pkt_len[n - 1] + 15ull) & ~15ull;
    const uint64_t end = pkt_off[n - 1] + (l16 < 16ull ? 16ull : l16);
    if (w == nwaves) { plan[w].k = n; plan[w].off = end; return; }
    const uint64_t target = pkt_off[0] + w * bytes_per_wave;
    uint64_t lo = 0, hi = n;                       /* lower_bound over the (increasing) offsets */
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pkt_off[mid] < target) lo = mid + 1; else hi = mid;
    }
    plan[w].k = lo;
    plan[w].off = (lo < n) ? pkt_off[lo] : end;
}

template <int DEPTH, bool MASKED, bool NT, bool EMIT = false>
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_packed_kernel(const uint8_t *__restrict__ arena, const uint64_t *__restrict__ pkt_off,
                       const uint32_t *__restrict__ pkt_len, const unsigned long long *__restrict__ bitmap,
                       const kmp_plan_entry *__restrict__ plan, const kmp_pattern_dev *__restrict__ patterns,
                       const uint32_t *__restrict__ pat_ids, unsigned long long *__restrict__ partials, Emitter em)
{
    __shared__ kmp_pattern_dev s_pat;
    __shared__ unsigned long long s_wave_cnt[KMP_BLOCK_WAVES];

    const uint32_t pid = pat_ids[blockIdx.y];
    const kmp_pattern_dev *gp = patterns + pid;
    if (threadIdx.x < sizeof(kmp_pattern_dev) / 4u)
        reinterpret_cast<uint32_t *>(&s_pat)[threadIdx.x] = reinterpret_cast<const uint32_t *>(gp)[threadIdx.x];
    __syncthreads();

    const PatConst pc = load_pat_const(gp);
    const uint32_t m = pc.m, first = pc.first, mask = pc.mask;
    const uint32_t lane = threadIdx.x & (KMP_WAVE - 1u);
    const uint32_t wave = sgpr(threadIdx.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + wave;
    if (EMIT) em.pattern = pid;
    const uint64_t k0 = plan[gw].k, k1 = plan[gw + 1].k;
    const uint64_t off0 = plan[gw].off;
    const uint32_t range = (k1 > k0) ? (uint32_t)(plan[gw + 1].off - off0) : 0u;    /* planner guarantees < 2^31 */

    uint32_t cnt = 0u;
    if (range) {
        const uint8_t *base = arena + off0;
        const i32x4    rsrc = make_rsrc(base, range);
        const uint32_t vo0 = lane * KMP_LANE_BYTES;
        /* packet-start bits of chunk j: bits [b0 + 64 j, +64) of the bitmap = words wi0+j, wi0+j+1 shifted by sh */
        const uint64_t b0 = off0 >> 4;
        const unsigned long long *bw = bitmap + (b0 >> 6);
        const uint32_t sh = (uint32_t)(b0 & 63ull);

        u32x4 buf[DEPTH];
        unsigned long long hiw[DEPTH];       /* bitmap word wi0 + j + 1 of the chunk in ring slot s */
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            flat_issue<NT>(buf[s], rsrc, vo0, (uint32_t)s * KMP_CHUNK);
            hiw[s] = bw[s + 1];
        }
        unsigned long long low = bw[0];      /* bitmap word wi0 + j of the chunk being consumed */
        uint64_t kbase = k0 - 1ull;          /* index of the last packet started before the chunk */
        bool     dead = false;
        uint32_t cb = 0u, j = 0u;

        while (cb < range) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                ring_wait<DEPTH - 2>(buf[s], buf[(s + 1) % DEPTH]);
                if (cb < range) {
                    const uint4    v   = make_uint4(buf[s].x, buf[s].y, buf[s].z, buf[s].w);
                    const u32x4    bn  = buf[(s + 1) % DEPTH];
                    const uint32_t w[5] = {v.x, v.y, v.z, v.w, wave_shl1(v.x, sgpr(bn.x))};
                    const unsigned long long hi = hiw[s];
                    const uint64_t st = sh ? ((low >> sh) | (hi << (64u - sh))) : low;   /* lanes where a packet starts */
                    low = hi;

                    const uint32_t zm = zero_byte_mask(w[0]) | zero_byte_mask(w[1]) | zero_byte_mask(w[2]) | zero_byte_mask(w[3]);
                    const uint32_t fm = filter_min<MASKED>(w, first, mask);
                    const uint64_t zl = ballot64(zm != 0u);
                    const uint64_t cl = ballot64(fm == 0u);
                    const bool dead_in = dead;
                    if (zl == 0ull) { if (st != 0ull) dead = false; }
                    else            dead = (st == 0ull) ? true : ((zl >> (63u - (uint32_t)__builtin_clzll(st))) != 0ull);

                    if (cl != 0ull) {
                        /* rare path: which packet does a candidate lane sit in? */
                        int32_t  maxi = -1;
                        uint32_t p0 = 0u, L = 0u;
                        uint64_t kl = 0ull;
                        if (fm == 0u) {
                            const uint64_t le = (2ull << lane) - 1ull;                  /* lanes <= own (lane 63: all ones) */
                            kl = kbase + (uint64_t)__builtin_popcountll(st & le);
                            const uint64_t po = pkt_off[kl];
                            L  = pkt_len[kl];
                            p0 = (uint32_t)(off0 + cb + vo0 - po);
                            maxi = (int32_t)L - (int32_t)m - (int32_t)p0;
                        }
                        if (zl != 0ull || dead_in) maxi = nul_limit(maxi, w, zl, st, dead_in, lane);
                        confirm_lanes<MASKED, EMIT>(w, v, bn, maxi, p0, L, pc, s_pat, cnt, kl, em);
                    }
                    kbase += (uint64_t)__builtin_popcountll(st);
                }
                __builtin_amdgcn_sched_barrier(0);
                flat_issue<NT>(buf[s], rsrc, vo0, cb + (uint32_t)DEPTH * KMP_CHUNK);
                hiw[s] = bw[j + (uint32_t)DEPTH + 1u];
                cb += KMP_CHUNK;
                ++j;
            }
        }
#pragma unroll
        for (int s = 0; s < DEPTH; s += 2) ring_wait<0>(buf[s], buf[(s + 1) % DEPTH]);
    }

    unsigned long long c64 = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c64 += __shfl_xor(c64, o);
    if (lane == 0u) s_wave_cnt[wave] = c64;
    __syncthreads();
    if (threadIdx.x == 0u) {
        unsigned long long t = 0ull;
#pragma unroll
        for (uint32_t i = 0; i < KMP_BLOCK_WAVES; ++i) t += s_wave_cnt[i];
        partials[(uint64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

/* ================================================================================================
 * Fused multi-pattern pass: every pattern of 2..20 bytes in ONE read of a packed
 * arena -- the reference re-reads every payload once per pattern (serial.c:154, openmp_data.c:163).
 *
 * Same streaming skeleton as kmp_scan_packed_kernel (buffer-load ring, packet-start bitmap, byte-
 * balanced plan).  Per chunk:
 *   - rem = payload bytes left from the lane's first byte (uniform loop over the packet starts of the
 *     chunk; lengths come by scalar loads, offsets are implied by the bitmap because the arena is packed);
 *   - level 1: the 2-byte window at each of the 16 start offsets indexes a 64 Kbit LDS bitmap "some
 *     pattern starts with these two bytes" -> 16-bit hit mask per lane;
 *   - level 2, per start offset that has a hit in some lane: hash the 2 bytes to a bucket, walk the
 *     bucket's short list of patterns, compare up to 20 bytes dword-wise with byte masks (text from
 *     registers, pattern records from LDS), check window-in-payload and the strlen() rule, and bump
 *     the pattern's counter in LDS.  Counters go to partials[unique pattern][block] at the end.
 * ============================================================================================== */
template <int DEPTH, bool NT>
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_multi_kernel(const uint8_t *__restrict__ arena, const uint32_t *__restrict__ pkt_len,
                      const unsigned long long *__restrict__ bitmap, const kmp_plan_entry *__restrict__ plan,
                      const uint32_t *__restrict__ tables, uint32_t table_words, uint32_t n_unique,
                      unsigned long long *__restrict__ partials)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_tab[];    /* tables, counters, then one chunk window per wavefront */
    uint32_t *s_cnt = s_tab + table_words;
    uint32_t *s_win = s_tab + ((table_words + n_unique + 3u) & ~3u);
    for (uint32_t i = threadIdx.x; i < table_words; i += KMP_BLOCK_THREADS) s_tab[i] = tables[i];
    for (uint32_t i = threadIdx.x; i < n_unique; i += KMP_BLOCK_THREADS) s_cnt[i] = 0u;
    __syncthreads();
    const uint16_t *s_bucket = reinterpret_cast<const uint16_t *>(s_tab + KMP_MULTI_BUCKET_W0);
    const uint32_t *s_entry  = s_tab + KMP_MULTI_ENTRY_W0;

    const uint32_t lane = threadIdx.x & (KMP_WAVE - 1u);
    const uint32_t wave = sgpr(threadIdx.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + wave;
    const uint64_t k0 = plan[gw].k, k1 = plan[gw + 1].k;
    const uint64_t off0 = plan[gw].off;
    const uint32_t range = (k1 > k0) ? (uint32_t)(plan[gw + 1].off - off0) : 0u;

    if (range) {
        const i32x4    rsrc = make_rsrc(arena + off0, range);
        const uint32_t vo0 = lane * KMP_LANE_BYTES;
        const uint64_t b0 = off0 >> 4;
        const unsigned long long *bw = bitmap + (b0 >> 6);
        const uint32_t sh = (uint32_t)(b0 & 63ull);

        u32x4 buf[DEPTH];
        unsigned long long hiw[DEPTH];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            flat_issue<NT>(buf[s], rsrc, vo0, (uint32_t)s * KMP_CHUNK);
            hiw[s] = bw[s + 1];
        }
        unsigned long long low = bw[0];
        uint64_t kcur = k0 - 1ull;           /* last packet that has started                                   */
        int32_t  remc = 0;                   /* payload bytes of that packet left at the chunk's first byte     */
        bool     dead = false;
        uint32_t cb = 0u, j = 0u;

        while (cb < range) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                ring_wait<DEPTH - 2>(buf[s], buf[(s + 1) % DEPTH]);
                if (cb < range) {
                    const uint4 v  = make_uint4(buf[s].x, buf[s].y, buf[s].z, buf[s].w);
                    const u32x4 bn = buf[(s + 1) % DEPTH];
                    const unsigned long long hi = hiw[s];
                    uint64_t st = sh ? ((low >> sh) | (hi << (64u - sh))) : low;
                    low = hi;
                    const uint32_t left = range - cb;
                    if (left < KMP_CHUNK) st &= (1ull << (left >> 4)) - 1ull;        /* bits past the range belong to the next wavefront */

                    const uint32_t w[5] = {v.x, v.y, v.z, v.w, wave_shl1(v.x, sgpr(bn.x))};

                    const uint32_t zm = zero_byte_mask(w[0]) | zero_byte_mask(w[1]) | zero_byte_mask(w[2]) | zero_byte_mask(w[3]);
                    const uint64_t zl = ballot64(zm != 0u);
                    const bool dead_in = dead;
                    if (zl == 0ull) { if (st != 0ull) dead = false; }
                    else            dead = (st == 0ull) ? true : ((zl >> (63u - (uint32_t)__builtin_clzll(st))) != 0ull);

                    /* payload bytes left from this lane's first byte (<= 0: slot padding) */
                    int32_t rem = remc - (int32_t)vo0;
                    int32_t remn = remc - (int32_t)KMP_CHUNK;
                    for (uint64_t sb = st; sb != 0ull; sb &= sb - 1ull) {
                        const uint32_t sl = (uint32_t)__builtin_ctzll(sb);
                        ++kcur;
                        const int32_t top = (int32_t)pkt_len[kcur] + (int32_t)(sl * KMP_LANE_BYTES);
                        if (lane >= sl) rem = top - (int32_t)vo0;
                        remn = top - (int32_t)KMP_CHUNK;
                    }
                    remc = remn;

                    /* level 1: which start offsets begin with the first two bytes of some pattern? */
                    uint32_t hm = 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const uint32_t d0 = a ? __builtin_amdgcn_alignbyte(w[q + 1], w[q], a) : w[q];
                            const uint32_t bi = (d0 & 0xFFFFu) * 0x9E3Bu;                              /* KMP_MULTI_BIT: bank spreading */
                            const uint32_t word = s_tab[(bi >> 5) & 0x7FFu];
                            hm |= ((word >> (bi & 31u)) & 1u) << (4 * q + a);
                        }
                    }
                    if (ballot64(hm != 0u) != 0ull) {
                        /* keep only the start offsets that can count: no 0x00 before them (strlen rule) and at
                         * least the shortest pattern still inside the payload */
                        int32_t lim = min(15, rem - (int32_t)KMP_MULTI_MIN_LEN);
                        if (zl != 0ull || dead_in) lim = min(lim, nul_limit(15, w, zl, st, dead_in, lane));
                        hm = (lim < 0) ? 0u : (hm & ((2u << lim) - 1u));
                        if (ballot64(hm != 0u) != 0ull) {
                            /* level 2.  Stage the chunk + 32 bytes of halo in this wavefront's LDS window so that a
                             * lane can fetch the 20 bytes behind ANY of its start offsets, then let every lane walk
                             * its own hits: iterations = the largest hit count of a lane, not 16. */
                            uint32_t *win = s_win + wave * (KMP_CHUNK / 4u + 8u);
                            *reinterpret_cast<uint4 *>(win + lane * 4u) = v;
                            if (lane < 2u) *reinterpret_cast<uint4 *>(win + KMP_CHUNK / 4u + lane * 4u) = make_uint4(bn.x, bn.y, bn.z, bn.w);
                            while (ballot64(hm != 0u) != 0ull) {
                                if (hm != 0u) {
                                    const uint32_t i = (uint32_t)__builtin_ctz(hm);
                                    hm &= hm - 1u;
                                    const uint32_t o = vo0 + i;                         /* byte offset inside the chunk window */
                                    const uint32_t *src = win + (o >> 2);
                                    const uint32_t sa = o & 3u;
                                    const uint32_t r0 = src[0], r1 = src[1];
                                    const uint32_t T0 = __builtin_amdgcn_alignbyte(r1, r0, sa);
                                    const uint32_t b2 = (T0 >> 16) & 0xFFu;             /* third text byte: cheap pre-check per entry */
                                    uint32_t e = s_bucket[KMP_MULTI_HASH(T0 & 0xFFFFu)];
                                    while (e != 0xFFFFu) {
                                        const uint32_t ent = s_entry[e];
                                        const uint32_t pb2 = (ent >> 8) & 0xFFu;
                                        if (pb2 == 0u || pb2 == b2) {
                                            /* rare: fetch the other 16 text bytes and the pattern record */
                                            const uint32_t uid = ent & 0xFFu;
                                            const uint32_t *rec = s_tab + KMP_MULTI_REC_W0 + uid * KMP_MULTI_REC_WORDS;
                                            uint32_t diff = (T0 ^ rec[0]) & rec[5];
                                            uint32_t prev = r1;
#pragma unroll
                                            for (int d = 1; d < 5; ++d) {
                                                const uint32_t nx = src[d + 1];
                                                diff |= (__builtin_amdgcn_alignbyte(nx, prev, sa) ^ rec[d]) & rec[5 + d];
                                                prev = nx;
                                            }
                                            if (diff == 0u && (int32_t)(i + rec[10]) <= rem) atomicAdd(&s_cnt[uid], 1u);
                                        }
                                        e = (ent & 0x80000000u) ? 0xFFFFu : e + 1u;
                                    }
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                flat_issue<NT>(buf[s], rsrc, vo0, cb + (uint32_t)DEPTH * KMP_CHUNK);
                hiw[s] = bw[j + (uint32_t)DEPTH + 1u];
                cb += KMP_CHUNK;
                ++j;
            }
        }
#pragma unroll
        for (int s = 0; s < DEPTH; s += 2) ring_wait<0>(buf[s], buf[(s + 1) % DEPTH]);
    }

    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_unique; i += KMP_BLOCK_THREADS)
        partials[(uint64_t)i * gridDim.x + blockIdx.x] = s_cnt[i];
}

/* ================================================================================================
 * On-device payload extraction: the raw capture file is uploaded as it is; the host
 * only walks the 16-byte record headers (frame offset + caplen).  Replaces the extraction phase
 * openmp_data.c:128-147 (parallel-for: dump_*_packet + malloc + memcpy per payload) by
 *   kmp_extract_kernel      one thread per frame: the accept/reject rule and payload bounds of
 *                           dump_UDP_packet / dump_TCP_packet (packet_dumping.h:87-139, 150-188)
 *   kmp_scan_*_kernel       exclusive scan of {slot bytes, valid} -> slot offsets and packet indices
 *   kmp_gather_kernel       one wavefront per payload: copy it to its 16-byte aligned slot, zero the pad
 * ============================================================================================== */
#define KMP_SCAN_ITEMS 4u                                         /* items per thread in the block scan */
#define KMP_SCAN_TILE  (KMP_BLOCK_THREADS * KMP_SCAN_ITEMS)

__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_extract_kernel(const uint8_t *__restrict__ file, const uint64_t *__restrict__ frame_off,
                   const uint32_t *__restrict__ caplen, uint64_t n, int tcp,
                   uint32_t *__restrict__ poff, uint32_t *__restrict__ plen)      /* plen = 0xFFFFFFFF: rejected */
{
    for (uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n; f += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *p = file + frame_off[f];
        const uint32_t cl = caplen[f];
        uint32_t o = 0u, l = 0xFFFFFFFFu;
        if (!tcp) {                                                    /* packet_dumping.h:87-139 */
            if (cl >= 14u + 20u) {                                     /* :94, :102 */
                const uint32_t rest = cl - 14u;
                const uint32_t ihl = (uint32_t)(p[14] & 0x0Fu) << 2;   /* :108 */
                if (rest >= ihl && p[23] == 17u && rest - ihl >= 8u) { /* :110, :116, :125 */
                    o = 14u + ihl + 8u;                                /* :133 */
                    l = rest - ihl - 8u;                               /* :136 */
                }
            }
        } else if (cl >= 15u) {                                        /* packet_dumping.h:150-188 */
            const uint32_t size_ip = (uint32_t)(p[14] & 0x0Fu) << 2;   /* :165 */
            const uint32_t tcp_at = 14u + size_ip;
            if (size_ip >= 20u && cl >= tcp_at + 13u) {                /* :166 */
                const uint32_t size_tcp = (uint32_t)(p[tcp_at + 12u] >> 4) << 2;    /* :175 */
                if (size_tcp >= 20u && cl >= tcp_at + size_tcp) {      /* :176; wrap of the reference's unsigned length rejected */
                    o = tcp_at + size_tcp;                             /* :181 */
                    l = cl - o;                                        /* :184 */
                }
            }
        }
        poff[f] = o; plen[f] = l;
    }
}

__device__ __forceinline__ uint64_t slot_bytes(uint32_t l) { return l == 0xFFFFFFFFu ? 0ull : (l ? (((uint64_t)l + 15ull) & ~15ull) : 16ull); }

/* Block-local exclusive scan of {slot bytes, valid}: local prefixes per frame, totals per block. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_local_kernel(const uint32_t *__restrict__ plen, uint64_t n, uint64_t *__restrict__ loc_off,
                      uint32_t *__restrict__ loc_idx, uint64_t *__restrict__ blk_bytes, uint32_t *__restrict__ blk_cnt)
{
    __shared__ uint64_t s_b[KMP_BLOCK_THREADS];
    __shared__ uint32_t s_c[KMP_BLOCK_THREADS];
    const uint64_t base = (uint64_t)blockIdx.x * KMP_SCAN_TILE + (uint64_t)threadIdx.x * KMP_SCAN_ITEMS;
    uint64_t b[KMP_SCAN_ITEMS], tb = 0;
    uint32_t c[KMP_SCAN_ITEMS], tc = 0;
#pragma unroll
    for (uint32_t i = 0; i < KMP_SCAN_ITEMS; i++) {
        const uint32_t l = (base + i < n) ? plen[base + i] : 0xFFFFFFFFu;
        b[i] = tb; c[i] = tc;
        tb += slot_bytes(l); tc += (l != 0xFFFFFFFFu);
    }
    s_b[threadIdx.x] = tb; s_c[threadIdx.x] = tc;
    __syncthreads();
    for (uint32_t d = 1; d < KMP_BLOCK_THREADS; d <<= 1) {            /* Hillis-Steele inclusive scan of the thread totals */
        const uint64_t vb = threadIdx.x >= d ? s_b[threadIdx.x - d] : 0ull;
        const uint32_t vc = threadIdx.x >= d ? s_c[threadIdx.x - d] : 0u;
        __syncthreads();
        s_b[threadIdx.x] += vb; s_c[threadIdx.x] += vc;
        __syncthreads();
    }
    const uint64_t eb = s_b[threadIdx.x] - tb;
    const uint32_t ec = s_c[threadIdx.x] - tc;
#pragma unroll
    for (uint32_t i = 0; i < KMP_SCAN_ITEMS; i++)
        if (base + i < n) { loc_off[base + i] = eb + b[i]; loc_idx[base + i] = ec + c[i]; }
    if (threadIdx.x == KMP_BLOCK_THREADS - 1u) { blk_bytes[blockIdx.x] = s_b[threadIdx.x]; blk_cnt[blockIdx.x] = s_c[threadIdx.x]; }
}

/* One block: exclusive scan of the block totals in place; totals[0] = bytes, totals[1] = payloads. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_totals_kernel(uint64_t *__restrict__ blk_bytes, uint32_t *__restrict__ blk_cnt, uint32_t nblk,
                       unsigned long long *__restrict__ totals)
{
    __shared__ uint64_t s_b[KMP_BLOCK_THREADS];
    __shared__ uint32_t s_c[KMP_BLOCK_THREADS];
    uint64_t cb = 0;
    uint64_t cc = 0;
    for (uint32_t base = 0; base < nblk; base += KMP_BLOCK_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t vb0 = i < nblk ? blk_bytes[i] : 0ull;
        const uint32_t vc0 = i < nblk ? blk_cnt[i] : 0u;
        s_b[threadIdx.x] = vb0; s_c[threadIdx.x] = vc0;
        __syncthreads();
        for (uint32_t d = 1; d < KMP_BLOCK_THREADS; d <<= 1) {
            const uint64_t vb = threadIdx.x >= d ? s_b[threadIdx.x - d] : 0ull;
            const uint32_t vc = threadIdx.x >= d ? s_c[threadIdx.x - d] : 0u;
            __syncthreads();
            s_b[threadIdx.x] += vb; s_c[threadIdx.x] += vc;
            __syncthreads();
        }
        if (i < nblk) { blk_bytes[i] = cb + s_b[threadIdx.x] - vb0; blk_cnt[i] = (uint32_t)(cc + s_c[threadIdx.x] - vc0); }
        cb += s_b[KMP_BLOCK_THREADS - 1u]; cc += s_c[KMP_BLOCK_THREADS - 1u];
        __syncthreads();
    }
    if (threadIdx.x == 0u) { totals[0] = cb; totals[1] = cc; }
}

/* Index of the payload arena + source address of every payload. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scatter_index_kernel(const uint64_t *__restrict__ frame_off, const uint32_t *__restrict__ poff,
                         const uint32_t *__restrict__ plen, const uint64_t *__restrict__ loc_off,
                         const uint32_t *__restrict__ loc_idx, const uint64_t *__restrict__ blk_bytes,
                         const uint32_t *__restrict__ blk_cnt, uint64_t n, uint64_t *__restrict__ pkt_off,
                         uint32_t *__restrict__ pkt_len, uint64_t *__restrict__ src_off)
{
    for (uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n; f += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = plen[f];
        if (l == 0xFFFFFFFFu) continue;                            /* serial.c:138-140: rejected frames are skipped */
        const uint64_t blk = f / KMP_SCAN_TILE;
        const uint64_t k = (uint64_t)blk_cnt[blk] + loc_idx[f];
        pkt_off[k] = blk_bytes[blk] + loc_off[f];
        pkt_len[k] = l;
        src_off[k] = frame_off[f] + poff[f];
    }
}

/* serial.c:125-127 (malloc + memcpy per payload): one wavefront copies one payload into its slot. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_gather_kernel(const uint8_t *__restrict__ file, const uint64_t *__restrict__ src_off, const uint64_t *__restrict__ pkt_off,
                  const uint32_t *__restrict__ pkt_len, uint64_t n_pkts, uint8_t *__restrict__ arena)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nw = (uint64_t)gridDim.x * KMP_BLOCK_WAVES;
    for (uint64_t k = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + (threadIdx.x >> 6); k < n_pkts; k += nw) {
        const uint8_t *src = file + src_off[k];
        uint32_t *dst = reinterpret_cast<uint32_t *>(arena + pkt_off[k]);
        const uint32_t L = pkt_len[k];
        const uint32_t nw4 = (L ? ((L + 15u) & ~15u) : 16u) / 4u;
        for (uint32_t w = lane; w < nw4; w += 64u) {
            const uint32_t i0 = w * 4u;
            uint32_t v = 0u;
            if (i0 + 4u <= L) __builtin_memcpy(&v, src + i0, 4);           /* unaligned source */
            else
                for (uint32_t b = 0; b < 4u && i0 + b < L; b++) v |= (uint32_t)src[i0 + b] << (8u * b);
            dst[w] = v;
        }
    }
}

/* counts[pat_ids[y]] = sum of that pattern's block partials. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_reduce_kernel(const unsigned long long *__restrict__ partials, uint32_t blocks_x,
                  const uint32_t *__restrict__ pat_ids, const uint32_t *__restrict__ rows,
                  unsigned long long *__restrict__ counts, int accumulate)
{
    __shared__ unsigned long long s[KMP_BLOCK_WAVES];
    unsigned long long t = 0ull;
    /* rows == nullptr: row y of partials belongs to pat_ids[y]; else row rows[y] (fused pass: duplicates share a row) */
    const unsigned long long *row = partials + (uint64_t)(rows ? rows[blockIdx.x] : blockIdx.x) * blocks_x;
    for (uint32_t i = threadIdx.x; i < blocks_x; i += KMP_BLOCK_THREADS) t += row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63u) == 0u) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0u) {
        unsigned long long r = 0ull;
        for (uint32_t i = 0; i < KMP_BLOCK_WAVES; ++i) r += s[i];
        /* accumulate: counts keep adding up over the batches of a streamed capture (openmp_task.c:172-175) */
        unsigned long long *dst = counts + pat_ids[blockIdx.x];
        *dst = accumulate ? *dst + r : r;
    }
}

/* Layout contract of kmpgpu.h for a device-resident index: err[0] |= 1 misaligned, |= 2 out of
 * bounds, |= 4 length >= 2^30; err[1] & 1: the arena is NOT uniform-stride, err[1] & 2: slots are NOT
 * packed back to back; info[0] += sum(len),
 * info[1..4] = offset of payload 0, stride (offset 1 - offset 0), length of payload 0, end of the last slot. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_validate_index_kernel(const uint64_t *__restrict__ pkt_off, const uint32_t *__restrict__ pkt_len, uint64_t n,
                          uint64_t arena_bytes, uint32_t *__restrict__ err, unsigned long long *__restrict__ info)
{
    __shared__ unsigned long long s[KMP_BLOCK_WAVES];
    unsigned long long sum = 0ull;
    uint32_t e = 0u, nonuni = 0u;
    const uint64_t off0 = pkt_off[0];
    const uint32_t len0 = pkt_len[0];
    const uint64_t l016 = ((uint64_t)len0 + 15ull) & ~15ull;
    const uint64_t stride = (n > 1) ? pkt_off[1] - off0 : (l016 < 16ull ? 16ull : l016);
    if ((n > 1 && pkt_off[1] < off0) || stride < (l016 < 16ull ? 16ull : l016) || (stride & 15ull)) nonuni |= 1u;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o = pkt_off[k];
        const uint64_t l = pkt_len[k];
        if (o & 15ull) e |= 1u;
        if (l >= (1ull << 30)) e |= 4u;
        const uint64_t l16 = (l + 15ull) & ~15ull;
        if (o > arena_bytes || (l16 < 16ull ? 16ull : l16) > arena_bytes - o) e |= 2u;   /* every payload owns >= 16 readable bytes */
        if (l != len0 || o != off0 + k * stride) nonuni |= 1u;
        if (k + 1 < n && pkt_off[k + 1] != o + (l16 < 16ull ? 16ull : l16)) nonuni |= 2u;      /* not packed back to back */
        sum += l;
    }
    if (e) atomicOr(err, e);
    if (nonuni) atomicOr(err + 1, nonuni);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63u) == 0u) s[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0u) {
        unsigned long long r = 0ull;
        for (uint32_t i = 0; i < KMP_BLOCK_WAVES; ++i) r += s[i];
        atomicAdd(info, r);
        if (blockIdx.x == 0u) {
            info[1] = off0; info[2] = stride; info[3] = len0;
            const uint64_t ll16 = ((uint64_t)pkt_len[n - 1] + 15ull) & ~15ull;
            info[4] = pkt_off[n - 1] + (ll16 < 16ull ? 16ull : ll16);
        }
    }
}

/* Synthetic payloads (kmp_synth.h): one wavefront per packet, one dword per lane per step. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_synth_fill_kernel(uint8_t *__restrict__ arena, const uint64_t *__restrict__ pkt_off,
                      const uint32_t *__restrict__ pkt_len, uint64_t first_pkt_id, uint64_t n, kmp_synth_params sp)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave0 = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * KMP_BLOCK_WAVES;
    for (uint64_t i = wave0; i < n; i += nwaves) {
        const uint64_t id = first_pkt_id + i;
        const uint32_t len = pkt_len[i];
        uint32_t *slot = reinterpret_cast<uint32_t *>(arena + pkt_off[i]);
        const uint32_t key = kmp_synth_pkt_key(sp.seed, id);
        uint32_t pos = 0u;
        const int planted = kmp_synth_plant(&sp, id, len, &pos);
        const uint32_t nw = ((len + 15u) & ~15u) / 4u;
        for (uint32_t w = lane; w < nw; w += 64u) slot[w] = kmp_synth_slot_word(&sp, key, w, len, planted, pos);
    }
}

__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_fixed_index_kernel(uint64_t *__restrict__ pkt_off, uint32_t *__restrict__ pkt_len, uint64_t n, uint32_t len,
                       uint64_t stride)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        pkt_off[k] = k * stride;
        pkt_len[k] = len;
    }
}

template <int DEPTH, bool MASKED, int MODE>
hipError_t launch_scan_t(const kmp_scan_args &a, hipStream_t st)
{
    dim3 grid(a.blocks_x, a.n_ids), block(KMP_BLOCK_THREADS);
    if (a.nontemporal)
        hipLaunchKernelGGL((kmp_scan_kernel<DEPTH, MASKED, MODE, true>), grid, block, 0, st, a.arena, a.pkt_off, a.pkt_len,
                           a.n_pkts, a.patterns, a.pat_ids, a.partials);
    else
        hipLaunchKernelGGL((kmp_scan_kernel<DEPTH, MASKED, MODE, false>), grid, block, 0, st, a.arena, a.pkt_off, a.pkt_len,
                           a.n_pkts, a.patterns, a.pat_ids, a.partials);
    return hipGetLastError();
}

template <bool MASKED, int MODE>
hipError_t launch_scan_d(const kmp_scan_args &a, hipStream_t st)
{
    switch (a.depth) {
    case 2: return launch_scan_t<2, MASKED, MODE>(a, st);
    case 3: return launch_scan_t<3, MASKED, MODE>(a, st);
    case 5: return launch_scan_t<5, MASKED, MODE>(a, st);
    case 6: return launch_scan_t<6, MASKED, MODE>(a, st);
    default: return launch_scan_t<4, MASKED, MODE>(a, st);
    }
}

}  // namespace

hipError_t kmp_launch_scan(const kmp_scan_args &a, hipStream_t st)
{
    if (a.n_ids == 0 || a.blocks_x == 0) return hipSuccess;
    if (a.mode == 1) return launch_scan_d<true, 1>(a, st);      /* MASKED is unused by the automaton */
    return a.masked ? launch_scan_d<true, 0>(a, st) : launch_scan_d<false, 0>(a, st);
}

namespace {
Emitter emitter_of(const kmp_scan_args &a)
{
    Emitter e;
    e.out = reinterpret_cast<uint4 *>(a.emit_out);
    e.counter = a.emit_counter;
    e.cap = a.emit_cap;
    e.pattern = 0;
    return e;
}

template <int DEPTH, bool MASKED>
hipError_t launch_flat_t(const kmp_scan_args &a, hipStream_t st)
{
    dim3 grid(a.blocks_x, a.n_ids), block(KMP_BLOCK_THREADS);
    const Emitter em = emitter_of(a);
#define KMP_FLAT_ARGS a.arena, a.n_pkts, a.uniform_stride, a.uniform_len, a.pkts_per_wave, a.patterns, a.pat_ids, a.partials, em
    if (a.emit_out)
        hipLaunchKernelGGL((kmp_scan_flat_kernel<4, MASKED, true, true>), grid, block, 0, st, KMP_FLAT_ARGS);
    else if (a.nontemporal)
        hipLaunchKernelGGL((kmp_scan_flat_kernel<DEPTH, MASKED, true>), grid, block, 0, st, KMP_FLAT_ARGS);
    else
        hipLaunchKernelGGL((kmp_scan_flat_kernel<DEPTH, MASKED, false>), grid, block, 0, st, KMP_FLAT_ARGS);
#undef KMP_FLAT_ARGS
    return hipGetLastError();
}
template <bool MASKED>
hipError_t launch_flat_d(const kmp_scan_args &a, hipStream_t st)
{
    switch (a.depth) {
    case 2: return launch_flat_t<2, MASKED>(a, st);
    case 3: return launch_flat_t<3, MASKED>(a, st);
    case 5: return launch_flat_t<5, MASKED>(a, st);
    case 6: return launch_flat_t<6, MASKED>(a, st);
    case 8: return launch_flat_t<8, MASKED>(a, st);
    default: return launch_flat_t<4, MASKED>(a, st);
    }
}
}  // namespace

namespace {
template <int DEPTH, bool MASKED>
hipError_t launch_packed_t(const kmp_scan_args &a, hipStream_t st)
{
    dim3 grid(a.blocks_x, a.n_ids), block(KMP_BLOCK_THREADS);
    const kmp_plan_entry *plan = reinterpret_cast<const kmp_plan_entry *>(a.plan);
    const Emitter em = emitter_of(a);
#define KMP_PACKED_ARGS a.arena, a.pkt_off, a.pkt_len, a.bitmap, plan, a.patterns, a.pat_ids, a.partials, em
    if (a.emit_out)
        hipLaunchKernelGGL((kmp_scan_packed_kernel<4, MASKED, true, true>), grid, block, 0, st, KMP_PACKED_ARGS);
    else if (a.nontemporal)
        hipLaunchKernelGGL((kmp_scan_packed_kernel<DEPTH, MASKED, true>), grid, block, 0, st, KMP_PACKED_ARGS);
    else
        hipLaunchKernelGGL((kmp_scan_packed_kernel<DEPTH, MASKED, false>), grid, block, 0, st, KMP_PACKED_ARGS);
#undef KMP_PACKED_ARGS
    return hipGetLastError();
}
}  // namespace

/* Flat streaming kernel for packed arenas of arbitrary payload lengths (bitmap + plan from kmp_launch_prepare_packed). */
hipError_t kmp_launch_scan_packed(const kmp_scan_args &a, hipStream_t st)
{
    if (a.n_ids == 0 || a.blocks_x == 0) return hipSuccess;
    switch (a.depth) {
    case 3: return a.masked ? launch_packed_t<3, true>(a, st) : launch_packed_t<3, false>(a, st);
    case 6: return a.masked ? launch_packed_t<6, true>(a, st) : launch_packed_t<6, false>(a, st);
    default: return a.masked ? launch_packed_t<4, true>(a, st) : launch_packed_t<4, false>(a, st);
    }
}

hipError_t kmp_launch_build_bitmap(const uint64_t *pkt_off, uint64_t n, unsigned long long *bitmap, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS;
    if (blocks > 4096u) blocks = 4096u;
    hipLaunchKernelGGL(kmp_build_bitmap_kernel, dim3((uint32_t)blocks), dim3(KMP_BLOCK_THREADS), 0, st, pkt_off, n, bitmap);
    return hipGetLastError();
}

hipError_t kmp_launch_plan(const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n, uint64_t nwaves, uint64_t bytes_per_wave,
                           void *plan, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (nwaves + 1 + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS;
    hipLaunchKernelGGL(kmp_plan_kernel, dim3((uint32_t)blocks), dim3(KMP_BLOCK_THREADS), 0, st, pkt_off, pkt_len, n, nwaves,
                       bytes_per_wave, reinterpret_cast<kmp_plan_entry *>(plan));
    return hipGetLastError();
}

/* Flat streaming kernel for uniform-stride arenas (a.arena already points at payload 0). */
hipError_t kmp_launch_scan_flat(const kmp_scan_args &a, hipStream_t st)
{
    if (a.n_ids == 0 || a.blocks_x == 0) return hipSuccess;
    return a.masked ? launch_flat_d<true>(a, st) : launch_flat_d<false>(a, st);
}

/* On-device extraction: frames of a raw capture file -> packed payload arena.  The temporaries live in
 * ws (caller-allocated, kmp_extract_ws_bytes(n) bytes).  Two phases because the arena size is only known
 * after the scan: phase 1 leaves {arena bytes, payload count} in ws->totals. */
size_t kmp_extract_ws_bytes(uint64_t n_frames)
{
    const uint64_t nblk = (n_frames + KMP_SCAN_TILE - 1) / KMP_SCAN_TILE;
    return (size_t)(n_frames * (4 + 4 + 8 + 4) + nblk * (8 + 4) + 64 + 256);
}

hipError_t kmp_launch_extract_phase1(const uint8_t *file, const uint64_t *frame_off, const uint32_t *caplen, uint64_t n, int tcp,
                                     uint8_t *ws, unsigned long long *totals, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint64_t nblk = (n + KMP_SCAN_TILE - 1) / KMP_SCAN_TILE;
    uint64_t *loc_off = reinterpret_cast<uint64_t *>(ws);
    uint64_t *blk_bytes = loc_off + n;
    uint32_t *poff = reinterpret_cast<uint32_t *>(blk_bytes + nblk);
    uint32_t *plen = poff + n, *loc_idx = plen + n, *blk_cnt = loc_idx + n;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS, 4096);
    hipLaunchKernelGGL(kmp_extract_kernel, dim3(blocks), dim3(KMP_BLOCK_THREADS), 0, st, file, frame_off, caplen, n, tcp, poff, plen);
    hipLaunchKernelGGL(kmp_scan_local_kernel, dim3((uint32_t)nblk), dim3(KMP_BLOCK_THREADS), 0, st, plen, n, loc_off, loc_idx, blk_bytes, blk_cnt);
    hipLaunchKernelGGL(kmp_scan_totals_kernel, dim3(1), dim3(KMP_BLOCK_THREADS), 0, st, blk_bytes, blk_cnt, (uint32_t)nblk, totals);
    return hipGetLastError();
}

hipError_t kmp_launch_extract_phase2(const uint8_t *file, const uint64_t *frame_off, uint64_t n, uint8_t *ws, uint64_t n_pkts,
                                     uint8_t *arena, uint64_t *pkt_off, uint32_t *pkt_len, uint64_t *src_off, hipStream_t st)
{
    if (n == 0 || n_pkts == 0) return hipSuccess;
    const uint64_t nblk = (n + KMP_SCAN_TILE - 1) / KMP_SCAN_TILE;
    uint64_t *loc_off = reinterpret_cast<uint64_t *>(ws);
    uint64_t *blk_bytes = loc_off + n;
    uint32_t *poff = reinterpret_cast<uint32_t *>(blk_bytes + nblk);
    uint32_t *plen = poff + n, *loc_idx = plen + n, *blk_cnt = loc_idx + n;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS, 4096);
    hipLaunchKernelGGL(kmp_scatter_index_kernel, dim3(blocks), dim3(KMP_BLOCK_THREADS), 0, st, frame_off, poff, plen, loc_off, loc_idx,
                       blk_bytes, blk_cnt, n, pkt_off, pkt_len, src_off);
    uint32_t gblocks = (uint32_t)std::min<uint64_t>((n_pkts + KMP_BLOCK_WAVES - 1) / KMP_BLOCK_WAVES, 8192);
    hipLaunchKernelGGL(kmp_gather_kernel, dim3(gblocks), dim3(KMP_BLOCK_THREADS), 0, st, file, src_off, pkt_off, pkt_len, n_pkts, arena);
    return hipGetLastError();
}

hipError_t kmp_launch_reduce(const unsigned long long *partials, uint32_t blocks_x, const uint32_t *pat_ids,
                             uint32_t n_ids, unsigned long long *counts, hipStream_t st, const uint32_t *rows, int accumulate)
{
    if (n_ids == 0) return hipSuccess;
    hipLaunchKernelGGL(kmp_reduce_kernel, dim3(n_ids), dim3(KMP_BLOCK_THREADS), 0, st, partials, blocks_x, pat_ids, rows, counts,
                       accumulate);
    return hipGetLastError();
}

/* Fused multi-pattern pass over a packed arena (bitmap + plan as for kmp_launch_scan_packed). */
hipError_t kmp_launch_scan_multi(const kmp_scan_args &a, const uint32_t *tables, uint32_t table_words, uint32_t n_unique,
                                 hipStream_t st)
{
    if (n_unique == 0 || a.blocks_x == 0) return hipSuccess;
    const kmp_plan_entry *plan = reinterpret_cast<const kmp_plan_entry *>(a.plan);
    const size_t lds = ((size_t)((table_words + n_unique + 3u) & ~3u) + KMP_BLOCK_WAVES * (KMP_CHUNK / 4u + 8u)) * sizeof(uint32_t);
    if (a.nontemporal)
        hipLaunchKernelGGL((kmp_scan_multi_kernel<4, true>), dim3(a.blocks_x), dim3(KMP_BLOCK_THREADS), lds, st, a.arena, a.pkt_len,
                           a.bitmap, plan, tables, table_words, n_unique, a.partials);
    else
        hipLaunchKernelGGL((kmp_scan_multi_kernel<4, false>), dim3(a.blocks_x), dim3(KMP_BLOCK_THREADS), lds, st, a.arena, a.pkt_len,
                           a.bitmap, plan, tables, table_words, n_unique, a.partials);
    return hipGetLastError();
}

hipError_t kmp_launch_validate(const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n, uint64_t arena_bytes,
                               uint32_t *err, unsigned long long *payload_bytes, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS);
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(kmp_validate_index_kernel, dim3(blocks), dim3(KMP_BLOCK_THREADS), 0, st, pkt_off, pkt_len, n,
                       arena_bytes, err, payload_bytes);
    return hipGetLastError();
}

hipError_t kmp_launch_synth_fill(uint8_t *arena, const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t first_pkt_id,
                                 uint64_t n, const kmp_synth_params &sp, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + KMP_BLOCK_WAVES - 1) / KMP_BLOCK_WAVES;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(kmp_synth_fill_kernel, dim3((uint32_t)blocks), dim3(KMP_BLOCK_THREADS), 0, st, arena, pkt_off,
                       pkt_len, first_pkt_id, n, sp);
    return hipGetLastError();
}

hipError_t kmp_launch_fixed_index(uint64_t *pkt_off, uint32_t *pkt_len, uint64_t n, uint32_t len, uint64_t stride,
                                  hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS;
    if (blocks > 4096u) blocks = 4096u;
    hipLaunchKernelGGL(kmp_fixed_index_kernel, dim3((uint32_t)blocks), dim3(KMP_BLOCK_THREADS), 0, st, pkt_off, pkt_len, n,
                       len, stride);
    return hipGetLastError();
}
