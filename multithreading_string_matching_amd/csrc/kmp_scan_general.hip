/*
 * kmp_scan_general.hip -- one-packet-per-wavefront kernel for arenas whose slots are not back to back
 * (gaps, reordered slots), and the pure KMP-automaton mode used as an independent second
 * implementation by the parity tests.  gfx950.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kmp_device.h"
#include "kmp_launch.h"
#include "kmp_dev_common.h"

namespace {

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

