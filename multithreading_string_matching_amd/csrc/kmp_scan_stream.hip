/*
 * kmp_scan_stream.hip -- the hot path on gfx950 (MI355X / CDNA4): per-pattern match counts over a
 * payload arena.  Replaces the reference loop serial.c:153-155 (= openmp_data.c:157-175) and its
 * kernel function kmp_matcher (serial.c:190-215).  Hand-written for 64-lane wavefronts; no MFMA
 * (byte scan, HBM-bound).
 *
 * Shape of the streaming kernels (flat: uniform stride, packed: any lengths)
 *   - every wavefront owns ONE contiguous byte range of the arena (whole packets) and streams it in 1 KiB chunks: one
 *     buffer_load_dwordx4 per lane (16 B), perfectly coalesced, DEPTH chunk loads in flight per wavefront in a register
 *     ring driven by hand-counted s_waitcnt vmcnt(N) (kmp_dev_common.h: ring_wait / flat_issue); the buffer resource's
 *     record count makes the tail chunk read zeros, so there is no clamping and no lane mask;
 *   - the ranges are SMALL and the grid is not persistent: ~6 KiB per wavefront for the flat kernel (four 1500-byte
 *     packets), ~16 KiB for the packed one, one block per four ranges, handed out in arena order by the hardware as CUs
 *     free up -- the whole chip reads one compact moving window of the arena (kmpgpu.hip grid_blocks,
 *     profiles/r02_flat_grid.txt: 0.89 of the HBM peak against 0.82 with one long resident range per wavefront);
 *   - still one packet per wavefront at a time: the packets of a range are scanned in order by the same
 *     wavefront, which keeps the strlen() rule (serial.c:191) wave-local state;
 *   - per chunk, every lane tests its 16 start offsets against the pattern's first dword with four v_mqsad_pk_u16_u8
 *     (masked quad byte-SAD: four offsets per instruction, byte alignment included; halo dword from the next lane by
 *     DPP wave_shl:1) and looks for 0x00 bytes with the has-zero trick; three ballots (zero lanes, packet-start lanes,
 *     candidate lanes); the common case -- no candidate in the chunk -- ends there;
 *   - rare path (confirm_sad, kmp_dev_common.h): every start offset of every candidate lane is compared in full,
 *     branch-free, by accumulating the pattern's further dwords into the filter's sums (16 bytes per block, further
 *     blocks by scalar loads); per lane the largest start index that still counts (window inside the payload, no 0x00
 *     before it) bars the rest.  The literal KMP automaton (kmp_matcher, serial.c:190-215) lives in kmp_scan_general.hip;
 *   - a start offset s counts iff s + m <= E, E = min(len, first 0x00);
 *   - counts: per-lane -> wave -> block, one partial per (block, pattern), summed by kmp_reduce_kernel
 *     (plain stores, deterministic).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kmp_device.h"
#include "kmp_launch.h"
#include "kmp_dev_common.h"

namespace {

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


template <int DEPTH, bool NT, bool EMIT = false>
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_flat_kernel(const uint8_t *__restrict__ arena, uint64_t n_pkts, uint32_t stride, uint32_t L,
                     uint32_t pkts_per_wave, const kmp_pattern_dev *__restrict__ patterns,
                     const uint32_t *__restrict__ pat_ids, unsigned long long *__restrict__ partials, unsigned long long *__restrict__ zero_counts, Emitter em)
{
    __shared__ unsigned long long s_wave_cnt[KMP_BLOCK_WAVES];

    const uint32_t lane = threadIdx.x & (KMP_WAVE - 1u);
    const uint32_t wave = sgpr(threadIdx.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + wave;
    /* this wavefront's packets [k0, k1) = bytes [0, range) behind base */
    const uint64_t k0 = gw * pkts_per_wave;
    const uint64_t k1 = min(n_pkts, k0 + pkts_per_wave);
    const uint32_t range = (k0 < n_pkts) ? (uint32_t)(k1 - k0) * stride : 0u;      /* host guarantees < 2^31 */
    const uint8_t *base = arena + ((k0 < n_pkts) ? k0 * (uint64_t)stride : 0ull);
    const uint32_t step_mod = KMP_CHUNK % stride;                                    /* p0 advance per chunk (mod stride) */

    /* The stream starts before anything else: the first DEPTH chunk loads need nothing but the range, and the
     * 2 us they take cover the fetch of the pattern record below (an empty range has a record count of 0: its
     * loads fetch nothing and return zeros). */
    const i32x4    rsrc = make_rsrc(base, range);
    const uint32_t vo0 = lane * KMP_LANE_BYTES;
    u32x4 buf[DEPTH];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) flat_issue<NT, true>(buf[s], rsrc, vo0, (uint32_t)s * KMP_CHUNK);

    const uint32_t pid = pat_ids[blockIdx.y];
    const kmp_pattern_dev *gp = patterns + pid;
    const PatConst pc = load_pat_const(gp);
    const uint32_t m = pc.m;
    if (EMIT) em.pattern = pid;

    uint32_t cnt = 0u;
    {
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
                    uint64_t S[4];
                    uint32_t t[4];
                    const uint32_t fz = zero_half_mask(filter_sad(w, pc.p[0], S, t));      /* != 0 iff the lane has a candidate */
                    const uint64_t zl = ballot64(zm != 0u);                   /* lanes holding a 0x00           */
                    const uint64_t st = ballot64(p0 == 0u);                   /* lanes where a packet starts    */
                    const uint64_t cl = ballot64(fz != 0u);                   /* lanes with a candidate         */
                    const bool dead_in = dead;
                    /* carry for the next chunk: is there a 0x00 at or after the last packet start of this chunk? */
                    if (zl == 0ull) { if (st != 0ull) dead = false; }
                    else            dead = (st == 0ull) ? true : ((zl >> (63u - (uint32_t)__builtin_clzll(st))) != 0ull);

                    if (cl != 0ull) {
                        /* rare path.  maxi = largest start index (0..15) of this lane that still counts:
                         * window inside the payload, no 0x00 before it, lane has a candidate at all. */
                        int32_t maxi = (int32_t)L - (int32_t)m - (int32_t)p0;
                        if (fz == 0u) maxi = -1;
                        if (zl != 0ull || dead_in) {
                            /* A 0x00 in the LAST lane of a packet (slot padding, a trailer) ends nothing but that lane's own
                             * later offsets -- and matters only if that lane has a candidate; the segmented form is for a
                             * 0x00 in mid-packet. */
                            const uint64_t last_lanes = ballot64(p0 + KMP_LANE_BYTES == stride);
                            if (dead_in || (zl & ~last_lanes) != 0ull) maxi = nul_limit(maxi, w, zl, st, dead_in, lane);
                            else if (ballot64(zm != 0u && fz != 0u) != 0ull) maxi = nul_limit(maxi, w, 0ull, st, false, lane);
                        }
                        const uint64_t pkt = EMIT ? (k0 + (uint64_t)((cb + vo0 - p0) / stride)) : 0ull;
                        confirm_sad<EMIT>(S, t, w, v, bn, fz, maxi, p0, pc, gp, cnt, pkt, em);
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
        /* a launch of tens of thousands of blocks is summed by several blocks of kmp_reduce_kernel, each of which ADDS its share
         * to the pattern's counter: unless the pass accumulates, the counter starts from 0 here, a kernel boundary ahead of them */
        if (zero_counts && blockIdx.x == 0u) zero_counts[pid] = 0ull;
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

__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_build_bitmap_kernel(const uint64_t *__restrict__ pkt_off, uint64_t n, unsigned long long *__restrict__ bitmap)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t slot = pkt_off[k] >> 4;
        atomicOr(bitmap + (slot >> 6), 1ull << (slot & 63ull));
    }
}

/* plan[w] = the packet whose start lies CLOSEST to off[0] + w * bytes_per_wave (w = 0..nwaves); plan[nwaves] = {n, end}.
 * Ranges are whole packets, so a range's length differs from bytes_per_wave by where packets happen to start; with the first start
 * at or behind the target (round 2) that was up to one packet at either end -- 9000 bytes on ~41 KB ranges for the Zipf lengths of
 * BASELINE configs[4], and a block is as slow as its slowest wavefront --, with the closest one it is half of that. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_plan_kernel(const uint64_t *__restrict__ pkt_off, const uint32_t *__restrict__ pkt_len, uint64_t n, uint64_t nwaves,
                kmp_plan_shape shape, kmp_plan_entry *__restrict__ plan)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwaves) return;
    const uint64_t l16 = ((uint64_t)pkt_len[n - 1] + 15ull) & ~15ull;
    const uint64_t end = pkt_off[n - 1] + (l16 < 16ull ? 16ull : l16);
    if (w == nwaves) { plan[w].k = n; plan[w].off = end; return; }
    uint64_t target = pkt_off[0];
    if (shape.units == 0u) target += w * shape.step;
    else {
        /* unit u of region r (kmp_launch.h): the big units first, then the small ones; what a region's units overshoot belongs to the next region */
        const uint64_t r = w / shape.units, u = w % shape.units;
        const uint64_t in = u <= shape.big_units ? u * shape.step : shape.big_units * shape.step + (u - shape.big_units) * shape.small;
        target += r * shape.region + (in < shape.region ? in : shape.region);
    }
    uint64_t lo = 0, hi = n;                       /* lower_bound over the (increasing) offsets */
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pkt_off[mid] < target) lo = mid + 1; else hi = mid;
    }
    if (lo > 0 && w > 0) {
        const uint64_t next = (lo < n) ? pkt_off[lo] : end;
        /* (a target behind the arena's end -- more wavefronts than bytes_per_wave-sized pieces -- keeps lo = n: an empty range) */
        if (next >= target && target - pkt_off[lo - 1] < next - target) --lo;      /* the start before the target is the closer one */
    }
    plan[w].k = lo;
    plan[w].off = (lo < n) ? pkt_off[lo] : end;
}

template <int DEPTH, bool NT, bool EMIT = false>
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_packed_kernel(const uint8_t *__restrict__ arena, const uint64_t *__restrict__ pkt_off,
                       const uint32_t *__restrict__ pkt_len, const unsigned long long *__restrict__ bitmap,
                       const kmp_plan_entry *__restrict__ plan, const kmp_pattern_dev *__restrict__ patterns,
                       const uint32_t *__restrict__ pat_ids, unsigned long long *__restrict__ partials, unsigned long long *__restrict__ zero_counts, Emitter em,
                       uint32_t pad_clean)
{
    __shared__ unsigned long long s_wave_cnt[KMP_BLOCK_WAVES];

    const uint32_t lane = threadIdx.x & (KMP_WAVE - 1u);
    const uint32_t wave = sgpr(threadIdx.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + wave;
    const uint64_t k0 = plan[gw].k, k1 = plan[gw + 1].k;
    /* The range starts at a packet start (16-byte aligned).  Streaming from there would make every 1 KiB chunk
     * load straddle nine 128-byte lines instead of covering eight, and the line shared by two consecutive
     * chunks is fetched from HBM twice under the streaming (nt) policy: +4.5 % traffic measured.  So the
     * stream starts on the line boundary below; the `pl` lanes of the first chunk that precede the first
     * packet belong to the previous wavefront and are blanked (zero bytes, no start bits). */
    const uint64_t off_first = plan[gw].off;
    const uint32_t pre = (uint32_t)(off_first & 127ull), pl = pre >> 4;
    const uint64_t off0 = off_first - pre;
    const uint32_t range = (k1 > k0) ? (uint32_t)(plan[gw + 1].off - off0) : 0u;    /* planner guarantees < 2^31 */

    /* the stream starts before the pattern record is staged (see kmp_scan_flat_kernel); an empty range fetches nothing */
    const i32x4    rsrc = make_rsrc(arena + off0, range);
    const uint32_t vo0 = lane * KMP_LANE_BYTES;
    u32x4 buf[DEPTH];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) flat_issue<NT, true>(buf[s], rsrc, vo0, (uint32_t)s * KMP_CHUNK);

    const uint32_t pid = pat_ids[blockIdx.y];
    const kmp_pattern_dev *gp = patterns + pid;
    const PatConst pc = load_pat_const(gp);
    const uint32_t m = pc.m;
    if (EMIT) em.pattern = pid;

    uint32_t cnt = 0u;
    if (range) {
        /* packet-start bits of chunk j: bits [b0 + 64 j, +64) of the bitmap = words wi0+j, wi0+j+1 shifted by sh */
        const uint64_t b0 = off0 >> 4;
        const unsigned long long *bw = bitmap + (b0 >> 6);
        const uint32_t sh = (uint32_t)(b0 & 63ull);

        unsigned long long hiw[DEPTH];       /* bitmap word wi0 + j + 1 of the chunk in ring slot s, fetched one group ahead */
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) hiw[s] = bw[s + 1];
        unsigned long long low = bw[0];      /* bitmap word wi0 + j of the chunk being consumed */
        uint64_t kbase = k0 - 1ull;          /* index of the last packet started before the chunk */
        bool     dead = false;
        uint32_t cb = 0u, j = 0u;

        while (cb < range) {
            /* Packet-start words, one GROUP of ring slots ahead.  Scalar loads return out of order, so the only
             * wait that covers them is lgkmcnt(0) -- which also waits for a load issued a moment ago.  Using this
             * group's words first (they have had a whole group of chunks to arrive) and only then asking for the
             * next group's keeps that wait off the critical path. */
            uint64_t st_[DEPTH];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                const unsigned long long hi = hiw[s];
                st_[s] = sh ? ((low >> sh) | (hi << (64u - sh))) : low;                   /* lanes where a packet starts */
                low = hi;
                asm volatile("" : "+s"(st_[s]));      /* computed HERE, not sunk below the loads that follow */
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) hiw[s] = bw[j + (uint32_t)DEPTH + 1u + (uint32_t)s];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                ring_wait<DEPTH - 2>(buf[s], buf[(s + 1) % DEPTH]);
                if (cb < range) {
                    uint4          v   = make_uint4(buf[s].x, buf[s].y, buf[s].z, buf[s].w);
                    const u32x4    bn  = buf[(s + 1) % DEPTH];
                    uint64_t       st  = st_[s];
                    if (s == 0 && cb == 0u && pl != 0u) {                               /* head of the range, see above */
                        if (lane < pl) v = make_uint4(0u, 0u, 0u, 0u);
                        st &= ~0ull << pl;
                    }
                    const uint32_t w[5] = {v.x, v.y, v.z, v.w, wave_shl1(v.x, sgpr(bn.x))};

                    const uint32_t zm = zero_byte_mask(w[0]) | zero_byte_mask(w[1]) | zero_byte_mask(w[2]) | zero_byte_mask(w[3]);
                    uint64_t S[4];
                    uint32_t t[4];
                    const uint32_t fz = zero_half_mask(filter_sad(w, pc.p[0], S, t));      /* != 0 iff the lane has a candidate */
                    const uint64_t zl = ballot64(zm != 0u);
                    const uint64_t cl = ballot64(fz != 0u);
                    const bool dead_in = dead;
                    if (zl == 0ull) { if (st != 0ull) dead = false; }
                    else            dead = (st == 0ull) ? true : ((zl >> (63u - (uint32_t)__builtin_clzll(st))) != 0ull);

                    if (cl != 0ull) {
                        /* rare path: which packet does a candidate lane sit in? */
                        int32_t  maxi = -1;
                        uint32_t p0 = 0u, L = 0u;
                        uint64_t kl = 0ull;
                        uint64_t nx;                                                    /* start bits of the next chunk */
                        if (s + 1 < DEPTH) nx = st_[(s + 1) % DEPTH];
                        else               nx = sh ? ((low >> sh) | (hiw[0] << (64u - sh))) : low;
                        const uint64_t last_lanes = (st >> 1) | (nx << 63);           /* the lanes behind which a packet starts */
                        if (!EMIT && pad_clean && m <= 16u && (cl & last_lanes) == 0ull) {
                            /* (see the next branch) no candidate sits in the last lane of its slot: every one of them has 32
                             * bytes of slot or more from its first byte, all 16 start offsets fit a pattern of up to 16 bytes */
                            maxi = fz != 0u ? 15 : -1;
                        } else if (!EMIT && pad_clean) {
                            /* Slot padding is all 0x00 (checked when the arena was loaded), so "the window lies inside
                             * the payload" = "it lies inside the slot and holds no 0x00": the distance to the next
                             * packet start, read off the bitmap, replaces the payload's offset and length -- no
                             * gather from the index, which costs a memory round trip per candidate chunk. */
                            if (fz != 0u) {
                                const uint64_t above = (st >> 1) >> lane;               /* starts at the lanes above own */
                                uint32_t d = 4096u;                                     /* 16-byte groups up to the next start */
                                if (above != 0ull) d = (uint32_t)__builtin_ctzll(above) + 1u;
                                else if (nx != 0ull) d = 64u - lane + (uint32_t)__builtin_ctzll(nx);
                                L = d * KMP_LANE_BYTES;                                 /* bytes from the lane's first to the slot's end */
                                maxi = (int32_t)L - (int32_t)m;
                            }
                        } else if (fz != 0u) {
                            const uint64_t le = (2ull << lane) - 1ull;                  /* lanes <= own (lane 63: all ones) */
                            kl = kbase + (uint64_t)__builtin_popcountll(st & le);
                            const uint64_t po = pkt_off[kl];
                            L  = pkt_len[kl];
                            p0 = (uint32_t)(off0 + cb + vo0 - po);
                            maxi = (int32_t)L - (int32_t)m - (int32_t)p0;
                        }
                        if (zl != 0ull || dead_in) {
                            /* a 0x00 in the last lane of a packet ends nothing but that lane's own later offsets (see kmp_scan_flat_kernel) */
                            if (dead_in || (zl & ~last_lanes) != 0ull) maxi = nul_limit(maxi, w, zl, st, dead_in, lane);
                            else if (ballot64(zm != 0u && fz != 0u) != 0ull) maxi = nul_limit(maxi, w, 0ull, st, false, lane);
                        }
                        confirm_sad<EMIT>(S, t, w, v, bn, fz, maxi, p0, pc, gp, cnt, kl, em);
                        /* leave nothing of the rare path's LDS/scalar reads "possibly in flight": merged into the
                         * common path that state costs an s_waitcnt lgkmcnt(0) per chunk, which would also wait
                         * for the bitmap words just asked for */
                        __builtin_amdgcn_s_waitcnt(0xC07F);      /* lgkmcnt(0) only */
                    }
                    kbase += (uint64_t)__builtin_popcountll(st);
                }
                __builtin_amdgcn_sched_barrier(0);
                flat_issue<NT>(buf[s], rsrc, vo0, cb + (uint32_t)DEPTH * KMP_CHUNK);
                cb += KMP_CHUNK;
                ++j;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < DEPTH; s += 2) ring_wait<0>(buf[s], buf[(s + 1) % DEPTH]);     /* nothing in flight when the wavefront ends */

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
        /* a launch of tens of thousands of blocks is summed by several blocks of kmp_reduce_kernel, each of which ADDS its share
         * to the pattern's counter: unless the pass accumulates, the counter starts from 0 here, a kernel boundary ahead of them */
        if (zero_counts && blockIdx.x == 0u) zero_counts[pid] = 0ull;
    }
}

}  // namespace

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

template <int DEPTH>
hipError_t launch_flat_t(const kmp_scan_args &a, hipStream_t st)
{
    dim3 grid(a.blocks_x, a.n_ids), block(KMP_BLOCK_THREADS);
    const Emitter em = emitter_of(a);
#define KMP_FLAT_ARGS a.arena, a.n_pkts, a.uniform_stride, a.uniform_len, a.pkts_per_wave, a.patterns, a.pat_ids, a.partials, a.zero_counts, em
    if (a.emit_out)
        hipLaunchKernelGGL((kmp_scan_flat_kernel<4, true, true>), grid, block, 0, st, KMP_FLAT_ARGS);
    else if (a.nontemporal)
        hipLaunchKernelGGL((kmp_scan_flat_kernel<DEPTH, true>), grid, block, 0, st, KMP_FLAT_ARGS);
    else
        hipLaunchKernelGGL((kmp_scan_flat_kernel<DEPTH, false>), grid, block, 0, st, KMP_FLAT_ARGS);
#undef KMP_FLAT_ARGS
    return hipGetLastError();
}
}  // namespace

namespace {
template <int DEPTH>
hipError_t launch_packed_t(const kmp_scan_args &a, hipStream_t st)
{
    dim3 grid(a.blocks_x, a.n_ids), block(KMP_BLOCK_THREADS);
    const kmp_plan_entry *plan = reinterpret_cast<const kmp_plan_entry *>(a.plan);
    const Emitter em = emitter_of(a);
#define KMP_PACKED_ARGS a.arena, a.pkt_off, a.pkt_len, a.bitmap, plan, a.patterns, a.pat_ids, a.partials, a.zero_counts, em, (a.pad_clean ? 1u : 0u)
    if (a.emit_out)
        hipLaunchKernelGGL((kmp_scan_packed_kernel<4, true, true>), grid, block, 0, st, KMP_PACKED_ARGS);
    else if (a.nontemporal)
        hipLaunchKernelGGL((kmp_scan_packed_kernel<DEPTH, true>), grid, block, 0, st, KMP_PACKED_ARGS);
    else
        hipLaunchKernelGGL((kmp_scan_packed_kernel<DEPTH, false>), grid, block, 0, st, KMP_PACKED_ARGS);
#undef KMP_PACKED_ARGS
    return hipGetLastError();
}
}  // namespace

/* Flat streaming kernel for packed arenas of arbitrary payload lengths (bitmap + plan from kmp_launch_prepare_packed). */
hipError_t kmp_launch_scan_packed(const kmp_scan_args &a, hipStream_t st)
{
    if (a.n_ids == 0 || a.blocks_x == 0) return hipSuccess;
    switch (a.depth) {          /* 0 = auto: 4 chunks in flight with the 16 KiB ranges (profiles/r02_flat_grid.txt; 3 with the persistent grid of round 1) */
    case 2: case 3: return launch_packed_t<3>(a, st);
    case 6: case 8: return launch_packed_t<6>(a, st);
    default: return launch_packed_t<4>(a, st);
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

hipError_t kmp_launch_plan(const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n, uint64_t nwaves, const kmp_plan_shape &shape,
                           void *plan, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (nwaves + 1 + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS;
    hipLaunchKernelGGL(kmp_plan_kernel, dim3((uint32_t)blocks), dim3(KMP_BLOCK_THREADS), 0, st, pkt_off, pkt_len, n, nwaves,
                       shape, reinterpret_cast<kmp_plan_entry *>(plan));
    return hipGetLastError();
}

/* Flat streaming kernel for uniform-stride arenas (a.arena already points at payload 0). */
hipError_t kmp_launch_scan_flat(const kmp_scan_args &a, hipStream_t st)
{
    if (a.n_ids == 0 || a.blocks_x == 0) return hipSuccess;
    switch (a.depth) {
    case 2: return launch_flat_t<2>(a, st);
    case 3: return launch_flat_t<3>(a, st);
    case 5: return launch_flat_t<5>(a, st);
    case 6: return launch_flat_t<6>(a, st);
    case 8: return launch_flat_t<8>(a, st);
    default: return launch_flat_t<4>(a, st);
    }
}

