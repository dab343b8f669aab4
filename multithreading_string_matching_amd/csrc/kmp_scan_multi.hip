/*
 * kmp_scan_multi.hip -- fused multi-pattern pass, gfx950.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kmp_device.h"
#include "kmp_launch.h"
#include "kmp_dev_common.h"

namespace {

/* ================================================================================================
 * Fused multi-pattern pass: every pattern of 2..99 bytes in ONE read of a packed
 * arena -- the reference re-reads every payload once per pattern (serial.c:154, openmp_data.c:163).
 *
 * Same streaming skeleton as kmp_scan_packed_kernel (buffer-load ring, packet-start bitmap, byte-
 * balanced plan).  Per chunk:
 *   - rem = payload bytes left from the lane's first byte (uniform loop over the packet starts of the
 *     chunk; lengths come by scalar loads, offsets are implied by the bitmap because the arena is packed);
 *   - level 1: the 3-byte window at each of the 16 start offsets is hashed into a 64 Kbit LDS filter "some
 *     pattern may start with these bytes" (2-byte patterns set all 256 third bytes) -> 16-bit hit mask per lane;
 *   - level 2, per start offset that has a hit in some lane: hash the 2 bytes to a bucket, walk the
 *     bucket's short list of patterns, compare up to 20 bytes dword-wise with byte masks (text from
 *     registers, pattern records from LDS), check window-in-payload and the strlen() rule, and bump
 *     the pattern's counter in LDS.  Counters go to partials[unique pattern][block] at the end.
 * ============================================================================================== */
template <int DEPTH, bool NT, bool CLEAN, bool EMIT = false>
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_multi_kernel(const uint8_t *__restrict__ arena, const uint32_t *__restrict__ pkt_len,
                      const unsigned long long *__restrict__ bitmap, const kmp_plan_entry *__restrict__ plan,
                      const uint32_t *__restrict__ tables, uint32_t table_words, uint32_t n_unique,
                      unsigned long long *__restrict__ partials, Emitter em, const uint32_t *__restrict__ uid_first,
                      const uint32_t *__restrict__ uid_ids, const kmp_pattern_dev *__restrict__ patterns)
{
    /* buckets, entries and the filter sit in static LDS: their offsets are compile-time constants that fold
     * into the ds_read offset field; records, counters and one chunk window per wavefront follow dynamically */
    __shared__ __attribute__((aligned(16))) uint32_t s_fix[KMP_MULTI_REC_W0];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const uint32_t rec_words = table_words - KMP_MULTI_REC_W0;
    uint32_t *s_rec = s_dyn;
    uint32_t *s_cnt = s_dyn + rec_words;
    uint32_t *s_win = s_dyn + ((rec_words + n_unique + 3u) & ~3u);
    const uint32_t lane = threadIdx.x & (KMP_WAVE - 1u);
    const uint32_t wave = sgpr(threadIdx.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + wave;
    const uint64_t k0 = plan[gw].k, k1 = plan[gw + 1].k;
    /* the stream starts on the 128-byte line below the first packet (see kmp_scan_packed_kernel) */
    const uint64_t off_first = plan[gw].off;
    const uint32_t pre = (uint32_t)(off_first & 127ull), pl = pre >> 4;
    const uint64_t off0 = off_first - pre;
    const uint32_t range = (k1 > k0) ? (uint32_t)(plan[gw + 1].off - off0) : 0u;

    /* The stream starts before the tables are copied: the first DEPTH chunk loads need nothing but the range, and
     * filling 16-30 KB of LDS from global memory takes longer than they do (an empty range has a record count of
     * 0: its loads fetch nothing and return zeros). */
    const i32x4    rsrc = make_rsrc(arena + off0, range);
    const uint32_t vo0 = lane * KMP_LANE_BYTES;
    u32x4 buf[DEPTH];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) flat_issue<NT>(buf[s], rsrc, vo0, (uint32_t)s * KMP_CHUNK);

    for (uint32_t i = threadIdx.x; i < KMP_MULTI_REC_W0; i += KMP_BLOCK_THREADS) s_fix[i] = tables[i];
    for (uint32_t i = threadIdx.x; i < rec_words; i += KMP_BLOCK_THREADS) s_rec[i] = tables[KMP_MULTI_REC_W0 + i];
    for (uint32_t i = threadIdx.x; i < n_unique; i += KMP_BLOCK_THREADS) s_cnt[i] = 0u;
    __syncthreads();
    const uint16_t *s_bucket = reinterpret_cast<const uint16_t *>(s_fix + KMP_MULTI_BUCKET_W0);
    const uint32_t *s_entry  = s_fix + KMP_MULTI_ENTRY_W0;
    const uint8_t  *s_filter = reinterpret_cast<const uint8_t *>(s_fix + KMP_MULTI_FILTER_W0);

    if (range) {
        const uint64_t b0 = off0 >> 4;
        const unsigned long long *bw = bitmap + (b0 >> 6);
        const uint32_t sh = (uint32_t)(b0 & 63ull);

        unsigned long long hiw[DEPTH];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) hiw[s] = bw[s + 1];
        unsigned long long low = bw[0];
        uint64_t kcur = k0 - 1ull;           /* last packet that has started                                   */
        uint64_t kbase = k0 - 1ull;          /* EMIT: the same, kept in both variants of the payload-end logic */
        uint32_t last_start = 0u;            /* EMIT: byte position (from the stream's first byte) of that packet's start */
        int32_t  remc = 0;                   /* payload bytes of that packet left at the chunk's first byte     */
        bool     dead = false;
        uint32_t cb = 0u, j = 0u;

        while (cb < range) {
            /* packet-start words: use this group's, then ask for the next group's (see kmp_scan_packed_kernel) */
            uint64_t st_[DEPTH];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                const unsigned long long hi = hiw[s];
                st_[s] = sh ? ((low >> sh) | (hi << (64u - sh))) : low;
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
                    uint4 v = make_uint4(buf[s].x, buf[s].y, buf[s].z, buf[s].w);
                    const u32x4 bn = buf[(s + 1) % DEPTH];
                    uint64_t st = st_[s];
                    if (s == 0 && cb == 0u && pl != 0u) {                               /* lanes before the first packet: not ours */
                        if (lane < pl) v = make_uint4(0u, 0u, 0u, 0u);
                        st &= ~0ull << pl;
                    }
                    const uint32_t left = range - cb;
                    if (left < KMP_CHUNK) st &= (1ull << (left >> 4)) - 1ull;        /* bits past the range belong to the next wavefront */

                    const uint32_t w[5] = {v.x, v.y, v.z, v.w, wave_shl1(v.x, sgpr(bn.x))};

                    const uint32_t zm = zero_byte_mask(w[0]) | zero_byte_mask(w[1]) | zero_byte_mask(w[2]) | zero_byte_mask(w[3]);
                    const uint64_t zl = ballot64(zm != 0u);
                    const bool dead_in = dead;
                    if (zl == 0ull) { if (st != 0ull) dead = false; }
                    else            dead = (st == 0ull) ? true : ((zl >> (63u - (uint32_t)__builtin_clzll(st))) != 0ull);

                    /* rem: payload bytes left from this lane's first byte */
                    int32_t rem;
                    uint64_t sg_all = 0ull, nx_all = 0ull;      /* CLEAN: start bits of this chunk (not cut at the range's end) and of the next */
                    if constexpr (CLEAN) {
                        /* Slot padding is all 0x00 (kmp_check_padding_kernel), so the payload's end can be replaced by
                         * the slot's end: a window that reaches into the padding holds a 0x00 and matches nothing.
                         * Only lanes within 2 x 16 bytes of the next packet start are constrained (patterns are at
                         * most 20 bytes): two lane masks from the start bitmap, no payload lengths, no loop. */
                        uint64_t nx;                                                /* start bits of the next chunk */
                        if (s + 1 < DEPTH) nx = st_[(s + 1) % DEPTH];
                        else               nx = sh ? ((low >> sh) | (hiw[0] << (64u - sh))) : low;
                        const uint64_t sg = st_[s];                                 /* not cut at the range's end: the next wavefront's first start ends our last slot */
                        const bool next1 = __builtin_amdgcn_inverse_ballot_w64((sg >> 1) | (nx << 63));     /* lane + 1 starts a packet */
                        const bool next2 = __builtin_amdgcn_inverse_ballot_w64((sg >> 2) | (nx << 62));     /* lane + 2 does            */
                        rem = next1 ? 16 : next2 ? 32 : (1 << 20);
                        sg_all = sg; nx_all = nx;
                    } else {
                        /* from the index: uniform loop over the packet starts of the chunk (<= 0: slot padding) */
                        rem = remc - (int32_t)vo0;
                        int32_t remn = remc - (int32_t)KMP_CHUNK;
                        for (uint64_t sb = st; sb != 0ull; sb &= sb - 1ull) {
                            const uint32_t sl = (uint32_t)__builtin_ctzll(sb);
                            ++kcur;
                            const int32_t top = (int32_t)pkt_len[kcur] + (int32_t)(sl * KMP_LANE_BYTES);
                            if (lane >= sl) rem = top - (int32_t)vo0;
                            remn = top - (int32_t)KMP_CHUNK;
                        }
                        remc = remn;
                    }

                    /* level 1: which start offsets may begin some pattern (filter over the first three bytes)? */
                    uint32_t hm = 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const uint32_t d0 = a ? __builtin_amdgcn_alignbyte(w[q + 1], w[q], a) : w[q];
                            const uint32_t pr = __umul24(d0, 0x9E3779u);                               /* KMP_MULTI_BIT = pr >> 16: hash of 3 text bytes */
                            const uint32_t fb = s_filter[pr >> 19];
                            hm |= __builtin_amdgcn_ubfe(fb, (pr >> 16) & 7u, 1u) << (4 * q + a);
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
                            uint32_t *win = s_win + wave * KMP_MULTI_WIN_WORDS;
                            *reinterpret_cast<uint4 *>(win + lane * 4u) = v;
                            if (lane < 7u) *reinterpret_cast<uint4 *>(win + KMP_CHUNK / 4u + lane * 4u) = make_uint4(bn.x, bn.y, bn.z, bn.w);
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
                                    uint32_t e = s_bucket[(__umul24(T0, 0x9E3Bu) >> 6) & (KMP_MULTI_BUCKETS - 1u)];   /* KMP_MULTI_HASH: bits 6..15 see 2 bytes only */
                                    while (e != 0xFFFFu) {
                                        const uint32_t ent = s_entry[e];
                                        const uint32_t pb2 = (ent >> 8) & 0xFFu;
                                        if (pb2 == 0u || pb2 == b2) {
                                            /* rare: fetch the other 16 text bytes and the pattern record */
                                            const uint32_t uid = ent & 0xFFu;
                                            const uint32_t *rec = s_rec + uid * KMP_MULTI_REC_WORDS;
                                            uint32_t diff = (T0 ^ rec[0]) & rec[5];
                                            uint32_t prev = r1;
#pragma unroll
                                            for (int d = 1; d < 5; ++d) {
                                                const uint32_t nx = src[d + 1];
                                                diff |= (__builtin_amdgcn_alignbyte(nx, prev, sa) ^ rec[d]) & rec[5 + d];
                                                prev = nx;
                                            }
                                            bool hit = diff == 0u && (int32_t)(i + rec[10]) <= rem;
                                            if (hit && (ent & 0x40000000u)) {
                                                /* a pattern of more than 20 bytes whose first 20 matched: exact room up to the slot's end
                                                 * (CLEAN: rem only tells 16 / 32 / more), then the remaining bytes, text from the LDS
                                                 * window, pattern from its kmp_pattern_dev */
                                                const uint32_t m = rec[10];
                                                if constexpr (CLEAN) {
                                                    const uint64_t above = (sg_all >> 1) >> lane;
                                                    uint32_t d = 4096u;
                                                    if (above != 0ull) d = (uint32_t)__builtin_ctzll(above) + 1u;
                                                    else if (nx_all != 0ull) d = 64u - lane + (uint32_t)__builtin_ctzll(nx_all);
                                                    hit = i + m <= d * KMP_LANE_BYTES;
                                                }
                                                if (hit) {
                                                    const uint8_t *pp = patterns[rec[11] - 1u].pat;
                                                    const uint8_t *tw = reinterpret_cast<const uint8_t *>(win) + o;
                                                    for (uint32_t b = KMP_MULTI_PREFIX; b < m; ++b)
                                                        if (tw[b] != pp[b]) { hit = false; break; }
                                                }
                                            }
                                            if (hit) {
                                                atomicAdd(&s_cnt[uid], 1u);
                                                if constexpr (EMIT) {
                                                    /* which packet, and how far into it: from the start bitmap (the packet that
                                                     * holds this lane started at the highest start bit at or below the lane, or
                                                     * before the chunk) */
                                                    const uint64_t st_le = st & ((2ull << lane) - 1ull);
                                                    const uint64_t pkt = kbase + (uint64_t)__builtin_popcountll(st_le);
                                                    const uint32_t pstart = st_le ? cb + (63u - (uint32_t)__builtin_clzll(st_le)) * KMP_LANE_BYTES : last_start;
                                                    const uint32_t offs = cb + o - pstart;
                                                    for (uint32_t t = uid_first[uid]; t < uid_first[uid + 1u]; ++t)      /* duplicates of a pattern are reported one by one */
                                                        emit_match_as<true>(true, pkt, offs, uid_ids[t], em);
                                                }
                                            }
                                        }
                                        e = (ent & 0x80000000u) ? 0xFFFFu : e + 1u;
                                    }
                                }
                            }
                        }
                    }
                    if constexpr (EMIT) {
                        if (st != 0ull) {                                            /* packets that started in this chunk */
                            last_start = cb + (63u - (uint32_t)__builtin_clzll(st)) * KMP_LANE_BYTES;
                            kbase += (uint64_t)__builtin_popcountll(st);
                        }
                    }
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

    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_unique; i += KMP_BLOCK_THREADS)
        partials[(uint64_t)i * gridDim.x + blockIdx.x] = s_cnt[i];
}

}  // namespace

/* Fused multi-pattern pass over a packed arena (bitmap + plan as for kmp_launch_scan_packed). */
hipError_t kmp_launch_scan_multi(const kmp_scan_args &a, const uint32_t *tables, uint32_t table_words, uint32_t n_unique,
                                 const uint32_t *uid_first, const uint32_t *uid_ids, hipStream_t st)
{
    if (n_unique == 0 || a.blocks_x == 0) return hipSuccess;
    const kmp_plan_entry *plan = reinterpret_cast<const kmp_plan_entry *>(a.plan);
    const size_t lds = ((size_t)((table_words - KMP_MULTI_REC_W0 + n_unique + 3u) & ~3u) + KMP_BLOCK_WAVES * KMP_MULTI_WIN_WORDS) * sizeof(uint32_t);
    const Emitter em{reinterpret_cast<uint4 *>(a.emit_out), a.emit_counter, a.emit_cap, 0u};
#define KMP_MULTI_LAUNCH(NT_, CLEAN_, EMIT_) hipLaunchKernelGGL((kmp_scan_multi_kernel<4, NT_, CLEAN_, EMIT_>), dim3(a.blocks_x), dim3(KMP_BLOCK_THREADS), lds, st, \
        a.arena, a.pkt_len, a.bitmap, plan, tables, table_words, n_unique, a.partials, em, uid_first, uid_ids, a.patterns)
    if (a.emit_out) { if (a.pad_clean) KMP_MULTI_LAUNCH(true, true, true); else KMP_MULTI_LAUNCH(true, false, true); }
    else if (a.pad_clean) { if (a.nontemporal) KMP_MULTI_LAUNCH(true, true, false); else KMP_MULTI_LAUNCH(false, true, false); }
    else                  { if (a.nontemporal) KMP_MULTI_LAUNCH(true, false, false); else KMP_MULTI_LAUNCH(false, false, false); }
#undef KMP_MULTI_LAUNCH
    return hipGetLastError();
}

