/*
 * kmp_scan_multi.hip -- fused multi-pattern pass, gfx950.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kmp_device.h"
#include "kmp_launch.h"
#include "kmp_dev_common.h"

namespace {

/* ================================================================================================
 * Fused multi-pattern pass: every pattern of 2..99 bytes in ONE read of a packed
 * arena -- the reference re-reads every payload once per pattern (serial.c:154, openmp_data.c:163).
 *
 * Same streaming skeleton as kmp_scan_packed_kernel (buffer-load ring, packet-start bitmap), but the
 * wavefronts share the arena out in work units as they go (WORK UNITS, below).  Per chunk:
 *   - rem = payload bytes left from the lane's first byte (CLEAN: two lane masks off the start bitmap; otherwise a
 *     uniform loop over the packet starts of the chunk with lengths by scalar loads);
 *   - level 1: "may some pattern start here?" for the 16 start offsets of the lane, TWO offsets per LDS lookup: the text
 *     goes by 5-bit codes (b & 31), an 8-byte entry addressed by the two middle bytes of a 4-byte window says which bytes
 *     may stand before that pair and which may follow it as the start of a pattern (kmp_device.h, pair table), so one
 *     ds_read_b64 decides the offsets at the window's first and second byte.  v_dot4_u32_u8 computes the entry's
 *     address, v_lshrrev_b32_sdwa shifts each word by its text byte, v_dot4 packs the verdicts into the hit mask.
 *     This level is the pass (280 of 300 us, profiles/r03_fused_ablation.txt; vector ALUs 87 % busy, the LDS 63 %);
 *   - hits are not looked at here.  Every lane that has one appends ONE 32-byte record -- its 24 text bytes, its hit
 *     mask, the room up to the payload's end, its position -- to the wavefront's queue in LDS (slot = mbcnt over the
 *     ballot of those lanes: one ballot and two writes per chunk, no loop, no dependent LDS read), and whenever the
 *     queue holds KMP_MULTI_QBATCH records
 *   - level 2 takes them, ONE RECORD PER LANE, every lane busy: for each hit of its record (one, rarely two or three)
 *     the lane cuts the eight text bytes behind the hit offset out of the record (select + v_alignbyte with a per-lane
 *     shift); their first three bytes (five bits of the third) hash to a bucket {first entry, count}; an entry carries
 *     the pattern's first three bytes and its id, v_msad_u8 compares them in one instruction (a 2-byte pattern has
 *     0x00 as third byte, which the masked SAD skips).  Patterns of 2 or 3 bytes -- most matches in text -- are
 *     decided right there; a longer one compares its first eight bytes, and only a pattern of nine bytes or more whose
 *     first eight match reads the rest of the text from the arena itself.  Matches bump the pattern's counter in LDS;
 *     counters go to partials[unique pattern][block].
 *     (Round 1 walked every lane's hits chunk by chunk: 2 rounds of ~40 dependent instructions and three LDS round trips
 *     with 7 % of the lanes doing work, 55 % of the kernel's time.)
 * ============================================================================================== */
/* tuning builds (-DKMP_MULTI_TUNING): cut the kernel after stage n_ (a scalar branch); nothing in the product build */
#ifdef KMP_MULTI_TUNING
#define KMP_MULTI_CUT(n_, stmt_) do { if (ablate != 0u) { asm volatile("" ::: "memory"); if (ablate == (n_)) { stmt_; } } } while (0)
#else
#define KMP_MULTI_CUT(n_, stmt_) do { (void)ablate; } while (0)
#endif
#ifdef KMP_TUNE_STAMPS
/* tuning builds only (tools/fused_timeline.py): when each wavefront entered, had its tables, took its first unit from the pool, left its
 * chunk loop and ended (s_memrealtime, 10 ns ticks) */
__device__ unsigned long long kmp_tune_stamps[16384 * 8];
#define KMP_STAMP(k_) do { if (lane == 0u && gw_ < 16384u) kmp_tune_stamps[gw_ * 8u + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KMP_STAMP(k_) do { } while (0)
#endif
/* static LDS of a block: tables up to the records, the work units' entries */
constexpr size_t KMP_MULTI_STATIC_BYTES = (size_t)KMP_MULTI_REC_W0 * sizeof(uint32_t) + KMP_MULTI_MAX_UNITS * 16u;
constexpr uint32_t QCAP = 80u;                  /* queue records per wavefront (32 bytes each): a batch + the lanes of one chunk */
constexpr uint32_t QBATCH = 64u;                /* records level 2 takes at a time */

/* (base + k) mod QCAP for base < QCAP (wave-uniform) and k < QCAP: no division */
__device__ __forceinline__ uint32_t ring_slot(uint32_t base, uint32_t k) { const uint32_t x = base + k; return min(x, x - QCAP); }
__device__ __forceinline__ uint32_t ring_wrap(uint32_t x) { return x >= QCAP ? x - QCAP : x; }           /* x < 2 * QCAP */

/* Byte D of acc = low byte of word >> (byte K of sel & 31); D == 0 also clears the other three bytes.  SDWA picks the shift
 * amount out of the text register and drops the result into its byte of the accumulator: one instruction per start offset. */
template <int K, int D>
__device__ __forceinline__ void shr_by_byte_into(uint32_t &acc, uint32_t word, uint32_t sel)
{
#define KMP_SHR_SDWA(K_, D_, UNUSED_) asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_" #D_ " dst_unused:" UNUSED_ " src0_sel:BYTE_" #K_ " src1_sel:DWORD" : "+v"(acc) : "v"(sel), "v"(word))
    static_assert((K == 0 && D == 0) || (K == 3 && D == 1) || (K == 2 && D == 2) || (K == 1 && D == 3), "the four start offsets of a dword");
    if constexpr (D == 0)      asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(acc) : "v"(sel), "v"(word));
    else if constexpr (D == 1) KMP_SHR_SDWA(3, 1, "UNUSED_PRESERVE");
    else if constexpr (D == 2) KMP_SHR_SDWA(2, 2, "UNUSED_PRESERVE");
    else                       KMP_SHR_SDWA(1, 3, "UNUSED_PRESERVE");
#undef KMP_SHR_SDWA
}

template <int DEPTH, bool NT, bool CLEAN, bool EMIT, bool ONES, uint32_t WAVES, bool CLASSED>
__device__ __forceinline__ void
kmp_scan_multi_body(const uint8_t *__restrict__ arena, const uint32_t *__restrict__ pkt_len,
                      const unsigned long long *__restrict__ bitmap, const kmp_plan_entry *__restrict__ plan,
                      const uint32_t *__restrict__ tables, uint32_t table_words, uint32_t n_unique, uint32_t cshift, uint32_t bmask,
                      uint32_t n_ones, uint32_t ones, uint32_t ablate, uint32_t n_units, uint32_t upb, uint32_t sides, uint32_t *__restrict__ pool_next, uint64_t span_end, uint32_t pstride, unsigned long long *__restrict__ partials, Emitter em, const uint32_t *__restrict__ uid_first,
                      const uint32_t *__restrict__ uid_ids, const kmp_pattern_dev *__restrict__ patterns)
{
    /* buckets, entries and the filter sit in static LDS: their offsets are compile-time constants that fold
     * into the ds_read offset field; records, counters and one hit queue per wavefront follow dynamically */
    __shared__ __attribute__((aligned(16))) uint32_t s_fix[KMP_MULTI_REC_W0];
    /* the block's work units {first chunk (bytes from the region's first), bytes, lanes before the first packet | bit 31: no such unit |
     * packet index, bits 32 up << 8, packet index}: static as well, an address that is a constant costs no register in the chunk loop */
    __shared__ uint4 s_unit[KMP_MULTI_MAX_UNITS];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const uint32_t rec_words = table_words - KMP_MULTI_REC_W0;
    uint32_t *s_rec = s_dyn;
    uint32_t *s_cnt = s_dyn + rec_words;
    uint4    *s_q   = reinterpret_cast<uint4 *>(s_dyn + ((rec_words + n_unique + 3u) & ~3u));
    const uint32_t lane = threadIdx.x & (KMP_WAVE - 1u);
    const uint32_t wave = sgpr(threadIdx.x >> 6);
    /* WORK UNITS.  The arena is cut into regions and every region into `upb` units of whole packets (plan[] holds first packet and
     * byte offset of every unit, kmp_plan_kernel): one large unit per wavefront, its own share, and a pool of units of a few tens of
     * KiB that the wavefronts TAKE one after the other off a counter when they are through their share, instead of owning a fixed
     * part of the arena each.  The SIMD issues the instructions of its oldest wavefront first, so of eight wavefronts with the same
     * work the first is done in 0.43 of the time the last one takes (profiles/r03_fused_timeline_static_ranges.txt: chunk loops of
     * 79 .. 184 us by hardware wave slot), the block's slot and LDS stay taken until its last wavefront ends.  With units every
     * wavefront runs at whatever speed it gets until the region is used up, and they all end within one pool unit of each other. */
    /* Regions go by PAIRS of blocks (sides == 2; a grid of one block: 1): blocks p and p + pairs share region p.  Its first units are
     * the wavefronts' own shares -- 16 for the one block, 16 for the other --, the rest is the pool both take from, off a counter in
     * global memory.  The hardware starts one block on every CU before it starts a second one anywhere, and the SIMDs serve
     * the wavefronts of the older block first: with a region of its own the first block of a CU was done at 0.64 of the time the
     * second one took (profiles/r03_fused_timeline_units_one_round.txt: blocks 0-255 end at 184 us, blocks 256-511 at 284 us).
     * Nothing here depends on where the blocks of a pair run: whichever is faster takes more of the pool. */
    const uint32_t pairs = gridDim.x / sides;
    const uint32_t pair = blockIdx.x % pairs, side = blockIdx.x / pairs;
    const uint32_t n_own = sides * WAVES;            /* units that are some wavefront's own share */
    const uint32_t ubeg = pair * upb;
#ifdef KMP_TUNE_STAMPS
    const uint32_t gw_ = blockIdx.x * WAVES + wave;
    uint32_t n_taken = 0u;
#endif
    KMP_STAMP(0);
    const uint32_t uhave = ubeg < n_units ? min(upb, n_units - ubeg) : 0u;
    /* positions in the hit queue count from the region's first chunk: a record outlives the unit it was made in */
    const uint64_t blk_off0_ = plan[min(ubeg, n_units)].off & ~(uint64_t)(KMP_CHUNK - 1u);
    const uint64_t blk_off0 = (uint64_t)sgpr((uint32_t)blk_off0_) | ((uint64_t)sgpr((uint32_t)(blk_off0_ >> 32)) << 32);
    /* the unit being read, in bytes from the region's first chunk: the chunk the loop below is at, and the unit's end ... */
    uint32_t pos = 0u, uend = 0u;
    uint64_t k0 = 0ull;                              /* ... its first packet ... */
    /* ... and: bits 0-5 the lanes of its first chunk that lie before its first packet (until that chunk is done), bit 6: the unit after
     * it has been taken off the counter, bits 8-: that unit's number */
    uint32_t aux = 0u;
    /* ONE buffer resource from the region's first chunk to the arena's last slot, never changed (a resource whose record count
     * moved from unit to unit was kept in spilled registers and read back before every load): a load behind the arena fetches
     * nothing and returns zeros; a unit's last chunk may reach into the unit behind it -- those lanes are put to zero where the
     * chunk is looked at (a zero lane can never be a candidate and only ends a packet that ends there anyway).  KMP_NOWHERE: an
     * offset behind every resource, for the loads of a ring round that has nothing to ask for. */
    constexpr uint32_t KMP_NOWHERE = 0x80000000u;
    const uint64_t to_end = span_end - blk_off0;
    const i32x4 rsrc = make_rsrc(arena + blk_off0, (uint32_t)(to_end < 0x7FFF0000ull ? to_end : 0x7FFF0000ull));
    /* What the loop below and level 2 need of the arena, they take from the resource (its base: the region's first chunk; its
     * record count: the bytes up to the arena's last slot, or more than a region has) and from the bitmap word of that chunk: the
     * chunk loop runs with every scalar register the hardware gives eight wavefronts per SIMD (80 less the compiler's reserve), and
     * what it does not hold it reads back from spill lanes -- the resource itself, before every load, at the worst. */
    const unsigned long long *const bwr = bitmap + (blk_off0 >> 10);
    uint32_t iob = 0u;                               /* the chunk at pos asks for the bytes at pos + iob of the region */
    const uint32_t vo0 = lane * KMP_LANE_BYTES;
    u32x4 buf[DEPTH];
    /* unit un of this block as the kernel wants it, from the plan (kmp_plan_kernel) */
    auto unit_of = [&](uint32_t gu) {
        const uint64_t ka = plan[gu].k, kb = plan[gu + 1].k;
        /* the stream starts on the 1 KiB boundary below the first packet (the packed kernel takes the 128-byte line): a chunk
         * is then exactly one 64-bit word of the packet-start bitmap, no funnel shift per chunk */
        const uint64_t off_first = plan[gu].off;
        const uint32_t pre = (uint32_t)(off_first & (uint64_t)(KMP_CHUNK - 1u));
        const uint64_t off0 = off_first - pre;
        return make_uint4((uint32_t)(off0 - blk_off0), (kb > ka) ? (uint32_t)(plan[gu + 1].off - off0) : 0u, (pre >> 4) | ((uint32_t)(ka >> 32) << 8), (uint32_t)ka);
    };
    /* (wave-uniform by construction; sgpr() says so) */
    auto adopt = [&](uint4 e) {
        const uint32_t z = sgpr(e.z);
        pos = sgpr(e.x);
        uend = pos + ((int32_t)z < 0 ? 0u : sgpr(e.y));
        aux = z & 63u;
        if constexpr (!CLEAN || EMIT) k0 = (uint64_t)sgpr(e.w) | ((uint64_t)((z & 0x7FFFFFFFu) >> 8) << 32);
        iob = (uint32_t)DEPTH * KMP_CHUNK;
    };
    /* the next unit of the pool, off the pair's counter (the units' entries are in LDS by then; a number past the region's last unit
     * finds an entry that says so): where its first chunk is.  (The returning atomic is a vector memory instruction: the compiler
     * waits for it with vmcnt(0), which the ring's loads -- the last chunks of the unit that is ending -- reach as well.) */
    auto claim = [&]() {
        uint32_t t = KMP_MULTI_MAX_UNITS;
        if (pool_next != nullptr && lane == 0u)          /* (regions without a pool: no counter, nobody resets one) */
            t = __hip_atomic_fetch_add(pool_next + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t un = min(n_own + min(sgpr(t), KMP_MULTI_MAX_UNITS), KMP_MULTI_MAX_UNITS - 1u);
        aux = (aux & 63u) | 64u | (un << 8);
        const uint4 e = s_unit[un];
        return ((int32_t)sgpr(e.z) >= 0 && sgpr(e.y) != 0u) ? sgpr(e.x) : KMP_NOWHERE;
    };
    auto next_exists = [&]() { return (int32_t)sgpr(s_unit[aux >> 8].z) >= 0; };
    auto adopt_next = [&]() { adopt(s_unit[aux >> 8]); };
    /* the first DEPTH chunk loads of the unit just adopted (every other unit's are issued by the unit before it) */
    auto prologue = [&](auto again) {
        const uint32_t at = pos < uend ? pos : KMP_NOWHERE;
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) flat_issue<NT, true, decltype(again)::value>(buf[s], rsrc, vo0, at + (uint32_t)s * KMP_CHUNK);
    };
    /* The stream starts before the tables are copied: the first DEPTH chunk loads need nothing but the range, and
     * filling 25-35 KB of LDS from global memory takes longer than they do.  The first units go by wavefront number: no LDS yet. */
    const uint4 no_unit = make_uint4(0u, 0u, 0x80000000u, 0u);
    const uint32_t own = side * WAVES + wave;
    adopt(own < uhave ? unit_of(ubeg + own) : no_unit);
    prologue(std::false_type{});
    {
        /* The block's tables into LDS -- ALL the loads first, then the stores: loop by loop (tables, records, units) every part was a
         * round trip to L2 of its own, four in a row, ~4 us before the first chunk could be looked at; a pass over 0.7 GB takes 140. */
        constexpr uint32_t NTHR = WAVES * KMP_WAVE;
        constexpr uint32_t T4 = KMP_MULTI_REC_W0 / 4u, NIT = (T4 + NTHR - 1u) / NTHR;
        const uint4 *t4 = reinterpret_cast<const uint4 *>(tables);
        uint4 *s4 = reinterpret_cast<uint4 *>(s_fix);
        const uint32_t tid = threadIdx.x;
        const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
        uint4 tt[NIT];
#pragma unroll
        for (uint32_t k = 0; k < NIT; ++k) tt[k] = (tid + k * NTHR < T4) ? t4[tid + k * NTHR] : zero4;
        const uint32_t r0 = tid < rec_words ? tables[KMP_MULTI_REC_W0 + tid] : 0u;
        uint4 ue = no_unit;
        if (tid < uhave) ue = unit_of(ubeg + tid);
#pragma unroll
        for (uint32_t k = 0; k < NIT; ++k) if (tid + k * NTHR < T4) s4[tid + k * NTHR] = tt[k];
        if (tid < rec_words) s_rec[tid] = r0;
        if (tid < KMP_MULTI_MAX_UNITS) s_unit[tid] = ue;
        for (uint32_t i = tid + NTHR; i < rec_words; i += NTHR) s_rec[i] = tables[KMP_MULTI_REC_W0 + i];      /* (more than 192 long patterns) */
        static_assert(KMP_MULTI_MAX_UNITS <= NTHR, "one unit entry per thread");
    }
    for (uint32_t i = threadIdx.x; i < n_unique; i += WAVES * KMP_WAVE) s_cnt[i] = 0u;
    KMP_STAMP(1);
    __syncthreads();
    KMP_STAMP(2);
    const uint32_t *s_bucket = s_fix + KMP_MULTI_BUCKET_W0;
    const uint32_t *s_entry  = s_fix + KMP_MULTI_ENTRY_W0;
    const uint8_t  *s_pair   = reinterpret_cast<const uint8_t *>(s_fix + KMP_MULTI_FILTER_W0);
    uint4 *q = s_q + wave * (2u * QCAP);         /* this wavefront's queue: a ring of 32-byte records {24 text bytes, hit mask | room << 16, position} */
    uint32_t q_head = 0u, q_count = 0u;          /* wave-uniform */
    uint32_t one_cnt[KMP_MULTI_MAX_ONES] = {0u, 0u, 0u, 0u};      /* this lane's matches of the 1-byte patterns that ride along */

    {
        /* state of the unit being read */
        unsigned long long hiw[DEPTH] = {};
        unsigned long long low = 0ull;
        uint64_t kcur = 0ull;                /* last packet that has started                                   */
        uint64_t kbase = 0ull;               /* EMIT: the same, kept in both variants of the payload-end logic */
        uint32_t last_start = 0u;            /* EMIT: byte position (from the region's first chunk) of that packet's start */
        int32_t  remc = 0;                   /* payload bytes of that packet left at the chunk's first byte     */
        bool     dead = false;
        /* EMIT only: the start bits of the chunk whose hits are in the queue (the queue is emptied after every chunk there) */
        uint64_t e_st = 0ull;

        /* a word of the packet-start bitmap (wave-uniform; the kernel that also stores offset records reads it with a vector load,
         * and has to be told so) */
        auto start_word = [&](const unsigned long long *p) {
            const unsigned long long x = *p;
            if constexpr (EMIT) return (unsigned long long)sgpr((uint32_t)x) | ((unsigned long long)sgpr((uint32_t)(x >> 32)) << 32);
            else return x;
        };
        /* one match of unique pattern uid at position pos_r of the region */
        auto count_match = [&](uint32_t uid, uint32_t pos_r) {
            atomicAdd(&s_cnt[uid], 1u);
            if constexpr (EMIT) {
                /* which packet, and how far into it: from the start bitmap of the chunk the hit lies in (the packet that
                 * holds the hit's lane started at the highest start bit at or below that lane, or before the chunk) */
                const uint32_t hl = (pos_r - pos) >> 4;             /* (the queue is empty between chunks here) */
                const uint64_t st_le = e_st & ((2ull << hl) - 1ull);
                const uint64_t pkt = kbase + (uint64_t)__builtin_popcountll(st_le);
                const uint32_t pstart = st_le ? pos + (63u - (uint32_t)__builtin_clzll(st_le)) * KMP_LANE_BYTES : last_start;
                for (uint32_t d = uid_first[uid]; d < uid_first[uid + 1u]; ++d)      /* duplicates of a pattern are reported one by one */
                    emit_match_as<true>(true, pkt, pos_r - pstart, uid_ids[d], em);
            }
        };
        /* the patterns of one bucket against the eight text bytes T0, T1 of a hit: bk = {first entry, further entries: count << 16 | first};
         * cw (a classed group, kmp_device.h): the word of the bucket's class -- KMP_MULTI_CLS_WORD: its short patterns, its first record,
         * its first id; a plain group: the number of its short patterns (the kernel's cshift argument is that number there) */
        auto walk = [&](uint32_t T0, uint32_t T1, uint32_t room, uint32_t pos, uint2 bk, uint32_t cw, bool act) {
            uint32_t ent = bk.x;
            uint32_t e = bk.y & 0xFFFFu;
            uint32_t n = act ? (bk.y >> 16) : 0u;                   /* entries left, this one included; a false hit of the filter usually finds an empty bucket */
            while (ballot64(n != 0u) != 0ull) {
                /* first three bytes (two for a 2-byte pattern: its third byte is 0x00 and skipped) */
                const bool m3 = n != 0u && __builtin_amdgcn_msad_u8(T0, ent & 0x00FFFFFFu, 0u) == 0u;
                const uint32_t uid_lo = ent >> 24, uid = CLASSED ? (cw >> 20) + uid_lo : uid_lo;
                const bool lng = uid_lo >= (CLASSED ? cw & 0x1FFu : cw);
                bool hit = m3 && !lng && ((ent & 0x00FF0000u) ? 3u : 2u) <= room;
                if (ballot64(m3 && lng) != 0ull) {
                    /* rare: the first three bytes of a pattern of four bytes or more */
                    if (m3 && lng) {
                        uint32_t m, pidx;
                        bool eight;
                        if constexpr (CLASSED) {
                            const uint2 rec = *reinterpret_cast<const uint2 *>(s_rec + KMP_MULTI_CLS_WORDS + (((cw >> 9) & 0x7FFu) + uid_lo - (cw & 0x1FFu)) * KMP_MULTI_CREC_WORDS);
                            m = (rec.x >> 8) & 0xFFu;
                            pidx = rec.x >> 16;
                            const uint32_t m47 = m >= 8u ? 0xFFFFFFFFu : m > 4u ? (1u << (8u * (m - 4u))) - 1u : 0u;      /* which of the bytes 4-7 the pattern has */
                            eight = (T0 >> 24) == (rec.x & 0xFFu) && ((T1 ^ rec.y) & m47) == 0u && m <= room;
                        } else {
                            const uint4 rec = *reinterpret_cast<const uint4 *>(s_rec + (uid - cw) * KMP_MULTI_REC_WORDS);
                            m = rec.w & 0xFFu;
                            pidx = rec.w >> 8;
                            eight = T0 == rec.x && ((T1 ^ rec.y) & rec.z) == 0u && m <= room;
                        }
                        hit = eight && m <= 8u;
                        if (eight && m > 8u) {
                            /* rarer: nine bytes or more, the first eight match: the rest straight from the arena (a 0x00 of
                             * the slot padding ends the comparison).  A window never leaves its payload (serial.c:193,198), so it
                             * never leaves the arena, which ends with the last slot (span_end): that bound is what holds for the LAST
                             * payload of the index, behind which no packet-start bit follows and the caller's memory may hold
                             * anything (kmpgpu.h: nothing is required, and nothing is read, behind the last slot). */
                            bool ok = pos + m <= (uint32_t)rsrc.z;      /* (positions count from the region's first chunk, as the resource does) */
                            if constexpr (CLEAN) {
                                /* room only tells 16 / 32 / more there: the exact distance to the next packet start */
                                const uint32_t b = (pos >> 4) + 1u;
                                const unsigned long long w0 = bwr[b >> 6], w1 = bwr[(b >> 6) + 1u];
                                const uint32_t s6 = b & 63u;
                                const uint64_t bits = s6 ? ((w0 >> s6) | (w1 << (64u - s6))) : w0;
                                if (bits != 0ull) ok = ok && m <= ((b + (uint32_t)__builtin_ctzll(bits)) << 4) - pos;
                            }
                            if (ok) {
                                const uint8_t *tp = reinterpret_cast<const uint8_t *>((uint64_t)(uint32_t)rsrc.x | ((uint64_t)(uint32_t)rsrc.y << 32)) + pos;
                                const uint8_t *pp = patterns[pidx].pat;
                                for (uint32_t b = 8u; b < m; ++b)
                                    if (tp[b] != pp[b]) { ok = false; break; }
                            }
                            hit = ok;
                        }
                    }
                }
                if (hit) count_match(uid, pos);
                n -= n != 0u ? 1u : 0u;
                if (ballot64(n != 0u) == 0ull) break;
                ent = s_entry[e];                                   /* a bucket with more than one pattern */
                ++e;
            }
        };
        /* Level 2 on the `nproc` oldest queue records, one per lane, ONE hit each: the rare record that holds another hit
         * (one in nine) goes back to the end of the queue with that hit left in its mask, so every round of this stage
         * runs with all its lanes busy instead of looping for the few records that need it. */
        auto process_batch = [&](uint32_t nproc) {
            const uint32_t slot = ring_slot(q_head, lane);
            const uint4 r0 = q[2u * slot], r1 = q[2u * slot + 1u];
            const uint32_t t[6] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y};        /* the lane's 16 text bytes and the 8 behind them */
            const bool act = lane < nproc;
            const uint32_t hm = act ? (r1.z & 0xFFFFu) : 0u;
            const uint32_t rem = r1.z >> 16;                                     /* payload bytes from the record's first text byte (clamped) */
            const uint32_t i = (uint32_t)__builtin_ctz(hm | 0x10000u);
            const uint32_t rest = hm & (hm - 1u);
            /* the eight text bytes behind offset i */
            const uint32_t q4 = i >> 2;
            const uint32_t x0 = q4 == 1u ? t[1] : q4 == 2u ? t[2] : q4 == 3u ? t[3] : t[0];
            const uint32_t x1 = q4 == 1u ? t[2] : q4 == 2u ? t[3] : q4 == 3u ? t[4] : t[1];
            const uint32_t x2 = q4 == 1u ? t[3] : q4 == 2u ? t[4] : q4 == 3u ? t[5] : t[2];
            const uint32_t T0 = __builtin_amdgcn_alignbyte(x1, x0, i);             /* shift = i & 3 bytes */
            const uint32_t T1 = __builtin_amdgcn_alignbyte(x2, x1, i);
            const uint32_t hx = (uint32_t)__umul24(T0 & bmask, KMP_MULTI_MUL) >> 22;                                           /* KMP_MULTI_HASH */
            const uint2 bk = reinterpret_cast<const uint2 *>(s_bucket)[hx];
            const uint32_t cw = CLASSED ? s_rec[hx >> cshift] : cshift;             /* the bucket's class */
            q_head = ring_wrap(q_head + nproc);
            q_count -= nproc;
            const uint64_t again = ballot64(rest != 0u);
            if (again != 0ull) {
                /* at most `nproc` (<= 64) records come back; QCAP - 64 stayed at most */
                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(again >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)again, ring_wrap(q_head + q_count)));
                if (rest != 0u) {
                    const uint32_t s2 = min(at, at - QCAP);
                    q[2u * s2]      = r0;
                    q[2u * s2 + 1u] = make_uint4(r1.x, r1.y, rest | (rem << 16), r1.w);
                }
                q_count += (uint32_t)__builtin_popcountll(again);
            }
            walk(T0, T1, rem > i ? rem - i : 0u, r1.w + i, bk, cw, act);
        };

        for (;;) {
        if (pos < uend) {
            {
                const unsigned long long *const bw = bwr + (pos >> 10);         /* one word per chunk; positions and bwr both count from the region's first chunk */
#pragma unroll
                for (int s = 0; s < DEPTH; ++s) hiw[s] = start_word(bw + s + 1);
                low = start_word(bw);
            }
            kcur = k0 - 1ull;
            kbase = k0 - 1ull;
            last_start = 0u;
            remc = 0;
            dead = false;
        }
        while (pos < uend) {
            if (pos + (uint32_t)DEPTH * KMP_CHUNK >= uend) {
                /* The last round of this unit: everything it has is loaded or in flight, and what this round asks for are the
                 * first chunks of the NEXT unit -- taken off the block's counter here, its entry read from LDS, the offset moved
                 * over to it.  The wavefront goes from unit to unit without emptying its ring: a unit
                 * that starts with plan entry, resource and first loads one after the other costs ~5 us that the other
                 * wavefronts of the SIMD do not cover (profiles/r03_tried_all_units_dynamic.txt). */
                asm volatile("" ::: "memory");
                iob = claim() - pos;
            }
            /* packet-start words: use this group's, then ask for the next group's (see kmp_scan_packed_kernel) */
            uint64_t st_[DEPTH];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                const unsigned long long hi = hiw[s];
                st_[s] = low;
                low = hi;
                asm volatile("" : "+s"(st_[s]));      /* computed HERE, not sunk below the loads that follow */
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const unsigned long long *const bwj = bwr + (pos >> 10);      /* (the word of this round's first chunk: one address, three offsets) */
#pragma unroll
                for (int s = 0; s < DEPTH; ++s) hiw[s] = start_word(bwj + (DEPTH + 1 + s));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                ring_wait<DEPTH - 2>(buf[s], buf[(s + 1) % DEPTH]);
                if (pos < uend) {
                    uint4 v = make_uint4(buf[s].x, buf[s].y, buf[s].z, buf[s].w);
                    const u32x4 bn = buf[(s + 1) % DEPTH];
                    uint64_t st = st_[s];
                    /* (real branches: both cases happen once per range, the selects they would otherwise become cost every chunk) */
                    if (s == 0 && (aux & 63u) != 0u) {                                  /* the unit's first chunk, lanes before its first packet: not ours */
                        asm volatile("" ::: "memory");
                        const uint32_t pl = aux & 63u;
                        aux &= ~63u;
                        if (lane < pl) v = make_uint4(0u, 0u, 0u, 0u);
                        st &= ~0ull << pl;
                    }
                    const uint32_t left = uend - pos;
                    if (left < KMP_CHUNK) {                                             /* the unit ends inside this chunk: what follows is another unit's */
                        asm volatile("" ::: "memory");
                        if (lane >= (left >> 4)) v = make_uint4(0u, 0u, 0u, 0u);
                        st &= (1ull << (left >> 4)) - 1ull;
                    }

                    const uint32_t w[5] = {v.x, v.y, v.z, v.w, wave_shl1(v.x, sgpr(bn.x))};

                    const uint32_t z[4] = {zero_byte_mask(w[0]), zero_byte_mask(w[1]), zero_byte_mask(w[2]), zero_byte_mask(w[3])};
                    const uint32_t zm = z[0] | z[1] | z[2] | z[3];
                    const uint64_t zl = ballot64(zm != 0u);
                    const bool dead_in = dead;
                    if (zl != 0ull) {
                        asm volatile("" ::: "memory");
                        dead = (st == 0ull) ? true : ((zl >> (63u - (uint32_t)__builtin_clzll(st))) != 0ull);
                    } else dead = dead && st == 0ull;

                    /* rem: payload bytes left from this lane's first byte; last_lanes: the lanes behind which a packet starts */
                    int32_t rem = 0;
                    uint64_t last_lanes;
                    {
                        uint64_t nx;                                                /* start bits of the next chunk */
                        if (s + 1 < DEPTH) nx = st_[(s + 1) % DEPTH];
                        else               nx = low;
                        const uint64_t sg = st_[s];                                 /* not cut at the range's end: the next wavefront's first start ends our last slot */
                        last_lanes = (sg >> 1) | (nx << 63);
                        if constexpr (CLEAN) {
                            /* Slot padding is all 0x00 (kmp_check_padding_kernel), so the payload's end can be replaced by
                             * the slot's end: a window that reaches into the padding holds a 0x00 and matches nothing.
                             * Only lanes within 2 x 16 bytes of the next packet start are constrained (the queue entry
                             * carries 8 text bytes; longer patterns get their exact room in level 2): two lane masks from
                             * the start bitmap, no payload lengths, no loop. */
                            const bool next1 = __builtin_amdgcn_inverse_ballot_w64(last_lanes);                      /* lane + 1 starts a packet */
                            const bool next2 = __builtin_amdgcn_inverse_ballot_w64((sg >> 2) | (nx << 62));         /* lane + 2 does            */
                            rem = next1 ? 16 : next2 ? 32 : (1 << 20);
                        }
                    }
                    if constexpr (!CLEAN) {
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

                    /* level 1: which start offsets may begin some pattern (filter over the first three bytes)?  The bytes go
                     * by their low five bits (x: the codes, x8: eight times the codes).  One 8-byte table entry, addressed by the
                     * two MIDDLE bytes of a 4-byte window (v_dot4_u32_u8 with the weights {0, 1, 33, 0} on x8 is the entry's
                     * byte offset), decides the start offsets at the window's first and second byte: word 0 has a bit per
                     * byte that may stand before the middle pair, word 1 a bit per byte that may follow it (kmp_device.h).
                     * Each word is shifted by its byte (SDWA picks the byte out of x, no extraction) and v_alignbit pushes
                     * bit 0 of the result into the hit mask: 8 lookups and ~52 VALU per 16 text bytes. */
                    uint32_t x8[5];
#pragma unroll
                    for (int q4 = 0; q4 < 5; ++q4) x8[q4] = (w[q4] & 0x1F1F1F1Fu) << 3;
                    uint2 e[8];
                    /* start offsets 4q, 4q + 1: the window is dword q (e[2q]); 4q + 2, 4q + 3: bytes 2 .. 5 from dword q (e[2q + 1]).  The even
                     * entries are asked for first: their addresses need no v_alignbyte, and the shifts below take them first.
                     * (Two 4-byte tables read by one ds_read2st64_b32 would save the five shifts -- the entry's offset is then a v_dot4 of
                     * the masked text -- and cost a quarter of the pass: 389 against 314 us, profiles/r03_tried_split_pair_table.txt;
                     * the LDS serves an 8-byte entry in one access, two words 4352 bytes apart in two.) */
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        e[2 * q4] = *reinterpret_cast<const uint2 *>(s_pair + __builtin_amdgcn_udot4(x8[q4], 0x00210100u, 0u, false));
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const uint32_t y8 = __builtin_amdgcn_alignbyte(x8[q4 + 1], x8[q4], 2u);
                        e[2 * q4 + 1] = *reinterpret_cast<const uint2 *>(s_pair + __builtin_amdgcn_udot4(y8, 0x00210100u, 0u, false));
                    }
                    /* all eight lookups are on their way before the first result is used */
                    __builtin_amdgcn_sched_barrier(0);
                    /* bit 0 of (word >> byte) is the verdict: the four of a dword land in the four bytes of one register, and a
                     * v_dot4 with the weights {1, 2, 4, 8} ({16, 32, 64, 128} for the odd dwords) packs them into the hit mask.  The
                     * sixteen shifts go round the four registers (an SDWA write into a register and the next instruction that touches it
                     * need a wait state between them: back to back, the assembler fills it with an s_nop -- an issue slot like any other) */
                    uint32_t hb[4];
                    /* two waits for the eight lookups (the LDS answers in order: four outstanding = the even entries are here), not one
                     * per first use of an entry */
                    __builtin_amdgcn_s_waitcnt(0xC47F);             /* lgkmcnt(4) */
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) shr_by_byte_into<0, 0>(hb[q4], e[2 * q4].x, w[q4]);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) shr_by_byte_into<3, 1>(hb[q4], e[2 * q4].y, w[q4]);
                    __builtin_amdgcn_s_waitcnt(0xC07F);             /* lgkmcnt(0) */
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) shr_by_byte_into<2, 2>(hb[q4], e[2 * q4 + 1].x, w[q4]);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) shr_by_byte_into<1, 3>(hb[q4], e[2 * q4 + 1].y, w[q4 + 1]);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) hb[q4] &= 0x01010101u;
                    uint32_t hmq[2];
                    hmq[0] = __builtin_amdgcn_udot4(hb[0], 0x08040201u, 0u, false);
                    hmq[1] = __builtin_amdgcn_udot4(hb[2], 0x08040201u, 0u, false);
                    hmq[0] = __builtin_amdgcn_udot4(hb[1], 0x80402010u, hmq[0], false);
                    hmq[1] = __builtin_amdgcn_udot4(hb[3], 0x80402010u, hmq[1], false);
                    uint32_t hm = hmq[0] | (hmq[1] << 8);
#if defined(KMP_MULTI_TUNING) && defined(KMP_TUNE_PAD_VALU)
                    /* sensitivity probe of tuning builds: KMP_TUNE_PAD_VALU extra two-operand VALU instructions per chunk */
#pragma unroll
                    for (int pad = 0; pad < KMP_TUNE_PAD_VALU; ++pad) asm volatile("v_and_b32 %0, %0, %0" : "+v"(hm));
#endif
#if defined(KMP_MULTI_TUNING) && defined(KMP_TUNE_PAD_SALU)
                    {   /* ... and KMP_TUNE_PAD_SALU extra scalar instructions */
                        uint32_t spad = pos;
#pragma unroll
                        for (int pad = 0; pad < KMP_TUNE_PAD_SALU; ++pad) asm volatile("s_and_b32 %0, %0, %0" : "+s"(spad) : : "scc");
                        asm volatile("" :: "s"(spad));
                    }
#endif
                    KMP_MULTI_CUT(1u, hm = 0u);
                    const uint64_t hl0 = ballot64(hm != 0u);
                    if (ONES || hl0 != 0ull) {
                        /* keep only the start offsets that can count: at least the shortest pattern still inside the payload
                         * and no 0x00 before them (strlen rule, serial.c:191).  Nearly every 0x00 of real traffic and all of the
                         * synthetic input's sit in the LAST lane of a packet (slot padding, trailers): those end nothing but
                         * their own lane, and there only for a lane that has a hit -- the general, segmented form
                         * (nul_limit) is kept for a 0x00 in mid-packet. */
                        int32_t nl = 15;                                             /* last start offset no 0x00 precedes */
                        /* (the 1-byte patterns over clean padding take care of a lane's own 0x00 themselves, below) */
                        constexpr bool ONES_BY_MASK = ONES && CLEAN && !EMIT;
                        const bool seg = dead_in || (zl & ~last_lanes) != 0ull;
                        /* (the hit mask is cut down where nl is worked out, inside these branches: applied behind them it became nine
                         * instructions of selects on every chunk, with nl == 15 in all lanes on nearly all of them) */
                        if (seg) {
                            asm volatile("" ::: "memory");
                            nl = nul_limit(15, w, zl, st, dead_in, lane);
                            hm = (nl < 0) ? 0u : (hm & ((2u << nl) - 1u));
                        } else if constexpr (ONES && !ONES_BY_MASK) {
                            if (zl != 0ull) {
                                asm volatile("" ::: "memory");
                                nl = nul_limit(15, w, 0ull, st, false, lane);                       /* the 1-byte patterns below count against nl */
                                hm = (nl < 0) ? 0u : (hm & ((2u << nl) - 1u));
                            }
                        } else if ((zl & hl0) != 0ull) {
                            /* A 0x00 in the last lane of its packet (slot padding, a trailer) bars that lane's own later start offsets and
                             * nothing else: zb has bit 4q + b for byte b of dword q (the has-zero masks hold 0x80 per zero byte; v_dot4 packs
                             * them), and the hits that stay are those below its lowest bit -- all of them in a lane without a 0x00.  Eleven
                             * instructions where the segmented form (nul_limit) takes ~40; with payloads of a few hundred bytes it ran
                             * on four chunks in ten. */
                            const uint32_t zlo = __builtin_amdgcn_udot4(z[1] >> 7, 0x80402010u, __builtin_amdgcn_udot4(z[0] >> 7, 0x08040201u, 0u, false), false);
                            const uint32_t zhi = __builtin_amdgcn_udot4(z[3] >> 7, 0x80402010u, __builtin_amdgcn_udot4(z[2] >> 7, 0x08040201u, 0u, false), false);
                            const uint32_t zb = zlo | (zhi << 8);
                            hm &= (zb - 1u) & ~zb;
                        }
                        /* (the payload's end is not applied to the hit mask: level 2 checks every hit's room, m <= rem - offset) */
                        if (ONES_BY_MASK && !seg) {
                            /* The 1-byte patterns, common case: no 0x00 but in the last lane of a packet, padding all 0x00.  A text byte
                             * counts iff it equals the pattern's byte and no 0x00 precedes it in its lane (the padding cannot match, the
                             * payload's end needs no test): vm[q] has 0x80 for the bytes of dword q below the lane's first 0x00 --
                             * ~z & (z - 1) & 0x80808080 per dword, nothing behind a dword that holds one; the haszero mask z may
                             * flag a 0x01 ABOVE a real 0x00, which is behind the first one anyway -- and the exact zero-byte test of
                             * w ^ byte, ~(((x & 0x7F..) + 0x7F..) | x), is ANDed with it in the same v_bitop3.  Five instructions
                             * and one v_bcnt per dword and pattern (the v_mqsad form below: 7.7 ns per v_mqsad alone). */
                            uint32_t vm[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
                            if (zl != 0ull) {
                                asm volatile("" ::: "memory");
                                vm[0] = ~z[0] & (z[0] - 1u) & 0x80808080u;
                                vm[1] = z[0] ? 0u : (~z[1] & (z[1] - 1u) & 0x80808080u);
                                vm[2] = (z[0] | z[1]) ? 0u : (~z[2] & (z[2] - 1u) & 0x80808080u);
                                vm[3] = (z[0] | z[1] | z[2]) ? 0u : (~z[3] & (z[3] - 1u) & 0x80808080u);
                            }
#pragma unroll
                            for (uint32_t k = 0; k < KMP_MULTI_MAX_ONES; ++k) {
                                if (k >= n_ones) break;
                                const uint32_t ref4 = ((ones >> (8u * k)) & 0xFFu) * 0x01010101u;
                                uint32_t c = one_cnt[k];
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) {
                                    const uint32_t x = w[q4] ^ ref4;
                                    const uint32_t t7 = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
                                    c += (uint32_t)__builtin_popcount(~(t7 | x) & vm[q4]);
                                }
                                one_cnt[k] = c;
                            }
                        } else if constexpr (ONES) {
                            /* the 1-byte patterns: their byte against all 16 start offsets (v_mqsad with a one-byte reference),
                             * counted where the offset lies inside the payload and before any 0x00 */
                            const int32_t lim1 = min(nl, rem - 1);
                            const bool barred = ballot64(lim1 < 15) != 0ull;
                            const uint32_t nv = (uint32_t)min(max(lim1 + 1, 0), 16);
                            const uint32_t nv2 = nv | (nv << 16);
#pragma unroll
                            for (uint32_t k = 0; k < KMP_MULTI_MAX_ONES; ++k) {
                                if (k >= n_ones) break;
                                const uint32_t ref = (ones >> (8u * k)) & 0xFFu;
                                uint32_t found = 0u;
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) {
                                    const uint64_t S = mqsad(w[q4], w[q4 + 1], ref, 0ull);
                                    if constexpr (!EMIT) {
                                        Emitter none{};
                                        uint32_t dummy = 0u;
                                        found = tally_group<false>(q4, S, barred, nv2, found, 0u, dummy, 0ull, none);
                                    } else {
#pragma unroll
                                        for (int a = 0; a < 4; ++a) {
                                            const bool ok = ((S >> (16 * a)) & 0xFFFFull) == 0ull && (4 * q4 + a) <= lim1;
                                            one_cnt[k] += ok ? 1u : 0u;
                                            if (ballot64(ok) != 0ull) {
                                                const uint64_t st_le = st & ((2ull << lane) - 1ull);
                                                const uint64_t pkt = kbase + (uint64_t)__builtin_popcountll(st_le);
                                                const uint32_t pstart = st_le ? pos + (63u - (uint32_t)__builtin_clzll(st_le)) * KMP_LANE_BYTES : last_start;
                                                const uint32_t row = n_unique - n_ones + k;
                                                for (uint32_t u = uid_first[row]; u < uid_first[row + 1u]; ++u)
                                                    emit_match_as<true>(ok, pkt, pos + vo0 + (uint32_t)(4 * q4 + a) - pstart, uid_ids[u], em);
                                            }
                                        }
                                    }
                                }
                                if constexpr (!EMIT) one_cnt[k] += (found & 0xFFFFu) + (found >> 16);
                            }
                        }
                        KMP_MULTI_CUT(2u, hm = 0u);
                        const uint64_t hl_ = ballot64(hm != 0u);                    /* the lanes that have a hit */
                        if (hl_ != 0ull) {
                            /* append one record per such lane to the queue */
                            const uint32_t nnew = (uint32_t)__builtin_popcountll(hl_);
                            while (q_count + nnew > QCAP) process_batch(min(q_count, 64u));  /* every batch resolves one hit per record: it ends */
                            const uint32_t w5 = wave_shl1(v.y, sgpr(bn.y));        /* text bytes 20..23 from the lane's first: a hit near its end carries 8 bytes too */
                            /* (the count of the lanes below starts from the queue's end: no add) */
                            const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(hl_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hl_, ring_wrap(q_head + q_count)));
                            if (hm != 0u) {
                                const uint32_t slot = min(at, at - QCAP);
                                q[2u * slot]      = v;
                                q[2u * slot + 1u] = make_uint4(w[4], w5, hm | ((uint32_t)min(max(rem, 0), 0xFFFF) << 16), pos + vo0);
                            }
                            q_count += nnew;
                            KMP_MULTI_CUT(3u, (q_head = 0u, q_count = 0u));
                            if constexpr (EMIT) {
                                e_st = st;
                                while (q_count != 0u) process_batch(min(q_count, 64u));
                            } else {
                                while (q_count >= QBATCH) process_batch(QBATCH);
                            }
                        }
                    }
                    if constexpr (EMIT) {
                        if (st != 0ull) {                                            /* packets that started in this chunk */
                            last_start = pos + (63u - (uint32_t)__builtin_clzll(st)) * KMP_LANE_BYTES;
                            kbase += (uint64_t)__builtin_popcountll(st);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                flat_issue<NT>(buf[s], rsrc, vo0, pos + iob);
                pos += KMP_CHUNK;
            }
        }
        if (aux & 64u) {
            /* the unit is read; the next one's first chunks are on their way (or nothing is, if there is none) */
            if (!next_exists()) break;
#ifdef KMP_TUNE_STAMPS
            if (n_taken++ == 0u) KMP_STAMP(6);
#endif
            adopt_next();
        } else {
            /* an empty unit (its first packet is longer than the unit, or a wavefront that got none at the start): its loads fetched nothing */
#pragma unroll
            for (int s = 0; s < DEPTH; s += 2) ring_wait<0>(buf[s], buf[(s + 1) % DEPTH]);
            (void)claim();                   /* (a wavefront without a share of its own: a region shorter than 32 units has no pool either) */
            if (!next_exists()) break;
            adopt_next();
            prologue(std::true_type{});
        }
        }
#pragma unroll
        for (int s = 0; s < DEPTH; s += 2) ring_wait<0>(buf[s], buf[(s + 1) % DEPTH]);     /* nothing in flight when the wavefront ends */
        KMP_STAMP(3);
        /* what is left in the queue */
        while (q_count != 0u) process_batch(min(q_count, 64u));
    }
    KMP_STAMP(4);
#ifdef KMP_TUNE_STAMPS
    if (lane == 0u && gw_ < 16384u) kmp_tune_stamps[gw_ * 8u + 7u] = n_taken;
#endif

#pragma unroll
    for (uint32_t k = 0; k < (ONES ? KMP_MULTI_MAX_ONES : 0u); ++k) {
        if (k >= n_ones) break;
        uint32_t c = one_cnt[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0u) atomicAdd(&s_cnt[n_unique - n_ones + k], c);
    }
    __syncthreads();
    /* partials[row][blocks_x]: there are fewer blocks here than columns; the columns nobody counts into are zeroed */
    for (uint32_t i = wave * KMP_WAVE + lane; i < n_unique; i += WAVES * KMP_WAVE) {       /* (not threadIdx.x: a register kept, or spilled, through the whole kernel) */
        partials[(uint64_t)i * pstride + blockIdx.x] = s_cnt[i];
        for (uint32_t c = blockIdx.x + gridDim.x; c < pstride; c += gridDim.x) partials[(uint64_t)i * pstride + c] = 0ull;
    }
    KMP_STAMP(5);
}

/* Three entry points.  The counting pass over clean padding (every arena this library builds) is held to 64 VGPRs, so that
 * two 16-wavefront blocks (8 wavefronts per SIMD) share a CU.  With 1-byte patterns riding along, or with the per-packet
 * length loop of unclean padding, it needs up to ~80 registers: "wide", 12-wavefront blocks, two of them = 6 wavefronts per
 * SIMD.  The pass that also writes offset records needs ~125: one 16-wavefront block per CU.  (The plan is cut for the
 * wavefronts that are resident at once, kmp_multi_resident_waves(); a block counts in whichever size its kernel has.) */
#define KMP_MULTI_PARAMS const uint8_t *__restrict__ arena, const uint32_t *__restrict__ pkt_len, const unsigned long long *__restrict__ bitmap,          \
                         const kmp_plan_entry *__restrict__ plan, const uint32_t *__restrict__ tables, uint32_t table_words, uint32_t n_unique,         \
                         uint32_t cshift, uint32_t bmask, uint32_t n_ones, uint32_t ones, uint32_t ablate, uint32_t n_units, uint32_t upb,             \
                         uint32_t sides, uint32_t *__restrict__ pool_next, uint64_t span_end, uint32_t pstride,                                                                                           \
                         unsigned long long *__restrict__ partials, Emitter em, const uint32_t *__restrict__ uid_first,                                 \
                         const uint32_t *__restrict__ uid_ids, const kmp_pattern_dev *__restrict__ patterns
#define KMP_MULTI_ARGS arena, pkt_len, bitmap, plan, tables, table_words, n_unique, cshift, bmask, n_ones, ones, ablate, n_units, upb, sides, pool_next, span_end, pstride, partials, em, uid_first, uid_ids, patterns

template <int DEPTH, bool NT, bool CLEAN, bool ONES, bool CLASSED>
__global__ void __launch_bounds__(KMP_MULTI_BLOCK_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
kmp_scan_multi_kernel(KMP_MULTI_PARAMS)
{
    kmp_scan_multi_body<DEPTH, NT, CLEAN, false, ONES, KMP_MULTI_BLOCK_WAVES, CLASSED>(KMP_MULTI_ARGS);
}

template <int DEPTH, bool NT, bool CLEAN, bool ONES, bool CLASSED>
__global__ void __launch_bounds__(KMP_MULTI_WIDE_WAVES * KMP_WAVE) __attribute__((amdgpu_waves_per_eu(6, 6)))
kmp_scan_multi_wide_kernel(KMP_MULTI_PARAMS)
{
    kmp_scan_multi_body<DEPTH, NT, CLEAN, false, ONES, KMP_MULTI_WIDE_WAVES, CLASSED>(KMP_MULTI_ARGS);
}

template <int DEPTH, bool NT, bool CLEAN, bool ONES, bool CLASSED>
__global__ void __launch_bounds__(KMP_MULTI_BLOCK_THREADS)
kmp_scan_multi_emit_kernel(KMP_MULTI_PARAMS)
{
    kmp_scan_multi_body<DEPTH, NT, CLEAN, true, ONES, KMP_MULTI_BLOCK_WAVES, CLASSED>(KMP_MULTI_ARGS);
}

}  // namespace

#ifdef KMP_TUNE_STAMPS
extern "C" int kmp_tune_read_stamps(unsigned long long *dst, size_t words)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(kmp_tune_stamps), words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

/* LDS one block of `waves` wavefronts of the fused pass takes: static tables + records + counters + one hit queue per wavefront. */
size_t kmp_multi_lds_bytes(uint32_t table_words, uint32_t n_unique, uint32_t waves)
{
    return KMP_MULTI_STATIC_BYTES + ((size_t)((table_words - KMP_MULTI_REC_W0 + n_unique + 3u) & ~3u) + waves * QCAP * 8u) * sizeof(uint32_t);
}

/* Which entry point a fused launch takes (0 counting, 1 wide, 2 with offset records), and how many wavefronts of it a CU
 * holds at once: what the wavefront plan of the scan is cut for. */
int kmp_multi_kind(bool emit, bool pad_clean, uint32_t n_ones) { return emit ? 2 : (n_ones != 0u || !pad_clean) ? 1 : 0; }
uint32_t kmp_multi_block_waves(int kind) { return kind == 1 ? KMP_MULTI_WIDE_WAVES : KMP_MULTI_BLOCK_WAVES; }
uint32_t kmp_multi_resident_waves(int kind, uint32_t table_words, uint32_t n_unique)
{
    const uint32_t bw = kmp_multi_block_waves(kind);
    const uint32_t by_regs = (kind == 0 ? 32u : kind == 1 ? 24u : 16u) / bw;             /* blocks per CU the registers allow: 8 / 6 / 4 wavefronts per SIMD */
    const uint32_t by_lds = (uint32_t)((160u * 1024u) / (kmp_multi_lds_bytes(table_words, n_unique, bw) + 512u));
    return bw * (by_lds < by_regs ? (by_lds ? by_lds : 1u) : by_regs);
}

/* Fused multi-pattern pass over a packed arena (packet-start bitmap as for kmp_launch_scan_packed; a.plan: the work units of
 * a.fused_blocks regions, a.units_per_block each). */
hipError_t kmp_launch_scan_multi(const kmp_scan_args &a, const uint32_t *tables, uint32_t table_words, uint32_t n_unique, uint32_t cshift, uint32_t bucket_mask, uint32_t n_ones, uint32_t ones,
                                 const uint32_t *uid_first, const uint32_t *uid_ids, hipStream_t st)
{
    if (n_unique == 0 || a.blocks_x == 0 || a.fused_blocks == 0) return hipSuccess;
    const kmp_plan_entry *plan = reinterpret_cast<const kmp_plan_entry *>(a.plan);
    const int kind = kmp_multi_kind(a.emit_out != nullptr, a.pad_clean, n_ones);
    const uint32_t bwaves = kmp_multi_block_waves(kind);
    const bool classed = a.fused_classed;
    const size_t lds = kmp_multi_lds_bytes(table_words, n_unique, bwaves) - KMP_MULTI_STATIC_BYTES;      /* the dynamic part */
    const Emitter em{reinterpret_cast<uint4 *>(a.emit_out), a.emit_counter, a.emit_cap, 0u};
    /* tuning builds only (make HIPFLAGS+=-DKMP_MULTI_TUNING; tools/fused_ablation.py, profiles/r02_fused_ablation.txt): cut the
     * kernel after a stage -- 1 = level 1 alone, 2 = + hit masking, 3 = + queueing; the counts are wrong then.  The product
     * build passes the constant 0. */
#ifdef KMP_MULTI_TUNING
    static const uint32_t ablate = []() { const char *e = getenv("KMP_MULTI_ABLATE"); return e ? (uint32_t)atoi(e) : 0u; }();
#else
    const uint32_t ablate = 0u;
#endif
#define KMP_MULTI_LAUNCH1(KERNEL_, NT_, CLEAN_, ONES_) KMP_MULTI_LAUNCH0(KERNEL_, NT_, CLEAN_, ONES_, false)
#define KMP_MULTI_LAUNCH0(KERNEL_, NT_, CLEAN_, ONES_, CLASSED_) hipLaunchKernelGGL((KERNEL_<3, NT_, CLEAN_, ONES_, CLASSED_>), dim3(a.fused_blocks), \
        dim3(bwaves * KMP_WAVE), lds, st, a.arena, a.pkt_len, a.bitmap, plan, tables, table_words, n_unique, cshift, bucket_mask, n_ones, ones, ablate, a.n_units, a.units_per_block, a.fused_sides, a.fused_pool, a.span_end, a.blocks_x, \
        a.partials, em, uid_first, uid_ids, a.patterns)
#define KMP_MULTI_LAUNCH(EMIT_K_, NT_, CLEAN_) do {                                                                               \
        if (classed) {        /* (a classed group has no 1-byte patterns riding along) */                                             \
            if (EMIT_K_) KMP_MULTI_LAUNCH0(kmp_scan_multi_emit_kernel, NT_, CLEAN_, false, true);                                     \
            else if (!(CLEAN_)) KMP_MULTI_LAUNCH0(kmp_scan_multi_wide_kernel, NT_, CLEAN_, false, true);                              \
            else KMP_MULTI_LAUNCH0(kmp_scan_multi_kernel, NT_, true, false, true);                                                    \
        } else if (EMIT_K_) { if (n_ones) KMP_MULTI_LAUNCH1(kmp_scan_multi_emit_kernel, NT_, CLEAN_, true); else KMP_MULTI_LAUNCH1(kmp_scan_multi_emit_kernel, NT_, CLEAN_, false); } \
        else if (n_ones) KMP_MULTI_LAUNCH1(kmp_scan_multi_wide_kernel, NT_, CLEAN_, true);                                           \
        else if (!(CLEAN_)) KMP_MULTI_LAUNCH1(kmp_scan_multi_wide_kernel, NT_, CLEAN_, false);                                       \
        else KMP_MULTI_LAUNCH1(kmp_scan_multi_kernel, NT_, true, false); } while (0)
    if (a.emit_out) { if (a.pad_clean) KMP_MULTI_LAUNCH(true, true, true); else KMP_MULTI_LAUNCH(true, true, false); }
    else if (a.pad_clean) { if (a.nontemporal) KMP_MULTI_LAUNCH(false, true, true); else KMP_MULTI_LAUNCH(false, false, true); }
    else                  { if (a.nontemporal) KMP_MULTI_LAUNCH(false, true, false); else KMP_MULTI_LAUNCH(false, false, false); }
#undef KMP_MULTI_LAUNCH
#undef KMP_MULTI_LAUNCH1
#undef KMP_MULTI_LAUNCH0
    return hipGetLastError();
}
