/* kmp_launch.h -- launch entry points of kmp_scan_*.hip / kmp_prep.hip, used by the C-ABI layer (kmpgpu.hip). */
#ifndef KMP_LAUNCH_H
#define KMP_LAUNCH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmp_device.h"

struct kmp_scan_args {
    const uint8_t         *arena;
    const uint64_t        *pkt_off;
    const uint32_t        *pkt_len;
    uint64_t               n_pkts;
    const kmp_pattern_dev *patterns;     /* all patterns, file order                         */
    const uint32_t        *pat_ids;      /* pattern index handled by blockIdx.y              */
    uint32_t               n_ids;        /* gridDim.y                                        */
    unsigned long long    *partials;     /* [n_ids][blocks_x]                                */
    unsigned long long    *zero_counts;  /* flat / packed kernels: counts[] to put to 0 for the patterns of this launch (sliced reduce without accumulation), or NULL */
    uint32_t               blocks_x;
    int                    depth;        /* chunk loads in flight per wavefront              */
    int                    mode;         /* 0 filter + confirm, 1 automaton only             */
    bool                   masked;       /* every pattern of this launch is shorter than 4   */
    bool                   nontemporal;
    /* flat kernel only: every payload has length uniform_len, payload k starts at arena + k * uniform_stride */
    uint32_t               uniform_stride;
    uint32_t               uniform_len;
    uint32_t               pkts_per_wave;
    /* packed kernel only */
    const unsigned long long *bitmap;   /* one bit per 16-byte slot of the arena: a payload starts here */
    const void            *plan;         /* kmp_plan_entry[waves + 1]; fused pass: [units + 1]          */
    /* fused pass only: the arena in `fused_blocks` regions of `units_per_block` work units each (kmp_plan_shape) */
    uint32_t               fused_blocks, units_per_block, n_units;   /* units_per_block: per REGION, which fused_sides (1 or 2) blocks share */
    uint32_t               fused_sides;
    bool                   fused_classed; /* the group of this launch is a classed one (kmp_device.h): cshift is a shift; a plain group: its short patterns */
    uint32_t              *fused_pool;   /* [regions] next unit of the region's pool, all 0 before the launch */
    uint64_t               span_end;     /* end of the last slot                                        */
    bool                   pad_clean;    /* every byte between a payload's end and the next slot is 0x00 */
    /* match-offset emission (streaming kernels only): kmpgpu_match[emit_cap], running counter */
    void                  *emit_out;
    unsigned long long    *emit_counter;
    unsigned long long     emit_cap;
};

hipError_t kmp_launch_scan(const kmp_scan_args &a, hipStream_t st);
hipError_t kmp_launch_scan_flat(const kmp_scan_args &a, hipStream_t st);
hipError_t kmp_launch_scan_packed(const kmp_scan_args &a, hipStream_t st);
hipError_t kmp_launch_build_bitmap(const uint64_t *pkt_off, uint64_t n, unsigned long long *bitmap, hipStream_t st);
/* Where the ranges of a plan are cut (before the closest packet start is taken).  units == 0: entry w at w * step bytes from the first
 * packet.  Otherwise the arena goes in regions of `region` bytes, one per block of the fused pass, and every region in `units` work
 * units: `big_units` of `step` bytes, then (the last part of a region, taken when its block is about to run out of work) units of `small`. */
struct kmp_plan_shape { uint64_t step; uint64_t region; uint32_t units, big_units, small; };
hipError_t kmp_launch_plan(const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n, uint64_t nwaves, const kmp_plan_shape &shape,
                           void *plan, hipStream_t st);
hipError_t kmp_launch_reduce(const unsigned long long *partials, uint32_t blocks_x, const uint32_t *pat_ids,
                             uint32_t n_ids, unsigned long long *counts, hipStream_t st, const uint32_t *rows = nullptr,
                             int accumulate = 0, bool counts_zeroed = false);
#define KMP_REDUCE_SLICE 4096u            /* partials one block of kmp_reduce_kernel adds up */
/* more than 16384 partials per pattern are summed by several blocks that ADD to counts[]: the scan kernel zeroes it first (zero_counts) */
static inline bool kmp_reduce_is_sliced(uint32_t blocks_x) { return blocks_x > 16384u; }
hipError_t kmp_launch_scan_multi(const kmp_scan_args &a, const uint32_t *tables, uint32_t table_words, uint32_t n_unique, uint32_t cshift, uint32_t bucket_mask, uint32_t n_ones, uint32_t ones,
                                 const uint32_t *uid_first, const uint32_t *uid_ids, hipStream_t st);
size_t kmp_multi_lds_bytes(uint32_t table_words, uint32_t n_unique, uint32_t waves);
int kmp_multi_kind(bool emit, bool pad_clean, uint32_t n_ones);
uint32_t kmp_multi_block_waves(int kind);
uint32_t kmp_multi_resident_waves(int kind, uint32_t table_words, uint32_t n_unique);
size_t kmp_extract_ws_bytes(uint64_t n_frames);
hipError_t kmp_launch_extract_phase1(const uint8_t *file, const uint64_t *frame_off, const uint32_t *caplen, uint64_t n, int tcp,
                                     uint8_t *ws, unsigned long long *totals, hipStream_t st);
hipError_t kmp_launch_extract_phase2(const uint8_t *file, const uint64_t *frame_off, uint64_t n, uint8_t *ws, uint64_t n_pkts,
                                     uint8_t *arena, uint64_t *pkt_off, uint32_t *pkt_len, uint64_t *src_off, hipStream_t st);
hipError_t kmp_launch_repack_phase1(const uint32_t *pkt_len, uint64_t n, uint8_t *ws, unsigned long long *totals, hipStream_t st);
hipError_t kmp_launch_repack_phase2(const uint8_t *old_arena, const uint64_t *old_off, const uint32_t *pkt_len, uint64_t n, uint8_t *ws,
                                    uint8_t *new_arena, uint64_t *new_off, hipStream_t st);
hipError_t kmp_launch_check_padding(uint8_t *arena, const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n, int fix,
                                    uint32_t *dirty, hipStream_t st);
hipError_t kmp_launch_effective_bytes(const uint8_t *arena, const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n,
                                      unsigned long long *out, hipStream_t st);
hipError_t kmp_launch_validate(const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n, uint64_t arena_bytes,
                               uint32_t *err, unsigned long long *payload_bytes, hipStream_t st);
hipError_t kmp_launch_synth_fill(uint8_t *arena, const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t first_pkt_id,
                                 uint64_t n, const kmp_synth_params &sp, hipStream_t st);
hipError_t kmp_launch_add_counts(unsigned long long *dst, const unsigned long long *src, uint32_t n, hipStream_t st);
hipError_t kmp_launch_fixed_index(uint64_t *pkt_off, uint32_t *pkt_len, uint64_t n, uint32_t len, uint64_t stride,
                                  hipStream_t st);

#endif
