/*
 * kmp_device.h -- layouts shared by the HIP kernels (kmp_scan_*.hip / kmp_prep.hip) and the C-ABI host code
 * (kmpgpu.hip).  gfx950 only.
 */
#ifndef KMP_DEVICE_H
#define KMP_DEVICE_H

#include <stdint.h>
#include "kmp_synth.h"

#define KMP_WAVE           64u
#define KMP_LANE_BYTES     16u                           /* one global_load_dwordx4 per lane     */
#define KMP_CHUNK          (KMP_WAVE * KMP_LANE_BYTES)   /* 1 KiB per wavefront load instruction */
#define KMP_BLOCK_THREADS  256u
#define KMP_BLOCK_WAVES    (KMP_BLOCK_THREADS / KMP_WAVE)
#define KMP_PAT_BYTES      112u                          /* 99 rounded up to 16                  */

/* One pattern as the kernels see it (240 B): replaces the reference's
 * (char *pattern, int *prefix_array) pair handed to kmp_matcher, serial.c:190. */
typedef struct __attribute__((aligned(16))) kmp_pattern_dev {
    uint8_t  pat[KMP_PAT_BYTES];    /* pattern bytes, zero padded                               */
    uint8_t  fail[KMP_PAT_BYTES];   /* KMP failure table (serial.c:217-238); values < m <= 99   */
    uint32_t m;                     /* pattern length, 1..99                                    */
    uint32_t first;                 /* first min(m,4) bytes, little-endian, zero padded         */
    uint32_t mask;                  /* 0xFF per byte of 'first' that is part of the pattern     */
    uint32_t reserved;
} kmp_pattern_dev;

/* Tables of the fused multi-pattern kernel (kmp_scan_multi_kernel), one blob of uint32 words built on
 * the host (kmpgpu_set_patterns) and copied into LDS by every block:
 *   [0, 2048)        two words per bucket KMP_MULTI_HASH(key): its first entry itself, then where its further entries start
 *                    (bits 0-15) | number of entries (bits 16-31) -- one 8-byte read decides a bucket of one pattern.
 *                    key = b0 | b1 << 8 | (b2 & 31) << 16 (bucket mask KMP_MULTI_KEYMASK): patterns that only share their
 *                    first two bytes ("de": depth, decode, defrag, detector, detail) land in different buckets, so a
 *                    walk is one entry long as a rule; a 2-byte pattern is entered under all 32 values of b2 & 31.  A
 *                    group with more than KMP_MULTI_MAX_TWO 2-byte patterns is bucketed by b0 | b1 << 8 alone (mask 0xFFFF).
 *   [2048, 2560)     entries: b0 | b1 << 8 | b2 << 16 | (unique-pattern id & 255) << 24; b2 = 0x00 for a 2-byte pattern
 *                    (v_msad_u8 skips a 0x00 reference byte: it matches whatever follows).  A group of up to 256 unique
 *                    patterns numbers them 0 .. 255, short ones (2 or 3 bytes) first: id < n_short means the entry decides
 *                    alone.  A CLASSED group holds up to 1024 (no 2-byte ones): the patterns of bucket class c = bucket >> 7
 *                    (eight classes, up to 256 patterns each) are numbered from the class's first id on, short ones first,
 *                    and the kernel adds that first id to the entry's eight bits (kmp_scan_multi_kernel's cshift: 7; a
 *                    plain group is one class, cshift 10).
 *   [2560, 4672)     filter over the first three text bytes, as a PAIR table: the bytes are taken by their low five bits
 *                    (codes c = b & 31), and entry KMP_MULTI_PAIR(c1, c2) = c1 + 33 * c2 (8 bytes, 1056 entries) holds
 *                      word 0: bit c0 set  <=>  some pattern starts c0 c1 c2      ("which byte may stand BEFORE c1 c2")
 *                      word 1: bit c3 set  <=>  some pattern starts c1 c2 c3      ("which byte may FOLLOW c1 c2")
 *                    so ONE ds_read_b64 at the two middle bytes of a 4-byte window decides the two start offsets at its
 *                    first and second byte: 8 LDS lookups per 16 text bytes instead of 16 (the lookups at random addresses
 *                    are what the LDS spends its cycles on: 32 lanes on 32 bank pairs, ~3.5 cycles per half wavefront).
 *                    A 2-byte pattern c0 c1 sets bit c0 of word 0 in all 32 entries (c1, *) and all of word 1 of entry
 *                    (c0, c1).  No hash: letters of one case never share a bit, text gives no false hit (round 2's first
 *                    filter, 16384 hashed slots: 42 % of its hits on lower-case text were collisions).  The factor 33
 *                    spreads the entries over the bank pairs by c1 + c2 (one letter alone covers 26 of 32).
 *   [4672, ...)      a plain group: one 4-word record per LONG unique pattern (4+ bytes; record = id - n_short): bytes 0-3, bytes 4-7,
 *                    the byte mask of bytes 4-7, m | pattern index << 8 (a pattern of nine bytes or more compares its
 *                    rest against kmp_pattern_dev[index].pat);
 *                    a classed group: [4672, 4680) one word per bucket class, KMP_MULTI_CLS_WORD: its short patterns | its
 *                    first record << 9 | its first id << 20, then one 2-word record per long pattern (record = first of its
 *                    class + the entry's eight bits - the class's short ones): byte 3 | m << 8 | pattern index << 16, bytes 4-7
 *                    (1024 records of 16 bytes would not leave room for two blocks per CU)
 * The first 4672 words live in static LDS (their offsets fold into the ds_read offset field). */
#define KMP_MULTI_BUCKETS     1024u
#define KMP_MULTI_BUCKET_W0   0u
#define KMP_MULTI_ENTRY_W0    2048u
#define KMP_MULTI_MAX_ENTRIES 512u
#define KMP_MULTI_MAX_ONES    4u       /* distinct 1-byte patterns the first group counts on the side */
#define KMP_MULTI_MAX_TWO     8u       /* 2-byte patterns a group may hold and still be bucketed by three bytes */
#define KMP_MULTI_FILTER_W0   2560u
#define KMP_MULTI_PAIR_ENTRIES 1056u
#define KMP_MULTI_REC_W0      (KMP_MULTI_FILTER_W0 + KMP_MULTI_PAIR_ENTRIES * 2u)
#define KMP_MULTI_CLS_WORDS   8u
#define KMP_MULTI_CLS_SHIFT   7u       /* class of a bucket in a classed group: bucket >> this */
#define KMP_MULTI_CLS_WORD(n_short, first_rec, first_id) ((uint32_t)(n_short) | ((uint32_t)(first_rec) << 9) | ((uint32_t)(first_id) << 20))
#define KMP_MULTI_REC_WORDS   4u
#define KMP_MULTI_CREC_WORDS  2u
#define KMP_MULTI_MAX_UNIQUE  256u       /* unique patterns of a plain group; a classed one: four times as many */
#define KMP_MULTI_MAX_UNITS   256u     /* work units a block's region is cut into at most (their entries sit in LDS, 16 bytes each) */
#define KMP_MULTI_MIN_LEN     2u
#define KMP_MULTI_SHORT_LEN   3u       /* patterns up to this length are decided by their bucket entry alone */
#define KMP_MULTI_MAX_LEN     99u
#define KMP_MULTI_KEYMASK     0x1FFFFFu                  /* b0, b1 and the low five bits of b2 */
#define KMP_MULTI_MUL         0x9E3779u
#define KMP_MULTI_PAIR(b1, b2) (((uint32_t)(b1) & 31u) + 33u * ((uint32_t)(b2) & 31u))
#define KMP_MULTI_BLOCK_WAVES   16u    /* the fused pass runs 1024-thread blocks: two of them share a CU (8 wavefronts per SIMD at 64 VGPRs, 2 x 60 KB of LDS) */
#define KMP_MULTI_BLOCK_THREADS (KMP_MULTI_BLOCK_WAVES * KMP_WAVE)
#define KMP_MULTI_WIDE_WAVES    12u    /* its variants that need up to ~80 VGPRs (1-byte patterns riding along, unclean padding): two blocks = 6 wavefronts per SIMD */
#define KMP_MULTI_HASH(key)   ((uint32_t)((uint32_t)(key) * KMP_MULTI_MUL) >> 22)                                           /* key already masked: v_mul_u32_u24 + v_lshrrev */

#endif
