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
 *   [0, 512)         1024 x uint16: first entry of the bucket hash(b0 | b1 << 8), 0xFFFF = empty
 *   [512, 1024)      entries, uint32 each: unique-pattern id (bits 0-7) | third pattern byte << 8 (0 when the
 *                    pattern has 2 bytes: nothing to pre-check) | 0x40000000 for a pattern of more than 20 bytes |
 *                    0x80000000 on the last entry of a bucket
 *   [1024, 3072)     64 Kbit filter over the first THREE text bytes: bit h = KMP_MULTI_BIT(b0 | b1 << 8 |
 *                    b2 << 16) (byte h >> 3, bit h & 7) is set for every pattern of 3+ bytes, and for all 256
 *                    values of b2 for a 2-byte pattern (so one lookup serves both).  The multiplicative
 *                    hash also spreads text, whose bytes share their high bits, over the LDS banks.
 *   [3072, ...)      one 12-word record per unique pattern: its first 20 bytes as 5 dwords, 5 byte masks, m, and for a
 *                    pattern of more than 20 bytes its index + 1 (the rest is compared against kmp_pattern_dev.pat)
 * The first 3072 words live in static LDS (their offsets fold into the ds_read offset field). */
#define KMP_MULTI_BUCKETS     1024u
#define KMP_MULTI_BUCKET_W0   0u
#define KMP_MULTI_ENTRY_W0    512u
#define KMP_MULTI_MAX_ENTRIES 512u
#define KMP_MULTI_FILTER_W0   1024u
#define KMP_MULTI_REC_W0      3072u
#define KMP_MULTI_REC_WORDS   12u
#define KMP_MULTI_MAX_UNIQUE  256u
#define KMP_MULTI_MIN_LEN     2u
#define KMP_MULTI_PREFIX      20u      /* bytes of a pattern held in its record */
#define KMP_MULTI_MAX_LEN     99u
#define KMP_MULTI_WIN_WORDS   (KMP_CHUNK / 4u + 28u)     /* per-wavefront LDS window: the chunk + 112 bytes of the next one */
#define KMP_MULTI_BIT(w24)    ((((uint32_t)(w24) & 0xFFFFFFu) * 0x9E3779u) >> 16)   /* v_mul_u32_u24, 16-bit hash */
#define KMP_MULTI_HASH(w16)   ((((uint32_t)(w16) * 0x9E3Bu) >> 6) & (KMP_MULTI_BUCKETS - 1u))

#endif
