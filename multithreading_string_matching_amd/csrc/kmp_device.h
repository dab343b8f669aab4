/*
 * kmp_device.h -- layouts shared by the HIP kernels (kmp_kernels.hip) and the C-ABI host code
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

#endif
