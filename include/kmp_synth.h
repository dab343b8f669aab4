/*
 * kmp_synth.h -- counter-based synthetic payload generator, shared verbatim by the host C code
 * (csrc/host/kmphost.c) and the HIP fill kernel (csrc/kmp_prep.hip), so that any shard of the
 * benchmark arena can be produced on the host or on the device with identical bytes.
 *
 * Not in the reference (it ships pcap files only); this is SURVEY.md section 8(d) input "S1/S2":
 * payload bytes uniform over [lo, lo+span) (default 'a'..'z': NUL-free and disjoint from the
 * needle), a needle planted in a packet with probability plant_permille/1000 at a uniformly
 * random offset that never touches the packet's last byte, optional NUL sprinkling.
 *
 *   byte(k, i) for packet k, payload byte i < len:
 *      planted(k) && pos(k) <= i < pos(k)+needle_len  ->  needle[i - pos(k)]
 *      else                                              base byte from hash(seed, k, i / 4)
 */
#ifndef KMP_SYNTH_H
#define KMP_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define KMP_HD __host__ __device__ static inline
#else
#define KMP_HD static inline
#endif

#define KMP_SYNTH_MAX_NEEDLE 100

typedef struct kmp_synth_params {
    uint32_t seed;
    uint32_t lo;              /* lowest byte value of the text alphabet (default 'a')     */
    uint32_t span;            /* number of values in the alphabet (default 26)            */
    uint32_t plant_permille;  /* probability*1000 that a packet carries the needle        */
    uint32_t nul_ppm;         /* per-byte probability*1e6 of a 0x00 (0 = NUL-free)        */
    uint32_t needle_len;      /* 0 = nothing planted                                      */
    uint8_t  needle[KMP_SYNTH_MAX_NEEDLE];
} kmp_synth_params;

KMP_HD uint32_t kmp_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

KMP_HD uint32_t kmp_synth_pkt_key(uint32_t seed, uint64_t k)
{
    return kmp_mix32(((uint32_t)k * 0x9E3779B1U) ^ ((uint32_t)(k >> 32) * 0x85EBCA77U) ^ seed);
}

/* Is packet k planted, and where?  Returns 1 and *pos when planted. */
KMP_HD int kmp_synth_plant(const kmp_synth_params *sp, uint64_t k, uint32_t len, uint32_t *pos)
{
    if (sp->needle_len == 0 || len < sp->needle_len + 2) return 0;
    uint32_t hp = kmp_mix32(kmp_synth_pkt_key(sp->seed, k) ^ 0xA511E9B3U);
    if ((hp % 1000U) >= sp->plant_permille) return 0;
    *pos = kmp_mix32(hp ^ 0x9E3779B9U) % (len - sp->needle_len - 1U);
    return 1;
}

/* The four base bytes of dword w of packet k (little-endian: byte 4w is the low byte). */
KMP_HD uint32_t kmp_synth_word(const kmp_synth_params *sp, uint32_t pkt_key, uint32_t w)
{
    uint32_t h = kmp_mix32(pkt_key + w * 0xC2B2AE3DU + 0x27D4EB2FU);
    uint32_t out = 0;
    for (int b = 0; b < 4; b++) {
        uint32_t x = (h >> (8 * b)) & 0xFFU;
        uint32_t ch = sp->lo + ((x * sp->span) >> 8);
        if (sp->nul_ppm) {
            uint32_t g = kmp_mix32(h ^ (0x68E31DA4U + (uint32_t)b * 0x1B873593U));
            if ((g % 1000000U) < sp->nul_ppm) ch = 0;
        }
        out |= (ch & 0xFFU) << (8 * b);
    }
    return out;
}

/* Dword w of packet k's 16-byte-padded slot: base bytes, needle overlay, zero beyond len. */
KMP_HD uint32_t kmp_synth_slot_word(const kmp_synth_params *sp, uint32_t pkt_key, uint32_t w,
                                    uint32_t len, int planted, uint32_t pos)
{
    uint32_t i0 = w * 4U;
    if (i0 >= len) return 0;
    uint32_t v = kmp_synth_word(sp, pkt_key, w);
    if (planted && i0 + 4U > pos && i0 < pos + sp->needle_len) {
        for (uint32_t b = 0; b < 4; b++) {
            uint32_t i = i0 + b;
            if (i >= pos && i < pos + sp->needle_len)
                v = (v & ~(0xFFU << (8 * b))) | ((uint32_t)sp->needle[i - pos] << (8 * b));
        }
    }
    if (i0 + 4U > len) v &= 0xFFFFFFFFU >> (8 * (i0 + 4U - len));
    return v;
}

#endif /* KMP_SYNTH_H */
