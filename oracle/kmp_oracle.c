/*
 * kmp_oracle.c -- CPU restatement of the reference's KMP packet-payload match-count path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there only
 * as the checker / the CPU baseline, never as the thing measured or shipped.  The product path
 * (multithreading_string_matching_amd/csrc) has its own host code and fails loudly without the
 * HIP library.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.  The
 * semantic is SURVEY.md Appendix A: the scanned text of a payload of length L is
 * payload[0 : E), E = min(L, index of first 0x00) -- the reference calls strlen() on an
 * unterminated heap buffer (serial.c:191); where that reads past L the reference is undefined
 * and this file follows the defined truncation.
 *
 * Parity pins: tests/test_oracle_*.py check this file against (1) SURVEY.md App. B count
 * vectors produced by the reference's compiled serial.c, via the JSON files in tests/golden, and (2) when
 * oracle/_ref/libkmpref.so is present, the reference's own kmp_matcher/kmp_prefix/dump_*_packet
 * object code compiled from /root/reference (oracle/Makefile).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <time.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ---- serial.c:217-238  kmp_prefix ------------------------------------------------------- */
/* prefix[i] = length of the longest proper prefix of pat[0..i] that is also its suffix. */
ORACLE_API void oracle_kmp_prefix(const uint8_t *pat, uint32_t m, int32_t *prefix)
{
    if (m == 0) return;                 /* serial.c:221 would write out of bounds; unreachable via fscanf("%s") */
    prefix[0] = 0;                      /* serial.c:221 */
    uint32_t i = 1, j = 0;
    while (i < m) {                     /* serial.c:223 */
        if (pat[i] == pat[j]) {         /* serial.c:224-227 */
            prefix[i] = (int32_t)(j + 1);
            j++; i++;
        } else if (j != 0) {            /* serial.c:229-230 */
            j = (uint32_t)prefix[j - 1];
        } else {                        /* serial.c:232-234 */
            prefix[i] = 0;
            i++;
        }
    }
}

/* ---- serial.c:191  strlen(text), bounded by the payload length (E_k) ------ */
ORACLE_API uint32_t oracle_text_len(const uint8_t *text, uint32_t len)
{
    const uint8_t *z = (const uint8_t *)memchr(text, 0, len);
    return z ? (uint32_t)(z - text) : len;
}

/* ---- serial.c:190-215  kmp_matcher ------------------------------------------------------- */
/* Same control flow as the reference, line for line in meaning: two passes (strlen, then the
 * automaton), overlapping matches counted because j falls back to prefix[j-1] after a hit. */
ORACLE_API int32_t oracle_kmp_matcher(const uint8_t *text, uint32_t len,
                                      const uint8_t *pat, uint32_t m, const int32_t *prefix)
{
    int32_t text_len = (int32_t)oracle_text_len(text, len);   /* serial.c:191 */
    int32_t pattern_len = (int32_t)m;                          /* serial.c:192 */
    if (text_len < pattern_len) return 0;                      /* serial.c:193-194 */
    if (pattern_len == 0) return 0;                            /* not reachable in the reference */
    int32_t i = 0, j = 0, occurrences = 0;                     /* serial.c:195-197 */
    while (i < text_len) {                                     /* serial.c:198 */
        if (pat[j] == text[i]) { j++; i++; }                   /* serial.c:199-202 */
        if (j == pattern_len) {                                /* serial.c:203-206 */
            occurrences++;
            j = prefix[j - 1];
        } else if (i < text_len && pat[j] != text[i]) {        /* serial.c:207-212 */
            if (j != 0) j = prefix[j - 1];
            else i++;
        }
    }
    return occurrences;                                        /* serial.c:214 */
}

/* Independent definition (second form): number of start offsets s with
 * s + m <= E and text[s:s+m] == pat.  Used by the tests to cross-check the automaton. */
ORACLE_API int32_t oracle_naive_count(const uint8_t *text, uint32_t len, const uint8_t *pat, uint32_t m)
{
    uint32_t E = oracle_text_len(text, len);
    if (m == 0 || E < m) return 0;
    int32_t c = 0;
    for (uint32_t s = 0; s + m <= E; s++)
        if (memcmp(text + s, pat, m) == 0) c++;
    return c;
}

/* ---- serial.c:148-155  hot loop over an arena ------------------------------------------- */
/* Payload k is arena[pkt_off[k] : pkt_off[k] + pkt_len[k]); pattern i is
 * pat_blob[pat_off[i] : pat_off[i] + pat_len[i]).  counts[i] accumulates like string_count[i]
 * (serial.c:101,155), widened to 64 bit. */
ORACLE_API void oracle_count_serial(const uint8_t *arena, const uint64_t *pkt_off, const uint32_t *pkt_len,
                                    uint64_t n_pkts, const uint8_t *pat_blob, const uint32_t *pat_off,
                                    const uint32_t *pat_len, uint32_t n_pat, uint64_t *counts)
{
    int32_t **prefix = (int32_t **)malloc(sizeof(int32_t *) * (n_pat ? n_pat : 1));
    for (uint32_t i = 0; i < n_pat; i++) {                     /* serial.c:150-152 */
        prefix[i] = (int32_t *)malloc(sizeof(int32_t) * (pat_len[i] ? pat_len[i] : 1));
        oracle_kmp_prefix(pat_blob + pat_off[i], pat_len[i], prefix[i]);
    }
    for (uint32_t i = 0; i < n_pat; i++) counts[i] = 0;        /* serial.c:101 calloc */
    for (uint64_t k = 0; k < n_pkts; k++)                      /* serial.c:153 */
        for (uint32_t i = 0; i < n_pat; i++)                   /* serial.c:154 */
            counts[i] += (uint64_t)oracle_kmp_matcher(arena + pkt_off[k], pkt_len[k],
                                                      pat_blob + pat_off[i], pat_len[i], prefix[i]);  /* :155 */
    for (uint32_t i = 0; i < n_pat; i++) free(prefix[i]);
    free(prefix);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- openmp_data.c:126-178  the CPU baseline: both timed phases --------------------------- */
/* Phase 1 (openmp_data.c:128-147): parallel-for schedule(guided), one malloc+memcpy per
 * payload into its own heap buffer (the reference keeps no lengths; we keep a one-byte 0x00
 * guard after each copy so that strlen() inside the matcher is defined -- App. A).
 * Phase 2 (openmp_data.c:157-175): parallel region, per-thread calloc'd counters, for
 * schedule(guided) collapse(2) over (payload, pattern), omp-atomic merge.
 * Returns the elapsed seconds of the reference's bracket (openmp_data.c:126 .. :178). */
ORACLE_API double oracle_count_openmp(const uint8_t *arena, const uint64_t *pkt_off, const uint32_t *pkt_len,
                                      uint64_t n_pkts, const uint8_t *pat_blob, const uint32_t *pat_off,
                                      const uint32_t *pat_len, uint32_t n_pat, int threads, uint64_t *counts)
{
    if (threads < 1) threads = 1;
    int64_t N = (int64_t)n_pkts;
    int64_t P = (int64_t)n_pat;
    uint8_t **payloads = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)(N ? N : 1));
    for (uint32_t i = 0; i < n_pat; i++) counts[i] = 0;

    double start = now_s();                                    /* openmp_data.c:126 */

#pragma omp parallel for num_threads(threads) schedule(guided)
    for (int64_t k = 0; k < N; k++) {                          /* openmp_data.c:128-147 */
        uint32_t L = pkt_len[k];
        payloads[k] = (uint8_t *)malloc((size_t)L + 1);        /* :140 (+1: defined strlen) */
        memcpy(payloads[k], arena + pkt_off[k], L);            /* :141 */
        payloads[k][L] = 0;
    }

    int32_t **prefix = (int32_t **)malloc(sizeof(int32_t *) * (size_t)(P ? P : 1));
    for (int64_t i = 0; i < P; i++) {                          /* openmp_data.c:153-155 (master thread) */
        prefix[i] = (int32_t *)malloc(sizeof(int32_t) * (pat_len[i] ? pat_len[i] : 1));
        oracle_kmp_prefix(pat_blob + pat_off[i], pat_len[i], prefix[i]);
    }

#pragma omp parallel num_threads(threads)
    {
        uint64_t *priv = (uint64_t *)calloc((size_t)(P ? P : 1), sizeof(uint64_t));   /* :159 */
#pragma omp for schedule(guided) collapse(2)
        for (int64_t k = 0; k < N; k++)                        /* openmp_data.c:161-164 */
            for (int64_t i = 0; i < P; i++)
                priv[i] += (uint64_t)oracle_kmp_matcher(payloads[k], pkt_len[k],
                                                        pat_blob + pat_off[i], pat_len[i], prefix[i]);
        for (int64_t i = 0; i < P; i++) {                      /* openmp_data.c:169-173 */
#pragma omp atomic
            counts[i] += priv[i];
        }
        free(priv);
    }

    double finish = now_s();                                   /* openmp_data.c:178 */

    for (int64_t i = 0; i < P; i++) free(prefix[i]);
    free(prefix);
    for (int64_t k = 0; k < N; k++) free(payloads[k]);
    free(payloads);
    return finish - start;
}

ORACLE_API int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- packet_dumping.h:87-139  dump_UDP_packet -------------------------------------------- */
/* Returns 1 and the payload's offset/length inside the frame, or 0 where the reference
 * returns NULL.  capture_len is what the caller passes (header->len in serial.c:120,
 * caplen in openmp_data.c:131).  No EtherType or IP-version check, as in the reference. */
ORACLE_API int oracle_dump_udp(const uint8_t *frame, uint32_t capture_len, uint32_t *payload_off, uint32_t *payload_len)
{
    uint32_t pos = 0;
    if (capture_len < 14) return 0;             /* :94-97  sizeof(struct ether_header) */
    pos += 14; capture_len -= 14;               /* :99-100 */
    if (capture_len < 20) return 0;             /* :102-105 sizeof(struct ip) */
    uint32_t ihl = (uint32_t)(frame[pos] & 0x0F) * 4;   /* :107-108 ip_hl (low nibble, little-endian bitfield) */
    if (capture_len < ihl) return 0;            /* :110-113 */
    if (frame[pos + 9] != 17) return 0;         /* :116-119 IPPROTO_UDP */
    pos += ihl; capture_len -= ihl;             /* :122-123 */
    if (capture_len < 8) return 0;              /* :125-128 sizeof(struct UDP_hdr) */
    pos += 8; capture_len -= 8;                 /* :133-134 sizeof(udp_h) = sizeof(pointer) = 8 on LP64 */
    *payload_off = pos;
    *payload_len = capture_len;                 /* :136 */
    return 1;
}

/* ---- packet_dumping.h:150-188  dump_TCP_packet ------------------------------------------- */
/* The reference does no bounds or protocol checks and its unsigned capture_len can wrap on
 * short frames (then serial.c:125 mallocs ~4 GiB and memcpy crashes).  Defined subset restated
 * here: accept iff IHL*4 >= 20 and data-offset*4 >= 20 (as the reference) AND the headers fit
 * inside the captured bytes; frames on which the reference would wrap are rejected (0). */
ORACLE_API int oracle_dump_tcp(const uint8_t *frame, uint32_t capture_len, uint32_t *payload_off, uint32_t *payload_len)
{
    if (capture_len < 14 + 1) return 0;                     /* need the IHL byte; reference: UB */
    uint32_t pos = 14;                                      /* :161 SIZE_ETHERNET */
    uint32_t size_ip = (uint32_t)(frame[pos] & 0x0F) * 4;   /* :165 */
    if (size_ip < 20) return 0;                             /* :166-169 */
    pos += size_ip;                                         /* :171 */
    if (capture_len < pos + 13) return 0;                   /* need th_offx2; reference: UB */
    uint32_t size_tcp = (uint32_t)((frame[pos + 12] & 0xF0) >> 4) * 4;   /* :175 */
    if (size_tcp < 20) return 0;                            /* :176-179 */
    pos += size_tcp;                                        /* :181 */
    if (capture_len < pos) return 0;                        /* reference: unsigned wrap at :182 */
    *payload_off = pos;
    *payload_len = capture_len - pos;                       /* :182-184 */
    return 1;
}
