/* Built with -fsanitize=address,undefined by tests/test_host.py: drives the host C library
 * (pcap reader, extractors, pattern loader, arena builder, batch reader, frame index, synthetic fill)
 * over the fixtures so that heap overflows / UB in the product's host code fail the CPU suite. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "kmphost.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    CHECK(argc == 4);
    const char *pcap = argv[1], *strings = argv[2];
    const unsigned long want_payloads = strtoul(argv[3], NULL, 10);
    char err[KMP_PCAP_ERRBUF];

    kmp_patterns pats;
    CHECK(kmp_patterns_load(strings, &pats) == 0 && pats.n > 0);
    for (uint32_t i = 0; i < pats.n; i++) {
        int32_t fail[KMP_MAX_PATTERN_LEN];
        kmp_failure_table(pats.blob + pats.off[i], pats.len[i], fail);
        CHECK(fail[0] == 0);
    }

    for (int proto = 0; proto < 2; proto++) {
        kmp_arena a;
        CHECK(kmp_arena_from_pcap(pcap, proto, NULL, NULL, &a, err) == 0);
        if (proto == KMP_PROTO_UDP) CHECK(a.n_pkts == want_payloads);
        uint64_t sum = 0;
        for (uint64_t k = 0; k < a.n_pkts; k++) { sum += a.len[k]; CHECK(a.off[k] % 16 == 0 && a.off[k] + a.len[k] <= a.nbytes); }
        CHECK(sum == a.payload_bytes);

        /* the batches of the streaming reader hold the same payloads, for awkward capacities too */
        const uint64_t caps[3] = {1 << 20, 4096, 700};
        for (int c = 0; c < 3; c++) {
            kmp_batch_reader *r = kmp_batch_open(pcap, proto, err);
            CHECK(r != NULL);
            uint8_t *buf = (uint8_t *)malloc(caps[c]);
            uint64_t off[8]; uint32_t len[8];
            uint64_t used = 0, frames = 0, k = 0;
            for (;;) {
                int64_t n = kmp_batch_next(r, buf, caps[c], off, len, 8, &used, &frames);
                CHECK(n >= 0);
                if (n == 0) break;
                for (int64_t i = 0; i < n; i++, k++) {
                    CHECK(k < a.n_pkts && len[i] == a.len[k]);
                    CHECK(memcmp(buf + off[i], a.bytes + a.off[k], len[i]) == 0);
                }
            }
            CHECK(k == a.n_pkts && frames == a.n_frames);
            free(buf);
            kmp_batch_close(r);
        }
        kmp_arena_free(&a);
    }

    kmp_frames fr;
    CHECK(kmp_frames_from_pcap(pcap, NULL, NULL, &fr, err) == 0);
    uint64_t acc = 0;
    for (uint64_t f = 0; f < fr.n; f++) {
        uint32_t po, pl;
        CHECK(fr.off[f] + fr.caplen[f] <= fr.nbytes);
        if (kmp_extract_udp(fr.bytes + fr.off[f], fr.caplen[f], &po, &pl)) acc++;
        (void)kmp_extract_tcp(fr.bytes + fr.off[f], fr.caplen[f], &po, &pl);
    }
    CHECK(acc == want_payloads);
    /* every truncation of the first frames: the extractors must never read past capture_len */
    for (uint64_t f = 0; f < fr.n && f < 50; f++)
        for (uint32_t cl = 0; cl <= fr.caplen[f] && cl < 80; cl++) {
            uint8_t *copy = (uint8_t *)malloc(cl ? cl : 1);
            uint32_t po, pl;
            memcpy(copy, fr.bytes + fr.off[f], cl);
            if (kmp_extract_udp(copy, cl, &po, &pl)) CHECK(po + pl == cl);
            if (kmp_extract_tcp(copy, cl, &po, &pl)) CHECK(po + pl == cl);
            free(copy);
        }
    kmp_frames_free(&fr);

    kmp_synth_params sp;
    memset(&sp, 0, sizeof sp);
    sp.seed = 7; sp.lo = 'a'; sp.span = 26; sp.plant_permille = 500; sp.needle_len = 5; memcpy(sp.needle, "NEEDL", 5);
    uint32_t lens[6] = {0, 1, 7, 16, 100, 1500};
    uint64_t off[6]; uint32_t ln[6];
    const uint64_t nb = kmp_arena_layout(lens, 0, 6, 16, off, ln);
    uint8_t *arena = (uint8_t *)malloc(nb);
    kmp_synth_fill_host(arena, off, ln, 0, 6, &sp, 2);
    CHECK(kmp_synth_count_planted(lens, 0, 0, 6, &sp) <= 6);
    free(arena);

    kmp_patterns_free(&pats);
    printf("sanitizer driver ok\n");
    return 0;
}
