/*
 * ref_dump_wrapper.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Translation unit that pulls the reference's header-only payload extractor
 * (/root/reference/packet_dumping.h:87-188, dump_UDP_packet / dump_TCP_packet) into
 * oracle/_ref/libkmpref.so.  The header is found through -I/root/reference (oracle/Makefile);
 * nothing of it is copied here.  The includes below are the ones the reference programs place
 * in front of it (serial.c:6-13) minus <pcap.h>, which the extractor does not use.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <netinet/ip.h>
#include <netinet/if_ether.h>
#include "packet_dumping.h"

/*
 * Driver for the CPU baseline of bench.py ("kind": "reference"): the reference's OWN kmp_prefix / kmp_matcher
 * object code (serial.c:190-238, compiled from the file where it lies, see oracle/Makefile) called once per
 * payload, the calls spread over OpenMP threads the way openmp_data.c:157-175 spreads them (guided schedule,
 * private counter, one merge).  Only this loop is ours.  kmp_matcher finds the text's end with strlen()
 * (serial.c:191), so every payload must be followed by a 0x00 inside its slot: true for the benchmark arena
 * (1500-byte payloads in 1504-byte slots, zero padded); the caller checks it.
 */
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif
int  kmp_matcher(char text[], char pattern[], int *prefix_array);
int *kmp_prefix(char pattern[]);

__attribute__((visibility("default")))
long long kmpref_count_arena(const char *arena, const uint64_t *off, uint64_t n, const char *pattern, int threads)
{
    int *prefix = kmp_prefix((char *)pattern);                       /* serial.c:148-152 */
    long long total = 0;
    if (threads < 1) threads = 1;
    (void)threads;
#pragma omp parallel num_threads(threads)
    {
        long long mine = 0;                                          /* openmp_data.c:152: private counter */
#pragma omp for schedule(guided) nowait
        for (int64_t k = 0; k < (int64_t)n; k++)
            mine += kmp_matcher((char *)arena + off[k], (char *)pattern, prefix);      /* openmp_data.c:164 */
#pragma omp atomic
        total += mine;                                               /* openmp_data.c:172-175 */
    }
    free(prefix);
    return total;
}

