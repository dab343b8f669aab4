/*
 * kmphost.h -- host side of the MI355X KMP packet-payload matcher (plain C, no GPU dependency).
 *
 * These are the pieces of the reference's programs that sit AROUND the hot loop and decide which
 * bytes reach it.  Each entry cites the reference interface it replaces (paths relative to the
 * reference repository).  Library: multithreading_string_matching_amd/lib/libkmphost.so.
 *
 * Conventions: every function returns 0 on success and a negative KMPHOST_E* code on failure;
 * nothing calls exit(); buffers handed in are borrowed, buffers handed out are owned by the
 * struct that carries them and released by its *_free().
 */
#ifndef KMPHOST_H
#define KMPHOST_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "kmp_synth.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KMPHOST_OK            0
#define KMPHOST_EIO          -1   /* file cannot be opened / read                       */
#define KMPHOST_EFORMAT      -2   /* not a classic pcap savefile                        */
#define KMPHOST_ENOMEM       -3
#define KMPHOST_EINVAL       -4
#define KMPHOST_ETOKEN       -5   /* pattern token longer than KMP_MAX_PATTERN_LEN      */

#define KMP_MAX_PATTERN_LEN  99   /* serial.c:64  char str[100]                         */
#define KMP_SLOT_ALIGN       16   /* every payload starts on a 16-byte boundary         */
#define KMP_ARENA_SLACK      64   /* readable bytes after the last slot                 */
#define KMP_PROTO_UDP         0   /* serial.c:16                                        */
#define KMP_PROTO_TCP         1   /* serial.c:17                                        */

/* ---- pcap savefile reader ---------------------------------------------------------------
 * Replaces the three libpcap calls the reference makes: pcap_open_offline (serial.c:91,
 * openmp_data.c:94), pcap_next_ex (serial.c:115, openmp_data.c:107), pcap_close.  Classic pcap
 * (magic a1b2c3d4 / a1b23c4d, either byte order) and pcapng (enhanced, simple and obsolete packet
 * blocks, either byte order, several sections) -- the two formats libpcap's pcap_open_offline reads;
 * libpcap itself is not in this image. */
typedef struct kmp_pcap kmp_pcap;
#define KMP_PCAP_ERRBUF 256                       /* PCAP_ERRBUF_SIZE analogue, serial.c:26 */
kmp_pcap *kmp_pcap_open(const char *path, char errbuf[KMP_PCAP_ERRBUF]);
/* 1 = packet returned (data valid until the next call), -2 = end of file, -1 = truncated record;
 * the reference's loop ends on either negative value (serial.c:115). */
int  kmp_pcap_next(kmp_pcap *p, uint32_t *caplen, uint32_t *len, const uint8_t **data);
uint32_t kmp_pcap_linktype(const kmp_pcap *p);
void kmp_pcap_close(kmp_pcap *p);

/* ---- payload extraction -------------------------------------------------------------------
 * Replace dump_UDP_packet (packet_dumping.h:87-139) and dump_TCP_packet (packet_dumping.h:150-188):
 * return 1 and the payload's offset/length inside the frame, or 0 where the reference returns
 * NULL.  Quirks kept: no EtherType / IP-version test, IHL from the low nibble of frame byte 14,
 * UDP header skipped as 8 bytes, payload = all remaining captured bytes.  The
 * TCP variant additionally rejects frames on which the reference's unsigned length would wrap. */
int kmp_extract_udp(const uint8_t *frame, uint32_t capture_len, uint32_t *payload_off, uint32_t *payload_len);
int kmp_extract_tcp(const uint8_t *frame, uint32_t capture_len, uint32_t *payload_off, uint32_t *payload_len);

/* ---- pattern list -------------------------------------------------------------------------
 * Replaces the fscanf("%s") loader of serial.c:54-87 / openmp_data.c:57-90: whitespace-separated
 * tokens in file order, duplicates kept. */
typedef struct kmp_patterns {
    uint32_t  n;          /* number of tokens                              */
    uint8_t  *blob;       /* all tokens back to back, each NUL-terminated  */
    uint32_t *off;        /* off[i]: start of token i in blob              */
    uint32_t *len;        /* len[i]: strlen of token i (1..99)             */
} kmp_patterns;
int  kmp_patterns_load(const char *path, kmp_patterns *out);                 /* KMPHOST_EIO: errno is set, as for fopen (serial.c:59-63) */
int  kmp_patterns_parse(const uint8_t *text, size_t n, kmp_patterns *out);
void kmp_patterns_free(kmp_patterns *p);

/* KMP failure function: replaces kmp_prefix (serial.c:217-238); prefix has room for m ints. */
void kmp_failure_table(const uint8_t *pat, uint32_t m, int32_t *prefix);

/* ---- payload arena -------------------------------------------------------------------------
 * Replaces char **array_of_payloads (serial.c:99,124-136; openmp_data.c:123,139-142): one
 * contiguous byte arena, payload k at bytes[off[k] .. off[k]+len[k]), off[k] % 16 == 0, the gap
 * up to the next slot zero-filled, KMP_ARENA_SLACK readable zero bytes after the last slot.
 * Memory comes from the allocator pair so the caller can make it pinned (kmpgpu_host_alloc). */
typedef void *(*kmp_alloc_fn)(size_t);
typedef void  (*kmp_free_fn)(void *);
typedef struct kmp_arena {
    uint8_t  *bytes;
    uint64_t  nbytes;         /* allocated size of bytes, slack included                */
    uint64_t *off;
    uint32_t *len;
    uint64_t  n_pkts;         /* payloads stored (invalid frames are skipped, serial.c:138-140) */
    uint64_t  payload_bytes;  /* sum of len[]                                            */
    uint64_t  n_frames;       /* records read from the savefile                          */
    kmp_free_fn free_fn;
} kmp_arena;
/* serial.c:115-141: read every record, extract, store.  capture length handed to the extractor is
 * caplen (openmp_data.c:116,131; equals serial.c's header->len whenever caplen == len). */
int  kmp_arena_from_pcap(const char *path, int proto, kmp_alloc_fn alloc_fn, kmp_free_fn free_fn,
                         kmp_arena *out, char errbuf[KMP_PCAP_ERRBUF]);
/* Build an arena from caller-supplied payloads (tests, synthetic inputs). */
int  kmp_arena_from_payloads(const uint8_t *const *payloads, const uint32_t *lens, uint64_t n,
                             kmp_alloc_fn alloc_fn, kmp_free_fn free_fn, kmp_arena *out);
/* Index for n fixed-length payloads / for given lengths; returns the arena size needed. */
uint64_t kmp_arena_layout(const uint32_t *lens, uint32_t fixed_len, uint64_t n, uint32_t slot_align,
                          uint64_t *off_out, uint32_t *len_out);
void kmp_arena_free(kmp_arena *a);

/* ---- raw frames for on-device extraction ------------------------------------------------------
 * The capture file as it is + where its frames lie: everything the host does when the payload
 * extraction (openmp_data.c:128-147) runs on the GPU (kmpgpu_load_frames).  Walking the record
 * headers is the only sequential part of a pcap file. */
typedef struct kmp_frames {
    uint8_t  *bytes;          /* the whole savefile                                  */
    uint64_t  nbytes;
    uint64_t *off;            /* off[f]: first byte of frame f inside bytes          */
    uint32_t *caplen;         /* caplen[f]: captured bytes of frame f                */
    uint64_t  n;
    kmp_free_fn free_fn;
    uint64_t  map_len;        /* != 0: bytes is the file mapping itself (no copy was made), released by munmap */
} kmp_frames;
/* alloc_fn == NULL: no copy at all -- bytes is the read-only mapping of the file (a host-to-device copy
 * straight from it runs at PCIe speed on the MI355X hosts, profiles/r01_h2d_probe.txt); with an allocator
 * the file is copied into its memory by several threads. */
int  kmp_frames_from_pcap(const char *path, kmp_alloc_fn alloc_fn, kmp_free_fn free_fn, kmp_frames *out,
                          char errbuf[KMP_PCAP_ERRBUF]);
void kmp_frames_free(kmp_frames *f);

/* ---- streamed capture: batches -----------------------------------------------------------------
 * Replaces the producer of openmp_task.c:126-155 (read up to N packets, extract, hand the batch to
 * a task).  The caller owns the (pinned) buffers; a batch ends when the next payload would not fit
 * in cap_bytes / cap_pkts.  The batch is a packed arena in the sense of kmpgpu.h. */
typedef struct kmp_batch_reader kmp_batch_reader;
kmp_batch_reader *kmp_batch_open(const char *path, int proto, char errbuf[KMP_PCAP_ERRBUF]);
/* Returns the number of payloads stored (0 = end of capture), or a negative KMPHOST_E* code
 * (KMPHOST_EINVAL: a single payload is larger than cap_bytes).  *used_bytes = arena bytes to upload
 * (slack included), *frames += records read. */
int64_t kmp_batch_next(kmp_batch_reader *r, uint8_t *arena, uint64_t cap_bytes, uint64_t *off, uint32_t *len,
                       uint64_t cap_pkts, uint64_t *used_bytes, uint64_t *frames);
/* The same producer with the extraction left to the GPU (kmpgpu_load_frames): the next records of the capture whose bytes
 * span at most max_span_bytes of the file (at least one record, at most cap_frames).  Only the record headers are read;
 * frame_off[f] / frame_caplen[f] locate the frames inside the buffer kmp_batch_file() returns (the mapped capture itself:
 * nothing is copied here).  Returns the number of frames (0 = end of capture).  A reader serves either kind of batch, not both. */
int64_t kmp_batch_next_frames(kmp_batch_reader *r, uint64_t max_span_bytes, uint64_t *frame_off, uint32_t *frame_caplen,
                              uint64_t cap_frames);
const uint8_t *kmp_batch_file(const kmp_batch_reader *r, uint64_t *nbytes);
/* memcpy by several threads (the CPUs this process may use, at most 16; KMPHOST_THREADS): stages a batch of raw frames from the mapped
 * capture into a pinned buffer -- one bulk copy per batch, no per-packet work. */
void kmp_copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t n);
void kmp_batch_close(kmp_batch_reader *r);

/* Synthetic fill on the host (same bytes as the device generator, kmp_synth.h). */
void kmp_synth_fill_host(uint8_t *arena, const uint64_t *off, const uint32_t *len, uint64_t first_pkt_id,
                         uint64_t n, const kmp_synth_params *sp, int threads);
/* Number of packets among ids [first, first+n) that carry the needle. */
uint64_t kmp_synth_count_planted(const uint32_t *len, uint32_t fixed_len, uint64_t first_pkt_id, uint64_t n,
                                 const kmp_synth_params *sp);

/* ---- report ---------------------------------------------------------------------------------
 * serial.c:163-169: header line, "%s: %d times!" for every pattern with a non-zero count in file
 * order, then "Elapsed time = %f seconds". */
void kmp_report(FILE *fp, const kmp_patterns *pats, const uint64_t *counts, double elapsed_seconds);

/* Write a classic little-endian pcap of Ethernet/IPv4/UDP frames around the arena's payloads
 * (test + benchmark tooling: proves arena-route == pcap-route). */
int kmp_write_udp_pcap(const char *path, const uint8_t *arena, const uint64_t *off, const uint32_t *len, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* KMPHOST_H */
