/*
 * kmpgpu.h -- C-ABI of the MI355X (gfx950) KMP match-count hot path.
 *
 * The reference has no FFI: its hot path is the loop
 *     string_count[i] += kmp_matcher(array_of_payloads[k], array_of_strings[i], prefix_array[i]);
 * at serial.c:153-155 (= openmp_data.c:157-175, mpi_dumping.c:198-200) with kmp_matcher at
 * serial.c:190-215 and kmp_prefix at serial.c:217-238.  A maintainer replaces that loop by the
 * calls below (INTEGRATION.md shows the patch).  Library:
 * multithreading_string_matching_amd/lib/libkmpgpu.so (hipcc, --offload-arch=gfx950).
 *
 * Semantics (bit-exact with compiled serial.c): for payload k of length L_k,
 * E_k = min(L_k, index of its first 0x00); count[i] = sum over k of the number of start offsets
 * s with s + m_i <= E_k and payload_k[s : s+m_i] == pattern_i (overlapping starts all count).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 or a negative KMPGPU_E*
 * code and never exits; kmpgpu_last_error() gives the text (per thread).  One context drives one GPU;
 * several GPUs = several contexts, in one process or one process each, whose counters are summed by
 * RCCL (kmpgpu_comm_*, the MPI_Reduce of mpi_dumping.c:202).  A context is not thread-safe; different
 * contexts may be driven from different threads.  Host buffers passed in are borrowed for the call; device buffers created
 * by the context are owned by it.
 */
#ifndef KMPGPU_H
#define KMPGPU_H

#include <stddef.h>
#include <stdint.h>

#include "kmp_synth.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KMPGPU_OK         0
#define KMPGPU_EHIP      -1   /* a HIP runtime call failed (no device, out of memory ...)  */
#define KMPGPU_EINVAL    -2   /* bad argument / layout contract violated                    */
#define KMPGPU_ESTATE    -3   /* call order: patterns or arena not set                      */
#define KMPGPU_ENOMEM    -4

#define KMPGPU_MAX_PATTERN_LEN 99      /* serial.c:64 */
#define KMPGPU_SLOT_ALIGN      16

typedef struct kmpgpu_ctx kmpgpu_ctx;

/* Timing of the last kmpgpu_scan / kmpgpu_load_arena on this context (HIP events on the
 * context's stream). */
typedef struct kmpgpu_timing {
    double   h2d_ms;          /* arena + index upload (kmpgpu_load_arena)                  */
    double   kernel_ms;       /* scan kernel(s) + partial-count reduce                     */
    double   d2h_ms;          /* counts download                                           */
    uint32_t launches;        /* scan-kernel launches in the last scan                     */
    uint32_t grid_blocks;     /* blocks per launch (x dimension)                           */
    uint64_t h2d_bytes;       /* bytes the last kmpgpu_load_arena / kmpgpu_load_frames copied to the device */
} kmpgpu_timing;

/* One reported match (kmpgpu_scan_offsets).  Not in the reference, which prints counts only
 * (serial.c:163-166); BASELINE north_star asks for offsets. */
typedef struct kmpgpu_match {
    uint64_t packet;          /* payload index in the arena                                */
    uint32_t offset;          /* start offset inside the payload                           */
    uint32_t pattern;         /* pattern index (file order)                                */
} kmpgpu_match;

/* Option keys for kmpgpu_set_option. */
#define KMPGPU_OPT_MODE          1   /* 0 auto (filter + KMP verify), 1 KMP automaton only */
#define KMPGPU_OPT_BLOCKS_PER_CU 2   /* grid = CUs * this (1..256); 0 = auto (default): the streaming kernels
                                        take ~6-16 KiB per wavefront, however many blocks that is */
#define KMPGPU_OPT_DEPTH         3   /* chunk loads in flight per wavefront: 2..6, 8; 0 = auto */
#define KMPGPU_OPT_FUSED         4   /* 1 = fused multi-pattern pass: the patterns of 2..99
                                        bytes are counted in ONE read of a packed arena per 1024
                                        distinct patterns (256 where 2-byte patterns are among
                                        them; up to four 1-byte patterns ride along,
                                        further ones keep one read each); 0 = off; 2 = auto
                                        (default): fused from 2 such unique patterns on         */
#define KMPGPU_OPT_KERNEL        5   /* 0 auto: slots back to back -> packed streaming kernel, or
                                        the flat streaming kernel when every payload has the same
                                        length of 512 bytes or more (also taken for equal slots
                                        with gaps between them); otherwise the general kernel.
                                        1 always the general one-packet-per-wavefront kernel;
                                        2 the packed kernel whenever the slots are back to back;
                                        3 the flat kernel whenever the stride is uniform          */

#define KMPGPU_OPT_ACCUMULATE    6   /* 1 = every pass ADDS to the counts buffer instead of
                                        overwriting it (batches of a streamed capture,
                                        openmp_task.c:172-175); kmpgpu_counts_reset() zeroes it */

#define KMPGPU_OPT_REPACK        7   /* 1 (default) = an arena whose slots are not back to back is
                                        copied once, on the device, into a packed arena owned by
                                        the context (streaming kernels); 0 = scan it in place
                                        with the general kernel                                 */
#define KMPGPU_OPT_FUSED_UNIT    8   /* fused pass: bytes of a work unit of the pool at the end of a region (a multiple
                                        of 1024 up to 1 MiB; 0 = auto, 32 KiB; larger where a region has more than ~220 of
                                        them): a wavefront that is through its own share of the region takes the pool's
                                        units one after the other */
#define KMPGPU_OPT_NONTEMPORAL 100   /* 1 (default) = arena loads carry the non-temporal hint (every
                                        byte is read once per pass; measured +10 % on MI355X), 0 =
                                        default cache policy                                   */

const char *kmpgpu_last_error(void);
int  kmpgpu_device_count(void);                        /* >= 0, or KMPGPU_EHIP                */

/* Create / destroy the context for one device.  Replaces nothing in the reference (its state
 * lives in main()'s locals, serial.c:99-101,148). */
int  kmpgpu_init(kmpgpu_ctx **ctx, int device);
void kmpgpu_destroy(kmpgpu_ctx *ctx);

/* Launch on a caller-owned HIP stream (hipStream_t, e.g. an explicit torch stream); NULL returns
 * to the context's own (non-blocking) stream.  The legacy default stream has the handle NULL and
 * therefore cannot be borrowed: create an explicit stream when work must be ordered with the
 * caller's (bench.py does). */
int  kmpgpu_set_stream(kmpgpu_ctx *ctx, void *hip_stream);
int  kmpgpu_set_option(kmpgpu_ctx *ctx, int key, int64_t value);

/* Pinned host memory for the arena (north_star: "pinned contiguous arena"). */
void *kmpgpu_host_alloc(size_t bytes);
void  kmpgpu_host_free(void *p);
/* Pin memory the caller already has -- e.g. (a window of) a capture file's read-only mapping -- so that uploads from it
 * (kmpgpu_load_frames, kmpgpu_load_arena) run as asynchronous DMA at PCIe speed instead of being staged through the runtime's
 * bounce buffers.  ptr page-aligned.  Returns KMPGPU_EHIP where the runtime refuses (then the memory simply stays pageable). */
int   kmpgpu_host_register(const void *ptr, size_t bytes);
int   kmpgpu_host_unregister(const void *ptr);

/* Replaces array_of_strings + prefix_array construction, serial.c:148-152 (kmp_prefix for every
 * pattern): copies the patterns, builds the failure tables on the host, uploads both.
 * 1 <= pat_len[i] <= 99, no 0x00 inside a pattern (fscanf("%s") + strlen cannot produce one). */
int  kmpgpu_set_patterns(kmpgpu_ctx *ctx, const uint8_t *const *pat, const uint32_t *pat_len, uint32_t n_pat);

/* Replaces array_of_payloads, serial.c:99,124-136: upload a host arena + index (H2D copy,
 * device copy owned by the context).  Contract: pkt_off[k] % 16 == 0 and
 * pkt_off[k] + max(16, round_up(pkt_len[k], 16)) <= arena_bytes for every k (checked): every
 * payload, also an empty one, owns at least one readable 16-byte slot.  The bytes between a payload's
 * end and the end of its slot may hold anything; when they are 0x00 (as kmp_arena builds them -- checked
 * at load time, and cleared in the copy kmpgpu_load_arena makes) the streaming kernels take a payload's
 * end from the packet-start bitmap and never read the index on the scan path.
 * BEHIND THE LAST SLOT nothing is required: arena_bytes need only reach the end of the last slot
 * (pkt_off[k] + max(16, round_up(pkt_len[k], 16)) of the payload that lies last), no kernel reads a byte at or
 * behind that end, and whatever a caller keeps there -- the rest of a larger buffer of which the index is a
 * prefix view, a longer batch loaded earlier into the same device buffers -- never enters a count: a window
 * never leaves its payload (serial.c:193,198).  (The host library still leaves KMP_ARENA_SLACK zero bytes
 * there; the kernels do not depend on them.) */
int  kmpgpu_load_arena(kmpgpu_ctx *ctx, const uint8_t *arena, uint64_t arena_bytes,
                       const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n_pkts);

/* Replaces the read loop AND the extraction phase (serial.c:115-141; openmp_data.c:128-147): upload the
 * raw capture file + the position of its frames (kmp_frames_from_pcap) and let the GPU apply the
 * dump_UDP_packet / dump_TCP_packet rules (packet_dumping.h:87-188) and pack the accepted payloads into
 * the context's arena.  tcp: 0 = UDP rule, 1 = TCP rule.  *n_payloads = payloads accepted. */
int  kmpgpu_load_frames(kmpgpu_ctx *ctx, const uint8_t *file_bytes, uint64_t file_nbytes, const uint64_t *frame_off,
                        const uint32_t *frame_caplen, uint64_t n_frames, int tcp, uint64_t *n_payloads);

/* The same in two steps, for a caller that keeps the copy engine busy (bin/openmp_task): _begin waits for the context's earlier passes,
 * then ENQUEUES the upload and the first extraction stage on the context's stream and returns; _finish waits for them, packs the
 * payloads and makes the new arena the context's current one.  Between the two the caller may begin a load on another context (its
 * upload queues up behind this one) or finish an earlier one.  The host buffers handed to _begin must stay valid and unchanged until
 * _finish has returned; scans of this context are only allowed again after _finish. */
int  kmpgpu_load_frames_begin(kmpgpu_ctx *ctx, const uint8_t *file_bytes, uint64_t file_nbytes, const uint64_t *frame_off,
                              const uint32_t *frame_caplen, uint64_t n_frames, int tcp);
int  kmpgpu_load_frames_finish(kmpgpu_ctx *ctx, uint64_t *n_payloads);
/* Between _begin and _finish: wait until the upload alone is through (the copy engine is free for the next context's). */
int  kmpgpu_load_frames_uploaded(kmpgpu_ctx *ctx);

/* Size the context's device buffers ahead of time for arenas of up to arena_bytes / n_pkts payloads and, when frame_bytes != 0, for
 * kmpgpu_load_frames calls of up to frame_bytes of capture / n_frames frames: a streamed capture (openmp_task.c:126-186) loads batch
 * after batch into the same buffers, and the first batch should not pay for a dozen device allocations.  Optional: every loader
 * grows what it needs. */
int  kmpgpu_reserve(kmpgpu_ctx *ctx, uint64_t arena_bytes, uint64_t n_pkts, uint64_t frame_bytes, uint64_t n_frames);

/* Same, for an arena already resident in device memory (borrowed; same contract, checked by a
 * device-side pass; an arena that is not packed is copied unless KMPGPU_OPT_REPACK is 0).  d_arena: uint8_t*, d_pkt_off: uint64_t*, d_pkt_len: uint32_t*.
 * IMMUTABILITY: the call derives state from the buffers -- packet-start bitmap, wavefront plan, uniform / packed
 * flags, the slot-padding check -- so arena AND index must stay unchanged for as long as they are attached.  A
 * device-resident batch ring that refills a buffer in place calls kmpgpu_attach_arena again after every refill (the
 * derived state is rebuilt by a few small kernels, no copy of the arena); scanning a rewritten buffer without it
 * counts against the OLD packet boundaries. */
int  kmpgpu_attach_arena(kmpgpu_ctx *ctx, const void *d_arena, uint64_t arena_bytes,
                         const void *d_pkt_off, const void *d_pkt_len, uint64_t n_pkts);

/* Replaces the hot loop serial.c:153-155 / openmp_data.c:157-175: counts_out[i] for every
 * pattern, in pattern order.  Synchronous; fills *t when non-NULL. */
int  kmpgpu_scan(kmpgpu_ctx *ctx, uint64_t *counts_out, kmpgpu_timing *t);

/* Asynchronous form: enqueue one full pass on the context's stream, no host synchronisation.
 * The counts (uint64_t[n_pat]) are written to d_counts_out, a caller-owned device buffer, or to
 * the context's own device buffer when d_counts_out is NULL.  That buffer is the operand of the
 * cross-GPU sum that replaces MPI_Reduce(local_string_count ...), mpi_dumping.c:202. */
int  kmpgpu_scan_enqueue(kmpgpu_ctx *ctx, void *d_counts_out);
/* Device address of the context's own counts buffer. */
void *kmpgpu_counts_device(kmpgpu_ctx *ctx);
/* Zero the context's own counts buffer (asynchronous, on the context's stream). */
int  kmpgpu_counts_reset(kmpgpu_ctx *ctx);
/* dst's counters += src's (both contexts on the same device, same patterns): the merge of contexts that share a GPU,
 * e.g. the two double-buffering contexts of a streamed capture (openmp_task.c:172-175).  Waits for src's stream, then
 * enqueues the addition on dst's. */
int  kmpgpu_counts_add(kmpgpu_ctx *dst, kmpgpu_ctx *src);
/* Timing of the context's last kmpgpu_load_arena / kmpgpu_load_frames / kmpgpu_scan. */
int  kmpgpu_last_timing(kmpgpu_ctx *ctx, kmpgpu_timing *t);
/* Wait for the context's stream and copy its own counts buffer to the host (uint64_t[n_pat]). */
int  kmpgpu_counts_read(kmpgpu_ctx *ctx, uint64_t *counts_out);
int  kmpgpu_sync(kmpgpu_ctx *ctx);

/* ---- the count reduce over GPUs: replaces MPI_Reduce(local_string_count, string_count, n, MPI_INT, MPI_SUM, 0, comm),
 * mpi_dumping.c:202 (and the scatter's rank/size bookkeeping, mpi_dumping.c:29-31) -- RCCL over xGMI.  librccl.so is
 * opened on the first call (no link-time dependency: single-GPU callers never load it).
 *
 * One rank per context, every context on its own device.
 *   kmpgpu_comm_init       all ranks live in THIS process (ncclCommInitAll): ctx[0..n_ctx) on n_ctx distinct devices;
 *   kmpgpu_comm_init_rank  one process per GPU (as mpirun starts mpi_dumping): rank `rank` of `n_ranks`, `unique_id`
 *                          (KMPGPU_COMM_ID_BYTES bytes) made once by kmpgpu_comm_unique_id and handed to every
 *                          rank by the launcher (file, environment, socket ...);
 *   kmpgpu_comm_allreduce_counts   enqueue ncclAllReduce(ncclUint64, ncclSum) in place over every local context's
 *                          counts buffer (kmpgpu_counts_device), on that context's stream, i.e. after the passes
 *                          enqueued so far (kmpgpu_scan_enqueue(ctx, NULL)); no host synchronisation: read the totals
 *                          with kmpgpu_counts_read afterwards (every rank holds them, like MPI_Allreduce).
 * All contexts of a communicator must hold the same number of patterns. */
#define KMPGPU_COMM_ID_BYTES 128
typedef struct kmpgpu_comm kmpgpu_comm;
int  kmpgpu_comm_init(kmpgpu_comm **comm, kmpgpu_ctx *const *ctx, int n_ctx);
int  kmpgpu_comm_unique_id(void *id_out);
int  kmpgpu_comm_init_rank(kmpgpu_comm **comm, kmpgpu_ctx *ctx, int n_ranks, int rank, const void *unique_id);
int  kmpgpu_comm_allreduce_counts(kmpgpu_comm *comm);
/* Lifetime: destroy the communicator BEFORE its contexts.  The other order is tolerated -- kmpgpu_destroy() of a context
 * that is still a rank waits for its stream and detaches it: kmpgpu_comm_allreduce_counts then fails with KMPGPU_ESTATE
 * and kmpgpu_comm_destroy only releases the RCCL handles -- but the peers of a detached rank must not enter another
 * collective.  A context belongs to at most one communicator.  Threads: kmpgpu_comm_init_rank may be called from one
 * thread per context at the same time (it blocks until every rank has joined); calls on ONE communicator are not
 * thread-safe.  While a communicator is being created the process's stdout (fd 1) points at stderr (RCCL's banner). */
void kmpgpu_comm_destroy(kmpgpu_comm *comm);
/* The device a context was created on. */
int  kmpgpu_device_of(kmpgpu_ctx *ctx);

/* Per-launch durations of the scan kernel, measured with HIP events on the launch stream.
 * begin(): start recording up to max_launches launches; end(): synchronise, write the
 * durations (ms) of the *n recorded launches, stop recording. */
int  kmpgpu_profile_begin(kmpgpu_ctx *ctx, uint32_t max_launches);
int  kmpgpu_profile_end(kmpgpu_ctx *ctx, float *ms_out, uint32_t *n);

/* Counts plus the matches themselves: every (packet, start offset, pattern) that counts, at most
 * cap of them written to out (host memory, unspecified order); *n_found is the total number found
 * (may exceed cap).  The pass writes its counts to a buffer of its own: the context's counters (a running total under
 * KMPGPU_OPT_ACCUMULATE, the result of a count reduce) stay as they are.  An arena that is scanned in place with slots
 * not back to back (KMPGPU_OPT_REPACK = 0) is packed on this call, once. */
int  kmpgpu_scan_offsets(kmpgpu_ctx *ctx, kmpgpu_match *out, uint64_t cap, uint64_t *n_found,
                         uint64_t *counts_out);

/* Fill a device arena with the synthetic payloads of kmp_synth.h (benchmark input S1/S2):
 * packet ids first_pkt_id .. first_pkt_id + n_pkts - 1 at the slots of the given device index. */
int  kmpgpu_synth_fill(kmpgpu_ctx *ctx, void *d_arena, const void *d_pkt_off, const void *d_pkt_len,
                       uint64_t first_pkt_id, uint64_t n_pkts, const kmp_synth_params *sp);

/* Device-side index for n fixed-length payloads at stride round_up(len, align): writes
 * d_pkt_off[k] = k * stride, d_pkt_len[k] = len. */
int  kmpgpu_fixed_index(kmpgpu_ctx *ctx, void *d_pkt_off, void *d_pkt_len, uint64_t n_pkts,
                        uint32_t len, uint32_t slot_align);

/* What the arena currently attached/loaded holds. */
int  kmpgpu_arena_info(kmpgpu_ctx *ctx, uint64_t *n_pkts, uint64_t *payload_bytes);
/* Sum over payloads of min(len, first 0x00 + 1): the bytes a strlen()-bounded scan (serial.c:191) has to
 * touch; equals payload_bytes on NUL-free input.  Reported beside the payload bytes for inputs that carry
 * NUL bytes; one extra pass over the arena, not part of kmpgpu_scan. */
int  kmpgpu_effective_bytes(kmpgpu_ctx *ctx, uint64_t *bytes_out);
/* Copy the context's device arena + index back to host buffers (tests: the on-device extraction must
 * build exactly the arena the host builds).  Buffers sized from kmpgpu_arena_info / arena_bytes. */
int  kmpgpu_arena_download(kmpgpu_ctx *ctx, uint8_t *arena_out, uint64_t arena_cap, uint64_t *arena_bytes,
                           uint64_t *pkt_off_out, uint32_t *pkt_len_out);

#ifdef __cplusplus
}
#endif
#endif /* KMPGPU_H */
