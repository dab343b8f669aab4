/*
 * kmp_cli.c -- the reference's command lines on top of the MI355X hot path.
 *
 * Built twice (csrc/Makefile):
 *   bin/serial       ./serial <file.pcap> <string.txt> [udp/tcp]                     serial.c:2-3,33-51
 *   bin/openmp_data  ./openmp_data <file.pcap> <string.txt> thread_number [tcp/udp]  openmp_data.c:1-2,35-54
 * In the second form thread_number is the number of GPU shards: the payloads are split into
 * contiguous ranges the way mpi_dumping.c:149-157 splits them over ranks (N/P each, remainder to
 * shard 0) and the per-shard counts are summed (mpi_dumping.c:202).
 *
 * stdout is byte-compatible with the reference (serial.c:163-169); throughput details go to
 * stderr.  There is no CPU fallback: without a gfx950 device the program fails with exit code 2.
 *
 * KMPGPU_DEVICE_EXTRACT=1: the raw capture is uploaded and the payload extraction
 * (openmp_data.c:128-147, packet_dumping.h:87-188) runs on the GPU (kmpgpu_load_frames).
 *
 * Extension (the reference prints counts only, serial.c:163-166): with the environment variable
 * KMPGPU_OFFSETS_FILE=<path> every match is also written to <path> as "payload,offset,pattern"
 * lines (payload = index among the extracted payloads, pattern = index in the pattern file).
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "kmpgpu.h"
#include "kmphost.h"

#ifndef KMP_CLI_OPENMP_FORM
#define KMP_CLI_OPENMP_FORM 0
#endif

#if KMP_CLI_OPENMP_FORM
#define PROG "./openmp_data"
#define ARGS "<file.pcap> <string.txt> thread_number [tcp/udp]"
#else
#define PROG "./serial"
#define ARGS "<file.pcap> <string.txt> [tcp/udp]"
#endif

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int parse_proto(const char *s, int *proto)
{
    if (strcmp(s, "udp") == 0) { *proto = KMP_PROTO_UDP; return 1; }       /* serial.c:38-41 */
    if (strcmp(s, "tcp") == 0) { *proto = KMP_PROTO_TCP; return 1; }
    return 0;
}

static void die_gpu(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, kmpgpu_last_error());
    exit(2);
}

/* The HIP runtime takes 0.2-0.3 s to come up.  It does so on a side thread while the main thread maps, indexes
 * and extracts the capture; the two meet before the first context is created. */
static void *warm_gpu(void *arg)
{
    (void)arg;
    if (kmpgpu_device_count() > 0) {
        kmpgpu_ctx *w = NULL;
        if (kmpgpu_init(&w, 0) == 0) kmpgpu_destroy(w);
    }
    return NULL;
}

/* One GPU shard: a contiguous range of the payloads (or, with KMPGPU_DEVICE_EXTRACT=1, of the frames). */
typedef struct shard_job {
    int device, tcp, threaded, rc;
    uint64_t lo, cnt;                       /* first unit and number of units of this shard */
    const kmp_arena *arena;
    const kmp_frames *frames;               /* non-NULL: extraction on the device */
    const uint8_t *const *pp;
    const kmp_patterns *pats;
    kmpgpu_ctx *ctx;
    kmpgpu_timing t;
    uint64_t n_payloads, payload_bytes;
    uint64_t *reb, *own;
    const char *what;
    char err[512];
    pthread_t thread;
} shard_job;

static void *shard_fail(shard_job *j, const char *what)
{
    j->rc = 1; j->what = what;
    snprintf(j->err, sizeof j->err, "%s", kmpgpu_last_error());            /* the error text is per thread */
    return NULL;
}

static void *shard_load(void *arg)
{
    shard_job *j = (shard_job *)arg;
    if (kmpgpu_init(&j->ctx, j->device)) return shard_fail(j, "kmpgpu_init");
    if (kmpgpu_set_patterns(j->ctx, j->pp, j->pats->len, j->pats->n)) return shard_fail(j, "kmpgpu_set_patterns");
    if (j->frames) {
        /* only the bytes this shard's frames span are uploaded (kmpgpu_load_frames) */
        if (kmpgpu_load_frames(j->ctx, j->frames->bytes, j->frames->nbytes, j->frames->off + j->lo, j->frames->caplen + j->lo, j->cnt, j->tcp,
                               &j->n_payloads))
            return shard_fail(j, "kmpgpu_load_frames");
        kmpgpu_arena_info(j->ctx, NULL, &j->payload_bytes);
    } else {
        const kmp_arena *a = j->arena;
        const uint64_t hi = j->lo + j->cnt;
        const uint64_t b0 = a->off[j->lo];
        const uint64_t b1 = (hi < a->n_pkts) ? a->off[hi] : a->nbytes;
        j->reb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)j->cnt);
        if (!j->reb) { j->rc = 1; j->what = "malloc"; snprintf(j->err, sizeof j->err, "out of memory"); return NULL; }
        for (uint64_t k = 0; k < j->cnt; k++) j->reb[k] = a->off[j->lo + k] - b0;
        /* slots are contiguous and at least 16 bytes each, so [b0, b1) holds the whole shard */
        if (kmpgpu_load_arena(j->ctx, a->bytes + b0, b1 - b0, j->reb, a->len + j->lo, j->cnt)) return shard_fail(j, "kmpgpu_load_arena");
    }
    if (kmpgpu_last_timing(j->ctx, &j->t)) return shard_fail(j, "kmpgpu_last_timing");
    return NULL;
}

int main(int argc, char *argv[])
{
    int proto = KMP_PROTO_UDP;                                              /* serial.c:31 */
    int shards = 1;
#if KMP_CLI_OPENMP_FORM
    if (argc == 4 || argc == 5) {                                           /* openmp_data.c:35 */
        shards = atoi(argv[3]);                                             /* openmp_data.c:38 */
        if (argc == 5 && !parse_proto(argv[4], &proto)) {
            printf("USAGE " PROG " " ARGS "\n");                            /* openmp_data.c:46 */
            exit(1);
        }
    } else {
        printf("USAGE: " PROG " " ARGS "\n");                               /* openmp_data.c:52 */
        exit(1);
    }
    if (shards < 1) shards = 1;
#else
    if (argc == 3 || argc == 4) {                                           /* serial.c:33 */
        if (argc == 4 && !parse_proto(argv[3], &proto)) {
            printf("USAGE " PROG " " ARGS "\n");                            /* serial.c:43 */
            exit(1);
        }
    } else {
        printf("USAGE: " PROG " " ARGS "\n");                               /* serial.c:49 */
        exit(1);
    }
#endif
    const char *pcap_path = argv[1], *strings_path = argv[2];

    kmp_patterns pats;
    int rc = kmp_patterns_load(strings_path, &pats);                        /* serial.c:54-87 */
    if (rc == KMPHOST_EIO) {
        perror("error opening file: ");                                     /* serial.c:61 */
        exit(1);
    }
    if (rc) {
        fprintf(stderr, "error reading pattern file: %s\n", rc == KMPHOST_ETOKEN ? "token longer than 99 bytes" : "out of memory");
        exit(1);
    }

#if !KMP_CLI_OPENMP_FORM
    const double t_start = now_s();                                         /* serial.c:110-111: before the file read */
#endif
    pthread_t warm;
    const int warming = pthread_create(&warm, NULL, warm_gpu, NULL) == 0;
    const double t_load0 = now_s();
    char errbuf[KMP_PCAP_ERRBUF];
    kmp_arena arena;
    kmp_frames frames;
    memset(&arena, 0, sizeof arena);
    memset(&frames, 0, sizeof frames);
    const char *dx = getenv("KMPGPU_DEVICE_EXTRACT");
    const int device_extract = dx && dx[0] == '1';
    /* One-shot buffers stay in ordinary memory: pinning 1.5 GB costs 0.22 s and unpinning 0.15 s, while the
     * host-to-device copy runs at PCIe speed from pageable memory and even straight from the mapped file on the
     * MI355X hosts (profiles/r01_h2d_probe.txt).  Pinned buffers pay off where they are reused (bin/openmp_task). */
    if (device_extract)
        rc = kmp_frames_from_pcap(pcap_path, NULL, NULL, &frames, errbuf);
    else
        rc = kmp_arena_from_pcap(pcap_path, proto, NULL, NULL, &arena, errbuf);                              /* serial.c:91-141 */
    const double t_loaded = now_s();
    if (warming) pthread_join(warm, NULL);                                  /* the runtime is up (or there is none: reported below) */
    const double t_warm = now_s();
    if (rc == KMPHOST_EIO || rc == KMPHOST_EFORMAT) {
        fprintf(stderr, "error reading pcap file: %s\n", errbuf);           /* serial.c:93 */
        exit(1);
    }
    if (rc) {
        fprintf(stderr, "error building the payload arena: %s\n", errbuf[0] ? errbuf : kmpgpu_last_error());
        exit(2);
    }
#if KMP_CLI_OPENMP_FORM
    const double t_start = now_s();                                         /* openmp_data.c:126: after the pre-load */
#endif

    const int ndev = kmpgpu_device_count();
    if (ndev <= 0) die_gpu("no MI355X device");

    uint64_t *counts = (uint64_t *)calloc(pats.n ? pats.n : 1, sizeof(uint64_t));
    uint64_t *part = (uint64_t *)calloc(pats.n ? pats.n : 1, sizeof(uint64_t));
    const uint8_t **pp = (const uint8_t **)malloc(sizeof(uint8_t *) * (pats.n ? pats.n : 1));
    for (uint32_t i = 0; i < pats.n; i++) pp[i] = pats.blob + pats.off[i];

    double kernel_ms = 0, h2d_ms = 0;
    /* KMPGPU_STATS=1: also report the bytes up to the first NUL of every payload (what strlen bounds, serial.c:191);
     * one extra pass over the arena, so it is opt-in and the "Elapsed time" line of a plain run stays comparable */
    const char *stats_env = getenv("KMPGPU_STATS");
    const int want_stats = stats_env && stats_env[0] && stats_env[0] != '0';
    uint64_t eff_bytes = 0, h2d_bytes = 0;
    int reduce_rccl = 0;
    kmpgpu_comm *td_comm = NULL; kmpgpu_ctx **td_ctxs = NULL; shard_job *td_job = NULL; int td_n = 0;      /* torn down after the report */
    const uint64_t units = device_extract ? frames.n : arena.n_pkts;       /* what is split over the shards */
    if (pats.n && units) {
        /* Shards: contiguous ranges, N/P each, the remainder to shard 0 (mpi_dumping.c:149-157); shard r on device
         * r % ndev.  Every shard is brought up by its own host thread -- context, patterns, upload (and extraction) --
         * so that the uploads of the shards run side by side, one PCIe link each (MPI_Scatterv, mpi_dumping.c:161). */
        if ((uint64_t)shards > units) shards = (int)units;
        shard_job *job = (shard_job *)calloc((size_t)shards, sizeof *job);
        uint64_t lo = 0;
        for (int r = 0; r < shards; r++) {
            const uint64_t cnt = units / (uint64_t)shards + (r == 0 ? units % (uint64_t)shards : 0);
            job[r].device = r % ndev; job[r].lo = lo; job[r].cnt = cnt;
            job[r].arena = &arena; job[r].frames = device_extract ? &frames : NULL; job[r].tcp = proto == KMP_PROTO_TCP;
            job[r].pp = pp; job[r].pats = &pats;
            lo += cnt;
        }
        for (int r = 1; r < shards; r++)
            if (pthread_create(&job[r].thread, NULL, shard_load, &job[r]) != 0) { job[r].threaded = 0; shard_load(&job[r]); } else job[r].threaded = 1;
        shard_load(&job[0]);
        for (int r = 1; r < shards; r++) if (job[r].threaded) pthread_join(job[r].thread, NULL);
        for (int r = 0; r < shards; r++) {
            if (job[r].rc) { fprintf(stderr, "%s: %s\n", job[r].what, job[r].err); exit(2); }
            h2d_ms += job[r].t.h2d_ms; h2d_bytes += job[r].t.h2d_bytes;
            if (device_extract) { arena.n_pkts += job[r].n_payloads; arena.payload_bytes += job[r].payload_bytes; }
        }
        if (device_extract) arena.n_frames = frames.n;

        /* The count reduce (mpi_dumping.c:202).  One shard per device: RCCL all-reduce over xGMI of the shards' device
         * counters (kmpgpu_comm_*), then ONE download.  Shards that share a device (more shards than GPUs): host sum.
         * KMPGPU_RCCL=0 forces the host sum, =1 asks for the communicator even with a single shard. */
        const char *rccl_env = getenv("KMPGPU_RCCL");
        kmpgpu_comm *comm = NULL;
        kmpgpu_ctx **ctxs = (kmpgpu_ctx **)calloc((size_t)shards, sizeof *ctxs);
        for (int r = 0; r < shards; r++) ctxs[r] = job[r].ctx;
        if (shards <= ndev && (shards > 1 || (rccl_env && rccl_env[0] == '1')) && !(rccl_env && rccl_env[0] == '0')) {
            /* no communicator is no reason to stop: the shards' own counters are summed on the host instead (below) */
            if (kmpgpu_comm_init(&comm, ctxs, shards)) {
                fprintf(stderr, "[kmpgpu] kmpgpu_comm_init: %s -- summing the shards' counts on the host\n", kmpgpu_last_error());
                comm = NULL;
            } else reduce_rccl = 1;
        }
        const double t_scan0 = now_s();
        /* every shard's pass is enqueued before any result is read, so the GPUs work side by side
         * (mpi_dumping.c:198-202: all ranks count, then one reduce) */
        for (int r = 0; r < shards; r++)
            if (kmpgpu_scan_enqueue(ctxs[r], NULL)) die_gpu("kmpgpu_scan_enqueue");
        if (comm) {
            /* the shards' own counts are kept before they are summed in place: the offset files need them again below, and
             * they are what the host sums should the all-reduce fail (n_patterns x 8 bytes per shard) */
            for (int r = 0; r < shards; r++) {
                job[r].own = (uint64_t *)malloc(sizeof(uint64_t) * (pats.n ? pats.n : 1));
                if (!job[r].own || kmpgpu_counts_read(ctxs[r], job[r].own)) die_gpu("kmpgpu_counts_read");
            }
            int bad = kmpgpu_comm_allreduce_counts(comm) != 0;
            if (!bad) bad = kmpgpu_counts_read(ctxs[0], counts) != 0;                          /* MPI_Reduce root 0 */
            for (int r = 1; r < shards && !bad; r++) bad = kmpgpu_sync(ctxs[r]) != 0;
            if (bad) {
                fprintf(stderr, "[kmpgpu] RCCL all-reduce of the counts: %s -- summing the shards' counts on the host\n", kmpgpu_last_error());
                reduce_rccl = 0;
                for (uint32_t i = 0; i < pats.n; i++) {
                    counts[i] = 0;
                    for (int r = 0; r < shards; r++) counts[i] += job[r].own[i];            /* mpi_dumping.c:202 MPI_SUM */
                }
            }
        } else {
            for (int r = 0; r < shards; r++) {
                if (kmpgpu_counts_read(ctxs[r], part)) die_gpu("kmpgpu_counts_read");
                for (uint32_t i = 0; i < pats.n; i++) counts[i] += part[i];     /* mpi_dumping.c:202 MPI_SUM */
                if (getenv("KMPGPU_OFFSETS_FILE")) {
                    job[r].own = (uint64_t *)malloc(sizeof(uint64_t) * pats.n);
                    memcpy(job[r].own, part, sizeof(uint64_t) * pats.n);
                }
            }
        }
        kernel_ms = (now_s() - t_scan0) * 1e3;                              /* wall time of the concurrent passes + reduce (mpi_dumping.c:206 MPI_MAX) */

        const char *off_path = getenv("KMPGPU_OFFSETS_FILE");
        if (off_path && off_path[0]) {
            FILE *off_fp = fopen(off_path, "w");
            if (!off_fp) { perror("KMPGPU_OFFSETS_FILE"); exit(1); }
            uint64_t shard_lo = 0;                                          /* payload index of the shard's first payload */
            for (int r = 0; r < shards; r++) {
                uint64_t total = 0, found = 0, np = 0;
                for (uint32_t i = 0; i < pats.n; i++) total += job[r].own[i];
                kmpgpu_match *mm = (kmpgpu_match *)malloc(sizeof *mm * (size_t)(total ? total : 1));
                if (!mm || kmpgpu_scan_offsets(ctxs[r], mm, total, &found, NULL)) die_gpu("kmpgpu_scan_offsets");
                for (uint64_t i = 0; i < found && i < total; i++)
                    fprintf(off_fp, "%llu,%u,%u\n", (unsigned long long)(mm[i].packet + shard_lo), mm[i].offset, mm[i].pattern);
                free(mm);
                kmpgpu_arena_info(ctxs[r], &np, NULL);
                shard_lo += np;
            }
            fclose(off_fp);
        }
        for (int r = 0; r < shards && want_stats; r++) {
            uint64_t e = 0;
            if (kmpgpu_effective_bytes(ctxs[r], &e)) die_gpu("kmpgpu_effective_bytes");
            eff_bytes += e;
        }
        /* (teardown after the report: serial.c:159-160 takes the time before it frees anything, :178-180) */
        td_comm = comm; td_ctxs = ctxs; td_job = job; td_n = shards;
    }
    const double t_finish = now_s();                                        /* serial.c:159-160 */

    kmp_report(stdout, &pats, counts, t_finish - t_start);                  /* serial.c:163-169 */
    if (td_comm) kmpgpu_comm_destroy(td_comm);
    for (int r = 0; r < td_n; r++) { kmpgpu_destroy(td_ctxs[r]); free(td_job[r].own); free(td_job[r].reb); }
    free(td_ctxs); free(td_job);

    if (kernel_ms > 0) {
        uint64_t total = 0;
        for (uint32_t i = 0; i < pats.n; i++) total += counts[i];
        const double bytes = (double)arena.payload_bytes * (double)pats.n;
        fprintf(stderr, "[kmpgpu] %llu frames, %llu payloads, %llu payload bytes, %u patterns, %d shard(s) on %d device(s), count reduce: %s, %llu bytes uploaded\n",
                (unsigned long long)arena.n_frames, (unsigned long long)arena.n_pkts, (unsigned long long)arena.payload_bytes,
                pats.n, shards, ndev, reduce_rccl ? "RCCL all-reduce" : (shards > 1 ? "host sum" : "none"), (unsigned long long)h2d_bytes);
        fprintf(stderr, "[kmpgpu] kernel %.3f ms (%.2f GB/s payload x patterns, %.3g matches/s), h2d %.3f ms\n", kernel_ms,
                bytes / (kernel_ms * 1e6), (double)total / (kernel_ms * 1e-3), h2d_ms);
        if (want_stats) {
            fprintf(stderr, "[kmpgpu] %llu of the %llu payload bytes lie at or before the first NUL of their payload\n",
                    (unsigned long long)eff_bytes, (unsigned long long)arena.payload_bytes);
            fprintf(stderr, "[kmpgpu] phases: capture -> host buffers %.3f s, waiting for the HIP runtime %.3f s, contexts + upload + scan %.3f s\n",
                    t_loaded - t_load0, t_warm - t_loaded, t_finish - t_warm);
        }
    }
    free(counts); free(part); free(pp);
    kmp_arena_free(&arena);
    kmp_frames_free(&frames);
    kmp_patterns_free(&pats);
    return 0;
}
