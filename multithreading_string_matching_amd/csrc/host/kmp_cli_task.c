/*
 * kmp_cli_task.c -- bin/openmp_task: the reference's streaming variant on the MI355X hot path.
 *
 *   ./openmp_task <file.pcap> <string.txt> thread_number [tcp/udp]          openmp_task.c:1-2,35-55
 *
 * openmp_task.c:126-186 lets one thread read the capture 100 packets at a time and spawns a task per
 * batch; the tasks add their private counters into the shared ones.  Here the producer (this thread)
 * reads batches of KMPGPU_BATCH_BYTES (default 64 MiB) into pinned buffers while consumer threads,
 * two per GPU shard (thread_number), upload and scan them: pcap read || H2D || scan.  Every consumer
 * has a context of its own (own stream, own device buffers) so that the upload of a batch
 * overlaps the scan of the previous one; the contexts accumulate counts over their batches
 * (KMPGPU_OPT_ACCUMULATE) and the totals are summed at the end (openmp_task.c:172-175).
 *
 * KMPGPU_DEVICE_EXTRACT=1 (SURVEY 8(f) N2 x N3): the producer only walks the record headers of the mapped capture
 * (kmp_batch_next_frames); a second host thread stages the RAW bytes of a batch -- headers, frames, everything, one bulk
 * copy by several threads -- into a pinned buffer while the producer walks the next batch; the consumers upload that
 * buffer and extract the payloads on the GPU (kmpgpu_load_frames, packet_dumping.h:87-188 on the device).  No per-packet
 * work on the host.  (Uploading straight from the mapping was measured and dropped: pageable pages go through the
 * runtime's bounce buffers at 15 GB/s, and pinning them costs 55 ms per GB: profiles/r03_end_to_end_pinning_experiment.txt.)
 *
 * stdout is byte-compatible with the reference (openmp_task.c:190-196).  No CPU fallback: exit code 2
 * without a gfx950 device.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "kmpgpu.h"
#include "kmphost.h"


typedef struct slot {
    uint8_t  *arena;                 /* pinned: the payload arena of a batch / the raw bytes of a frame batch */
    uint64_t *off;                   /* payload offsets in arena / frame offsets in the capture */
    uint32_t *len;                   /* payload lengths / captured lengths of the frames       */
    uint64_t  used, n;
    uint64_t  lo;                    /* frame batches: file offset of the batch's first staged byte */
    int       state;                 /* 0 free, 1 filled (frame batches: 1 walked, 2 staged) */
} slot;

#define KMP_MAX_DEVICES 64
typedef struct shared {
    pthread_mutex_t mu;
    pthread_cond_t  cv;
    slot           *slots;
    int             n_slots;
    int             done;            /* producer finished */
    uint64_t        next_fill, next_take;     /* ring positions */
    uint64_t        next_walk, next_copy;     /* frame batches: walked / staged so far */
    int             walked_all;
    double          copy_s;
    const kmp_patterns *pats;
    const uint8_t **pp;
    int             ndev;
    int             failed;
    int             ready;           /* consumers whose contexts are up */
    /* KMPGPU_DEVICE_EXTRACT=1: batches of raw frames of the mapped capture */
    int             frames_mode, tcp;
    const uint8_t  *file;
    uint64_t        file_bytes;
    uint64_t        batch_bytes, cap_pkts;
    int             per_shard;       /* consumer threads (contexts) per GPU shard */
    int             serial_uploads;  /* raw frames: an upload is enqueued only when the one before it on that device is through (default) */
    pthread_mutex_t up_mu[KMP_MAX_DEVICES];
    kmpgpu_ctx     *up_last[KMP_MAX_DEVICES];
} shared;

typedef struct consumer {
    shared   *sh;
    int       id;
    kmpgpu_ctx *total;               /* the context that holds this consumer's counts when it is done */
    kmpgpu_ctx *extra;               /* raw frames: its second context */
    double    kernel_ms, h2d_ms;
    uint64_t  batches, payloads, bytes;
    double    load_s, wait_s;        /* time inside the load calls / waiting for a filled slot */
} consumer;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void die_gpu(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, kmpgpu_last_error());
    exit(2);
}

/* KMPGPU_DEVICE_EXTRACT=1, second stage: the raw bytes of a walked batch go from the mapped capture into the slot's pinned buffer
 * (kmp_copy_bytes: several threads), the frame offsets are rebased to that buffer, and the batch is handed to the consumers. */
static void *stage_batches(void *arg)
{
    shared *sh = (shared *)arg;
    for (;;) {
        pthread_mutex_lock(&sh->mu);
        while (sh->next_copy == sh->next_walk && !sh->walked_all) pthread_cond_wait(&sh->cv, &sh->mu);
        if (sh->next_copy == sh->next_walk && sh->walked_all) {
            sh->done = 1;
            pthread_cond_broadcast(&sh->cv);
            pthread_mutex_unlock(&sh->mu);
            return NULL;
        }
        slot *s = &sh->slots[sh->next_copy % (uint64_t)sh->n_slots];
        pthread_mutex_unlock(&sh->mu);
        const double t0 = now_s();
        kmp_copy_bytes(s->arena, sh->file + s->lo, s->used);
        for (uint64_t k = 0; k < s->n; k++) s->off[k] -= s->lo;
        pthread_mutex_lock(&sh->mu);
        sh->copy_s += now_s() - t0;
        s->state = 2;
        sh->next_copy++;
        sh->next_fill++;
        pthread_cond_broadcast(&sh->cv);
        pthread_mutex_unlock(&sh->mu);
    }
}

static kmpgpu_ctx *make_context(const shared *sh, int device)
{
    kmpgpu_ctx *c = NULL;
    if (kmpgpu_init(&c, device)) die_gpu("kmpgpu_init");
    if (kmpgpu_set_patterns(c, sh->pp, sh->pats->len, sh->pats->n)) die_gpu("kmpgpu_set_patterns");
    if (kmpgpu_set_option(c, KMPGPU_OPT_ACCUMULATE, 1) || kmpgpu_counts_reset(c)) die_gpu("kmpgpu_set_option");
    /* the device buffers of a batch, once: the first batch does not pay for a dozen allocations under the clock */
    if (kmpgpu_reserve(c, sh->batch_bytes + 64, sh->cap_pkts, sh->frames_mode ? sh->batch_bytes + 64 : 0, sh->frames_mode ? sh->cap_pkts : 0)) die_gpu("kmpgpu_reserve");
    if (kmpgpu_sync(c)) die_gpu("kmpgpu_sync");
    return c;
}

/* the next filled slot, or NULL when the capture is through */
static slot *take_slot(consumer *me)
{
    shared *sh = me->sh;
    const double tw0 = now_s();
    pthread_mutex_lock(&sh->mu);
    while (sh->next_take == sh->next_fill && !sh->done) pthread_cond_wait(&sh->cv, &sh->mu);
    me->wait_s += now_s() - tw0;
    slot *s = NULL;
    if (sh->next_take != sh->next_fill) { s = &sh->slots[sh->next_take % (uint64_t)sh->n_slots]; sh->next_take++; }
    pthread_mutex_unlock(&sh->mu);
    return s;
}

static void free_slot(shared *sh, slot *s)
{
    pthread_mutex_lock(&sh->mu);
    s->state = 0;                                            /* the pinned buffer may be refilled */
    pthread_cond_broadcast(&sh->cv);
    pthread_mutex_unlock(&sh->mu);
}

static void *consume(void *arg)
{
    consumer *me = (consumer *)arg;
    shared *sh = me->sh;
    const int device = (me->id / sh->per_shard) % sh->ndev;
    /* Contexts have their own stream and device buffers.  Payload batches: one context per consumer thread, two threads per shard
     * (the upload of one overlaps the scan of the other).  Raw frames: ONE thread per shard drives TWO contexts and keeps the copy
     * engine fed -- the upload of batch i + 1 is enqueued (kmpgpu_load_frames_begin) before the thread waits for batch i's
     * extraction (kmpgpu_load_frames_finish), so the uploads follow each other without a gap and the extraction and the scans
     * run behind them. */
    kmpgpu_ctx *ctx[2] = {make_context(sh, device), sh->frames_mode ? make_context(sh, device) : NULL};
    me->total = ctx[0]; me->extra = ctx[1];
    pthread_mutex_lock(&sh->mu);                               /* the contexts are up: the clock may start (main waits for every consumer) */
    sh->ready++;
    pthread_cond_broadcast(&sh->cv);
    pthread_mutex_unlock(&sh->mu);

    if (sh->frames_mode) {
        slot *pend[2] = {NULL, NULL};
        int turn = 0;
        for (;;) {
            slot *s = take_slot(me);
            const double tl0 = now_s();
            if (s) {
                /* ONE upload at a time per device: two in flight (one per context's stream, or one per shard that shares the GPU) split the
                 * link between them and both finish late -- 47 ms for the 1.56 GB capture, 41 ms with each upload enqueued when the one
                 * before it is through (profiles/r03_end_to_end.txt).  The token is the context whose upload was enqueued last. */
                pthread_mutex_lock(&sh->up_mu[device]);
                if (sh->serial_uploads && sh->up_last[device] && kmpgpu_load_frames_uploaded(sh->up_last[device])) die_gpu("kmpgpu_load_frames_uploaded");
                if (kmpgpu_load_frames_begin(ctx[turn], s->arena, s->used, s->off, s->len, s->n, sh->tcp)) die_gpu("kmpgpu_load_frames");
                sh->up_last[device] = ctx[turn];
                pthread_mutex_unlock(&sh->up_mu[device]);
                pend[turn] = s;
            }
            /* then the batch begun before this one (the other context); at the end of the capture, whatever is still pending */
            const int order[2] = {turn ^ 1, turn};
            for (int i = 0; i < (s ? 1 : 2); i++) {
                const int k = order[i];
                if (!pend[k]) continue;
                uint64_t np = 0, pb = 0;
                if (kmpgpu_load_frames_finish(ctx[k], &np)) die_gpu("kmpgpu_load_frames");
                kmpgpu_arena_info(ctx[k], NULL, &pb);
                me->payloads += np; me->bytes += np ? pb : 0;
                { kmpgpu_timing tt; if (kmpgpu_last_timing(ctx[k], &tt) == 0) me->h2d_ms += tt.h2d_ms; }
                me->batches++;
                free_slot(sh, pend[k]);
                pend[k] = NULL;
                if (kmpgpu_scan_enqueue(ctx[k], NULL)) die_gpu("kmpgpu_scan_enqueue");
            }
            me->load_s += now_s() - tl0;
            if (!s) break;
            turn ^= 1;
        }
        /* the two contexts' running totals are merged on the device (openmp_task.c:172-175) */
        if (kmpgpu_counts_add(ctx[0], ctx[1])) die_gpu("kmpgpu_counts_add");
    } else {
        kmpgpu_ctx *c = ctx[0];
        for (;;) {
            slot *s = take_slot(me);
            if (!s) break;
            /* waits for this context's previous scan, uploads (the other contexts' work keeps running) */
            const double tl0 = now_s();
            /* (one load at a time per device, as for the raw frames above: kmpgpu_load_arena returns when its upload and the index behind it
             * are through; the scan that follows runs beside the other thread's upload) */
            if (sh->serial_uploads) pthread_mutex_lock(&sh->up_mu[device]);
            const int lrc = kmpgpu_load_arena(c, s->arena, s->used, s->off, s->len, s->n);
            if (sh->serial_uploads) pthread_mutex_unlock(&sh->up_mu[device]);
            if (lrc) die_gpu("kmpgpu_load_arena");
            me->payloads += s->n;
            me->load_s += now_s() - tl0;
            { kmpgpu_timing tt; if (kmpgpu_last_timing(c, &tt) == 0) me->h2d_ms += tt.h2d_ms; }
            me->batches++;
            free_slot(sh, s);
            if (kmpgpu_scan_enqueue(c, NULL)) die_gpu("kmpgpu_scan_enqueue");
        }
    }
    if (kmpgpu_sync(ctx[0])) die_gpu("kmpgpu_sync");
    return NULL;
}

int main(int argc, char *argv[])
{
    int proto = KMP_PROTO_UDP;
    int shards = 1;
    if (argc == 4 || argc == 5) {                                            /* openmp_task.c:35 */
        shards = atoi(argv[3]);                                              /* openmp_task.c:38 */
        if (argc == 5) {
            if (strcmp(argv[4], "udp") == 0) proto = KMP_PROTO_UDP;
            else if (strcmp(argv[4], "tcp") == 0) proto = KMP_PROTO_TCP;
            else {
                printf("USAGE ./openmp_task <file.pcap> <string.txt> thread_number [tcp/udp]\n");   /* openmp_task.c:46 */
                exit(1);
            }
        }
    } else {
        printf("USAGE: ./openmp_task <file.pcap> <string.txt> [tcp/udp]\n");                       /* openmp_task.c:52 (sic) */
        exit(1);
    }
    if (shards < 1) shards = 1;

    kmp_patterns pats;
    int rc = kmp_patterns_load(argv[2], &pats);                                                     /* openmp_task.c:57-96 */
    if (rc == KMPHOST_EIO) { perror("error opening file: "); exit(1); }
    if (rc) { fprintf(stderr, "error reading pattern file\n"); exit(1); }

    char errbuf[KMP_PCAP_ERRBUF];
    kmp_batch_reader *rd = kmp_batch_open(argv[1], proto, errbuf);                                 /* openmp_task.c:104-108 */
    if (!rd) { fprintf(stderr, "error reading pcap file: %s\n", errbuf); exit(1); }

    int ndev = kmpgpu_device_count();
    if (ndev <= 0) die_gpu("no MI355X device");
    if (ndev > KMP_MAX_DEVICES) ndev = KMP_MAX_DEVICES;

    uint64_t batch_bytes = 64ull << 20;
    const char *env = getenv("KMPGPU_BATCH_BYTES");
    if (env && atoll(env) >= (1 << 16)) batch_bytes = (uint64_t)atoll(env);                 /* 64 KiB and up: a frame (<= 64 KiB captured) always fits */
    const char *dx = getenv("KMPGPU_DEVICE_EXTRACT");
    const int frames_mode = dx && dx[0] == '1';
    /* a frame record takes 16 bytes of header and up; a payload slot 16 bytes and up, 64 on average or more in practice */
    const uint64_t cap_pkts = frames_mode ? batch_bytes / 32 : batch_bytes / 64;

    /* consumer threads (contexts) per GPU shard: two by default; KMPGPU_STREAM_CONTEXTS = 1..8 */
    int per_shard = 2;
    { const char *e = getenv("KMPGPU_STREAM_CONTEXTS"); if (e && atoi(e) >= 1 && atoi(e) <= 8) per_shard = atoi(e); }
    { const char *dx0 = getenv("KMPGPU_DEVICE_EXTRACT"); if (dx0 && dx0[0] == '1') per_shard = 1; }       /* raw frames: one thread per shard drives two contexts */
    shared sh;
    memset(&sh, 0, sizeof sh);
    pthread_mutex_init(&sh.mu, NULL);
    pthread_cond_init(&sh.cv, NULL);
    /* one being walked / read, one being staged, one or two per consumer being uploaded, one spare */
    sh.n_slots = (frames_mode ? 5 : 3 + per_shard) * shards;
    sh.slots = (slot *)calloc((size_t)sh.n_slots, sizeof(slot));
    sh.pats = &pats; sh.ndev = ndev;
    sh.frames_mode = frames_mode; sh.tcp = proto == KMP_PROTO_TCP;
    sh.batch_bytes = batch_bytes; sh.cap_pkts = cap_pkts; sh.per_shard = per_shard;
    { const char *e = getenv("KMPGPU_SERIAL_UPLOADS"); sh.serial_uploads = e ? atoi(e) != 0 : 1; }
    for (int d = 0; d < KMP_MAX_DEVICES; d++) { pthread_mutex_init(&sh.up_mu[d], NULL); sh.up_last[d] = NULL; }
    sh.file = kmp_batch_file(rd, &sh.file_bytes);
    sh.pp = (const uint8_t **)malloc(sizeof(uint8_t *) * (pats.n ? pats.n : 1));
    for (uint32_t i = 0; i < pats.n; i++) sh.pp[i] = pats.blob + pats.off[i];
    for (int i = 0; i < sh.n_slots; i++) {
        sh.slots[i].arena = (uint8_t *)kmpgpu_host_alloc((size_t)batch_bytes + 64);
        sh.slots[i].off = (uint64_t *)kmpgpu_host_alloc((size_t)cap_pkts * sizeof(uint64_t));
        sh.slots[i].len = (uint32_t *)kmpgpu_host_alloc((size_t)cap_pkts * sizeof(uint32_t));
        if (!sh.slots[i].arena || !sh.slots[i].off || !sh.slots[i].len) die_gpu("kmpgpu_host_alloc");
    }

    /* The consumers and their GPU contexts (streams, pattern tables) are set up before the clock starts, like the pinned
     * buffers above and like everything openmp_task.c does before :124 (pattern load, pcap_open_offline, allocations). */
    const int n_cons = per_shard * shards;
    consumer *cons = (consumer *)calloc((size_t)n_cons, sizeof(consumer));
    pthread_t *th = (pthread_t *)calloc((size_t)n_cons, sizeof(pthread_t));
    if (pats.n) {
        for (int r = 0; r < n_cons; r++) {
            cons[r].sh = &sh; cons[r].id = r;
            if (pthread_create(&th[r], NULL, consume, &cons[r]) != 0) { fprintf(stderr, "cannot start a consumer thread\n"); exit(2); }
        }
        pthread_mutex_lock(&sh.mu);
        while (sh.ready < n_cons) pthread_cond_wait(&sh.cv, &sh.mu);
        pthread_mutex_unlock(&sh.mu);
    }
    const double t_start = now_s();                                                                 /* openmp_task.c:124 */
    pthread_t stage_th;
    const int staging = frames_mode && pats.n && pthread_create(&stage_th, NULL, stage_batches, &sh) == 0;
    if (frames_mode && pats.n && !staging) { fprintf(stderr, "cannot start the staging thread\n"); exit(2); }
    uint64_t frames = 0, payloads = 0, bytes = 0, batches = 0;
    double prod_walk_s = 0, prod_wait_s = 0;
    for (;;) {                                                                                      /* openmp_task.c:130-155: the producer */
        slot *s = &sh.slots[(frames_mode ? sh.next_walk : sh.next_fill) % (uint64_t)sh.n_slots];
        const double tp0 = now_s();
        pthread_mutex_lock(&sh.mu);
        while (s->state != 0) pthread_cond_wait(&sh.cv, &sh.mu);
        pthread_mutex_unlock(&sh.mu);
        const double tp1 = now_s();
        prod_wait_s += tp1 - tp0;
        /* the first batches are small (8, 16, 32 MiB ...): the upload of the first one starts a millisecond sooner */
        uint64_t this_batch = batch_bytes;
        if (batches < 4 && ((8ull << 20) << batches) < batch_bytes) this_batch = (8ull << 20) << batches;
        int64_t n;
        if (frames_mode) {
            n = kmp_batch_next_frames(rd, this_batch, s->off, s->len, cap_pkts);                    /* record headers only */
            if (n > 0) {
                frames += (uint64_t)n;
                s->lo = s->off[0] & ~(uint64_t)15;                                                  /* (keeps the frames' alignment in the staged copy) */
                s->used = s->off[n - 1] + s->len[n - 1] - s->lo;
                if (s->used > batch_bytes + 64) n = -1;                                             /* one record larger than a batch */
            }
        } else
            n = kmp_batch_next(rd, s->arena, this_batch, s->off, s->len, cap_pkts, &s->used, &frames);
        prod_walk_s += now_s() - tp1;
        if (n < 0) { fprintf(stderr, "error reading pcap file: a payload exceeds KMPGPU_BATCH_BYTES\n"); exit(1); }
        if (n == 0) break;
        s->n = (uint64_t)n;
        batches++;
        if (!frames_mode) {
            payloads += s->n;
            for (uint64_t k = 0; k < s->n; k++) bytes += s->len[k];
        }
        if (!pats.n) continue;
        pthread_mutex_lock(&sh.mu);
        s->state = 1;
        if (frames_mode) sh.next_walk++;             /* on to the staging thread, which hands it to the consumers */
        else sh.next_fill++;
        pthread_cond_broadcast(&sh.cv);
        pthread_mutex_unlock(&sh.mu);
    }
    pthread_mutex_lock(&sh.mu);
    if (staging) sh.walked_all = 1;                  /* the staging thread announces the end once it has caught up */
    else sh.done = 1;
    pthread_cond_broadcast(&sh.cv);
    pthread_mutex_unlock(&sh.mu);
    if (staging) pthread_join(stage_th, NULL);

    uint64_t *counts = (uint64_t *)calloc(pats.n ? pats.n : 1, sizeof(uint64_t));
    int reduce_rccl = 0;
    if (pats.n) {
        kmpgpu_ctx **tot = (kmpgpu_ctx **)calloc((size_t)shards, sizeof *tot);
        for (int r = 0; r < n_cons; r++) {
            pthread_join(th[r], NULL);
            if (frames_mode) { payloads += cons[r].payloads; bytes += cons[r].bytes; }              /* what the GPUs extracted */
        }
        for (int r = 0; r < shards; r++) {
            /* a shard's contexts' running totals are merged on the device (openmp_task.c:172-175); the shard's counters stay
             * there for the reduce over the shards */
            for (int k = 1; k < per_shard; k++)
                if (kmpgpu_counts_add(cons[per_shard * r].total, cons[per_shard * r + k].total)) die_gpu("kmpgpu_counts_add");
            if (kmpgpu_sync(cons[per_shard * r].total)) die_gpu("kmpgpu_sync");
            tot[r] = cons[per_shard * r].total;
        }
        /* The sum over the shards (mpi_dumping.c:202): one shard per device -> RCCL all-reduce of the device counters and
         * one download; shards that share a device -> host sum.  KMPGPU_RCCL=0 / 1 as in bin/openmp_data. */
        const char *rccl_env = getenv("KMPGPU_RCCL");
        kmpgpu_comm *comm = NULL;
        /* every shard's own totals first: what the host sums when there is no communicator or the all-reduce fails */
        uint64_t *part = (uint64_t *)calloc((size_t)shards * pats.n, sizeof(uint64_t));
        for (int r = 0; r < shards; r++)
            if (kmpgpu_counts_read(tot[r], part + (size_t)r * pats.n)) die_gpu("kmpgpu_counts_read");
        if (shards <= ndev && (shards > 1 || (rccl_env && rccl_env[0] == '1')) && !(rccl_env && rccl_env[0] == '0')) {
            int bad = kmpgpu_comm_init(&comm, tot, shards) != 0;
            if (!bad) bad = kmpgpu_comm_allreduce_counts(comm) != 0;
            if (!bad) bad = kmpgpu_counts_read(tot[0], counts) != 0;
            for (int r = 1; r < shards && !bad; r++) bad = kmpgpu_sync(tot[r]) != 0;
            if (bad) fprintf(stderr, "[kmpgpu] RCCL count reduce: %s -- summing the shards' counts on the host\n", kmpgpu_last_error());
            else reduce_rccl = 1;
            if (comm) kmpgpu_comm_destroy(comm);
        }
        if (!reduce_rccl) {
            for (uint32_t i = 0; i < pats.n; i++) {
                counts[i] = 0;
                for (int r = 0; r < shards; r++) counts[i] += part[(size_t)r * pats.n + i];          /* mpi_dumping.c:202 MPI_SUM */
            }
        }
        free(part);
        free(tot);
    }
    const double t_finish = now_s();                                                                /* openmp_task.c:188 */

    kmp_report(stdout, &pats, counts, t_finish - t_start);                                         /* openmp_task.c:190-196 */
    fprintf(stderr, "[kmpgpu] streamed %llu frames, %llu payloads, %llu payload bytes in %llu batch(es) of <= %llu MiB (%s) over %d shard(s), count reduce: %s: %.3f s, %.2f GB/s end to end\n",
            (unsigned long long)frames, (unsigned long long)payloads, (unsigned long long)bytes, (unsigned long long)batches,
            (unsigned long long)(batch_bytes >> 20), frames_mode ? "raw frames, extraction on the GPU" : "payloads extracted on the host", shards, reduce_rccl ? "RCCL all-reduce" : (shards > 1 ? "host sum" : "none"), t_finish - t_start,
            (double)bytes / (t_finish - t_start) / 1e9);

    { const char *st = getenv("KMPGPU_STATS");
      if (st && st[0] && st[0] != '0') {
          double load_s = 0, wait_s = 0, h2d = 0;
          for (int r = 0; r < (pats.n ? n_cons : 0); r++) { load_s += cons[r].load_s; wait_s += cons[r].wait_s; h2d += cons[r].h2d_ms; }
          fprintf(stderr, "[kmpgpu] phases: producer %.3f s building batches + %.3f s waiting for a free slot; staging copies %.3f s; consumers %.3f s in the load calls "
                          "(uploads by the events: %.3f s), %.3f s waiting for a batch\n",
                  prod_walk_s, prod_wait_s, sh.copy_s, load_s, h2d * 1e-3, wait_s);
      } }
    /* teardown, after the clock has stopped (openmp_task.c:188 takes the time before it frees anything) */
    if (pats.n) for (int r = 0; r < n_cons; r++) { kmpgpu_destroy(cons[r].total); if (cons[r].extra) kmpgpu_destroy(cons[r].extra); }
    kmp_batch_close(rd);
    for (int i = 0; i < sh.n_slots; i++) {
        kmpgpu_host_free(sh.slots[i].arena); kmpgpu_host_free(sh.slots[i].off); kmpgpu_host_free(sh.slots[i].len);
    }
    return 0;
}
