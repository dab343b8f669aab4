/*
 * kmphost.c -- host side of the matcher (include/kmphost.h): pcap savefile reader, payload
 * extraction, pattern loader, arena builder, report.  Plain C, no GPU dependency.  Citations are
 * relative to the reference repository.
 */
#define _GNU_SOURCE
#include "kmphost.h"

#include <errno.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================ pcap savefile reader =====================================
 * What the reference obtains from libpcap: pcap_open_offline (serial.c:91), pcap_next_ex
 * (serial.c:115).  Classic pcap: 24-byte global header {magic, ver_major, ver_minor, thiszone,
 * sigfigs, snaplen, linktype}, then records {ts_sec, ts_frac, caplen, len} + caplen bytes.
 *
 * The file is mapped, not read: records are handed out (and copied from, by several threads) in
 * place.  One walker serves the record reader, the arena builder, the frame index of the on-device
 * extraction and the batch producer, so that all of them end at the same record of a damaged file. */
typedef struct kmp_view {
    const uint8_t *base;
    uint64_t       size;
    int            mapped;       /* 1: munmap, 0: free (fallback for what cannot be mapped, e.g. a pipe) */
} kmp_view;

static int host_threads(void);

/* A large mapping is walked record by record right after it is made; taking its ~250 000 page faults per GB on
 * one thread costs more than the walk itself, so several threads touch the pages first. */
static void view_prefault(const kmp_view *v)
{
    if (v->size < (64u << 20)) return;
    const int nt = host_threads();
    (void)nt;
    const int64_t pages = (int64_t)((v->size + 4095u) >> 12);
    uint64_t sink = 0;
#pragma omp parallel for num_threads(nt) schedule(static) reduction(+ : sink)
    for (int64_t p = 0; p < pages; p++) sink += v->base[(uint64_t)p << 12];
    __asm__ volatile("" :: "r"(sink));
}

static int view_open(const char *path, kmp_view *v, char errbuf[KMP_PCAP_ERRBUF])
{
    memset(v, 0, sizeof *v);
    if (errbuf) errbuf[0] = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "%s: %s", path, strerror(errno));
        return KMPHOST_EIO;
    }
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
        v->size = (uint64_t)st.st_size;
        if (v->size) {
            void *m = mmap(NULL, (size_t)v->size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)v->size, MADV_WILLNEED);
                v->base = (const uint8_t *)m; v->mapped = 1;
                close(fd);
                view_prefault(v);
                return KMPHOST_OK;
            }
        } else { close(fd); return KMPHOST_OK; }
    }
    /* not mappable: read it */
    size_t cap = 1u << 20, n = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    for (;;) {
        if (!buf) { close(fd); if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory"); return KMPHOST_ENOMEM; }
        const ssize_t got = read(fd, buf + n, cap - n);
        if (got < 0) { if (errno == EINTR) continue; free(buf); close(fd); if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "%s: %s", path, strerror(errno)); return KMPHOST_EIO; }
        if (got == 0) break;
        n += (size_t)got;
        if (n == cap) { cap *= 2; uint8_t *nb = (uint8_t *)realloc(buf, cap); if (!nb) free(buf); buf = nb; }
    }
    close(fd);
    v->base = buf; v->size = n; v->mapped = 0;
    return KMPHOST_OK;
}

static void view_close(kmp_view *v)
{
    if (v->base) { if (v->mapped) munmap((void *)v->base, (size_t)v->size); else free((void *)v->base); }
    memset(v, 0, sizeof *v);
}

/* Worker threads for the copy loops: the CPUs this process may use (affinity, cgroup quota), at most 16 --
 * a memcpy stream per core saturates the memory system well before that. */
static int host_threads(void)
{
    static int cached = 0;
    if (cached) return cached;
    int n = 1;
#ifdef _OPENMP
    n = omp_get_num_procs();
#endif
    FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (fp) {
        char q[32]; long period = 0;
        if (fscanf(fp, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            if (quota > 0 && (quota + period - 1) / period < n) n = (int)((quota + period - 1) / period);
        }
        fclose(fp);
    } else {
        long quota = -1, period = 0;
        FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fpd = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fpd && fscanf(fq, "%ld", &quota) == 1 && fscanf(fpd, "%ld", &period) == 1 && quota > 0 && period > 0 &&
            (quota + period - 1) / period < n) n = (int)((quota + period - 1) / period);
        if (fq) fclose(fq);
        if (fpd) fclose(fpd);
    }
    const char *e = getenv("KMPHOST_THREADS");
    if (e && atoi(e) > 0) n = atoi(e);
    if (n > 16 && !(e && atoi(e) > 0)) n = 16;
    if (n < 1) n = 1;
    cached = n;
    return n;
}

/* Fork-join over [0, n) with plain threads that exist for the call only.  The copy loops of a streamed capture run between
 * GPU uploads; an OpenMP team would keep spinning on its cores after every loop (libgomp's wait policy, fixed when the library
 * is loaded), and on a host that grants this process a CPU quota those spinning threads are taken out of the share of the
 * threads that feed the GPU. */
typedef void (*kmp_range_fn)(void *arg, int64_t lo, int64_t hi);
typedef struct kmp_range_job { kmp_range_fn fn; void *arg; int64_t lo, hi; } kmp_range_job;
static void *range_thread(void *p) { kmp_range_job *j = (kmp_range_job *)p; j->fn(j->arg, j->lo, j->hi); return NULL; }
static void run_parallel(int64_t n, int nthreads, kmp_range_fn fn, void *arg)
{
    if (nthreads > 64) nthreads = 64;
    if (nthreads < 1 || n < 2 * nthreads) nthreads = 1;
    kmp_range_job job[64];
    pthread_t th[64];
    int started[64];
    for (int t = 0; t < nthreads; t++) {
        job[t].fn = fn; job[t].arg = arg; job[t].lo = n * t / nthreads; job[t].hi = n * (t + 1) / nthreads;
        started[t] = t > 0 && pthread_create(&th[t], NULL, range_thread, &job[t]) == 0;
    }
    for (int t = 0; t < nthreads; t++) if (!started[t]) fn(arg, job[t].lo, job[t].hi);      /* thread 0, and whatever could not be started */
    for (int t = 1; t < nthreads; t++) if (started[t]) pthread_join(th[t], NULL);
}

#define PCAPNG_SHB 0x0A0D0D0Au
#define PCAPNG_BOM 0x1A2B3C4Du

static uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

typedef struct kmp_walk {
    const uint8_t *b;
    uint64_t sz, pos;
    int      swap;          /* file byte order differs from the host's */
    int      ng;            /* pcapng (libpcap's pcap_open_offline reads both formats) */
    uint32_t linktype, snaplen;
} kmp_walk;

/* The checks pcap_open_offline makes on the file header (message texts as libpcap words them). */
static int walk_init(kmp_walk *w, const uint8_t *b, uint64_t sz, char errbuf[KMP_PCAP_ERRBUF])
{
    memset(w, 0, sizeof *w);
    w->b = b; w->sz = sz;
    if (sz < 24u) {
        if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "truncated dump file; tried to read %zu file header bytes", (size_t)24);
        return KMPHOST_EIO;
    }
    const uint32_t magic = rd32(b), bom = rd32(b + 8);
    if (magic == 0xA1B2C3D4u || magic == 0xA1B23C4Du) w->swap = 0;
    else if (bswap32(magic) == 0xA1B2C3D4u || bswap32(magic) == 0xA1B23C4Du) w->swap = 1;
    else if (magic == PCAPNG_SHB && (bom == PCAPNG_BOM || bswap32(bom) == PCAPNG_BOM)) { w->ng = 1; w->swap = (bom != PCAPNG_BOM); }
    else {
        if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "unknown file format");
        return KMPHOST_EFORMAT;
    }
    if (w->ng) { w->snaplen = 0; w->linktype = 1; w->pos = 0; }      /* the block walker starts at the section header */
    else {
        w->snaplen = w->swap ? bswap32(rd32(b + 16)) : rd32(b + 16);
        w->linktype = w->swap ? bswap32(rd32(b + 20)) : rd32(b + 20);
        w->pos = 24;
    }
    return KMPHOST_OK;
}

/* One pcapng block body -> a packet?  1 and the packet, 0 for other blocks, -1 for a damaged one. */
static int pcapng_packet(kmp_walk *w, uint32_t type, uint32_t body, const uint8_t *b, uint32_t *caplen, uint32_t *len, const uint8_t **data)
{
    if ((type == 6u || type == 2u) && body >= 20u) {   /* Enhanced Packet Block / obsolete Packet Block */
        const uint32_t cl = w->swap ? bswap32(rd32(b + 12)) : rd32(b + 12), ln = w->swap ? bswap32(rd32(b + 16)) : rd32(b + 16);
        if (cl > body - 20u) return -1;
        *caplen = cl; *len = ln; *data = b + 20;
        return 1;
    }
    if (type == 3u && body >= 4u) {                    /* Simple Packet Block */
        const uint32_t ln = w->swap ? bswap32(rd32(b)) : rd32(b);
        uint32_t cl = ln;
        if (w->snaplen && cl > w->snaplen) cl = w->snaplen;
        if (cl > body - 4u) cl = body - 4u;
        *caplen = cl; *len = ln; *data = b + 4;
        return 1;
    }
    if (type == 1u && body >= 8u) {                    /* Interface Description Block: link type + snaplen of interface 0 */
        uint16_t lt;
        memcpy(&lt, b, 2);
        if (w->snaplen == 0) {
            w->linktype = w->swap ? (uint32_t)((lt >> 8) | ((lt & 0xFF) << 8)) : lt;
            w->snaplen = w->swap ? bswap32(rd32(b + 4)) : rd32(b + 4);
        }
    }
    return 0;
}

/* pcap_next_ex: 1 = packet (data points into the file), -2 = end of file, -1 = truncated / damaged record. */
static int walk_next(kmp_walk *w, uint32_t *caplen, uint32_t *len, const uint8_t **data)
{
    if (!w->ng) {
        if (w->pos >= w->sz) return -2;                 /* clean end of file */
        if (w->sz - w->pos < 16u) return -1;            /* truncated record header */
        const uint8_t *r = w->b + w->pos;
        const uint32_t cl = w->swap ? bswap32(rd32(r + 8)) : rd32(r + 8);
        const uint32_t ln = w->swap ? bswap32(rd32(r + 12)) : rd32(r + 12);
        if (cl > (64u << 20)) return -1;                /* corrupt record */
        if (w->sz - w->pos - 16u < cl) return -1;       /* truncated packet */
        *caplen = cl; *len = ln; *data = r + 16;
        w->pos += 16u + (uint64_t)cl;
        /* The walk touches 16 bytes per record, one cache line in a new place every time: it is bound by memory latency unless the
         * next headers are asked for early.  Captures are mostly runs of equal-sized records, so the headers 8 and 16 records ahead
         * are guessed from this record's size (a wrong guess costs a useless prefetch). */
        {
            const uint64_t step = 16u + (uint64_t)cl;
            if (w->pos + 16u * step < w->sz) {
                __builtin_prefetch(w->b + w->pos + 8u * step);
                __builtin_prefetch(w->b + w->pos + 16u * step);
            }
        }
        return 1;
    }
    for (;;) {
        if (w->pos >= w->sz) return -2;
        if (w->sz - w->pos < 8u) return -1;
        const uint8_t *h = w->b + w->pos;
        uint32_t type = rd32(h), total = rd32(h + 4);
        if (type == PCAPNG_SHB) {                       /* a new section may change the byte order */
            if (w->sz - w->pos < 12u) return -1;
            const uint32_t bom = rd32(h + 8);
            if (bom == PCAPNG_BOM) w->swap = 0;
            else if (bswap32(bom) == PCAPNG_BOM) w->swap = 1;
            else return -1;
            total = w->swap ? bswap32(total) : total;
            if (total < 16u || (total & 3u)) return -1;
            w->pos += total;                            /* a section header that runs past the end ends the file cleanly, as a seek does */
            w->snaplen = 0;
            continue;
        }
        if (w->swap) { type = bswap32(type); total = bswap32(total); }
        if (total < 12u || (total & 3u) || total > (64u << 20)) return -1;
        const uint32_t body = total - 12u;
        if (w->sz - w->pos - 8u < (uint64_t)body + 4u) return -1;      /* body + trailing length */
        const int r = pcapng_packet(w, type, body, h + 8, caplen, len, data);
        w->pos += total;
        if (r != 0) return r;
    }
}

struct kmp_pcap {
    kmp_view view;
    kmp_walk walk;
};

kmp_pcap *kmp_pcap_open(const char *path, char errbuf[KMP_PCAP_ERRBUF])
{
    kmp_pcap *p = (kmp_pcap *)calloc(1, sizeof *p);
    if (!p) { if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory"); return NULL; }
    if (view_open(path, &p->view, errbuf) || walk_init(&p->walk, p->view.base, p->view.size, errbuf)) {
        view_close(&p->view);
        free(p);
        return NULL;
    }
    return p;
}

int kmp_pcap_next(kmp_pcap *p, uint32_t *caplen, uint32_t *len, const uint8_t **data)
{
    return walk_next(&p->walk, caplen, len, data);
}

uint32_t kmp_pcap_linktype(const kmp_pcap *p) { return p->walk.linktype; }

void kmp_pcap_close(kmp_pcap *p)
{
    if (!p) return;
    view_close(&p->view);
    free(p);
}

/* ============================ payload extraction ======================================== */

/* packet_dumping.h:87-139 */
int kmp_extract_udp(const uint8_t *f, uint32_t cl, uint32_t *poff, uint32_t *plen)
{
    if (cl < 14u) return 0;                              /* :94  Ethernet header            */
    uint32_t rest = cl - 14u;
    if (rest < 20u) return 0;                            /* :102 minimal IP header          */
    const uint32_t ihl = (uint32_t)(f[14] & 0x0Fu) << 2; /* :108 no version / EtherType test */
    if (rest < ihl) return 0;                            /* :110                            */
    if (f[14 + 9] != 17u) return 0;                      /* :116 protocol field             */
    rest -= ihl;
    if (rest < 8u) return 0;                             /* :125 UDP header                 */
    *poff = 14u + ihl + 8u;                              /* :133 (sizeof(pointer) == 8)     */
    *plen = rest - 8u;                                   /* :136 everything that was captured */
    return 1;
}

/* packet_dumping.h:150-188; frames on which the reference's unsigned arithmetic would wrap
 * (headers longer than the captured bytes) are rejected instead of crashing. */
int kmp_extract_tcp(const uint8_t *f, uint32_t cl, uint32_t *poff, uint32_t *plen)
{
    if (cl < 15u) return 0;
    const uint32_t size_ip = (uint32_t)(f[14] & 0x0Fu) << 2;    /* :165 */
    if (size_ip < 20u) return 0;                                 /* :166 */
    const uint32_t tcp_at = 14u + size_ip;
    if (cl < tcp_at + 13u) return 0;
    const uint32_t size_tcp = (uint32_t)(f[tcp_at + 12u] >> 4) << 2;   /* :175 */
    if (size_tcp < 20u) return 0;                                /* :176 */
    const uint32_t start = tcp_at + size_tcp;
    if (cl < start) return 0;
    *poff = start;                                               /* :181 */
    *plen = cl - start;                                          /* :184 */
    return 1;
}

/* ============================ pattern list ============================================== */

static int is_c_space(uint8_t b) { return b == ' ' || (b >= '\t' && b <= '\r'); }

/* serial.c:66 fscanf(fp, "%s", str): tokens are maximal runs of non-whitespace bytes. */
int kmp_patterns_parse(const uint8_t *text, size_t n, kmp_patterns *out)
{
    memset(out, 0, sizeof *out);
    /* pass 1: count and measure */
    uint32_t cnt = 0;
    size_t bytes = 0, i = 0;
    while (i < n) {
        while (i < n && is_c_space(text[i])) i++;
        size_t s = i;
        while (i < n && !is_c_space(text[i])) i++;
        if (i > s) {
            size_t tl = i - s;
            const uint8_t *z = (const uint8_t *)memchr(text + s, 0, tl);   /* strlen() view, serial.c:69 */
            if (z) tl = (size_t)(z - (text + s));
            if (tl > KMP_MAX_PATTERN_LEN) return KMPHOST_ETOKEN;           /* would overflow char str[100] */
            if (tl == 0) continue;
            cnt++; bytes += tl + 1;
        }
    }
    out->blob = (uint8_t *)malloc(bytes ? bytes : 1);
    out->off = (uint32_t *)malloc(sizeof(uint32_t) * (cnt ? cnt : 1));
    out->len = (uint32_t *)malloc(sizeof(uint32_t) * (cnt ? cnt : 1));
    if (!out->blob || !out->off || !out->len) { kmp_patterns_free(out); return KMPHOST_ENOMEM; }
    /* pass 2: copy */
    uint32_t k = 0;
    size_t w = 0;
    i = 0;
    while (i < n) {
        while (i < n && is_c_space(text[i])) i++;
        size_t s = i;
        while (i < n && !is_c_space(text[i])) i++;
        if (i > s) {
            size_t tl = i - s;
            const uint8_t *z = (const uint8_t *)memchr(text + s, 0, tl);
            if (z) tl = (size_t)(z - (text + s));
            if (tl == 0) continue;
            out->off[k] = (uint32_t)w;
            out->len[k] = (uint32_t)tl;
            memcpy(out->blob + w, text + s, tl);
            out->blob[w + tl] = 0;
            w += tl + 1;
            k++;
        }
    }
    out->n = cnt;
    return KMPHOST_OK;
}

int kmp_patterns_load(const char *path, kmp_patterns *out)
{
    memset(out, 0, sizeof *out);
    FILE *fp = fopen(path, "rb");                       /* serial.c:59 */
    if (!fp) return KMPHOST_EIO;
    size_t cap = 1 << 16, n = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) { fclose(fp); return KMPHOST_ENOMEM; }
    for (;;) {
        size_t got = fread(buf + n, 1, cap - n, fp);
        n += got;
        if (got == 0) break;
        if (n == cap) {
            uint8_t *nb = (uint8_t *)realloc(buf, cap * 2);
            if (!nb) { free(buf); fclose(fp); return KMPHOST_ENOMEM; }
            buf = nb; cap *= 2;
        }
    }
    fclose(fp);
    int rc = kmp_patterns_parse(buf, n, out);
    free(buf);
    return rc;
}

void kmp_patterns_free(kmp_patterns *p)
{
    if (!p) return;
    free(p->blob); free(p->off); free(p->len);
    memset(p, 0, sizeof *p);
}

/* serial.c:217-238 */
void kmp_failure_table(const uint8_t *pat, uint32_t m, int32_t *prefix)
{
    if (!m) return;
    prefix[0] = 0;
    uint32_t k = 0;                                    /* length of the current border */
    for (uint32_t q = 1; q < m; q++) {
        while (k > 0 && pat[k] != pat[q]) k = (uint32_t)prefix[k - 1];
        if (pat[k] == pat[q]) k++;
        prefix[q] = (int32_t)k;
    }
}

/* ============================ arena ===================================================== */

static uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

uint64_t kmp_arena_layout(const uint32_t *lens, uint32_t fixed_len, uint64_t n, uint32_t slot_align,
                          uint64_t *off_out, uint32_t *len_out)
{
    if (slot_align < KMP_SLOT_ALIGN) slot_align = KMP_SLOT_ALIGN;
    uint64_t pos = 0;
    for (uint64_t k = 0; k < n; k++) {
        const uint32_t l = lens ? lens[k] : fixed_len;
        if (off_out) off_out[k] = pos;
        if (len_out) len_out[k] = l;
        pos += round_up(l ? l : 1, slot_align);         /* an empty payload still owns a slot */
    }
    return pos + KMP_ARENA_SLACK;
}

static int arena_alloc(kmp_arena *a, uint64_t nbytes, uint64_t n, kmp_alloc_fn alloc_fn, kmp_free_fn free_fn, int zero)
{
    memset(a, 0, sizeof *a);
    a->free_fn = alloc_fn ? free_fn : free;
    a->bytes = (uint8_t *)(alloc_fn ? alloc_fn((size_t)nbytes) : aligned_alloc(4096, (size_t)round_up(nbytes, 4096)));
    a->off = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    a->len = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    if (!a->bytes || !a->off || !a->len) { kmp_arena_free(a); return KMPHOST_ENOMEM; }
    if (zero) memset(a->bytes, 0, (size_t)nbytes);
    a->nbytes = nbytes;
    return KMPHOST_OK;
}

void kmp_arena_free(kmp_arena *a)
{
    if (!a) return;
    if (a->bytes && a->free_fn) a->free_fn(a->bytes);
    free(a->off); free(a->len);
    memset(a, 0, sizeof *a);
}

int kmp_arena_from_payloads(const uint8_t *const *payloads, const uint32_t *lens, uint64_t n,
                            kmp_alloc_fn alloc_fn, kmp_free_fn free_fn, kmp_arena *out)
{
    const uint64_t nbytes = kmp_arena_layout(lens, 0, n, KMP_SLOT_ALIGN, NULL, NULL);
    int rc = arena_alloc(out, nbytes, n, alloc_fn, free_fn, 1);
    if (rc) return rc;
    kmp_arena_layout(lens, 0, n, KMP_SLOT_ALIGN, out->off, out->len);
    for (uint64_t k = 0; k < n; k++) {
        if (lens[k]) memcpy(out->bytes + out->off[k], payloads[k], lens[k]);
        out->payload_bytes += lens[k];
    }
    out->n_pkts = n;
    out->n_frames = n;
    return KMPHOST_OK;
}

/* serial.c:115-141.  Two passes over the savefile: size the arena, then fill it, so the arena is
 * one allocation of the final size (it may be pinned memory). */
/* Every record of the mapped file: where its captured bytes lie.  Sequential (a record's position depends on
 * all the records before it), but it touches 16 bytes per record only. */
typedef struct kmp_index { uint64_t *off; uint32_t *caplen; uint64_t n, cap; } kmp_index;

static int index_push(kmp_index *ix, uint64_t off, uint32_t cl)
{
    if (ix->n == ix->cap) {
        const uint64_t nc = ix->cap ? ix->cap * 2 : 1u << 16;
        uint64_t *no = (uint64_t *)realloc(ix->off, sizeof(uint64_t) * (size_t)nc);
        if (no) ix->off = no;
        uint32_t *nl = (uint32_t *)realloc(ix->caplen, sizeof(uint32_t) * (size_t)nc);
        if (nl) ix->caplen = nl;
        if (!no || !nl) return KMPHOST_ENOMEM;
        ix->cap = nc;
    }
    ix->off[ix->n] = off; ix->caplen[ix->n] = cl; ix->n++;
    return KMPHOST_OK;
}

static int index_file(const kmp_view *v, kmp_index *ix, char errbuf[KMP_PCAP_ERRBUF])
{
    memset(ix, 0, sizeof *ix);
    kmp_walk w;
    int rc = walk_init(&w, v->base, v->size, errbuf);
    if (rc) return rc;
    uint32_t cl, ln;
    const uint8_t *data;
    while (walk_next(&w, &cl, &ln, &data) >= 0)                       /* serial.c:115: -1 and -2 both end the loop */
        if ((rc = index_push(ix, (uint64_t)(data - v->base), cl)) != 0) {
            free(ix->off); free(ix->caplen); memset(ix, 0, sizeof *ix);
            if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory");
            return rc;
        }
    return KMPHOST_OK;
}

int kmp_arena_from_pcap(const char *path, int proto, kmp_alloc_fn alloc_fn, kmp_free_fn free_fn,
                        kmp_arena *out, char errbuf[KMP_PCAP_ERRBUF])
{
    memset(out, 0, sizeof *out);
    kmp_view v;
    kmp_index ix;
    int rc = view_open(path, &v, errbuf);
    if (rc) return rc;
    if ((rc = index_file(&v, &ix, errbuf)) != 0) { view_close(&v); return rc; }
    const int64_t nf = (int64_t)ix.n;
    const int nt = host_threads();
    (void)nt;
    /* the extraction rule for every frame (serial.c:119-122 / openmp_data.c:128-147 do this per packet too) */
    uint32_t *po = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(nf ? nf : 1));
    uint32_t *pl = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(nf ? nf : 1));
    uint64_t *dst = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(nf ? nf : 1));
    if (!po || !pl || !dst) rc = KMPHOST_ENOMEM;
    if (!rc) {
#pragma omp parallel for num_threads(nt) schedule(static)
        for (int64_t f = 0; f < nf; f++) {
            uint32_t o = 0, l = 0;
            const int ok = (proto == KMP_PROTO_TCP) ? kmp_extract_tcp(v.base + ix.off[f], ix.caplen[f], &o, &l)
                                                    : kmp_extract_udp(v.base + ix.off[f], ix.caplen[f], &o, &l);
            po[f] = o; pl[f] = ok ? l : UINT32_MAX;                 /* UINT32_MAX: skipped (serial.c:138-140) */
        }
        uint64_t n = 0, pos = 0, payload = 0;
        for (int64_t f = 0; f < nf; f++) {
            if (pl[f] == UINT32_MAX) continue;
            dst[f] = pos;
            pos += round_up(pl[f] ? pl[f] : 1, KMP_SLOT_ALIGN);
            payload += pl[f];
            n++;
        }
        rc = arena_alloc(out, pos + KMP_ARENA_SLACK, n, alloc_fn, free_fn, 0);
        if (!rc) {
            uint64_t k = 0;
            for (int64_t f = 0; f < nf; f++) {
                if (pl[f] == UINT32_MAX) continue;
                out->off[k] = dst[f]; out->len[k] = pl[f]; k++;
            }
            /* payload copies (serial.c:125-127) + zeroed slot padding, one stream per thread */
#pragma omp parallel for num_threads(nt) schedule(static)
            for (int64_t f = 0; f < nf; f++) {
                if (pl[f] == UINT32_MAX) continue;
                const uint64_t slot = round_up(pl[f] ? pl[f] : 1, KMP_SLOT_ALIGN);
                uint8_t *d = out->bytes + dst[f];
                if (pl[f]) memcpy(d, v.base + ix.off[f] + po[f], pl[f]);
                if (slot > pl[f]) memset(d + pl[f], 0, (size_t)(slot - pl[f]));
            }
            memset(out->bytes + pos, 0, KMP_ARENA_SLACK);
            out->n_pkts = n;
            out->n_frames = (uint64_t)nf;
            out->payload_bytes = payload;
        }
    }
    if (rc == KMPHOST_ENOMEM && errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory");
    free(po); free(pl); free(dst);
    free(ix.off); free(ix.caplen);
    view_close(&v);
    return rc;
}

/* ============================ raw frames (on-device extraction) ========================= */

int kmp_frames_from_pcap(const char *path, kmp_alloc_fn alloc_fn, kmp_free_fn free_fn, kmp_frames *out,
                         char errbuf[KMP_PCAP_ERRBUF])
{
    memset(out, 0, sizeof *out);
    kmp_view v;
    kmp_index ix;
    int rc = view_open(path, &v, errbuf);
    if (rc) return rc;
    if ((rc = index_file(&v, &ix, errbuf)) != 0) { view_close(&v); return rc; }
    if (!alloc_fn && v.mapped) {                        /* hand the mapping itself out */
        out->bytes = (uint8_t *)v.base; out->nbytes = v.size; out->map_len = v.size; out->free_fn = NULL;
        out->off = ix.off; out->caplen = ix.caplen; out->n = ix.n;
        if (!out->off) {
            out->off = (uint64_t *)malloc(sizeof(uint64_t)); out->caplen = (uint32_t *)malloc(sizeof(uint32_t));
            if (!out->off || !out->caplen) { kmp_frames_free(out); return KMPHOST_ENOMEM; }
        }
        return KMPHOST_OK;
    }
    out->free_fn = alloc_fn ? free_fn : free;
    out->nbytes = v.size;
    out->bytes = (uint8_t *)(alloc_fn ? alloc_fn((size_t)v.size + 64) : malloc((size_t)v.size + 64));
    if (!out->bytes) {
        free(ix.off); free(ix.caplen); view_close(&v);
        if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory");
        return KMPHOST_ENOMEM;
    }
    /* the file as it is, copied by several threads (the destination is usually pinned memory) */
    const int nt = host_threads();
    (void)nt;
    const uint64_t piece = 4u << 20;
    const int64_t pieces = (int64_t)((v.size + piece - 1) / piece);
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t i = 0; i < pieces; i++) {
        const uint64_t o = (uint64_t)i * piece, l = (v.size - o < piece) ? v.size - o : piece;
        memcpy(out->bytes + o, v.base + o, (size_t)l);
    }
    memset(out->bytes + v.size, 0, 64);
    out->off = ix.off; out->caplen = ix.caplen; out->n = ix.n;
    if (!out->off) {                                     /* no record: keep the arrays non-NULL */
        out->off = (uint64_t *)malloc(sizeof(uint64_t)); out->caplen = (uint32_t *)malloc(sizeof(uint32_t));
        if (!out->off || !out->caplen) { kmp_frames_free(out); view_close(&v); return KMPHOST_ENOMEM; }
    }
    view_close(&v);
    return KMPHOST_OK;
}

void kmp_frames_free(kmp_frames *f)
{
    if (!f) return;
    if (f->bytes && f->map_len) munmap(f->bytes, (size_t)f->map_len);
    else if (f->bytes && f->free_fn) f->free_fn(f->bytes);
    free(f->off); free(f->caplen);
    memset(f, 0, sizeof *f);
}

/* ============================ streamed capture: batches =================================
 * The producer half of openmp_task.c:126-155. */
struct kmp_batch_reader {
    kmp_view  view;
    kmp_walk  walk;
    int       proto;
    int       pending;          /* a payload has been found but did not fit into the previous batch */
    uint64_t  p_src;            /* its bytes in the file, its length */
    uint32_t  p_len;
    int       eof;
    uint64_t *src;              /* scratch: source offset of every payload of the batch being built */
    uint64_t  src_cap;
};

kmp_batch_reader *kmp_batch_open(const char *path, int proto, char errbuf[KMP_PCAP_ERRBUF])
{
    kmp_batch_reader *r = (kmp_batch_reader *)calloc(1, sizeof *r);
    if (!r) { if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory"); return NULL; }
    if (view_open(path, &r->view, errbuf) || walk_init(&r->walk, r->view.base, r->view.size, errbuf)) {
        view_close(&r->view);
        free(r);
        return NULL;
    }
    r->proto = proto;
    return r;
}

typedef struct batch_copy_job { uint8_t *arena; const uint8_t *file; const uint64_t *src, *off; const uint32_t *len; } batch_copy_job;
static void batch_copy_range(void *arg, int64_t lo, int64_t hi)
{
    const batch_copy_job *j = (const batch_copy_job *)arg;
    for (int64_t k = lo; k < hi; k++) {
        const uint64_t slot = round_up(j->len[k] ? j->len[k] : 1, KMP_SLOT_ALIGN);
        uint8_t *d = j->arena + j->off[k];
        if (j->len[k]) memcpy(d, j->file + j->src[k], j->len[k]);
        if (slot > j->len[k]) memset(d + j->len[k], 0, (size_t)(slot - j->len[k]));
    }
}

int64_t kmp_batch_next(kmp_batch_reader *r, uint8_t *arena, uint64_t cap_bytes, uint64_t *off, uint32_t *len,
                       uint64_t cap_pkts, uint64_t *used_bytes, uint64_t *frames)
{
    uint64_t n = 0, pos = 0;
    if (cap_bytes < KMP_ARENA_SLACK + KMP_SLOT_ALIGN) return KMPHOST_EINVAL;
    const uint64_t room = cap_bytes - KMP_ARENA_SLACK;
    /* 1: walk the records and lay the batch out (sequential: headers only) */
    while (!r->eof) {
        if (!r->pending) {
            uint32_t cl, ln, po, pl;
            const uint8_t *data;
            if (walk_next(&r->walk, &cl, &ln, &data) < 0) { r->eof = 1; break; }          /* openmp_task.c:135 */
            if (frames) (*frames)++;
            const int ok = (r->proto == KMP_PROTO_TCP) ? kmp_extract_tcp(data, cl, &po, &pl)
                                                       : kmp_extract_udp(data, cl, &po, &pl);   /* openmp_task.c:139-142 */
            if (!ok) continue;                          /* invalid frames cannot match anything (openmp_task.c:150-153 stores " ") */
            r->p_src = (uint64_t)(data - r->view.base) + po; r->p_len = pl; r->pending = 1;
        }
        const uint64_t slot = round_up(r->p_len ? r->p_len : 1, KMP_SLOT_ALIGN);
        if (slot > room) return KMPHOST_EINVAL;
        if (pos + slot > room || n == cap_pkts) break;                /* keep the payload for the next batch */
        if (n == r->src_cap) {
            const uint64_t nc = r->src_cap ? r->src_cap * 2 : 1u << 16;
            uint64_t *ns = (uint64_t *)realloc(r->src, sizeof(uint64_t) * (size_t)nc);
            if (!ns) return KMPHOST_ENOMEM;
            r->src = ns; r->src_cap = nc;
        }
        r->src[n] = r->p_src;
        off[n] = pos; len[n] = r->p_len;
        pos += slot; n++;
        r->pending = 0;
    }
    /* 2: copy the payloads, several threads (the arena is usually pinned memory) */
    batch_copy_job bj = {arena, r->view.base, r->src, off, len};
    run_parallel((int64_t)n, host_threads(), batch_copy_range, &bj);
    if (n) memset(arena + pos, 0, KMP_ARENA_SLACK);
    if (used_bytes) *used_bytes = n ? pos + KMP_ARENA_SLACK : 0;
    return (int64_t)n;
}

/* openmp_task.c:130-137 with the extraction left to the device: record headers only. */
int64_t kmp_batch_next_frames(kmp_batch_reader *r, uint64_t max_span_bytes, uint64_t *frame_off, uint32_t *frame_caplen,
                              uint64_t cap_frames)
{
    if (!r || !frame_off || !frame_caplen || cap_frames == 0) return KMPHOST_EINVAL;
    uint64_t n = 0, first = 0;
    while (!r->eof && n < cap_frames) {
        const uint64_t before = r->walk.pos;
        uint32_t cl, ln;
        const uint8_t *data;
        if (walk_next(&r->walk, &cl, &ln, &data) < 0) { r->eof = 1; break; }                  /* openmp_task.c:135 */
        const uint64_t o = (uint64_t)(data - r->view.base);
        if (n == 0) first = o;
        else if (o + cl - first > max_span_bytes) { r->walk.pos = before; break; }           /* this record opens the next batch */
        frame_off[n] = o; frame_caplen[n] = cl; n++;
    }
    return (int64_t)n;
}

const uint8_t *kmp_batch_file(const kmp_batch_reader *r, uint64_t *nbytes)
{
    if (nbytes) *nbytes = r ? r->view.size : 0;
    return r ? r->view.base : NULL;
}

/* One piece of kmp_copy_bytes.  The destination is a pinned staging buffer that the GPU's copy engine reads next and the CPU never
 * reads again: streaming (non-temporal) stores write it past the caches -- no read-for-ownership of the destination lines, and the
 * DMA that follows finds the bytes in DRAM instead of probing dirty lines out of sixteen cores' caches. */
static void copy_streaming(uint8_t *dst, const uint8_t *src, uint64_t n)
{
#if defined(__SSE2__)
    uint64_t head = (16u - ((uintptr_t)dst & 15u)) & 15u;
    if (head > n) head = n;
    if (head) { memcpy(dst, src, (size_t)head); dst += head; src += head; n -= head; }
    const uint64_t body = n & ~(uint64_t)63;
    for (uint64_t o = 0; o < body; o += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(src + o)), b = _mm_loadu_si128((const __m128i *)(src + o + 16));
        const __m128i c = _mm_loadu_si128((const __m128i *)(src + o + 32)), d = _mm_loadu_si128((const __m128i *)(src + o + 48));
        _mm_stream_si128((__m128i *)(dst + o), a); _mm_stream_si128((__m128i *)(dst + o + 16), b);
        _mm_stream_si128((__m128i *)(dst + o + 32), c); _mm_stream_si128((__m128i *)(dst + o + 48), d);
    }
    if (n > body) memcpy(dst + body, src + body, (size_t)(n - body));
    _mm_sfence();
#else
    memcpy(dst, src, (size_t)n);
#endif
}

typedef struct bytes_copy_job { uint8_t *dst; const uint8_t *src; uint64_t n, piece; } bytes_copy_job;
static void bytes_copy_range(void *arg, int64_t lo, int64_t hi)
{
    const bytes_copy_job *j = (const bytes_copy_job *)arg;
    for (int64_t i = lo; i < hi; i++) {
        const uint64_t o = (uint64_t)i * j->piece, l = (j->n - o < j->piece) ? j->n - o : j->piece;
        copy_streaming(j->dst + o, j->src + o, l);
    }
}

void kmp_copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t n)
{
    const uint64_t piece = 1u << 20;
    bytes_copy_job bj = {dst, src, n, piece};
    run_parallel((int64_t)((n + piece - 1) / piece), host_threads(), bytes_copy_range, &bj);
}

void kmp_batch_close(kmp_batch_reader *r)
{
    if (!r) return;
    view_close(&r->view);
    free(r->src);
    free(r);
}

/* ============================ synthetic payloads ======================================== */

void kmp_synth_fill_host(uint8_t *arena, const uint64_t *off, const uint32_t *len, uint64_t first_pkt_id,
                         uint64_t n, const kmp_synth_params *sp, int threads)
{
    if (threads < 1) threads = 1;
    (void)threads;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const uint64_t id = first_pkt_id + (uint64_t)i;
        const uint32_t L = len[i];
        const uint32_t key = kmp_synth_pkt_key(sp->seed, id);
        uint32_t pos = 0;
        const int planted = kmp_synth_plant(sp, id, L, &pos);
        uint32_t *slot = (uint32_t *)(arena + off[i]);
        const uint32_t nw = ((L + 15u) & ~15u) / 4u;
        for (uint32_t w = 0; w < nw; w++) slot[w] = kmp_synth_slot_word(sp, key, w, L, planted, pos);
    }
}

uint64_t kmp_synth_count_planted(const uint32_t *len, uint32_t fixed_len, uint64_t first_pkt_id, uint64_t n,
                                 const kmp_synth_params *sp)
{
    uint64_t c = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t pos;
        c += (uint64_t)kmp_synth_plant(sp, first_pkt_id + i, len ? len[i] : fixed_len, &pos);
    }
    return c;
}

/* ============================ report ==================================================== */

/* serial.c:163-169 (the misspelling is the reference's). */
void kmp_report(FILE *fp, const kmp_patterns *pats, const uint64_t *counts, double elapsed_seconds)
{
    fprintf(fp, "Printing the number of appereances of each string throughout the entire pcap file:\n");
    for (uint32_t i = 0; i < pats->n; i++)
        if (counts[i] != 0) {
            /* The reference counts in int (serial.c:101) and prints %d (serial.c:166): beyond INT_MAX its own counter
             * has overflowed (undefined).  The line keeps the %d form -- the low 32 bits, as a wrapped int prints --
             * and the exact 64-bit count goes to stderr. */
            if (counts[i] > 2147483647ull)
                fprintf(stderr, "[kmphost] warning: %s matched %llu times, more than an int holds; the report line shows the wrapped value the reference's int counter would print\n",
                        (const char *)(pats->blob + pats->off[i]), (unsigned long long)counts[i]);
            fprintf(fp, "%s: %d times!\n", (const char *)(pats->blob + pats->off[i]), (int)counts[i]);
        }
    fprintf(fp, "Elapsed time = %f seconds\n", elapsed_seconds);
}

/* ============================ pcap writer (tooling) ===================================== */

int kmp_write_udp_pcap(const char *path, const uint8_t *arena, const uint64_t *off, const uint32_t *len, uint64_t n)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return KMPHOST_EIO;
    const uint32_t gh[6] = {0xA1B2C3D4u, 0x00040002u, 0, 0, 262144u, 1u};   /* v2.4, Ethernet */
    fwrite(gh, sizeof gh, 1, fp);
    uint8_t hdr[42];
    memset(hdr, 0, sizeof hdr);
    for (int i = 0; i < 12; i++) hdr[i] = (uint8_t)(i + 1);
    hdr[12] = 0x08; hdr[13] = 0x00;             /* IPv4 */
    hdr[14] = 0x45;                             /* version 4, IHL 5 */
    hdr[22] = 64;                               /* TTL */
    hdr[23] = 17;                               /* UDP */
    for (uint64_t k = 0; k < n; k++) {
        const uint32_t L = len[k], tot = 42u + L;
        const uint32_t rh[4] = {(uint32_t)(k / 1000000u), (uint32_t)(k % 1000000u), tot, tot};
        hdr[16] = (uint8_t)((28u + L) >> 8); hdr[17] = (uint8_t)(28u + L);
        hdr[38] = (uint8_t)((8u + L) >> 8); hdr[39] = (uint8_t)(8u + L);
        if (fwrite(rh, sizeof rh, 1, fp) != 1 || fwrite(hdr, sizeof hdr, 1, fp) != 1 ||
            (L && fwrite(arena + off[k], L, 1, fp) != 1)) {
            fclose(fp);
            return KMPHOST_EIO;
        }
    }
    return fclose(fp) ? KMPHOST_EIO : KMPHOST_OK;
}
