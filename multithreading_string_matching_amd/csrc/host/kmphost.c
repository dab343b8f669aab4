/*
 * kmphost.c -- host side of the matcher (include/kmphost.h): pcap savefile reader, payload
 * extraction, pattern loader, arena builder, report.  Plain C, no GPU dependency.  Citations are
 * relative to the reference repository.
 */
#define _GNU_SOURCE
#include "kmphost.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================ pcap savefile reader =====================================
 * What the reference obtains from libpcap: pcap_open_offline (serial.c:91), pcap_next_ex
 * (serial.c:115).  Classic pcap: 24-byte global header {magic, ver_major, ver_minor, thiszone,
 * sigfigs, snaplen, linktype}, then records {ts_sec, ts_frac, caplen, len} + caplen bytes. */
struct kmp_pcap {
    FILE    *fp;
    int      swap;          /* file byte order differs from the host's */
    int      ng;            /* pcapng (libpcap's pcap_open_offline reads both formats) */
    uint32_t linktype;
    uint32_t snaplen;
    uint8_t *buf;
    size_t   cap;
};

#define PCAPNG_SHB 0x0A0D0D0Au
#define PCAPNG_BOM 0x1A2B3C4Du

static uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

kmp_pcap *kmp_pcap_open(const char *path, char errbuf[KMP_PCAP_ERRBUF])
{
    if (errbuf) errbuf[0] = 0;
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "%s: %s", path, strerror(errno));
        return NULL;
    }
    uint32_t h[6];
    if (fread(h, 1, sizeof h, fp) != sizeof h) {
        if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "truncated dump file; tried to read %zu file header bytes", sizeof h);
        fclose(fp);
        return NULL;
    }
    int swap, ng = 0;
    if (h[0] == 0xA1B2C3D4u || h[0] == 0xA1B23C4Du) swap = 0;
    else if (bswap32(h[0]) == 0xA1B2C3D4u || bswap32(h[0]) == 0xA1B23C4Du) swap = 1;
    else if (h[0] == PCAPNG_SHB && (h[2] == PCAPNG_BOM || bswap32(h[2]) == PCAPNG_BOM)) { ng = 1; swap = (h[2] != PCAPNG_BOM); }
    else {
        if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "unknown file format");
        fclose(fp);
        return NULL;
    }
    kmp_pcap *p = (kmp_pcap *)calloc(1, sizeof *p);
    if (!p) { fclose(fp); if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory"); return NULL; }
    p->fp = fp;
    p->swap = swap;
    p->ng = ng;
    if (ng) {
        p->snaplen = 0; p->linktype = 1;
        fseek(fp, 0, SEEK_SET);                     /* the block walker starts at the section header */
    } else {
        p->snaplen = swap ? bswap32(h[4]) : h[4];
        p->linktype = swap ? bswap32(h[5]) : h[5];
    }
    return p;
}

/* One pcapng block body in p->buf -> a packet?  Returns 1 and the packet, or 0 for other blocks. */
static int pcapng_packet(kmp_pcap *p, uint32_t type, uint32_t body, uint32_t *caplen, uint32_t *len, const uint8_t **data)
{
    const uint8_t *b = p->buf;
    uint32_t w[5];
    if (type == 6u && body >= 20u) {                 /* Enhanced Packet Block */
        memcpy(w, b, 20);
        const uint32_t cl = p->swap ? bswap32(w[3]) : w[3], ln = p->swap ? bswap32(w[4]) : w[4];
        if (cl > body - 20u) return -1;
        *caplen = cl; *len = ln; *data = b + 20;
        return 1;
    }
    if (type == 2u && body >= 20u) {                 /* obsolete Packet Block */
        memcpy(w, b, 20);
        const uint32_t cl = p->swap ? bswap32(w[3]) : w[3], ln = p->swap ? bswap32(w[4]) : w[4];
        if (cl > body - 20u) return -1;
        *caplen = cl; *len = ln; *data = b + 20;
        return 1;
    }
    if (type == 3u && body >= 4u) {                  /* Simple Packet Block */
        memcpy(w, b, 4);
        const uint32_t ln = p->swap ? bswap32(w[0]) : w[0];
        uint32_t cl = ln;
        if (p->snaplen && cl > p->snaplen) cl = p->snaplen;
        if (cl > body - 4u) cl = body - 4u;
        *caplen = cl; *len = ln; *data = b + 4;
        return 1;
    }
    if (type == 1u && body >= 8u) {                  /* Interface Description Block: link type + snaplen of interface 0 */
        uint16_t lt;
        memcpy(&lt, b, 2);
        memcpy(w, b + 4, 4);
        if (p->snaplen == 0) {
            p->linktype = p->swap ? (uint32_t)((lt >> 8) | ((lt & 0xFF) << 8)) : lt;
            p->snaplen = p->swap ? bswap32(w[0]) : w[0];
        }
    }
    return 0;
}

static int pcapng_next(kmp_pcap *p, uint32_t *caplen, uint32_t *len, const uint8_t **data)
{
    for (;;) {
        uint32_t hd[2];
        size_t got = fread(hd, 1, sizeof hd, p->fp);
        if (got == 0) return -2;
        if (got != sizeof hd) return -1;
        uint32_t type = hd[0], total = hd[1];
        if (type == PCAPNG_SHB) {                    /* a new section may change the byte order */
            uint32_t bom;
            if (fread(&bom, 1, 4, p->fp) != 4) return -1;
            if (bom == PCAPNG_BOM) p->swap = 0;
            else if (bswap32(bom) == PCAPNG_BOM) p->swap = 1;
            else return -1;
            total = p->swap ? bswap32(total) : total;
            if (total < 16u || (total & 3u) || fseek(p->fp, (long)total - 12, SEEK_CUR) != 0) return -1;
            p->snaplen = 0;
            continue;
        }
        if (p->swap) { type = bswap32(type); total = bswap32(total); }
        if (total < 12u || (total & 3u) || total > (64u << 20)) return -1;
        const uint32_t body = total - 12u;
        if (body + 4u > p->cap) {
            uint8_t *nb = (uint8_t *)realloc(p->buf, (size_t)body + 4096u);
            if (!nb) return -1;
            p->buf = nb; p->cap = (size_t)body + 4096u;
        }
        if (fread(p->buf, 1, (size_t)body + 4u, p->fp) != (size_t)body + 4u) return -1;    /* body + trailing length */
        const int r = pcapng_packet(p, type, body, caplen, len, data);
        if (r != 0) return r;
    }
}

int kmp_pcap_next(kmp_pcap *p, uint32_t *caplen, uint32_t *len, const uint8_t **data)
{
    if (p->ng) return pcapng_next(p, caplen, len, data);
    uint32_t r[4];
    size_t got = fread(r, 1, sizeof r, p->fp);
    if (got == 0) return -2;                    /* clean end of file */
    if (got != sizeof r) return -1;             /* truncated record header */
    uint32_t cl = p->swap ? bswap32(r[2]) : r[2];
    uint32_t ln = p->swap ? bswap32(r[3]) : r[3];
    if (cl > (64u << 20)) return -1;            /* corrupt record */
    if (cl > p->cap) {
        size_t nc = cl + 4096u;
        uint8_t *nb = (uint8_t *)realloc(p->buf, nc);
        if (!nb) return -1;
        p->buf = nb; p->cap = nc;
    }
    if (cl && fread(p->buf, 1, cl, p->fp) != cl) return -1;   /* truncated packet */
    *caplen = cl; *len = ln; *data = p->buf;
    return 1;
}

uint32_t kmp_pcap_linktype(const kmp_pcap *p) { return p->linktype; }

void kmp_pcap_close(kmp_pcap *p)
{
    if (!p) return;
    if (p->fp) fclose(p->fp);
    free(p->buf);
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

static int arena_alloc(kmp_arena *a, uint64_t nbytes, uint64_t n, kmp_alloc_fn alloc_fn, kmp_free_fn free_fn)
{
    memset(a, 0, sizeof *a);
    a->free_fn = alloc_fn ? free_fn : free;
    a->bytes = (uint8_t *)(alloc_fn ? alloc_fn((size_t)nbytes) : aligned_alloc(4096, (size_t)round_up(nbytes, 4096)));
    a->off = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    a->len = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    if (!a->bytes || !a->off || !a->len) { kmp_arena_free(a); return KMPHOST_ENOMEM; }
    memset(a->bytes, 0, (size_t)nbytes);
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
    int rc = arena_alloc(out, nbytes, n, alloc_fn, free_fn);
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
int kmp_arena_from_pcap(const char *path, int proto, kmp_alloc_fn alloc_fn, kmp_free_fn free_fn,
                        kmp_arena *out, char errbuf[KMP_PCAP_ERRBUF])
{
    memset(out, 0, sizeof *out);
    uint64_t n = 0, frames = 0, bytes = 0;
    for (int pass = 0; pass < 2; pass++) {
        kmp_pcap *p = kmp_pcap_open(path, errbuf);
        if (!p) return errbuf && !strcmp(errbuf, "unknown file format") ? KMPHOST_EFORMAT : KMPHOST_EIO;
        uint32_t cl, ln;
        const uint8_t *data;
        uint64_t k = 0, pos = 0;
        while (kmp_pcap_next(p, &cl, &ln, &data) >= 0) {            /* serial.c:115 */
            uint32_t po, pl;
            const int ok = (proto == KMP_PROTO_TCP) ? kmp_extract_tcp(data, cl, &po, &pl)
                                                    : kmp_extract_udp(data, cl, &po, &pl);   /* :119-122 */
            if (pass == 0) frames++;
            if (!ok) continue;                                      /* serial.c:138-140: skipped */
            if (pass == 1) {
                out->off[k] = pos;
                out->len[k] = pl;
                if (pl) memcpy(out->bytes + pos, data + po, pl);    /* serial.c:125-127 */
                out->payload_bytes += pl;
            }
            pos += round_up(pl ? pl : 1, KMP_SLOT_ALIGN);
            k++;
        }
        kmp_pcap_close(p);
        if (pass == 0) {
            n = k; bytes = pos + KMP_ARENA_SLACK;
            int rc = arena_alloc(out, bytes, n, alloc_fn, free_fn);
            if (rc) { if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory"); return rc; }
        }
    }
    out->n_pkts = n;
    out->n_frames = frames;
    return KMPHOST_OK;
}

/* ============================ raw frames (on-device extraction) ========================= */

int kmp_frames_from_pcap(const char *path, kmp_alloc_fn alloc_fn, kmp_free_fn free_fn, kmp_frames *out,
                         char errbuf[KMP_PCAP_ERRBUF])
{
    memset(out, 0, sizeof *out);
    if (errbuf) errbuf[0] = 0;
    FILE *fp = fopen(path, "rb");
    if (!fp) { if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "%s: %s", path, strerror(errno)); return KMPHOST_EIO; }
    fseek(fp, 0, SEEK_END);
    const long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz < 24) { fclose(fp); if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "truncated dump file; tried to read 24 file header bytes"); return KMPHOST_EIO; }
    out->free_fn = alloc_fn ? free_fn : free;
    out->nbytes = (uint64_t)sz;
    out->bytes = (uint8_t *)(alloc_fn ? alloc_fn((size_t)sz + 64) : malloc((size_t)sz + 64));
    if (!out->bytes) { fclose(fp); if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory"); return KMPHOST_ENOMEM; }
    if (fread(out->bytes, 1, (size_t)sz, fp) != (size_t)sz) {
        fclose(fp); kmp_frames_free(out);
        if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "%s: short read", path);
        return KMPHOST_EIO;
    }
    fclose(fp);
    memset(out->bytes + sz, 0, 64);
    uint32_t magic;
    memcpy(&magic, out->bytes, 4);
    int swap, ng = 0;
    uint32_t bom = 0;
    memcpy(&bom, out->bytes + 8, 4);
    if (magic == 0xA1B2C3D4u || magic == 0xA1B23C4Du) swap = 0;
    else if (bswap32(magic) == 0xA1B2C3D4u || bswap32(magic) == 0xA1B23C4Du) swap = 1;
    else if (magic == PCAPNG_SHB && (bom == PCAPNG_BOM || bswap32(bom) == PCAPNG_BOM)) { ng = 1; swap = 0; }
    else { kmp_frames_free(out); if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "unknown file format"); return KMPHOST_EFORMAT; }
    for (int pass = 0; pass < 2; pass++) {
        uint64_t pos = ng ? 0 : 24, n = 0;
        uint32_t snap = 0;
        while (ng && pos + 12 <= (uint64_t)sz) {                               /* pcapng: walk the blocks */
            uint32_t type, total, w[5];
            memcpy(&type, out->bytes + pos, 4);
            memcpy(&total, out->bytes + pos + 4, 4);
            if (type == PCAPNG_SHB) {
                memcpy(&bom, out->bytes + pos + 8, 4);
                if (bom == PCAPNG_BOM) swap = 0; else if (bswap32(bom) == PCAPNG_BOM) swap = 1; else break;
                snap = 0;
            }
            if (swap) { type = bswap32(type); total = bswap32(total); }
            if (total < 12u || (total & 3u) || pos + total > (uint64_t)sz) break;
            const uint32_t body = total - 12u;
            const uint8_t *b = out->bytes + pos + 8;
            uint64_t doff = 0; uint32_t cl = 0; int pkt = 0;
            if ((type == 6u || type == 2u) && body >= 20u) {
                memcpy(w, b, 20);
                cl = swap ? bswap32(w[3]) : w[3];
                if (cl > body - 20u) break;
                doff = pos + 8 + 20; pkt = 1;
            } else if (type == 3u && body >= 4u) {
                memcpy(w, b, 4);
                cl = swap ? bswap32(w[0]) : w[0];
                if (snap && cl > snap) cl = snap;
                if (cl > body - 4u) cl = body - 4u;
                doff = pos + 8 + 4; pkt = 1;
            } else if (type == 1u && body >= 8u && snap == 0) {
                memcpy(w, b + 4, 4);
                snap = swap ? bswap32(w[0]) : w[0];
            }
            if (pkt) { if (pass) { out->off[n] = doff; out->caplen[n] = cl; } n++; }
            pos += total;
        }
        while (!ng && pos + 16 <= (uint64_t)sz) {
            uint32_t cl;
            memcpy(&cl, out->bytes + pos + 8, 4);
            if (swap) cl = bswap32(cl);
            if (cl > (64u << 20) || pos + 16 + cl > (uint64_t)sz) break;      /* truncated / corrupt record ends the walk (serial.c:115) */
            if (pass) { out->off[n] = pos + 16; out->caplen[n] = cl; }
            pos += 16 + (uint64_t)cl; n++;
        }
        if (!pass) {
            out->n = n;
            out->off = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
            out->caplen = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
            if (!out->off || !out->caplen) { kmp_frames_free(out); return KMPHOST_ENOMEM; }
        }
    }
    return KMPHOST_OK;
}

void kmp_frames_free(kmp_frames *f)
{
    if (!f) return;
    if (f->bytes && f->free_fn) f->free_fn(f->bytes);
    free(f->off); free(f->caplen);
    memset(f, 0, sizeof *f);
}

/* ============================ streamed capture: batches =================================
 * The producer half of openmp_task.c:126-155. */
struct kmp_batch_reader {
    kmp_pcap      *p;
    int            proto;
    int            pending;          /* a record has been read but did not fit into the previous batch */
    uint32_t       cl;
    const uint8_t *data;
    int            eof;
};

kmp_batch_reader *kmp_batch_open(const char *path, int proto, char errbuf[KMP_PCAP_ERRBUF])
{
    kmp_pcap *p = kmp_pcap_open(path, errbuf);
    if (!p) return NULL;
    kmp_batch_reader *r = (kmp_batch_reader *)calloc(1, sizeof *r);
    if (!r) { kmp_pcap_close(p); if (errbuf) snprintf(errbuf, KMP_PCAP_ERRBUF, "out of memory"); return NULL; }
    r->p = p; r->proto = proto;
    return r;
}

int64_t kmp_batch_next(kmp_batch_reader *r, uint8_t *arena, uint64_t cap_bytes, uint64_t *off, uint32_t *len,
                       uint64_t cap_pkts, uint64_t *used_bytes, uint64_t *frames)
{
    uint64_t n = 0, pos = 0;
    if (cap_bytes < KMP_ARENA_SLACK + KMP_SLOT_ALIGN) return KMPHOST_EINVAL;
    const uint64_t room = cap_bytes - KMP_ARENA_SLACK;
    while (!r->eof) {
        if (!r->pending) {
            uint32_t ln;
            if (kmp_pcap_next(r->p, &r->cl, &ln, &r->data) < 0) { r->eof = 1; break; }   /* openmp_task.c:135 */
            if (frames) (*frames)++;
            r->pending = 1;
        }
        uint32_t po, pl;
        const int ok = (r->proto == KMP_PROTO_TCP) ? kmp_extract_tcp(r->data, r->cl, &po, &pl)
                                                   : kmp_extract_udp(r->data, r->cl, &po, &pl);   /* openmp_task.c:139-142 */
        if (!ok) { r->pending = 0; continue; }      /* invalid frames cannot match anything (openmp_task.c:150-153 stores " ") */
        const uint64_t slot = round_up(pl ? pl : 1, KMP_SLOT_ALIGN);
        if (slot > room) return KMPHOST_EINVAL;
        if (pos + slot > room || n == cap_pkts) break;           /* keep the record for the next batch */
        off[n] = pos; len[n] = pl;
        if (pl) memcpy(arena + pos, r->data + po, pl);
        if (slot > pl) memset(arena + pos + pl, 0, (size_t)(slot - pl));
        pos += slot; n++;
        r->pending = 0;
    }
    if (n) memset(arena + pos, 0, KMP_ARENA_SLACK);
    if (used_bytes) *used_bytes = n ? pos + KMP_ARENA_SLACK : 0;
    return (int64_t)n;
}

void kmp_batch_close(kmp_batch_reader *r)
{
    if (!r) return;
    kmp_pcap_close(r->p);
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
        if (counts[i] != 0)
            fprintf(fp, "%s: %d times!\n", (const char *)(pats->blob + pats->off[i]), (int)counts[i]);
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
