/*
 * kmpgpu.hip -- C-ABI layer over the gfx950 kernels (include/kmpgpu.h).  Host code only; the
 * kernels live in kmp_scan_*.hip / kmp_prep.hip.  Replaces the state the reference keeps in main()'s locals
 * (array_of_strings / prefix_array / array_of_payloads / string_count, serial.c:54,99,101,148)
 * and the hot loop serial.c:153-155.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>          /* types and enums only: the library itself is opened with dlopen (kmpgpu_comm_*) */

#include "kmpgpu.h"
#include "kmp_device.h"
#include "kmp_launch.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(KMPGPU_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

/* Failure function, as kmp_prefix (serial.c:217-238). */
void failure_table(const uint8_t *pat, uint32_t m, uint8_t *out)
{
    out[0] = 0;
    uint32_t j = 0;
    for (uint32_t i = 1; i < m;) {
        if (pat[i] == pat[j]) { out[i++] = (uint8_t)++j; }
        else if (j) { j = out[j - 1]; }
        else { out[i++] = 0; }
    }
}

}  // namespace

struct kmpgpu_ctx {
    int          device = 0;
    hipStream_t  own_stream = nullptr;
    hipStream_t  stream = nullptr;
    int          cu_count = 256;

    /* patterns */
    uint32_t              n_pat = 0;
    kmp_pattern_dev      *d_patterns = nullptr;
    uint32_t             *d_ids = nullptr;          /* [n_pat]: long patterns (m >= 4) first, then short */
    uint32_t              n_long = 0, n_short = 0;
    /* fused multi-pattern pass: unique patterns of 2..20 bytes share one arena read */
    /* fused multi-pattern pass: the eligible patterns in groups of at most KMP_MULTI_MAX_UNIQUE distinct ones, one read
     * of the arena per group */
    struct FusedGroup {
        uint32_t *d_tables = nullptr;        /* layout: kmp_device.h KMP_MULTI_*                        */
        uint32_t *d_ids = nullptr;           /* [n_ids] pattern indices counted by this group            */
        uint32_t *d_rows = nullptr;          /* [n_ids] their unique-pattern row                         */
        uint32_t *d_uid_first = nullptr, *d_uid_ids = nullptr;   /* row -> pattern indices (offset emission): CSR */
        uint32_t  words = 0, n_unique = 0, cshift = 0, bmask = 0, n_ones = 0, ones = 0, n_ids = 0;   /* cshift: a plain group's short patterns, a classed one's class shift */
        bool      classed = false;
    };
    std::vector<FusedGroup> fused_groups;
    uint32_t              n_multi_unique = 0;          /* distinct eligible patterns over all groups */
    uint32_t             *d_rest_ids = nullptr;      /* [rest_long + rest_short] everything else, long first  */
    uint32_t              rest_long = 0, rest_short = 0;

    /* arena */
    const uint8_t  *d_arena = nullptr;
    const uint64_t *d_off = nullptr;
    const uint32_t *d_len = nullptr;
    uint64_t        arena_bytes = 0, n_pkts = 0, payload_bytes = 0;
    bool            uniform = false;                  /* every payload has the same length, slots back to back */
    bool            packed = false;                   /* slots back to back (any lengths): flat streaming with bitmap + plan */
    bool            pad_clean = false;                /* packed arena whose slot padding is all 0x00 (kmp_check_padding_kernel) */
    bool            pad_known_clean = false;          /* set around prepare_packed by a loader that wrote the padding itself: no check pass */
    uint64_t        span_end = 0;                     /* end offset of the last slot */
    unsigned long long *d_bitmap = nullptr;           /* one bit per 16-byte slot: a payload starts here */
    void           *d_plan = nullptr;                 /* kmp_plan_entry[plan_waves + 1] */
    uint64_t        plan_waves = 0, plan_cap = 0;
    uint32_t       *d_pool = nullptr;                 /* fused pass: next pool unit of every region */
    uint64_t        pool_cap = 0;
    void           *d_uplan = nullptr;                /* fused pass: kmp_plan_entry[uplan_units + 1], the work units of its blocks' regions */
    uint64_t        uplan_units = 0, uplan_cap = 0;
    kmp_plan_shape  uplan_shape{};                    /* what d_uplan was cut for */
    int             fused_unit = 0;                   /* KMPGPU_OPT_FUSED_UNIT */
    uint64_t        uni_off0 = 0;
    uint32_t        uni_stride = 0, uni_len = 0;
    void           *owned_arena = nullptr, *owned_off = nullptr, *owned_len = nullptr;
    uint64_t        cap_arena = 0, cap_pkts = 0;      /* capacities of the owned buffers (reused by the next load) */
    uint64_t        bitmap_cap = 0;                   /* words d_bitmap holds (kept from load to load: a streamed capture loads batch after batch) */
    bool            bitmap_live = false;              /* d_bitmap describes the arena that is attached now */
    /* scratch of kmpgpu_load_frames, kept between calls for the same reason (hipMalloc / hipFree per batch would synchronise
     * the device under the other context's scan): the frames' bytes, their offsets / captured lengths, the scan workspace, the
     * payloads' source offsets, the totals */
    uint8_t        *fr_file = nullptr, *fr_ws = nullptr;
    uint64_t       *fr_off = nullptr, *fr_src = nullptr;
    uint32_t       *fr_cl = nullptr;
    unsigned long long *fr_tot = nullptr;
    bool            fr_pending = false;               /* kmpgpu_load_frames_begin has enqueued an upload that kmpgpu_load_frames_finish has not taken yet */
    uint64_t        fr_n = 0, fr_span = 0, fr_span_lo = 0;
    int             fr_tcp = 0;
    uint64_t        fr_file_cap = 0, fr_off_cap = 0, fr_cl_cap = 0, fr_ws_cap = 0, fr_src_cap = 0;

    /* results */
    unsigned long long *d_partials = nullptr;
    size_t              partials_cap = 0;             /* elements */
    unsigned long long *d_counts = nullptr;
    uint32_t           *d_err = nullptr;              /* [2] validation flags */
    unsigned long long *d_sum = nullptr;              /* [6] payload bytes, offset 0, stride, length 0, end of last slot */
    uint64_t           *h_counts = nullptr;           /* pinned */
    size_t              h_counts_cap = 0;
    unsigned long long *h_small = nullptr;            /* pinned, 16 words: where the loaders read small device results back (a copy to pageable
                                                         memory goes through the runtime's blocking staging path) */

    /* options */
    int mode = 0, blocks_per_cu = 0 /* auto */, depth = 0 /* auto */, nontemporal = 1, kernel_sel = 0, fused = 2 /* auto */, accumulate = 0, repack = 1;

    /* timing */
    hipEvent_t  ev[4] = {nullptr, nullptr, nullptr, nullptr};
    kmpgpu_timing last{};
    std::vector<hipEvent_t> prof_ev;                  /* pairs */
    uint32_t    prof_cap = 0, prof_n = 0;
    bool        profiling = false;

    struct kmpgpu_comm *comm = nullptr;               /* the communicator this context is a rank of (kmpgpu_comm_*), if any */
};

static void comm_forget(kmpgpu_comm *k, kmpgpu_ctx *c);

namespace {

/* Uniform-stride arenas: the flat kernel from 512-byte payloads on; below that a chunk holds several packets and
 * the packed kernel's bitmap beats the flat kernel's arithmetic by 3-6 % (profiles/r01_flat_vs_packed.txt). */
bool use_flat(const kmpgpu_ctx *c)
{
    if (!c->uniform || c->mode != 0) return false;
    if (c->kernel_sel == 3) return true;
    return c->kernel_sel == 0 && (c->uni_len >= 512u || !c->packed || !c->bitmap_live);
}
/* Fused multi-pattern pass: explicit (1) or automatic (2): from 2 unique eligible patterns on -- 0.24 ms against
 * 2 x 0.23 ms as streaming passes over 1.5 GB (profiles/r02_multipattern.txt); the 1-byte patterns that ride along
 * do not count, a set of one eligible pattern plus 1-byte patterns keeps its streaming passes. */
bool use_fused(const kmpgpu_ctx *c)
{
    if (!c->packed || !c->bitmap_live || c->mode != 0 || c->kernel_sel == 1 || c->fused_groups.empty()) return false;
    if (c->fused == 1) return c->n_multi_unique >= 2;
    return c->fused == 2 && c->n_multi_unique >= 2;
}

bool use_packed(const kmpgpu_ctx *c)
{
    return c->packed && c->bitmap_live && c->mode == 0 && (c->kernel_sel == 2 || ((c->kernel_sel == 0 || c->kernel_sel == 3) && !use_flat(c)));
}

uint32_t grid_blocks(const kmpgpu_ctx *c, bool emit = false)
{
    /* In units of 4-wavefront blocks.  An explicit KMPGPU_OPT_BLOCKS_PER_CU means CUs x that many (the shape of rounds 1-2:
     * a persistent grid, 4 per CU for the flat kernel, 6 for the packed one); automatic: the flat and the packed kernel
     * take small ranges and as many blocks as that needs (below), the general kernel 8 per CU, the fused pass what
     * fits a CU (two of its 16-wavefront blocks = 8 of these units). */
    const bool streaming = use_flat(c) || use_packed(c);
    int fused_bpc = 7;
    if (use_fused(c)) {
        /* as many wavefronts as a CU holds of the kernel the pass will take (registers and the 160 KB of LDS), counted here in 4-wavefront blocks,
         * the unit the plan is cut in.  Only the first group carries 1-byte patterns; the plan follows it. */
        uint32_t waves = 64u;
        for (const kmpgpu_ctx::FusedGroup &g : c->fused_groups)
            waves = std::min(waves, kmp_multi_resident_waves(kmp_multi_kind(emit, c->pad_clean, c->fused_groups.front().n_ones), g.words, g.n_unique));
        /* ONE round of blocks: inside a block the wavefronts share its region out among themselves as they go (work units, enqueue_pass), so no
         * block ends long before the others and a second round has nothing to even out (with fixed ranges it had: 325 -> 317 us; with
         * units one round 303 / 146 / 154 us, two rounds 303 / 158 / 165 us on 1500-byte, Zipf and 64-byte packets, profiles/r03_fused_units_sweep2.txt) */
        fused_bpc = (int)std::max<uint32_t>(1u, waves / KMP_BLOCK_WAVES);
    }
    const int bpc = c->blocks_per_cu > 0 ? c->blocks_per_cu
                  : use_fused(c) ? fused_bpc : !streaming ? 8 : use_flat(c) ? 4 : 6;
    uint64_t need = (c->n_pkts + KMP_BLOCK_WAVES - 1) / KMP_BLOCK_WAVES;
    if (c->blocks_per_cu <= 0 && use_flat(c) && !use_fused(c) && c->uni_stride) {
        /* The flat kernel does NOT run as a persistent grid: a wavefront takes ~6 KiB (four 1500-byte packets) and the grid is
         * one block per four such ranges.  The hardware hands the blocks out in order as CUs free up, so at any moment the
         * whole chip reads one compact, moving window of the arena and nobody waits for a straggler at the end: 209-211 us per
         * 1.5 GB (0.89 of the HBM peak) against 223-234 us with four resident blocks per CU and 366 KB per wavefront
         * (profiles/r02_flat_grid.txt).  Capped so that blocks x patterns stays below 2^22 (partial counts; a launch of 2^30 threads). */
        uint64_t g = c->uni_stride, r = 128;
        while (r) { const uint64_t t = g % r; g = r; r = t; }                     /* gcd(stride, 128): ranges start on 128-byte lines */
        const uint64_t q = c->uni_stride >= 4096u ? 1 : 128 / g;                  /* (long payloads: a shared line per range is noise, a range of several is not) */
        const uint64_t ppw = std::max<uint64_t>(6144 / c->uni_stride / q * q, q);        /* about 6 KiB, a multiple of q packets (1504-byte slots: 4) */
        uint64_t bx = (c->n_pkts + KMP_BLOCK_WAVES * ppw - 1) / (KMP_BLOCK_WAVES * ppw);
        const uint64_t max_bx = std::max<uint64_t>((1ull << 22) / std::max<uint32_t>(c->n_pat, 1u), (uint64_t)c->cu_count * 4u);
        bx = std::min(bx, max_bx);
        return (uint32_t)std::max<uint64_t>(bx, 1);
    }
    if (c->blocks_per_cu <= 0 && use_packed(c) && !use_fused(c)) {
        /* The packed kernel likewise: ~16 KiB per wavefront and as many blocks as that takes, handed out in order by the
         * hardware (its per-range set-up -- plan entry, bitmap words -- is heavier than the flat kernel's, 6 KiB ranges cost
         * more than they gain): Zipf 64..9000 B 109 -> 105 us per 0.67 GB, 64-byte payloads 155 -> 143 us per 0.77 GB
         * (profiles/r02_flat_grid.txt). */
        const uint64_t span = c->span_end - c->uni_off0;
        uint64_t bx = std::min<uint64_t>(need, (span + KMP_BLOCK_WAVES * 16384ull - 1) / (KMP_BLOCK_WAVES * 16384ull));
        bx = std::min<uint64_t>(bx, std::max<uint64_t>((1ull << 22) / std::max<uint32_t>(c->n_pat, 1u), (uint64_t)c->cu_count * 6u));
        return (uint32_t)std::max<uint64_t>(bx, 1);
    }
    if (streaming && c->blocks_per_cu <= 0) {
        /* small captures: give every wavefront at least 8 KiB to stream instead of launching
         * thousands of nearly empty wavefronts per pattern */
        const uint64_t span = c->span_end - c->uni_off0;
        const uint64_t per_wave = use_fused(c) ? 2048ull : 8192ull;      /* the fused pass does ~10x the work per byte */
        need = std::min<uint64_t>(need, (span + KMP_BLOCK_WAVES * per_wave - 1) / (KMP_BLOCK_WAVES * per_wave));
    }
    uint64_t cap = (uint64_t)c->cu_count * (uint64_t)bpc;
    uint64_t b = std::min(need, cap);
    return (uint32_t)std::max<uint64_t>(b, 1);
}

void free_fused_groups(kmpgpu_ctx *c)
{
    for (kmpgpu_ctx::FusedGroup &g : c->fused_groups)
        for (uint32_t *p : {g.d_tables, g.d_ids, g.d_rows, g.d_uid_first, g.d_uid_ids})
            if (p) (void)hipFree(p);
    c->fused_groups.clear();
    c->n_multi_unique = 0;
}

int ensure_partials(kmpgpu_ctx *c, size_t elems)
{
    if (elems <= c->partials_cap) return KMPGPU_OK;
    if (c->d_partials) HIP_TRY(hipFree(c->d_partials));
    c->d_partials = nullptr; c->partials_cap = 0;
    HIP_TRY(hipMalloc(&c->d_partials, elems * sizeof(unsigned long long)));
    c->partials_cap = elems;
    return KMPGPU_OK;
}

void release_arena(kmpgpu_ctx *c, bool keep_buffers = false)
{
    if (!keep_buffers) {
        if (c->owned_arena) (void)hipFree(c->owned_arena);
        if (c->owned_off) (void)hipFree(c->owned_off);
        if (c->owned_len) (void)hipFree(c->owned_len);
        c->owned_arena = c->owned_off = c->owned_len = nullptr;
        c->cap_arena = c->cap_pkts = 0;
    }
    c->d_arena = nullptr; c->d_off = nullptr; c->d_len = nullptr;
    c->arena_bytes = c->n_pkts = c->payload_bytes = 0;
    c->uniform = false; c->packed = false; c->pad_clean = false; c->plan_waves = 0; c->uplan_units = 0;
    c->bitmap_live = false;                           /* the buffer itself (1/128 of an arena) is kept for the next arena */
}

/* Host ranges pinned through kmpgpu_host_register.  One copy must not straddle two registrations (the runtime refuses it), and a
 * capture is pinned window by window: uploads are cut at the registrations' boundaries. */
std::mutex g_pinned_mu;
std::map<uintptr_t, size_t> g_pinned;

hipError_t upload_split(void *dst, const uint8_t *src, uint64_t n, hipStream_t st)
{
    std::vector<std::pair<uint64_t, uint64_t>> pieces;          /* (offset, length) */
    {
        std::lock_guard<std::mutex> lock(g_pinned_mu);
        if (g_pinned.empty()) pieces.emplace_back(0, n);
        else {
            uint64_t pos = 0;
            while (pos < n) {
                const uintptr_t a = (uintptr_t)src + pos;
                uint64_t len = n - pos;
                auto it = g_pinned.upper_bound(a);                /* first registration that starts behind a */
                if (it != g_pinned.begin()) {
                    auto in = std::prev(it);
                    if (a < in->first + in->second) len = std::min<uint64_t>(len, in->first + in->second - a);       /* inside one: up to its end */
                    else if (it != g_pinned.end()) len = std::min<uint64_t>(len, it->first - a);                     /* pageable stretch up to the next */
                } else if (it != g_pinned.end()) len = std::min<uint64_t>(len, it->first - a);
                pieces.emplace_back(pos, len);
                pos += len;
            }
        }
    }
    for (const auto &p : pieces) {
        const hipError_t e = hipMemcpyAsync((uint8_t *)dst + p.first, src + p.first, p.second, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

/* a device buffer of at least `want` elements, kept between calls: grown (with an eighth of headroom) only when it is too small */
template <typename T>
hipError_t grow_buffer(T **p, uint64_t *cap, uint64_t want)
{
    if (*p && *cap >= want) return hipSuccess;
    if (*p) { const hipError_t e = hipFree(*p); *p = nullptr; *cap = 0; if (e != hipSuccess) return e; }
    const uint64_t take = want + want / 8 + 1;
    const hipError_t e = hipMalloc((void **)p, (size_t)take * sizeof(T));
    if (e == hipSuccess) *cap = take;
    return e;
}

void release_frame_scratch(kmpgpu_ctx *c)
{
    for (void *p : {(void *)c->fr_file, (void *)c->fr_ws, (void *)c->fr_off, (void *)c->fr_src, (void *)c->fr_cl, (void *)c->fr_tot})
        if (p) (void)hipFree(p);
    c->fr_file = c->fr_ws = nullptr; c->fr_off = c->fr_src = nullptr; c->fr_cl = nullptr; c->fr_tot = nullptr;
    c->fr_file_cap = c->fr_off_cap = c->fr_cl_cap = c->fr_ws_cap = c->fr_src_cap = 0;
}

/* An arena whose slots are not back to back (gaps, shuffled order) is copied once into a packed one owned
 * by the context, so that the streaming kernels apply to it too (KMPGPU_OPT_REPACK, default on). */
int repack_arena(kmpgpu_ctx *c)
{
    if (c->packed || !c->repack || c->n_pkts == 0) return KMPGPU_OK;
    uint8_t *ws = nullptr, *na = nullptr;
    uint64_t *noff = nullptr;
    uint32_t *nlen = nullptr;
    unsigned long long *d_tot = nullptr, tot[2] = {0, 0};
    auto drop = [&]() { if (ws) (void)hipFree(ws); if (d_tot) (void)hipFree(d_tot); };
#define KMP_TRY3(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { drop(); if (na) (void)hipFree(na); if (noff) (void)hipFree(noff); if (nlen) (void)hipFree(nlen); \
        return fail(KMPGPU_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    KMP_TRY3(hipMalloc(&ws, kmp_extract_ws_bytes(c->n_pkts)));
    KMP_TRY3(hipMalloc(&d_tot, 2 * sizeof(unsigned long long)));
    KMP_TRY3(kmp_launch_repack_phase1(c->d_len, c->n_pkts, ws, d_tot, c->stream));
    KMP_TRY3(hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, c->stream));
    KMP_TRY3(hipStreamSynchronize(c->stream));
    const uint64_t nbytes = tot[0] + 64;
    KMP_TRY3(hipMalloc(&na, nbytes));
    KMP_TRY3(hipMalloc(&noff, c->n_pkts * sizeof(uint64_t)));
    KMP_TRY3(hipMalloc(&nlen, c->n_pkts * sizeof(uint32_t)));
    KMP_TRY3(hipMemsetAsync(na + tot[0], 0, 64, c->stream));
    KMP_TRY3(hipMemcpyAsync(nlen, c->d_len, c->n_pkts * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    KMP_TRY3(kmp_launch_repack_phase2(c->d_arena, c->d_off, c->d_len, c->n_pkts, ws, na, noff, c->stream));
    KMP_TRY3(hipStreamSynchronize(c->stream));
#undef KMP_TRY3
    drop();
    /* the context now owns the packed copy; a borrowed or uploaded original is released */
    if (c->owned_arena) (void)hipFree(c->owned_arena);
    if (c->owned_off) (void)hipFree(c->owned_off);
    if (c->owned_len) (void)hipFree(c->owned_len);
    c->owned_arena = na; c->owned_off = noff; c->owned_len = nlen;
    c->cap_arena = nbytes; c->cap_pkts = c->n_pkts;
    c->d_arena = na; c->d_off = noff; c->d_len = nlen;
    c->arena_bytes = nbytes;
    c->packed = true; c->uniform = false;
    c->uni_off0 = 0; c->span_end = tot[0];
    return KMPGPU_OK;
}

/* Side tables of the packed streaming kernel: start bitmap now, wavefront plan on first use. */
int prepare_packed(kmpgpu_ctx *c)
{
    int rc = repack_arena(c);
    if (rc) return rc;
    if (!c->packed || c->n_pkts == 0) return KMPGPU_OK;
    const size_t words = (size_t)(c->arena_bytes / KMP_CHUNK) + 32;      /* the group prefetch reads up to 2 * DEPTH + 1 words past the end */
    if (c->bitmap_cap < words) {
        if (c->d_bitmap) HIP_TRY(hipFree(c->d_bitmap));
        c->d_bitmap = nullptr; c->bitmap_cap = 0;
        HIP_TRY(hipMalloc(&c->d_bitmap, words * sizeof(unsigned long long)));
        c->bitmap_cap = words;
    }
    HIP_TRY(hipMemsetAsync(c->d_bitmap, 0, words * sizeof(unsigned long long), c->stream));
    HIP_TRY(kmp_launch_build_bitmap(c->d_off, c->n_pkts, c->d_bitmap, c->stream));
    c->bitmap_live = true;
    /* slot padding: checked once; cleared when the arena is the context's own copy, otherwise the packed
     * kernel keeps fetching offset and length of a candidate's payload from the index */
    const bool own = c->owned_arena && c->d_arena == (const uint8_t *)c->owned_arena;
    if (own && c->pad_known_clean) { c->pad_clean = true; return KMPGPU_OK; }
    uint32_t dirty = 0;
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(kmp_launch_check_padding(const_cast<uint8_t *>(c->d_arena), c->d_off, c->d_len, c->n_pkts, own ? 1 : 0, c->d_err, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_small, c->d_err, sizeof dirty, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(&dirty, c->h_small, sizeof dirty);
    c->pad_clean = own || dirty == 0;
    return KMPGPU_OK;
}

/* Enqueue one full pass: scan launches (patterns grouped by "shorter than 4 bytes") + reduce. */
struct EmitTarget { void *out = nullptr; unsigned long long *counter = nullptr; unsigned long long cap = 0; };

int enqueue_pass(kmpgpu_ctx *c, uint32_t *launches, unsigned long long *d_out, const EmitTarget *emit = nullptr)
{
    if (!d_out) d_out = c->d_counts;
    if (!c->d_patterns || c->n_pat == 0) return fail(KMPGPU_ESTATE, "kmpgpu_scan: no patterns set");
    if (!c->d_off && c->n_pkts) return fail(KMPGPU_ESTATE, "kmpgpu_scan: no arena loaded");
    uint32_t nl = 0;
    if (c->n_pkts == 0) {
        if (!c->accumulate) HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(unsigned long long) * c->n_pat, c->stream));
        if (launches) *launches = 0;
        return KMPGPU_OK;
    }
    const uint32_t bx = grid_blocks(c, emit != nullptr);
    /* partial counts: a row per pattern -- or, where the fused pass runs, a row per id of its largest group (a classed group numbers
     * its patterns by bucket class, up to 1024 ids however few patterns it has) and one per pattern that keeps a pass of its own */
    size_t part_rows = c->n_pat;
    if (use_fused(c)) {
        size_t max_u = 0;
        for (const kmpgpu_ctx::FusedGroup &g : c->fused_groups) max_u = std::max<size_t>(max_u, g.n_unique);
        part_rows = std::max<size_t>(part_rows, max_u + c->rest_long + c->rest_short);
    }
    int rc = ensure_partials(c, (size_t)bx * part_rows);
    if (rc) return rc;

    kmp_scan_args a{};
    a.arena = c->d_arena; a.pkt_off = c->d_off; a.pkt_len = c->d_len; a.n_pkts = c->n_pkts;
    a.patterns = c->d_patterns; a.blocks_x = bx; a.depth = c->depth; /* 0: the launcher's own default */ a.mode = c->mode;
    a.nontemporal = c->nontemporal != 0;
    a.pad_clean = c->pad_clean;
    if (emit) { a.emit_out = emit->out; a.emit_counter = emit->counter; a.emit_cap = emit->cap; }
    /* uniform-stride arenas take the flat streaming kernel (contiguous packet run per wavefront) */
    const uint64_t nwaves = (uint64_t)bx * KMP_BLOCK_WAVES;
    uint64_t ppw = (c->n_pkts + nwaves - 1) / nwaves;
    if (c->uniform && c->uni_stride) {
        /* start every wavefront's range on a 128-byte line so that neighbouring ranges share no cache line */
        uint64_t g = c->uni_stride, r = 128;
        while (r) { const uint64_t t = g % r; g = r; r = t; }      /* gcd(stride, 128) */
        const uint64_t q = c->uni_stride >= 4096u ? 1 : 128 / g;
        ppw = (ppw + q - 1) / q * q;
    }
    const bool flat = use_flat(c) && ppw * c->uni_stride < (1ull << 31);
    if (flat) {
        a.arena = c->d_arena + c->uni_off0;
        a.uniform_stride = c->uni_stride; a.uniform_len = c->uni_len; a.pkts_per_wave = (uint32_t)ppw;
    }
    /* packed arenas: byte-balanced wavefront ranges (plan) + packet-start bitmap, used by the packed
     * streaming kernel (mixed lengths) and by the fused multi-pattern pass */
    const bool fused = use_fused(c);
    bool packed = !flat && use_packed(c);
    bool do_fused = false;
    if (fused) {
        /* The fused pass: one region of the arena per PAIR of blocks, in work units their wavefronts take one after the other
         * (kmp_scan_multi.hip).  Every wavefront starts with one large unit of its own, and the second half of the region
         * lies in a pool of 32 KiB units for whoever is done first (profiles/r03_fused_units_sweep4_pair_pools.txt): the SIMDs serve their wavefronts in order of age, so the
         * first wavefront of a block is through its share when the last one has a third to go, and the block that came to a CU first
         * is done when the second one has a third to go.  (Small units throughout cost more than they balance: a unit begins
         * with the dependent chain counter - entry - first loads, which the other wavefronts of the SIMD do not cover --
         * profiles/r03_tried_all_units_dynamic.txt.) */
        const uint32_t bwaves = kmp_multi_block_waves(kmp_multi_kind(emit != nullptr, c->pad_clean, c->fused_groups.front().n_ones));
        uint64_t fblocks = ((uint64_t)bx * KMP_BLOCK_WAVES + bwaves - 1u) / bwaves;
        const uint32_t sides = fblocks >= 2 ? 2u : 1u;
        fblocks -= fblocks % sides;
        const uint64_t regions = fblocks / sides;
        const uint64_t span = c->span_end - c->uni_off0;
        kmp_plan_shape sh{};
        sh.region = (((span + regions - 1) / regions) + 1023ull) & ~1023ull;
        uint64_t small = c->fused_unit ? (uint64_t)c->fused_unit : 32768ull, pool_num = 1, pool_div = 2, big = 0;
#ifdef KMP_MULTI_TUNING
        if (const char *e = getenv("KMP_FUSED_UNIT")) big = strtoull(e, nullptr, 0) & ~1023ull;          /* 0: from the pool's share */
        if (const char *e = getenv("KMP_FUSED_SMALL")) small = std::max<uint64_t>(1024ull, strtoull(e, nullptr, 0) & ~1023ull);
        if (const char *e = getenv("KMP_FUSED_TAIL_DIV")) { pool_num = 1; pool_div = std::max<uint64_t>(1ull, strtoull(e, nullptr, 0)); }
        if (const char *e = getenv("KMP_FUSED_TAIL_NUM")) pool_num = std::min<uint64_t>(pool_div, strtoull(e, nullptr, 0));
#endif
        uint64_t big_units = (uint64_t)sides * bwaves;
        /* (a small region -- a capture of a few hundred KB per block -- goes to the wavefronts whole: a unit of the pool costs a round trip
         * to the counter in global memory, which a pass of 10 us does not have) */
        const bool no_pool = sh.region < (1ull << 20) && !big;
        const uint64_t own_bytes = no_pool ? sh.region : sh.region - sh.region / pool_div * pool_num;
        sh.step = big ? big : std::max<uint64_t>(1024ull, no_pool ? ((own_bytes + big_units - 1) / big_units + 1023ull) & ~1023ull : (own_bytes / big_units) & ~1023ull);
        if (!no_pool && big_units * sh.step > sh.region) big_units = sh.region / sh.step;
        uint64_t rest = no_pool ? 0ull : sh.region - big_units * sh.step;         /* (without a pool the shares reach the region's end: kmp_plan_kernel cuts them there) */
        /* (a small region: at least two units of the pool per wavefront, or the last unit is all that is left to do for a long time) */
        if (!c->fused_unit) small = std::min<uint64_t>(small, std::max<uint64_t>(1024ull, (rest / (2ull * big_units ? 2ull * big_units : 1ull)) & ~1023ull));
        /* a block holds the entries of its units in LDS, KMP_MULTI_MAX_UNITS of them: a large region has larger pool units */
        const uint64_t room = KMP_MULTI_MAX_UNITS - 1u - big_units;                  /* (one entry stays free: "no such unit") */
        if ((rest + small - 1) / small > room) small = (((rest + room - 1) / room) + 1023ull) & ~1023ull;
        sh.small = (uint32_t)small;
        const uint64_t upb = big_units + (rest + small - 1) / small;
        const uint64_t n_units = regions * upb;
        if (sh.region < 0x7FE00000ull && n_units < (1ull << 31)) {                   /* (positions inside a region are 32-bit, kmp_scan_multi.hip) */
            sh.big_units = (uint32_t)big_units; sh.units = (uint32_t)upb;
            const kmp_plan_shape &o = c->uplan_shape;
            if (c->uplan_units != n_units || o.step != sh.step || o.region != sh.region || o.units != sh.units || o.big_units != sh.big_units || o.small != sh.small) {
                if (c->uplan_cap < n_units + 1) {
                    if (c->d_uplan) HIP_TRY(hipFree(c->d_uplan));
                    c->d_uplan = nullptr; c->uplan_cap = 0;
                    HIP_TRY(hipMalloc(&c->d_uplan, (n_units + 1) * 16));
                    c->uplan_cap = n_units + 1;
                }
                HIP_TRY(kmp_launch_plan(c->d_off, c->d_len, c->n_pkts, n_units, sh, c->d_uplan, c->stream));
                c->uplan_units = n_units; c->uplan_shape = sh;
            }
            if (c->pool_cap < regions) {
                if (c->d_pool) HIP_TRY(hipFree(c->d_pool));
                c->d_pool = nullptr; c->pool_cap = 0;
                HIP_TRY(hipMalloc(&c->d_pool, regions * sizeof(uint32_t)));
                c->pool_cap = regions;
            }
            a.fused_blocks = (uint32_t)fblocks; a.units_per_block = (uint32_t)upb; a.n_units = (uint32_t)n_units; a.span_end = c->span_end;
            a.fused_sides = sides; a.fused_pool = c->d_pool;
            a.bitmap = c->d_bitmap;
            do_fused = true;
        }
    }
    if (packed) {
        const uint64_t span = c->span_end - c->uni_off0;
        uint64_t bpw = (((span + nwaves - 1) / nwaves) + 15ull) & ~15ull;
        /* whole chunks per range where the ranges are several chunks long: equal-length small payloads then start every range on a
         * 1 KiB boundary (12 M x 64 B: 16 320-byte ranges 154-160 us, 16 384-byte ranges 142-144 us) */
        if (bpw >= 8192ull) bpw = (bpw + 1023ull) & ~1023ull;
        if (bpw >= (1ull << 30)) packed = false;
        else {
            if (c->plan_waves != nwaves) {
                if (c->plan_cap < nwaves + 1) {
                    if (c->d_plan) HIP_TRY(hipFree(c->d_plan));
                    c->d_plan = nullptr; c->plan_cap = 0;
                    HIP_TRY(hipMalloc(&c->d_plan, (nwaves + 1) * 16));
                    c->plan_cap = nwaves + 1;
                }
                kmp_plan_shape sh{};
                sh.step = bpw ? bpw : 16;
                HIP_TRY(kmp_launch_plan(c->d_off, c->d_len, c->n_pkts, nwaves, sh, c->d_plan, c->stream));
                c->plan_waves = nwaves;
            }
            a.bitmap = c->d_bitmap; a.plan = c->d_plan;
        }
    }

    auto record = [&](hipEvent_t &e0, hipEvent_t &e1) -> hipError_t {
        e0 = e1 = nullptr;
        if (c->profiling && c->prof_n < c->prof_cap) {
            e0 = c->prof_ev[2 * c->prof_n]; e1 = c->prof_ev[2 * c->prof_n + 1];
            return hipEventRecord(e0, c->stream);
        }
        return hipSuccess;
    };

    const uint32_t *ids = c->d_ids;
    uint32_t n_long = c->n_long, n_short = c->n_short;
    size_t part_base = 0;                         /* partial rows already used */
    if (do_fused) {
        /* one read of the arena for every group of unique patterns of 2..99 bytes (up to 256, classed groups up to 1024) */
        uint32_t max_u = 0;
        for (const kmpgpu_ctx::FusedGroup &g : c->fused_groups) {
            kmp_scan_args f = a;
            f.arena = c->d_arena;
            f.plan = c->d_uplan;
            f.fused_classed = g.classed;
            f.partials = c->d_partials;
            /* the regions' pool counters start from 0; regions without a pool (every unit is some wavefront's own) get no counter at all --
             * one that is never reset would come round after 2^32 takes, and a streamed capture is a pass per batch */
            f.fused_pool = nullptr;
            if (a.units_per_block > a.fused_sides * kmp_multi_block_waves(kmp_multi_kind(emit != nullptr, c->pad_clean, g.n_ones))) {
                HIP_TRY(hipMemsetAsync(c->d_pool, 0, (size_t)(a.fused_blocks / a.fused_sides) * sizeof(uint32_t), c->stream));
                f.fused_pool = c->d_pool;
            }
            hipEvent_t e0, e1;
            HIP_TRY(record(e0, e1));
            HIP_TRY(kmp_launch_scan_multi(f, g.d_tables, g.words, g.n_unique, g.cshift, g.bmask, g.n_ones, g.ones, g.d_uid_first, g.d_uid_ids, c->stream));
            if (e0) { HIP_TRY(hipEventRecord(e1, c->stream)); c->prof_n++; }
            HIP_TRY(kmp_launch_reduce(c->d_partials, bx, g.d_ids, g.n_ids, d_out, c->stream, g.d_rows, c->accumulate));
            ++nl;
            max_u = std::max(max_u, g.n_unique);
        }
        part_base = max_u;
        ids = c->d_rest_ids; n_long = c->rest_long; n_short = c->rest_short;
    }

    struct Group { uint32_t first, n; bool masked; } groups[2] = {{0, n_long, false}, {n_long, n_short, true}};
    for (const Group &g : groups) {
        /* gridDim.y is limited to 65535 */
        for (uint32_t done = 0; done < g.n; done += 65535u) {
            const uint32_t n = std::min(65535u, g.n - done);
            a.pat_ids = ids + g.first + done;
            a.n_ids = n;
            a.partials = c->d_partials + (part_base + g.first + done) * bx;
            a.masked = g.masked;
            /* tens of thousands of partials per pattern are added up by several blocks, which add to the counter: it starts from 0 */
            const bool sliced = (flat || packed) && kmp_reduce_is_sliced(bx);
            a.zero_counts = (sliced && !c->accumulate) ? d_out : nullptr;
            hipEvent_t e0, e1;
            HIP_TRY(record(e0, e1));
            if (emit && !flat && !packed)
                return fail(KMPGPU_EINVAL, "kmpgpu_scan_offsets: the arena could not be brought into the streaming kernels' layout");
            HIP_TRY(flat ? kmp_launch_scan_flat(a, c->stream) : packed ? kmp_launch_scan_packed(a, c->stream) : kmp_launch_scan(a, c->stream));
            if (e0) { HIP_TRY(hipEventRecord(e1, c->stream)); c->prof_n++; }
            HIP_TRY(kmp_launch_reduce(a.partials, bx, a.pat_ids, n, d_out, c->stream, nullptr, c->accumulate, sliced));
            ++nl;
        }
    }
    if (launches) *launches = nl;
    return KMPGPU_OK;
}

}  // namespace

extern "C" {

const char *kmpgpu_last_error(void) { return g_err.c_str(); }

int kmpgpu_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(KMPGPU_EHIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

int kmpgpu_init(kmpgpu_ctx **out, int device)
{
    if (!out) return fail(KMPGPU_EINVAL, "kmpgpu_init: ctx is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0) return fail(KMPGPU_EHIP, "kmpgpu_init: no HIP device visible");
    if (device < 0 || device >= n) return fail(KMPGPU_EINVAL, "kmpgpu_init: device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(KMPGPU_EHIP, "kmpgpu_init: device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    kmpgpu_ctx *c = new (std::nothrow) kmpgpu_ctx();
    if (!c) return fail(KMPGPU_ENOMEM, "kmpgpu_init: out of host memory");
    c->device = device;
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        c->stream = c->own_stream;
        for (auto &ev : c->ev) if (e == hipSuccess) e = hipEventCreate(&ev);
    }
    if (e == hipSuccess) e = hipMalloc(&c->d_err, 2 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&c->d_sum, 6 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_small, 16 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e != hipSuccess) {
        kmpgpu_destroy(c);
        return fail(KMPGPU_EHIP, "kmpgpu_init: %s", hipGetErrorString(e));
    }
    *out = c;
    return KMPGPU_OK;
}

void kmpgpu_destroy(kmpgpu_ctx *c)
{
    if (!c) return;
    if (c->comm) comm_forget(c->comm, c);           /* a communicator outliving one of its contexts: it keeps device + stream handle only */
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    release_arena(c);
    release_frame_scratch(c);
    if (c->d_bitmap) (void)hipFree(c->d_bitmap);
    if (c->d_patterns) (void)hipFree(c->d_patterns);
    if (c->d_ids) (void)hipFree(c->d_ids);
    if (c->d_partials) (void)hipFree(c->d_partials);
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_plan) (void)hipFree(c->d_plan);
    if (c->d_uplan) (void)hipFree(c->d_uplan);
    if (c->d_pool) (void)hipFree(c->d_pool);
    free_fused_groups(c);
    if (c->d_rest_ids) (void)hipFree(c->d_rest_ids);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_sum) (void)hipFree(c->d_sum);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->h_small) (void)hipHostFree(c->h_small);
    for (auto ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : c->prof_ev) if (ev) (void)hipEventDestroy(ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int kmpgpu_set_stream(kmpgpu_ctx *c, void *hip_stream)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_set_stream: ctx is NULL");
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return KMPGPU_OK;
}

int kmpgpu_set_option(kmpgpu_ctx *c, int key, int64_t value)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_set_option: ctx is NULL");
    switch (key) {
    case KMPGPU_OPT_MODE:
        if (value != 0 && value != 1) return fail(KMPGPU_EINVAL, "mode must be 0 or 1");
        c->mode = (int)value; return KMPGPU_OK;
    case KMPGPU_OPT_BLOCKS_PER_CU:
        if (value < 0 || value > 256) return fail(KMPGPU_EINVAL, "blocks per CU must be 0 (auto) or 1..256");
        c->blocks_per_cu = (int)value; return KMPGPU_OK;
    case KMPGPU_OPT_DEPTH:
        if (value != 0 && (value < 2 || value > 8 || value == 7)) return fail(KMPGPU_EINVAL, "depth must be 0 (auto), 2..6 or 8");
        c->depth = (int)value; return KMPGPU_OK;
    case KMPGPU_OPT_FUSED:
        if (value < 0 || value > 2) return fail(KMPGPU_EINVAL, "fused must be 0, 1 or 2");
        c->fused = (int)value; return KMPGPU_OK;
    case KMPGPU_OPT_REPACK:
        c->repack = value ? 1 : 0; return KMPGPU_OK;
    case KMPGPU_OPT_ACCUMULATE:
        c->accumulate = value ? 1 : 0; return KMPGPU_OK;
    case KMPGPU_OPT_KERNEL:
        if (value < 0 || value > 3) return fail(KMPGPU_EINVAL, "kernel selection must be 0, 1, 2 or 3");
        c->kernel_sel = (int)value; return KMPGPU_OK;
    case KMPGPU_OPT_NONTEMPORAL:
        c->nontemporal = value ? 1 : 0; return KMPGPU_OK;
    case KMPGPU_OPT_FUSED_UNIT:
        if (value != 0 && (value < 1024 || value > (1 << 20) || (value & 1023))) return fail(KMPGPU_EINVAL, "fused unit must be 0 (auto) or a multiple of 1024 up to 1 MiB");
        c->fused_unit = (int)value; return KMPGPU_OK;
    default:
        return fail(KMPGPU_EINVAL, "unknown option %d", key);
    }
}

void *kmpgpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail(KMPGPU_EHIP, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void kmpgpu_host_free(void *p) { if (p) (void)hipHostFree(p); }

int kmpgpu_host_register(const void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(KMPGPU_EINVAL, "kmpgpu_host_register: NULL / empty range");
    if ((uintptr_t)ptr & 4095u) return fail(KMPGPU_EINVAL, "kmpgpu_host_register: the range must start on a page boundary");
    /* portable: visible to every device's context (several shards upload from one mapping); the memory may be a PROT_READ mapping */
    hipError_t e = hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterPortable | hipHostRegisterReadOnly);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterPortable); }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(KMPGPU_EHIP, "hipHostRegister(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
    std::lock_guard<std::mutex> lock(g_pinned_mu);
    g_pinned[(uintptr_t)ptr] = bytes;
    return KMPGPU_OK;
}

int kmpgpu_host_unregister(const void *ptr)
{
    if (!ptr) return fail(KMPGPU_EINVAL, "kmpgpu_host_unregister: NULL");
    { std::lock_guard<std::mutex> lock(g_pinned_mu); g_pinned.erase((uintptr_t)ptr); }
    HIP_TRY(hipHostUnregister(const_cast<void *>(ptr)));
    return KMPGPU_OK;
}

int kmpgpu_set_patterns(kmpgpu_ctx *c, const uint8_t *const *pat, const uint32_t *pat_len, uint32_t n_pat)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_set_patterns: ctx is NULL");
    if (n_pat && (!pat || !pat_len)) return fail(KMPGPU_EINVAL, "kmpgpu_set_patterns: NULL pattern arrays");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<kmp_pattern_dev> host(n_pat ? n_pat : 1);
    std::vector<uint32_t> ids_long, ids_short;
    for (uint32_t i = 0; i < n_pat; i++) {
        const uint32_t m = pat_len[i];
        if (m < 1 || m > KMPGPU_MAX_PATTERN_LEN) return fail(KMPGPU_EINVAL, "pattern %u: length %u not in 1..99", i, m);
        if (!pat[i]) return fail(KMPGPU_EINVAL, "pattern %u is NULL", i);
        if (memchr(pat[i], 0, m)) return fail(KMPGPU_EINVAL, "pattern %u contains a 0x00 byte", i);
        kmp_pattern_dev &d = host[i];
        memset(&d, 0, sizeof d);
        memcpy(d.pat, pat[i], m);
        failure_table(d.pat, m, d.fail);
        d.m = m;
        const uint32_t f = m < 4 ? m : 4;
        for (uint32_t b = 0; b < f; b++) { d.first |= (uint32_t)d.pat[b] << (8 * b); d.mask |= 0xFFu << (8 * b); }
        (m >= 4 ? ids_long : ids_short).push_back(i);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_patterns) { HIP_TRY(hipFree(c->d_patterns)); c->d_patterns = nullptr; }
    if (c->d_ids) { HIP_TRY(hipFree(c->d_ids)); c->d_ids = nullptr; }
    if (c->d_counts) { HIP_TRY(hipFree(c->d_counts)); c->d_counts = nullptr; }
    c->n_pat = n_pat; c->n_long = (uint32_t)ids_long.size(); c->n_short = (uint32_t)ids_short.size();
    const size_t np = n_pat ? n_pat : 1;
    HIP_TRY(hipMalloc(&c->d_patterns, np * sizeof(kmp_pattern_dev)));
    HIP_TRY(hipMalloc(&c->d_ids, np * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->d_counts, np * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->d_counts, 0, np * sizeof(unsigned long long)));
    if (n_pat) {
        std::vector<uint32_t> ids(ids_long);
        ids.insert(ids.end(), ids_short.begin(), ids_short.end());
        HIP_TRY(hipMemcpy(c->d_patterns, host.data(), n_pat * sizeof(kmp_pattern_dev), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_ids, ids.data(), n_pat * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (c->h_counts_cap < np) {
        if (c->h_counts) (void)hipHostFree(c->h_counts);
        c->h_counts = nullptr; c->h_counts_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&c->h_counts, np * sizeof(uint64_t), hipHostMallocDefault));
        c->h_counts_cap = np;
    }

    /* ---- tables of the fused multi-pattern pass (layout: kmp_device.h) ------------------------- */
    free_fused_groups(c);
    if (c->d_rest_ids) { HIP_TRY(hipFree(c->d_rest_ids)); c->d_rest_ids = nullptr; }
    c->rest_long = c->rest_short = 0;
    /* a group: its distinct patterns, the row (unique-pattern id of the kernel) of each, the patterns counted by it and their rows.
     * `classed`: more than 256 rows -- an entry has eight bits for an id, the kernel adds the first id of the bucket's class
     * (= bucket >> 7: eight classes of up to 256 patterns each, 1024 in all; kmp_device.h) */
    constexpr uint32_t NCLS = KMP_MULTI_CLS_WORDS;
    struct HostGroup { std::vector<std::string> uniq; std::vector<uint32_t> gidx, row; std::vector<uint32_t> ids, rows; bool classed = false;
                       uint32_t n_cls[NCLS] = {}, overflow = 0; std::vector<uint8_t> bucket_used; };
    std::vector<HostGroup> hg;
    std::vector<uint32_t> rest_l, rest_s;
    /* 1-byte patterns: up to KMP_MULTI_MAX_ONES distinct ones ride along with the first fused group (counted straight
     * off the text registers, no filter, no queue); further ones keep one streaming pass each */
    std::vector<uint8_t> one_bytes;
    std::vector<std::pair<uint32_t, uint32_t>> one_ids;            /* (pattern index, slot) */
    std::vector<std::string> uniq_all;                             /* the distinct eligible patterns, file order */
    std::vector<uint32_t> first_pat;                               /* ... and the first pattern of the list that is each of them */
    std::unordered_map<std::string, uint32_t> uniq_of;
    std::vector<std::pair<uint32_t, uint32_t>> elig;               /* (pattern index, its distinct pattern) */
    for (uint32_t i = 0; i < n_pat; i++) {
        const uint32_t m = pat_len[i];
        if (m == 1) {
            size_t k = 0;
            while (k < one_bytes.size() && one_bytes[k] != pat[i][0]) k++;
            if (k == one_bytes.size() && k < KMP_MULTI_MAX_ONES) one_bytes.push_back(pat[i][0]);
            if (k < one_bytes.size()) { one_ids.emplace_back(i, (uint32_t)k); continue; }
        }
        if (m < KMP_MULTI_MIN_LEN || m > KMP_MULTI_MAX_LEN) { (m >= 4 ? rest_l : rest_s).push_back(i); continue; }
        const std::string key((const char *)pat[i], m);
        auto it = uniq_of.find(key);
        if (it == uniq_of.end()) {
            /* (a record names the pattern whose bytes 8 .. m-1 the kernel compares against in 16 bits: a pattern of nine bytes or
             * more that first occurs behind the 65 536th of the list keeps a pass of its own) */
            if (m > 8 && i > 0xFFFFu) { rest_l.push_back(i); continue; }
            it = uniq_of.emplace(key, (uint32_t)uniq_all.size()).first; uniq_all.push_back(key); first_pat.push_back(i);
        }
        elig.emplace_back(i, it->second);
    }
    if (uniq_all.size() < 2) {                    /* nothing to fuse: every pattern keeps its own pass (c->d_ids) */
        return KMPGPU_OK;
    }
    /* Which group a distinct pattern goes to.  Up to 256 of them: one group, rows in file order (short ones first, below).  More: the
     * 2-byte patterns (entered under every third byte: 32 buckets each, in all classes) and, if 1-byte patterns ride along, the first
     * patterns of the file fill plain groups of 256; everything else goes to classed groups of up to 1024 -- first fit, a pattern
     * whose class is full (256) or whose bucket would overflow the entry list waits for the next group. */
    std::vector<std::pair<uint32_t, uint32_t>> place(uniq_all.size());        /* distinct pattern -> (group, index in its uniq) */
    auto key_class = [](const std::string &p) {
        const uint32_t w24 = (uint8_t)p[0] | ((uint32_t)(uint8_t)p[1] << 8) | ((uint32_t)(uint8_t)p[2] << 16);
        return KMP_MULTI_HASH(w24 & KMP_MULTI_KEYMASK);
    };
    {
        const bool big = uniq_all.size() > KMP_MULTI_MAX_UNIQUE;
        std::vector<uint32_t> plain, classed;
        for (uint32_t u = 0; u < uniq_all.size(); u++) (!big || uniq_all[u].size() == 2 ? plain : classed).push_back(u);
        if (big && plain.empty() && !one_bytes.empty()) {                     /* the 1-byte patterns need a plain first group */
            const size_t take = std::min<size_t>(classed.size(), KMP_MULTI_MAX_UNIQUE);
            plain.assign(classed.begin(), classed.begin() + take);
            classed.erase(classed.begin(), classed.begin() + take);
        }
        for (uint32_t u : plain) {
            if (hg.empty() || hg.back().uniq.size() == KMP_MULTI_MAX_UNIQUE) hg.emplace_back();
            place[u] = {(uint32_t)hg.size() - 1u, (uint32_t)hg.back().uniq.size()};
            hg.back().uniq.push_back(uniq_all[u]); hg.back().gidx.push_back(u);
        }
        const size_t first_classed = hg.size();
        for (uint32_t u : classed) {
            const std::string &p = uniq_all[u];
            const uint32_t b = key_class(p), cl = b >> KMP_MULTI_CLS_SHIFT;
            size_t gi = first_classed;
            for (; gi < hg.size(); gi++) {
                HostGroup &h = hg[gi];
                if (h.uniq.size() < 4u * KMP_MULTI_MAX_UNIQUE && h.n_cls[cl] < 256u && h.overflow + (h.bucket_used[b] ? 1u : 0u) <= KMP_MULTI_MAX_ENTRIES) break;
            }
            if (gi == hg.size()) { hg.emplace_back(); hg.back().classed = true; hg.back().bucket_used.assign(KMP_MULTI_BUCKETS, 0); }
            HostGroup &h = hg[gi];
            h.n_cls[cl]++;
            if (h.bucket_used[b]) h.overflow++; else h.bucket_used[b] = 1;
            place[u] = {(uint32_t)gi, (uint32_t)h.uniq.size()};
            h.uniq.push_back(p); h.gidx.push_back(u);
        }
    }
    bool first_group = true;
    for (HostGroup &h : hg) {
        const uint32_t U = (uint32_t)h.uniq.size();
        const uint32_t n_ones = first_group ? (uint32_t)one_bytes.size() : 0u;
        uint32_t ones = 0;
        for (uint32_t k = 0; k < n_ones; k++) ones |= (uint32_t)one_bytes[k] << (8 * k);
        first_group = false;
        /* rows: class by class, short patterns (2 or 3 bytes: decided by their bucket entry alone) first in each (a plain group is one class) */
        uint32_t cls_short[NCLS] = {}, cls_n[NCLS] = {}, rec_base[NCLS] = {}, row_base[NCLS] = {};
        std::vector<uint32_t> cls_of(U, 0u), in_cls(U, 0u);
        h.row.assign(U, 0u);
        for (uint32_t u = 0; u < U; u++) {
            cls_of[u] = h.classed ? key_class(h.uniq[u]) >> KMP_MULTI_CLS_SHIFT : 0u;
            cls_n[cls_of[u]]++;
            if (h.uniq[u].size() <= KMP_MULTI_SHORT_LEN) cls_short[cls_of[u]]++;
        }
        for (uint32_t cl = 1; cl < NCLS; cl++) {
            row_base[cl] = row_base[cl - 1] + cls_n[cl - 1];
            rec_base[cl] = rec_base[cl - 1] + (cls_n[cl - 1] - cls_short[cl - 1]);
        }
        {
            uint32_t next_short[NCLS] = {}, next_long[NCLS];
            for (uint32_t cl = 0; cl < NCLS; cl++) next_long[cl] = cls_short[cl];
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t cl = cls_of[u];
                in_cls[u] = h.uniq[u].size() <= KMP_MULTI_SHORT_LEN ? next_short[cl]++ : next_long[cl]++;
                h.row[u] = row_base[cl] + in_cls[u];
            }
        }
        const uint32_t rows_n = U;
        const uint32_t n_long = rec_base[NCLS - 1] + (cls_n[NCLS - 1] - cls_short[NCLS - 1]);
        std::vector<uint32_t> tab(KMP_MULTI_REC_W0 + (h.classed ? KMP_MULTI_CLS_WORDS + (size_t)n_long * KMP_MULTI_CREC_WORDS : (size_t)n_long * KMP_MULTI_REC_WORDS), 0u);
        if (h.classed)
            for (uint32_t cl = 0; cl < NCLS; cl++) tab[KMP_MULTI_REC_W0 + cl] = KMP_MULTI_CLS_WORD(cls_short[cl], rec_base[cl], row_base[cl]);
        uint32_t *bucket = tab.data() + KMP_MULTI_BUCKET_W0;
        uint32_t *entry = tab.data() + KMP_MULTI_ENTRY_W0;
        std::vector<std::vector<uint32_t>> lists(KMP_MULTI_BUCKETS);
        uint32_t n_two = 0;
        for (uint32_t u = 0; u < U; u++) n_two += h.uniq[u].size() == 2 ? 1u : 0u;
        const uint32_t bmask = n_two <= KMP_MULTI_MAX_TWO ? KMP_MULTI_KEYMASK : 0xFFFFu;        /* bucket key: three bytes, or two when 2-byte patterns abound */
        for (uint32_t u = 0; u < U; u++) {
            const std::string &p = h.uniq[u];
            const uint32_t w16 = (uint8_t)p[0] | ((uint32_t)(uint8_t)p[1] << 8);
            uint32_t *pair = tab.data() + KMP_MULTI_FILTER_W0;
            if (p.size() >= 3) {
                pair[2u * KMP_MULTI_PAIR(p[1], p[2])]      |= 1u << ((uint8_t)p[0] & 31u);       /* p0 may stand before p1 p2 */
                pair[2u * KMP_MULTI_PAIR(p[0], p[1]) + 1u] |= 1u << ((uint8_t)p[2] & 31u);       /* p2 may follow p0 p1       */
            } else {
                for (uint32_t t = 0; t < 32u; t++) pair[2u * KMP_MULTI_PAIR(p[1], t)] |= 1u << ((uint8_t)p[0] & 31u);
                pair[2u * KMP_MULTI_PAIR(p[0], p[1]) + 1u] = 0xFFFFFFFFu;                       /* whatever follows        */
            }
            for (uint32_t t = 0; t < 32u; t++) {                      /* a 2-byte pattern matches whatever follows it */
                const uint32_t third = p.size() >= 3 ? (uint32_t)(uint8_t)p[2] : t;
                const uint32_t w24 = w16 | (third << 16);
                std::vector<uint32_t> &l = lists[KMP_MULTI_HASH(w24 & bmask)];
                if (l.empty() || l.back() != u) l.push_back(u);
                if (p.size() >= 3) break;
            }
            if (p.size() <= KMP_MULTI_SHORT_LEN) continue;
            const uint32_t cl = cls_of[u];
            if (h.classed) {
                uint32_t *rec = tab.data() + KMP_MULTI_REC_W0 + KMP_MULTI_CLS_WORDS + (size_t)(rec_base[cl] + in_cls[u] - cls_short[cl]) * KMP_MULTI_CREC_WORDS;
                rec[0] = (uint32_t)(uint8_t)p[3] | ((uint32_t)p.size() << 8) | (first_pat[h.gidx[u]] << 16);      /* byte 3 (the entry has bytes 0-2), the length, a pattern that has the rest */
                for (uint32_t b = 4; b < p.size() && b < 8u; b++) rec[1] |= (uint32_t)(uint8_t)p[b] << (8 * (b & 3));
            } else {
                uint32_t *rec = tab.data() + KMP_MULTI_REC_W0 + (size_t)(in_cls[u] - cls_short[0]) * KMP_MULTI_REC_WORDS;
                for (uint32_t b = 0; b < p.size() && b < 8u; b++) {
                    rec[b >> 2] |= (uint32_t)(uint8_t)p[b] << (8 * (b & 3));
                    if (b >= 4u) rec[2] |= 0xFFu << (8 * (b & 3));
                }
                rec[3] = (uint32_t)p.size() | (first_pat[h.gidx[u]] << 8);         /* the rest of it: kmp_pattern_dev[that index].pat */
            }
        }
        uint32_t pos = 0;
        for (uint32_t hh = 0; hh < KMP_MULTI_BUCKETS; hh++) {
            for (size_t q = 0; q < lists[hh].size(); q++) {
                const uint32_t u = lists[hh][q];
                const std::string &p = h.uniq[u];
                const uint32_t third = p.size() >= 3 ? (uint32_t)(uint8_t)p[2] : 0u;      /* never 0x00 inside a pattern */
                const uint32_t ent = (uint32_t)(uint8_t)p[0] | ((uint32_t)(uint8_t)p[1] << 8) | (third << 16) | (in_cls[u] << 24);
                if (q == 0) { bucket[2 * hh] = ent; continue; }       /* the first entry sits in the bucket itself */
                if (pos >= KMP_MULTI_MAX_ENTRIES) return fail(KMPGPU_EINVAL, "kmpgpu_set_patterns: fused tables: entry list overflow");
                entry[pos++] = ent;
            }
            const uint32_t extra = lists[hh].empty() ? 0u : (uint32_t)lists[hh].size() - 1u;
            bucket[2 * hh + 1] = (pos - extra) | ((uint32_t)lists[hh].size() << 16);
        }
        /* the patterns this group counts, and the row of each */
        for (const auto &e : elig) {
            const auto &pl = place[e.second];
            if (&hg[pl.first] != &h) continue;
            h.ids.push_back(e.first);
            h.rows.push_back(h.row[pl.second]);
        }
        /* the 1-byte patterns that ride along: rows behind the group's own */
        if (n_ones)
            for (const auto &oi : one_ids) { h.ids.push_back(oi.first); h.rows.push_back(rows_n + oi.second); }
        /* row -> the pattern indices that share it, for the offset records (duplicates are reported one by one) and for the
         * rest of a pattern of nine bytes or more (kmp_pattern_dev[first of them].pat) */
        std::vector<uint32_t> uid_first(rows_n + n_ones + 1, 0u), uid_ids(h.ids.size());
        for (uint32_t r : h.rows) uid_first[r + 1]++;
        for (uint32_t u = 0; u < rows_n + n_ones; u++) uid_first[u + 1] += uid_first[u];
        { std::vector<uint32_t> fill(uid_first.begin(), uid_first.end() - 1);
          for (size_t i = 0; i < h.ids.size(); i++) uid_ids[fill[h.rows[i]]++] = h.ids[i]; }
        c->fused_groups.emplace_back();
        kmpgpu_ctx::FusedGroup &g = c->fused_groups.back();
        auto up = [&](uint32_t **d, const std::vector<uint32_t> &v) -> hipError_t {
            hipError_t e = hipMalloc(d, (v.size() ? v.size() : 1) * sizeof(uint32_t));
            if (e == hipSuccess && !v.empty()) e = hipMemcpy(*d, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            return e;
        };
        HIP_TRY(up(&g.d_tables, tab));
        HIP_TRY(up(&g.d_ids, h.ids));
        HIP_TRY(up(&g.d_rows, h.rows));
        HIP_TRY(up(&g.d_uid_first, uid_first));
        HIP_TRY(up(&g.d_uid_ids, uid_ids));
        g.words = (uint32_t)tab.size(); g.n_unique = rows_n + n_ones; g.cshift = h.classed ? KMP_MULTI_CLS_SHIFT : cls_short[0]; g.classed = h.classed; g.bmask = bmask; g.n_ones = n_ones; g.ones = ones; g.n_ids = (uint32_t)h.ids.size();
        c->n_multi_unique += U;
    }
    std::vector<uint32_t> rest(rest_l);
    rest.insert(rest.end(), rest_s.begin(), rest_s.end());
    HIP_TRY(hipMalloc(&c->d_rest_ids, (rest.size() ? rest.size() : 1) * sizeof(uint32_t)));
    if (!rest.empty()) HIP_TRY(hipMemcpy(c->d_rest_ids, rest.data(), rest.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->rest_long = (uint32_t)rest_l.size(); c->rest_short = (uint32_t)rest_s.size();
    return KMPGPU_OK;
}

int kmpgpu_load_arena(kmpgpu_ctx *c, const uint8_t *arena, uint64_t arena_bytes, const uint64_t *pkt_off,
                      const uint32_t *pkt_len, uint64_t n_pkts)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_load_arena: ctx is NULL");
    if (n_pkts && (!arena || !pkt_off || !pkt_len)) return fail(KMPGPU_EINVAL, "kmpgpu_load_arena: NULL buffers");
    uint64_t payload = 0;
    for (uint64_t k = 0; k < n_pkts; k++) {         /* the layout contract the kernels rely on */
        const uint64_t o = pkt_off[k], l16 = ((uint64_t)pkt_len[k] + 15u) & ~15ull;
        if (o & 15u) return fail(KMPGPU_EINVAL, "payload %llu: offset %llu is not 16-byte aligned", (unsigned long long)k, (unsigned long long)o);
        if (pkt_len[k] >= (1u << 30)) return fail(KMPGPU_EINVAL, "payload %llu: length %u is not below 2^30", (unsigned long long)k, pkt_len[k]);
        if (o > arena_bytes || std::max<uint64_t>(l16, 16) > arena_bytes - o)
            return fail(KMPGPU_EINVAL, "payload %llu: [%llu, +%llu) padded to 16 B (at least one 16-byte slot) exceeds the arena (%llu B)", (unsigned long long)k,
                        (unsigned long long)o, (unsigned long long)pkt_len[k], (unsigned long long)arena_bytes);
        payload += pkt_len[k];
    }
    bool packed = n_pkts > 0;
    for (uint64_t k = 0; packed && k + 1 < n_pkts; k++)
        if (pkt_off[k + 1] != pkt_off[k] + std::max<uint64_t>(((uint64_t)pkt_len[k] + 15u) & ~15ull, 16)) packed = false;
    bool uniform = n_pkts > 0;
    uint64_t ustride = 0;
    if (uniform) {
        const uint64_t l16 = std::max<uint64_t>(((uint64_t)pkt_len[0] + 15u) & ~15ull, 16);
        ustride = n_pkts > 1 ? pkt_off[1] - pkt_off[0] : l16;
        if (n_pkts > 1 && pkt_off[1] < pkt_off[0]) uniform = false;
        if (ustride < l16 || (ustride & 15u) || ustride >= (1ull << 31)) uniform = false;
        for (uint64_t k = 0; uniform && k < n_pkts; k++)
            if (pkt_len[k] != pkt_len[0] || pkt_off[k] != pkt_off[0] + k * ustride) uniform = false;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    /* streamed captures load batch after batch: keep the device buffers when the next batch fits */
    const bool reuse = c->owned_arena && c->cap_arena >= arena_bytes && c->cap_pkts >= n_pkts && n_pkts > 0;
    release_arena(c, reuse);
    c->last.h2d_ms = 0; c->last.h2d_bytes = 0;
    if (n_pkts == 0) return KMPGPU_OK;
    if (arena_bytes < 16) return fail(KMPGPU_EINVAL, "arena smaller than 16 bytes");
    if (!reuse) {
        HIP_TRY(hipMalloc(&c->owned_arena, arena_bytes));
        HIP_TRY(hipMalloc(&c->owned_off, n_pkts * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&c->owned_len, n_pkts * sizeof(uint32_t)));
        c->cap_arena = arena_bytes; c->cap_pkts = n_pkts;
    }
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(upload_split(c->owned_arena, arena, arena_bytes, c->stream));
    HIP_TRY(hipMemcpyAsync(c->owned_off, pkt_off, n_pkts * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->owned_len, pkt_len, n_pkts * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->last.h2d_ms = ms;
    c->last.h2d_bytes = arena_bytes + n_pkts * (sizeof(uint64_t) + sizeof(uint32_t));
    c->d_arena = (const uint8_t *)c->owned_arena;
    c->d_off = (const uint64_t *)c->owned_off;
    c->d_len = (const uint32_t *)c->owned_len;
    c->arena_bytes = arena_bytes; c->n_pkts = n_pkts; c->payload_bytes = payload;
    c->uniform = uniform; c->uni_off0 = pkt_off[0]; c->uni_stride = (uint32_t)ustride; c->uni_len = pkt_len[0];
    c->packed = packed;
    c->span_end = pkt_off[n_pkts - 1] + std::max<uint64_t>(((uint64_t)pkt_len[n_pkts - 1] + 15u) & ~15ull, 16);
    return prepare_packed(c);
}

/* Shared tail of the loaders that produce the index on the device: contract + uniform/packed
 * detection + payload sum from kmp_validate_index_kernel, then the packed kernels' side tables. */
static int finish_device_index(kmpgpu_ctx *c, const char *who)
{
    HIP_TRY(hipMemsetAsync(c->d_err, 0, 2 * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_sum, 0, 6 * sizeof(unsigned long long), c->stream));
    HIP_TRY(kmp_launch_validate(c->d_off, c->d_len, c->n_pkts, c->arena_bytes, c->d_err, c->d_sum, c->stream));
    uint32_t err[2] = {0, 0};
    unsigned long long info[6] = {0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(c->h_small, c->d_sum, sizeof info, hipMemcpyDeviceToHost, c->stream));          /* (pinned: no staging) */
    HIP_TRY(hipMemcpyAsync(c->h_small + 8, c->d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(info, c->h_small, sizeof info);
    memcpy(err, c->h_small + 8, sizeof err);
    if (err[0]) return fail(KMPGPU_EINVAL, "%s: the payload index violates the layout contract (flags %u)", who, err[0]);
    c->payload_bytes = info[0];
    c->uniform = ((err[1] & 1u) == 0) && info[2] >= 16 && info[2] < (1ull << 31);
    c->uni_off0 = info[1]; c->uni_stride = (uint32_t)info[2]; c->uni_len = (uint32_t)info[3];
    c->packed = (err[1] & 2u) == 0;
    c->span_end = info[4];
    return prepare_packed(c);
}

int kmpgpu_load_frames_begin(kmpgpu_ctx *c, const uint8_t *file_bytes, uint64_t file_nbytes, const uint64_t *frame_off,
                             const uint32_t *frame_caplen, uint64_t n_frames, int tcp)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_load_frames: ctx is NULL");
    if (n_frames && (!file_bytes || !frame_off || !frame_caplen)) return fail(KMPGPU_EINVAL, "kmpgpu_load_frames: NULL buffers");
    if (c->fr_pending) return fail(KMPGPU_ESTATE, "kmpgpu_load_frames_begin: the previous load has not been finished");
    /* Only the bytes these frames span are uploaded: a shard of the frames (mpi_dumping.c:149-161 scatters shares, not
     * the whole capture) or a batch of a streamed capture (openmp_task.c:126-155) costs its share of PCIe time and HBM. */
    uint64_t span_lo = file_nbytes, span_hi = 0;
    for (uint64_t f = 0; f < n_frames; f++) {
        if (frame_off[f] > file_nbytes || frame_caplen[f] > file_nbytes - frame_off[f])
            return fail(KMPGPU_EINVAL, "kmpgpu_load_frames: frame %llu lies outside the file buffer", (unsigned long long)f);
        span_lo = std::min<uint64_t>(span_lo, frame_off[f]);
        span_hi = std::max<uint64_t>(span_hi, frame_off[f] + frame_caplen[f]);
    }
    if (span_hi < span_lo) span_lo = span_hi = 0;
    span_lo &= ~(uint64_t)15;                           /* keeps the frames' alignment relative to the device buffer */
    const uint64_t span = span_hi - span_lo;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));           /* the passes over the arena that is about to be replaced */
    release_arena(c, /* keep_buffers = */ true);        /* batch after batch: device buffers are reused when the next batch fits */
    c->last.h2d_ms = 0; c->last.h2d_bytes = 0;
    c->fr_pending = true; c->fr_n = n_frames; c->fr_tcp = tcp; c->fr_span = span; c->fr_span_lo = span_lo;
    if (n_frames == 0) return KMPGPU_OK;

    /* scratch, grown on demand and kept (no hipMalloc / hipFree per batch: either synchronises the whole device) */
    HIP_TRY(grow_buffer(&c->fr_file, &c->fr_file_cap, span + 64));
    HIP_TRY(grow_buffer(&c->fr_off, &c->fr_off_cap, n_frames));
    HIP_TRY(grow_buffer(&c->fr_cl, &c->fr_cl_cap, n_frames));
    HIP_TRY(grow_buffer(&c->fr_ws, &c->fr_ws_cap, (uint64_t)kmp_extract_ws_bytes(n_frames)));
    if (!c->fr_tot) HIP_TRY(hipMalloc(&c->fr_tot, 2 * sizeof(unsigned long long)));

    /* the device buffer holds the file's bytes [span_lo, span_hi): the kernels address it through the pointer that stands for
     * the file's first byte, so the frame offsets go up as they are (no rebased copy of them on the host) */
    const uint8_t *d_file0 = c->fr_file - span_lo;
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(upload_split(c->fr_file, file_bytes + span_lo, span, c->stream));
    HIP_TRY(hipMemcpyAsync(c->fr_off, frame_off, n_frames * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->fr_cl, frame_caplen, n_frames * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(kmp_launch_extract_phase1(d_file0, c->fr_off, c->fr_cl, n_frames, tcp, c->fr_ws, c->fr_tot, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_small, c->fr_tot, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));         /* (pinned: no staging) */
    return KMPGPU_OK;
}

int kmpgpu_load_frames_uploaded(kmpgpu_ctx *c)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_load_frames_uploaded: ctx is NULL");
    if (!c->fr_pending || c->fr_n == 0) return KMPGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev[1]));
    return KMPGPU_OK;
}

int kmpgpu_load_frames_finish(kmpgpu_ctx *c, uint64_t *n_payloads)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_load_frames: ctx is NULL");
    if (n_payloads) *n_payloads = 0;
    if (!c->fr_pending) return fail(KMPGPU_ESTATE, "kmpgpu_load_frames_finish: no load has been begun");
    c->fr_pending = false;
    const uint64_t n_frames = c->fr_n, span = c->fr_span;
    if (n_frames == 0) return KMPGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long tot[2] = {0, 0};
    memcpy(tot, c->h_small, sizeof tot);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->last.h2d_ms = ms;
    c->last.h2d_bytes = span + n_frames * (sizeof(uint64_t) + sizeof(uint32_t));
    const uint64_t n_pkts = tot[1], arena_bytes = tot[0] + 64;
    if (n_payloads) *n_payloads = n_pkts;
    if (n_pkts == 0) return KMPGPU_OK;
    if (!(c->owned_arena && c->cap_arena >= arena_bytes && c->cap_pkts >= n_pkts)) {
        if (c->owned_arena) HIP_TRY(hipFree(c->owned_arena));
        if (c->owned_off) HIP_TRY(hipFree(c->owned_off));
        if (c->owned_len) HIP_TRY(hipFree(c->owned_len));
        c->owned_arena = c->owned_off = c->owned_len = nullptr; c->cap_arena = c->cap_pkts = 0;
        const uint64_t take_b = arena_bytes + arena_bytes / 8, take_n = n_pkts + n_pkts / 8;
        HIP_TRY(hipMalloc(&c->owned_arena, take_b));
        HIP_TRY(hipMalloc(&c->owned_off, take_n * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&c->owned_len, take_n * sizeof(uint32_t)));
        c->cap_arena = take_b; c->cap_pkts = take_n;
    }
    HIP_TRY(grow_buffer(&c->fr_src, &c->fr_src_cap, n_pkts));
    const uint8_t *d_file0 = c->fr_file - c->fr_span_lo;
    HIP_TRY(hipMemsetAsync((uint8_t *)c->owned_arena + tot[0], 0, 64, c->stream));
    HIP_TRY(kmp_launch_extract_phase2(d_file0, c->fr_off, n_frames, c->fr_ws, n_pkts, (uint8_t *)c->owned_arena, (uint64_t *)c->owned_off,
                                      (uint32_t *)c->owned_len, c->fr_src, c->stream));
    c->d_arena = (const uint8_t *)c->owned_arena;
    c->d_off = (const uint64_t *)c->owned_off;
    c->d_len = (const uint32_t *)c->owned_len;
    c->arena_bytes = arena_bytes; c->n_pkts = n_pkts;
    c->pad_known_clean = true;                          /* kmp_gather_kernel writes every slot whole: payload, then 0x00 up to the slot's end */
    const int rc = finish_device_index(c, "kmpgpu_load_frames");
    c->pad_known_clean = false;
    /* one capture uploaded whole: its bytes are not kept around (a streamed capture's batches are small and the next one
     * reuses the buffer) */
    if (c->fr_file_cap > (1ull << 30)) { (void)hipFree(c->fr_file); c->fr_file = nullptr; c->fr_file_cap = 0; }
    return rc;
}

int kmpgpu_load_frames(kmpgpu_ctx *c, const uint8_t *file_bytes, uint64_t file_nbytes, const uint64_t *frame_off,
                       const uint32_t *frame_caplen, uint64_t n_frames, int tcp, uint64_t *n_payloads)
{
    if (n_payloads) *n_payloads = 0;
    const int rc = kmpgpu_load_frames_begin(c, file_bytes, file_nbytes, frame_off, frame_caplen, n_frames, tcp);
    return rc ? rc : kmpgpu_load_frames_finish(c, n_payloads);
}

int kmpgpu_reserve(kmpgpu_ctx *c, uint64_t arena_bytes, uint64_t n_pkts, uint64_t frame_bytes, uint64_t n_frames)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_reserve: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (arena_bytes && n_pkts && !(c->owned_arena && c->cap_arena >= arena_bytes && c->cap_pkts >= n_pkts)) {
        if (c->d_arena == (const uint8_t *)c->owned_arena) release_arena(c, true);        /* the arena in use lives in these buffers */
        if (c->owned_arena) HIP_TRY(hipFree(c->owned_arena));
        if (c->owned_off) HIP_TRY(hipFree(c->owned_off));
        if (c->owned_len) HIP_TRY(hipFree(c->owned_len));
        c->owned_arena = c->owned_off = c->owned_len = nullptr; c->cap_arena = c->cap_pkts = 0;
        HIP_TRY(hipMalloc(&c->owned_arena, arena_bytes));
        HIP_TRY(hipMalloc(&c->owned_off, n_pkts * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&c->owned_len, n_pkts * sizeof(uint32_t)));
        c->cap_arena = arena_bytes; c->cap_pkts = n_pkts;
    }
    if (arena_bytes) {
        const uint64_t words = arena_bytes / KMP_CHUNK + 32;
        if (c->bitmap_cap < words) {
            const bool live = c->bitmap_live;
            if (live) return fail(KMPGPU_ESTATE, "kmpgpu_reserve: an arena larger than the reserved size is attached");
            if (c->d_bitmap) HIP_TRY(hipFree(c->d_bitmap));
            c->d_bitmap = nullptr; c->bitmap_cap = 0;
            HIP_TRY(hipMalloc(&c->d_bitmap, words * sizeof(unsigned long long)));
            c->bitmap_cap = words;
        }
    }
    if (frame_bytes && n_frames) {
        HIP_TRY(grow_buffer(&c->fr_file, &c->fr_file_cap, frame_bytes + 64));
        HIP_TRY(grow_buffer(&c->fr_off, &c->fr_off_cap, n_frames));
        HIP_TRY(grow_buffer(&c->fr_cl, &c->fr_cl_cap, n_frames));
        HIP_TRY(grow_buffer(&c->fr_ws, &c->fr_ws_cap, (uint64_t)kmp_extract_ws_bytes(n_frames)));
        HIP_TRY(grow_buffer(&c->fr_src, &c->fr_src_cap, n_frames));
        if (!c->fr_tot) HIP_TRY(hipMalloc(&c->fr_tot, 2 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(c->fr_file, 0, frame_bytes + 64, c->stream));      /* first touch now: the first upload into fresh device memory runs at 3/4 of the rate */
    }
    if (c->owned_arena && arena_bytes && c->d_arena != (const uint8_t *)c->owned_arena) HIP_TRY(hipMemsetAsync(c->owned_arena, 0, std::min<uint64_t>(arena_bytes, c->cap_arena), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return KMPGPU_OK;
}

int kmpgpu_attach_arena(kmpgpu_ctx *c, const void *d_arena, uint64_t arena_bytes, const void *d_pkt_off,
                        const void *d_pkt_len, uint64_t n_pkts)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_attach_arena: ctx is NULL");
    if (n_pkts && (!d_arena || !d_pkt_off || !d_pkt_len)) return fail(KMPGPU_EINVAL, "kmpgpu_attach_arena: NULL buffers");
    if (((uintptr_t)d_arena & 15u) || ((uintptr_t)d_pkt_off & 7u) || ((uintptr_t)d_pkt_len & 3u))
        return fail(KMPGPU_EINVAL, "kmpgpu_attach_arena: misaligned device pointer");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    release_arena(c);
    if (n_pkts == 0) return KMPGPU_OK;
    if (arena_bytes < 16) return fail(KMPGPU_EINVAL, "arena smaller than 16 bytes");
    HIP_TRY(hipMemsetAsync(c->d_err, 0, 2 * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_sum, 0, 6 * sizeof(unsigned long long), c->stream));
    HIP_TRY(kmp_launch_validate((const uint64_t *)d_pkt_off, (const uint32_t *)d_pkt_len, n_pkts, arena_bytes, c->d_err, c->d_sum, c->stream));
    uint32_t err[2] = {0, 0};
    unsigned long long info[6] = {0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(err, c->d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(info, c->d_sum, sizeof info, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err[0] & 1u) return fail(KMPGPU_EINVAL, "kmpgpu_attach_arena: a payload offset is not 16-byte aligned");
    if (err[0] & 2u) return fail(KMPGPU_EINVAL, "kmpgpu_attach_arena: a payload (padded to 16 B) exceeds the arena");
    if (err[0] & 4u) return fail(KMPGPU_EINVAL, "kmpgpu_attach_arena: a payload length is not below 2^30");
    c->d_arena = (const uint8_t *)d_arena;
    c->d_off = (const uint64_t *)d_pkt_off;
    c->d_len = (const uint32_t *)d_pkt_len;
    c->arena_bytes = arena_bytes; c->n_pkts = n_pkts; c->payload_bytes = info[0];
    c->uniform = ((err[1] & 1u) == 0) && info[2] >= 16 && info[2] < (1ull << 31);
    c->uni_off0 = info[1]; c->uni_stride = (uint32_t)info[2]; c->uni_len = (uint32_t)info[3];
    c->packed = (err[1] & 2u) == 0;
    c->span_end = info[4];
    return prepare_packed(c);
}

int kmpgpu_scan_enqueue(kmpgpu_ctx *c, void *d_counts_out)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_scan_enqueue: ctx is NULL");
    if ((uintptr_t)d_counts_out & 7u) return fail(KMPGPU_EINVAL, "kmpgpu_scan_enqueue: d_counts_out is not 8-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    return enqueue_pass(c, nullptr, (unsigned long long *)d_counts_out);
}

void *kmpgpu_counts_device(kmpgpu_ctx *c) { return c ? (void *)c->d_counts : nullptr; }

int kmpgpu_counts_reset(kmpgpu_ctx *c)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_counts_reset: ctx is NULL");
    if (!c->d_counts) return fail(KMPGPU_ESTATE, "kmpgpu_counts_reset: no patterns set");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->d_counts, 0, sizeof(unsigned long long) * (c->n_pat ? c->n_pat : 1), c->stream));
    return KMPGPU_OK;
}

int kmpgpu_counts_add(kmpgpu_ctx *dst, kmpgpu_ctx *src)
{
    if (!dst || !src || dst == src) return fail(KMPGPU_EINVAL, "kmpgpu_counts_add: bad arguments");
    if (dst->device != src->device) return fail(KMPGPU_EINVAL, "kmpgpu_counts_add: the contexts sit on devices %d and %d (sum across devices with kmpgpu_comm_allreduce_counts)", dst->device, src->device);
    if (!dst->d_counts || !src->d_counts || dst->n_pat != src->n_pat) return fail(KMPGPU_ESTATE, "kmpgpu_counts_add: the contexts do not hold the same patterns");
    HIP_TRY(hipSetDevice(dst->device));
    HIP_TRY(hipStreamSynchronize(src->stream));                 /* src's passes have landed in its counters */
    HIP_TRY(kmp_launch_add_counts(dst->d_counts, src->d_counts, dst->n_pat, dst->stream));
    return KMPGPU_OK;
}

int kmpgpu_last_timing(kmpgpu_ctx *c, kmpgpu_timing *t)
{
    if (!c || !t) return fail(KMPGPU_EINVAL, "kmpgpu_last_timing: NULL argument");
    *t = c->last;
    return KMPGPU_OK;
}

int kmpgpu_counts_read(kmpgpu_ctx *c, uint64_t *counts_out)
{
    if (!c || (!counts_out && c->n_pat)) return fail(KMPGPU_EINVAL, "kmpgpu_counts_read: NULL argument");
    if (!c->d_counts) return fail(KMPGPU_ESTATE, "kmpgpu_counts_read: no patterns set");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, sizeof(uint64_t) * c->n_pat, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->n_pat) memcpy(counts_out, c->h_counts, sizeof(uint64_t) * c->n_pat);
    return KMPGPU_OK;
}

int kmpgpu_sync(kmpgpu_ctx *c)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_sync: ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return KMPGPU_OK;
}

int kmpgpu_scan(kmpgpu_ctx *c, uint64_t *counts_out, kmpgpu_timing *t)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_scan: ctx is NULL");
    if (!counts_out && c->n_pat) return fail(KMPGPU_EINVAL, "kmpgpu_scan: counts_out is NULL");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t launches = 0;
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    int rc = enqueue_pass(c, &launches, nullptr);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, sizeof(uint64_t) * c->n_pat, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float k_ms = 0, d_ms = 0;
    HIP_TRY(hipEventElapsedTime(&k_ms, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&d_ms, c->ev[1], c->ev[2]));
    memcpy(counts_out, c->h_counts, sizeof(uint64_t) * c->n_pat);
    c->last.kernel_ms = k_ms; c->last.d2h_ms = d_ms; c->last.launches = launches;
    c->last.grid_blocks = c->n_pkts ? grid_blocks(c) : 0;
    if (t) *t = c->last;
    return KMPGPU_OK;
}

int kmpgpu_profile_begin(kmpgpu_ctx *c, uint32_t max_launches)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_profile_begin: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    while (c->prof_ev.size() < 2 * (size_t)max_launches) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->prof_ev.push_back(e);
    }
    c->prof_cap = max_launches; c->prof_n = 0; c->profiling = true;
    return KMPGPU_OK;
}

int kmpgpu_profile_end(kmpgpu_ctx *c, float *ms_out, uint32_t *n)
{
    if (!c || !n) return fail(KMPGPU_EINVAL, "kmpgpu_profile_end: NULL argument");
    c->profiling = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < c->prof_n; i++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
        if (ms_out) ms_out[i] = ms;
    }
    *n = c->prof_n;
    c->prof_n = 0;
    return KMPGPU_OK;
}

int kmpgpu_scan_offsets(kmpgpu_ctx *c, kmpgpu_match *out, uint64_t cap, uint64_t *n_found, uint64_t *counts_out)
{
    if (!c || !n_found) return fail(KMPGPU_EINVAL, "kmpgpu_scan_offsets: NULL argument");
    if (cap && !out) return fail(KMPGPU_EINVAL, "kmpgpu_scan_offsets: out is NULL");
    if (c->mode != 0 || c->kernel_sel == 1) return fail(KMPGPU_EINVAL, "kmpgpu_scan_offsets runs on the streaming kernels only (mode 0, kernel 0, 2 or 3)");
    static_assert(sizeof(kmpgpu_match) == 16, "kmpgpu_match is a 16-byte record");
    HIP_TRY(hipSetDevice(c->device));
    *n_found = 0;
    if (!c->packed && c->n_pkts) {
        /* an arena kept in place (KMPGPU_OPT_REPACK = 0) whose slots are not back to back: the offsets come from the
         * streaming kernels, so it is packed now, once (the context scans its packed copy from here on) */
        HIP_TRY(hipStreamSynchronize(c->stream));
        const int keep = c->repack;
        c->repack = 1;
        const int rr = prepare_packed(c);
        c->repack = keep;
        if (rr) return rr;
    }
    void *d_out = nullptr;
    unsigned long long *d_cnt = nullptr;             /* [0] matches found; [1 ..] this pass's counts */
    const size_t np = c->n_pat ? c->n_pat : 1;
    HIP_TRY(hipMalloc(&d_out, (cap ? cap : 1) * sizeof(kmpgpu_match)));
    hipError_t e = hipMalloc(&d_cnt, (1 + np) * sizeof(unsigned long long));
    int rc = KMPGPU_OK;
    unsigned long long found = 0;
    if (e != hipSuccess) rc = fail(KMPGPU_EHIP, "hipMalloc failed: %s", hipGetErrorString(e));
    if (!rc && (e = hipMemsetAsync(d_cnt, 0, (1 + np) * sizeof(unsigned long long), c->stream)) != hipSuccess)
        rc = fail(KMPGPU_EHIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    if (!rc) {
        /* The pass writes its counts to a buffer of its own and never accumulates: the context's counters (a running
         * total under KMPGPU_OPT_ACCUMULATE, or the result of a count reduce) are left as they are. */
        EmitTarget t;
        t.out = d_out; t.counter = d_cnt; t.cap = cap;
        const int acc = c->accumulate;
        c->accumulate = 0;
        rc = enqueue_pass(c, nullptr, d_cnt + 1, &t);
        c->accumulate = acc;
    }
    if (!rc && (e = hipMemcpyAsync(&found, d_cnt, sizeof found, hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        rc = fail(KMPGPU_EHIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
    if (!rc && c->n_pat && counts_out &&
        (e = hipMemcpyAsync(c->h_counts, d_cnt + 1, sizeof(uint64_t) * c->n_pat, hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        rc = fail(KMPGPU_EHIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
    if (!rc && (e = hipStreamSynchronize(c->stream)) != hipSuccess) rc = fail(KMPGPU_EHIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    if (!rc) {
        const unsigned long long n = found < cap ? found : cap;
        if (n && (e = hipMemcpy(out, d_out, n * sizeof(kmpgpu_match), hipMemcpyDeviceToHost)) != hipSuccess)
            rc = fail(KMPGPU_EHIP, "hipMemcpy failed: %s", hipGetErrorString(e));
        if (counts_out && c->n_pat) memcpy(counts_out, c->h_counts, sizeof(uint64_t) * c->n_pat);
        *n_found = found;
    }
    (void)hipFree(d_out);
    if (d_cnt) (void)hipFree(d_cnt);
    return rc;
}

int kmpgpu_synth_fill(kmpgpu_ctx *c, void *d_arena, const void *d_pkt_off, const void *d_pkt_len, uint64_t first_pkt_id,
                      uint64_t n_pkts, const kmp_synth_params *sp)
{
    if (!c || !sp) return fail(KMPGPU_EINVAL, "kmpgpu_synth_fill: NULL argument");
    if (n_pkts && (!d_arena || !d_pkt_off || !d_pkt_len)) return fail(KMPGPU_EINVAL, "kmpgpu_synth_fill: NULL buffers");
    if (sp->needle_len > KMP_SYNTH_MAX_NEEDLE || sp->span == 0 || sp->span > 256 || sp->lo + sp->span > 256)
        return fail(KMPGPU_EINVAL, "kmpgpu_synth_fill: bad generator parameters");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(kmp_launch_synth_fill((uint8_t *)d_arena, (const uint64_t *)d_pkt_off, (const uint32_t *)d_pkt_len, first_pkt_id,
                                  n_pkts, *sp, c->stream));
    return KMPGPU_OK;
}

int kmpgpu_fixed_index(kmpgpu_ctx *c, void *d_pkt_off, void *d_pkt_len, uint64_t n_pkts, uint32_t len, uint32_t slot_align)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_fixed_index: ctx is NULL");
    if (slot_align < 16 || (slot_align & (slot_align - 1))) return fail(KMPGPU_EINVAL, "slot_align must be a power of two >= 16");
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t stride = ((uint64_t)len + slot_align - 1) & ~((uint64_t)slot_align - 1);
    HIP_TRY(kmp_launch_fixed_index((uint64_t *)d_pkt_off, (uint32_t *)d_pkt_len, n_pkts, len, stride ? stride : slot_align, c->stream));
    return KMPGPU_OK;
}

int kmpgpu_arena_info(kmpgpu_ctx *c, uint64_t *n_pkts, uint64_t *payload_bytes)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_arena_info: ctx is NULL");
    if (n_pkts) *n_pkts = c->n_pkts;
    if (payload_bytes) *payload_bytes = c->payload_bytes;
    return KMPGPU_OK;
}

int kmpgpu_effective_bytes(kmpgpu_ctx *c, uint64_t *bytes_out)
{
    if (!c || !bytes_out) return fail(KMPGPU_EINVAL, "kmpgpu_effective_bytes: NULL argument");
    *bytes_out = 0;
    if (c->n_pkts == 0) return KMPGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long *d = nullptr, h = 0ull;
    HIP_TRY(hipMalloc(&d, sizeof h));
    hipError_t e = hipMemsetAsync(d, 0, sizeof h, c->stream);
    if (e == hipSuccess) e = kmp_launch_effective_bytes(c->d_arena, c->d_off, c->d_len, c->n_pkts, d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(KMPGPU_EHIP, "kmpgpu_effective_bytes: %s", hipGetErrorString(e));
    *bytes_out = h;
    return KMPGPU_OK;
}

int kmpgpu_arena_download(kmpgpu_ctx *c, uint8_t *arena_out, uint64_t arena_cap, uint64_t *arena_bytes, uint64_t *pkt_off_out,
                          uint32_t *pkt_len_out)
{
    if (!c) return fail(KMPGPU_EINVAL, "kmpgpu_arena_download: ctx is NULL");
    if (arena_bytes) *arena_bytes = c->arena_bytes;
    if (c->n_pkts == 0) return KMPGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (arena_out) {
        if (arena_cap < c->arena_bytes) return fail(KMPGPU_EINVAL, "kmpgpu_arena_download: buffer too small");
        HIP_TRY(hipMemcpy(arena_out, c->d_arena, c->arena_bytes, hipMemcpyDeviceToHost));
    }
    if (pkt_off_out) HIP_TRY(hipMemcpy(pkt_off_out, c->d_off, c->n_pkts * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (pkt_len_out) HIP_TRY(hipMemcpy(pkt_len_out, c->d_len, c->n_pkts * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return KMPGPU_OK;
}

}  // extern "C"

/* ---- RCCL count reduce (mpi_dumping.c:202) -------------------------------------------------------------- */
namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load()
{
    static std::mutex mu;           /* contexts are per thread (bin/openmp_data brings its shards up on threads); the library handle is not */
    std::lock_guard<std::mutex> lock(mu);
    static int state = 0;           /* 0 untried, 1 loaded, -1 failed */
    if (state == 1) return KMPGPU_OK;
    if (state == -1) return fail(KMPGPU_EHIP, "librccl.so could not be loaded");
    /* RCCL writes its version banner and NCCL_DEBUG output to stdout unless told otherwise; stdout belongs to the
     * caller (the drop-in programs' report, serial.c:163-169, is compared byte for byte) */
    setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) { state = -1; return fail(KMPGPU_EHIP, "cannot load librccl.so: %s", dlerror()); }
    g_rccl.handle = h;
#define KMP_RCCL_SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); \
        if (!g_rccl.field) { state = -1; return fail(KMPGPU_EHIP, "librccl.so lacks %s", name); } } while (0)
    KMP_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    KMP_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    KMP_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    KMP_RCCL_SYM(AllReduce, "ncclAllReduce");
    KMP_RCCL_SYM(GroupStart, "ncclGroupStart");
    KMP_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    KMP_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    KMP_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef KMP_RCCL_SYM
    state = 1;
    return KMPGPU_OK;
}
/* RCCL prints its version banner (NCCL_DEBUG=VERSION and above) with printf on the first communicator: stdout belongs
 * to the caller -- the drop-in programs' report (serial.c:163-169) is compared byte for byte -- so file descriptor 1
 * points at stderr while a communicator is being created.  Guards may overlap (one thread per GPU inside
 * kmpgpu_comm_init_rank, which blocks until every rank has joined): the first one in saves and redirects, the last one
 * out restores, under a mutex.  (Another thread of the caller that writes to stdout in exactly that window lands on
 * stderr too.) */
struct StdoutToStderr {
    static std::mutex &mu() { static std::mutex m; return m; }
    static int &depth() { static int d = 0; return d; }
    static int &saved() { static int fd = -1; return fd; }
    StdoutToStderr()
    {
        std::lock_guard<std::mutex> lock(mu());
        if (depth()++ != 0) return;
        fflush(stdout);
        saved() = dup(1);
        if (saved() >= 0 && dup2(2, 1) < 0) { close(saved()); saved() = -1; }
    }
    ~StdoutToStderr()
    {
        std::lock_guard<std::mutex> lock(mu());
        if (--depth() != 0 || saved() < 0) return;
        fflush(stdout);
        (void)dup2(saved(), 1);
        close(saved());
        saved() = -1;
    }
};
#define RCCL_TRY(expr)                                                                                        \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return fail(KMPGPU_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)
}  // namespace

struct kmpgpu_comm {
    std::vector<kmpgpu_ctx *> ctx;          /* the local ranks' contexts (nullptr: destroyed before the communicator) */
    std::vector<ncclComm_t>   comm;         /* one communicator handle per local rank */
    std::vector<int>          device;       /* copies: kmpgpu_comm_destroy must not need the contexts */
    int n_ranks = 0;
};

static void comm_forget(kmpgpu_comm *k, kmpgpu_ctx *c)
{
    for (size_t i = 0; i < k->ctx.size(); i++)
        if (k->ctx[i] == c) {
            /* the rank's collectives were enqueued on the context's stream, which is about to go: let them finish */
            (void)hipSetDevice(c->device);
            if (c->stream) (void)hipStreamSynchronize(c->stream);
            k->ctx[i] = nullptr;
        }
    c->comm = nullptr;
}

extern "C" {

int kmpgpu_device_of(kmpgpu_ctx *c) { return c ? c->device : fail(KMPGPU_EINVAL, "kmpgpu_device_of: ctx is NULL"); }

int kmpgpu_comm_init(kmpgpu_comm **out, kmpgpu_ctx *const *ctx, int n_ctx)
{
    if (!out || !ctx || n_ctx < 1) return fail(KMPGPU_EINVAL, "kmpgpu_comm_init: bad arguments");
    *out = nullptr;
    std::vector<int> devs;
    for (int i = 0; i < n_ctx; i++) {
        if (!ctx[i]) return fail(KMPGPU_EINVAL, "kmpgpu_comm_init: context %d is NULL", i);
        if (ctx[i]->comm) return fail(KMPGPU_EINVAL, "kmpgpu_comm_init: context %d already belongs to a communicator", i);
        if (ctx[i]->n_pat != ctx[0]->n_pat) return fail(KMPGPU_EINVAL, "kmpgpu_comm_init: context %d holds %u patterns, context 0 %u", i, ctx[i]->n_pat, ctx[0]->n_pat);
        for (int d : devs)
            if (d == ctx[i]->device)
                return fail(KMPGPU_EINVAL, "kmpgpu_comm_init: two contexts on device %d (one rank per device: sum the counts of contexts that share a GPU on the host)", d);
        devs.push_back(ctx[i]->device);
    }
    int rc = rccl_load();
    if (rc) return rc;
    kmpgpu_comm *k = new (std::nothrow) kmpgpu_comm();
    if (!k) return fail(KMPGPU_ENOMEM, "kmpgpu_comm_init: out of host memory");
    k->ctx.assign(ctx, ctx + n_ctx);
    k->comm.assign((size_t)n_ctx, nullptr);
    k->n_ranks = n_ctx;
    ncclResult_t r;
    { StdoutToStderr guard; r = g_rccl.CommInitAll(k->comm.data(), n_ctx, devs.data()); }
    if (r != ncclSuccess) { delete k; return fail(KMPGPU_EHIP, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(r)); }
    k->device = devs;
    for (int i = 0; i < n_ctx; i++) ctx[i]->comm = k;
    *out = k;
    return KMPGPU_OK;
}

int kmpgpu_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(KMPGPU_EINVAL, "kmpgpu_comm_unique_id: NULL argument");
    static_assert(sizeof(ncclUniqueId) == KMPGPU_COMM_ID_BYTES, "KMPGPU_COMM_ID_BYTES");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    { StdoutToStderr guard; RCCL_TRY(g_rccl.GetUniqueId(&id)); }
    memcpy(id_out, &id, sizeof id);
    return KMPGPU_OK;
}

int kmpgpu_comm_init_rank(kmpgpu_comm **out, kmpgpu_ctx *ctx, int n_ranks, int rank, const void *unique_id)
{
    if (!out || !ctx || !unique_id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(KMPGPU_EINVAL, "kmpgpu_comm_init_rank: bad arguments");
    *out = nullptr;
    if (ctx->comm) return fail(KMPGPU_EINVAL, "kmpgpu_comm_init_rank: the context already belongs to a communicator");
    int rc = rccl_load();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    kmpgpu_comm *k = new (std::nothrow) kmpgpu_comm();
    if (!k) return fail(KMPGPU_ENOMEM, "kmpgpu_comm_init_rank: out of host memory");
    k->ctx.push_back(ctx);
    k->comm.push_back(nullptr);
    k->n_ranks = n_ranks;
    ncclResult_t r;
    { StdoutToStderr guard; r = g_rccl.CommInitRank(&k->comm[0], n_ranks, id, rank); }
    if (r != ncclSuccess) { delete k; return fail(KMPGPU_EHIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
    k->device.push_back(ctx->device);
    ctx->comm = k;
    *out = k;
    return KMPGPU_OK;
}

int kmpgpu_comm_allreduce_counts(kmpgpu_comm *k)
{
    if (!k) return fail(KMPGPU_EINVAL, "kmpgpu_comm_allreduce_counts: comm is NULL");
    for (kmpgpu_ctx *c : k->ctx)
        if (!c) return fail(KMPGPU_ESTATE, "kmpgpu_comm_allreduce_counts: a context of this communicator has been destroyed");
    const uint32_t n = k->ctx[0]->n_pat;
    for (kmpgpu_ctx *c : k->ctx) {
        if (!c->d_counts) return fail(KMPGPU_ESTATE, "kmpgpu_comm_allreduce_counts: a context has no patterns set");
        if (c->n_pat != n) return fail(KMPGPU_EINVAL, "kmpgpu_comm_allreduce_counts: the contexts hold different numbers of patterns");
    }
    if (n == 0) return KMPGPU_OK;
    RCCL_TRY(g_rccl.GroupStart());
    for (size_t i = 0; i < k->ctx.size(); i++) {
        kmpgpu_ctx *c = k->ctx[i];
        ncclResult_t r = g_rccl.AllReduce(c->d_counts, c->d_counts, n, ncclUint64, ncclSum, k->comm[i], c->stream);
        if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(KMPGPU_EHIP, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r)); }
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return KMPGPU_OK;
}

void kmpgpu_comm_destroy(kmpgpu_comm *k)
{
    if (!k) return;
    for (size_t i = 0; i < k->comm.size(); i++) {
        if (k->ctx[i]) {                                    /* (a context destroyed earlier has waited for its stream itself) */
            (void)hipSetDevice(k->ctx[i]->device);
            (void)hipStreamSynchronize(k->ctx[i]->stream);
            k->ctx[i]->comm = nullptr;
        }
        if (!k->comm[i]) continue;
        if (i < k->device.size()) (void)hipSetDevice(k->device[i]);
        (void)g_rccl.CommDestroy(k->comm[i]);
    }
    delete k;
}

}  // extern "C"
