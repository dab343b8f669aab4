/*
 * kmp_prep.hip -- everything around the scan kernels: partial-count reduce, layout validation,
 * on-device payload extraction, synthetic benchmark input.  gfx950.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kmp_device.h"
#include "kmp_launch.h"
#include "kmp_dev_common.h"

namespace {

/* ================================================================================================
 * On-device payload extraction: the raw capture file is uploaded as it is; the host
 * only walks the 16-byte record headers (frame offset + caplen).  Replaces the extraction phase
 * openmp_data.c:128-147 (parallel-for: dump_*_packet + malloc + memcpy per payload) by
 *   kmp_extract_kernel      one thread per frame: the accept/reject rule and payload bounds of
 *                           dump_UDP_packet / dump_TCP_packet (packet_dumping.h:87-139, 150-188)
 *   kmp_scan_*_kernel       exclusive scan of {slot bytes, valid} -> slot offsets and packet indices
 *   kmp_gather_kernel       one wavefront per payload: copy it to its 16-byte aligned slot, zero the pad
 * ============================================================================================== */
#define KMP_SCAN_ITEMS 4u                                         /* items per thread in the block scan */
#define KMP_SCAN_TILE  (KMP_BLOCK_THREADS * KMP_SCAN_ITEMS)

__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_extract_kernel(const uint8_t *__restrict__ file, const uint64_t *__restrict__ frame_off,
                   const uint32_t *__restrict__ caplen, uint64_t n, int tcp,
                   uint32_t *__restrict__ poff, uint32_t *__restrict__ plen)      /* plen = 0xFFFFFFFF: rejected */
{
    for (uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n; f += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *p = file + frame_off[f];
        const uint32_t cl = caplen[f];
        uint32_t o = 0u, l = 0xFFFFFFFFu;
        if (!tcp) {                                                    /* packet_dumping.h:87-139 */
            if (cl >= 14u + 20u) {                                     /* :94, :102 */
                const uint32_t rest = cl - 14u;
                const uint32_t ihl = (uint32_t)(p[14] & 0x0Fu) << 2;   /* :108 */
                if (rest >= ihl && p[23] == 17u && rest - ihl >= 8u) { /* :110, :116, :125 */
                    o = 14u + ihl + 8u;                                /* :133 */
                    l = rest - ihl - 8u;                               /* :136 */
                }
            }
        } else if (cl >= 15u) {                                        /* packet_dumping.h:150-188 */
            const uint32_t size_ip = (uint32_t)(p[14] & 0x0Fu) << 2;   /* :165 */
            const uint32_t tcp_at = 14u + size_ip;
            if (size_ip >= 20u && cl >= tcp_at + 13u) {                /* :166 */
                const uint32_t size_tcp = (uint32_t)(p[tcp_at + 12u] >> 4) << 2;    /* :175 */
                if (size_tcp >= 20u && cl >= tcp_at + size_tcp) {      /* :176; wrap of the reference's unsigned length rejected */
                    o = tcp_at + size_tcp;                             /* :181 */
                    l = cl - o;                                        /* :184 */
                }
            }
        }
        poff[f] = o; plen[f] = l;
    }
}

__device__ __forceinline__ uint64_t slot_bytes(uint32_t l) { return l == 0xFFFFFFFFu ? 0ull : (l ? (((uint64_t)l + 15ull) & ~15ull) : 16ull); }

/* Block-local exclusive scan of {slot bytes, valid}: local prefixes per frame, totals per block. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_local_kernel(const uint32_t *__restrict__ plen, uint64_t n, uint64_t *__restrict__ loc_off,
                      uint32_t *__restrict__ loc_idx, uint64_t *__restrict__ blk_bytes, uint32_t *__restrict__ blk_cnt)
{
    __shared__ uint64_t s_b[KMP_BLOCK_THREADS];
    __shared__ uint32_t s_c[KMP_BLOCK_THREADS];
    const uint64_t base = (uint64_t)blockIdx.x * KMP_SCAN_TILE + (uint64_t)threadIdx.x * KMP_SCAN_ITEMS;
    uint64_t b[KMP_SCAN_ITEMS], tb = 0;
    uint32_t c[KMP_SCAN_ITEMS], tc = 0;
#pragma unroll
    for (uint32_t i = 0; i < KMP_SCAN_ITEMS; i++) {
        const uint32_t l = (base + i < n) ? plen[base + i] : 0xFFFFFFFFu;
        b[i] = tb; c[i] = tc;
        tb += slot_bytes(l); tc += (l != 0xFFFFFFFFu);
    }
    s_b[threadIdx.x] = tb; s_c[threadIdx.x] = tc;
    __syncthreads();
    for (uint32_t d = 1; d < KMP_BLOCK_THREADS; d <<= 1) {            /* Hillis-Steele inclusive scan of the thread totals */
        const uint64_t vb = threadIdx.x >= d ? s_b[threadIdx.x - d] : 0ull;
        const uint32_t vc = threadIdx.x >= d ? s_c[threadIdx.x - d] : 0u;
        __syncthreads();
        s_b[threadIdx.x] += vb; s_c[threadIdx.x] += vc;
        __syncthreads();
    }
    const uint64_t eb = s_b[threadIdx.x] - tb;
    const uint32_t ec = s_c[threadIdx.x] - tc;
#pragma unroll
    for (uint32_t i = 0; i < KMP_SCAN_ITEMS; i++)
        if (base + i < n) { loc_off[base + i] = eb + b[i]; loc_idx[base + i] = ec + c[i]; }
    if (threadIdx.x == KMP_BLOCK_THREADS - 1u) { blk_bytes[blockIdx.x] = s_b[threadIdx.x]; blk_cnt[blockIdx.x] = s_c[threadIdx.x]; }
}

/* One block: exclusive scan of the block totals in place; totals[0] = bytes, totals[1] = payloads. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scan_totals_kernel(uint64_t *__restrict__ blk_bytes, uint32_t *__restrict__ blk_cnt, uint32_t nblk,
                       unsigned long long *__restrict__ totals)
{
    __shared__ uint64_t s_b[KMP_BLOCK_THREADS];
    __shared__ uint32_t s_c[KMP_BLOCK_THREADS];
    uint64_t cb = 0;
    uint64_t cc = 0;
    for (uint32_t base = 0; base < nblk; base += KMP_BLOCK_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t vb0 = i < nblk ? blk_bytes[i] : 0ull;
        const uint32_t vc0 = i < nblk ? blk_cnt[i] : 0u;
        s_b[threadIdx.x] = vb0; s_c[threadIdx.x] = vc0;
        __syncthreads();
        for (uint32_t d = 1; d < KMP_BLOCK_THREADS; d <<= 1) {
            const uint64_t vb = threadIdx.x >= d ? s_b[threadIdx.x - d] : 0ull;
            const uint32_t vc = threadIdx.x >= d ? s_c[threadIdx.x - d] : 0u;
            __syncthreads();
            s_b[threadIdx.x] += vb; s_c[threadIdx.x] += vc;
            __syncthreads();
        }
        if (i < nblk) { blk_bytes[i] = cb + s_b[threadIdx.x] - vb0; blk_cnt[i] = (uint32_t)(cc + s_c[threadIdx.x] - vc0); }
        cb += s_b[KMP_BLOCK_THREADS - 1u]; cc += s_c[KMP_BLOCK_THREADS - 1u];
        __syncthreads();
    }
    if (threadIdx.x == 0u) { totals[0] = cb; totals[1] = cc; }
}

/* Index of the payload arena + source address of every payload. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_scatter_index_kernel(const uint64_t *__restrict__ frame_off, const uint32_t *__restrict__ poff,
                         const uint32_t *__restrict__ plen, const uint64_t *__restrict__ loc_off,
                         const uint32_t *__restrict__ loc_idx, const uint64_t *__restrict__ blk_bytes,
                         const uint32_t *__restrict__ blk_cnt, uint64_t n, uint64_t *__restrict__ pkt_off,
                         uint32_t *__restrict__ pkt_len, uint64_t *__restrict__ src_off)
{
    for (uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n; f += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = plen[f];
        if (l == 0xFFFFFFFFu) continue;                            /* serial.c:138-140: rejected frames are skipped */
        const uint64_t blk = f / KMP_SCAN_TILE;
        const uint64_t k = (uint64_t)blk_cnt[blk] + loc_idx[f];
        pkt_off[k] = blk_bytes[blk] + loc_off[f];
        pkt_len[k] = l;
        src_off[k] = frame_off[f] + poff[f];
    }
}

/* New (packed) slot offsets of an existing index: new_off[k] = scanned slot bytes before payload k. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_repack_index_kernel(const uint64_t *__restrict__ loc_off, const uint64_t *__restrict__ blk_bytes, uint64_t n,
                        uint64_t *__restrict__ new_off)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
        new_off[k] = blk_bytes[k / KMP_SCAN_TILE] + loc_off[k];
}

/* serial.c:125-127 (malloc + memcpy per payload): one wavefront copies one payload into its slot. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_gather_kernel(const uint8_t *__restrict__ file, const uint64_t *__restrict__ src_off, const uint64_t *__restrict__ pkt_off,
                  const uint32_t *__restrict__ pkt_len, uint64_t n_pkts, uint8_t *__restrict__ arena)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nw = (uint64_t)gridDim.x * KMP_BLOCK_WAVES;
    for (uint64_t k = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + (threadIdx.x >> 6); k < n_pkts; k += nw) {
        const uint8_t *src = file + src_off[k];
        uint32_t *dst = reinterpret_cast<uint32_t *>(arena + pkt_off[k]);
        const uint32_t L = pkt_len[k];
        const uint32_t nw4 = (L ? ((L + 15u) & ~15u) : 16u) / 4u;
        for (uint32_t w = lane; w < nw4; w += 64u) {
            const uint32_t i0 = w * 4u;
            uint32_t v = 0u;
            if (i0 + 4u <= L) __builtin_memcpy(&v, src + i0, 4);           /* unaligned source */
            else
                for (uint32_t b = 0; b < 4u && i0 + b < L; b++) v |= (uint32_t)src[i0 + b] << (8u * b);
            dst[w] = v;
        }
    }
}

/* counts[pat_ids[x]] = sum of that pattern's block partials.  The persistent grids leave a thousand partials or two per pattern: one
 * 256-thread block each.  The flat kernel's grid is one block per 16 packets -- 62 500 partials per million packets, 500 000 for an
 * 8 M-packet shard --, which one block would take 60 us to add up: the row is cut into gridDim.y slices of KMP_REDUCE_SLICE partials
 * and every slice block ADDS its sum to the counter with an atomic add that returns nothing (integer sums: the result does not
 * depend on the order).  The counter was put to 0 by the scan kernel itself, one kernel boundary earlier, unless the pass
 * accumulates.  (Round 2 left the slice sums in a scratch row and let the block that drew the last ticket add them up: one
 * returning atomic and a dependent read more, 6.2 us per launch.) */
#define KMP_REDUCE_THREADS 1024u
__global__ void __launch_bounds__(KMP_REDUCE_THREADS)
kmp_reduce_kernel(const unsigned long long *__restrict__ partials, uint32_t blocks_x,
                  const uint32_t *__restrict__ pat_ids, const uint32_t *__restrict__ rows,
                  unsigned long long *__restrict__ counts, int accumulate)
{
    __shared__ unsigned long long s[KMP_REDUCE_THREADS / KMP_WAVE];
    unsigned long long t = 0ull;
    /* rows == nullptr: row x of partials belongs to pat_ids[x]; else row rows[x] (fused pass: duplicates share a row) */
    const unsigned long long *row = partials + (uint64_t)(rows ? rows[blockIdx.x] : blockIdx.x) * blocks_x;
    const uint32_t slices = gridDim.y;
    const uint32_t per = (blocks_x + slices - 1u) / slices;
    const uint32_t lo = blockIdx.y * per, hi = min(blocks_x, lo + per);
    uint32_t i = lo + threadIdx.x;
    for (; i + 7u * blockDim.x < hi; i += 8u * blockDim.x) {                 /* eight loads in flight per thread */
        unsigned long long v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) v[u] = row[i + u * blockDim.x];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) t += v[u];
    }
    for (; i < hi; i += blockDim.x) t += row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63u) == 0u) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0u) {
        unsigned long long r = 0ull;
        for (uint32_t w = 0; w < blockDim.x / KMP_WAVE; ++w) r += s[w];
        unsigned long long *dst = counts + pat_ids[blockIdx.x];
        if (slices > 1u) { if (r) __hip_atomic_fetch_add(dst, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else *dst = accumulate ? *dst + r : r;       /* accumulate: counts keep adding up over the batches of a streamed capture (openmp_task.c:172-175) */
    }
}

/* Layout contract of kmpgpu.h for a device-resident index: err[0] |= 1 misaligned, |= 2 out of
 * bounds, |= 4 length >= 2^30; err[1] & 1: the arena is NOT uniform-stride, err[1] & 2: slots are NOT
 * packed back to back; info[0] += sum(len),
 * info[1..4] = offset of payload 0, stride (offset 1 - offset 0), length of payload 0, end of the last slot. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_validate_index_kernel(const uint64_t *__restrict__ pkt_off, const uint32_t *__restrict__ pkt_len, uint64_t n,
                          uint64_t arena_bytes, uint32_t *__restrict__ err, unsigned long long *__restrict__ info)
{
    __shared__ unsigned long long s[KMP_BLOCK_WAVES];
    unsigned long long sum = 0ull;
    uint32_t e = 0u, nonuni = 0u;
    const uint64_t off0 = pkt_off[0];
    const uint32_t len0 = pkt_len[0];
    const uint64_t l016 = ((uint64_t)len0 + 15ull) & ~15ull;
    const uint64_t stride = (n > 1) ? pkt_off[1] - off0 : (l016 < 16ull ? 16ull : l016);
    if ((n > 1 && pkt_off[1] < off0) || stride < (l016 < 16ull ? 16ull : l016) || (stride & 15ull)) nonuni |= 1u;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o = pkt_off[k];
        const uint64_t l = pkt_len[k];
        if (o & 15ull) e |= 1u;
        if (l >= (1ull << 30)) e |= 4u;
        const uint64_t l16 = (l + 15ull) & ~15ull;
        if (o > arena_bytes || (l16 < 16ull ? 16ull : l16) > arena_bytes - o) e |= 2u;   /* every payload owns >= 16 readable bytes */
        if (l != len0 || o != off0 + k * stride) nonuni |= 1u;
        if (k + 1 < n && pkt_off[k + 1] != o + (l16 < 16ull ? 16ull : l16)) nonuni |= 2u;      /* not packed back to back */
        sum += l;
    }
    if (e) atomicOr(err, e);
    if (nonuni) atomicOr(err + 1, nonuni);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63u) == 0u) s[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0u) {
        unsigned long long r = 0ull;
        for (uint32_t i = 0; i < KMP_BLOCK_WAVES; ++i) r += s[i];
        atomicAdd(info, r);
        if (blockIdx.x == 0u) {
            info[1] = off0; info[2] = stride; info[3] = len0;
            const uint64_t ll16 = ((uint64_t)pkt_len[n - 1] + 15ull) & ~15ull;
            info[4] = pkt_off[n - 1] + (ll16 < 16ull ? 16ull : ll16);
        }
    }
}

/* Synthetic payloads (kmp_synth.h): one wavefront per packet, one dword per lane per step. */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_synth_fill_kernel(uint8_t *__restrict__ arena, const uint64_t *__restrict__ pkt_off,
                      const uint32_t *__restrict__ pkt_len, uint64_t first_pkt_id, uint64_t n, kmp_synth_params sp)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave0 = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * KMP_BLOCK_WAVES;
    for (uint64_t i = wave0; i < n; i += nwaves) {
        const uint64_t id = first_pkt_id + i;
        const uint32_t len = pkt_len[i];
        uint32_t *slot = reinterpret_cast<uint32_t *>(arena + pkt_off[i]);
        const uint32_t key = kmp_synth_pkt_key(sp.seed, id);
        uint32_t pos = 0u;
        const int planted = kmp_synth_plant(&sp, id, len, &pos);
        const uint32_t nw = ((len + 15u) & ~15u) / 4u;
        for (uint32_t w = lane; w < nw; w += 64u) slot[w] = kmp_synth_slot_word(&sp, key, w, len, planted, pos);
    }
}

__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_fixed_index_kernel(uint64_t *__restrict__ pkt_off, uint32_t *__restrict__ pkt_len, uint64_t n, uint32_t len,
                       uint64_t stride)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        pkt_off[k] = k * stride;
        pkt_len[k] = len;
    }
}

}  // namespace

/* On-device extraction: frames of a raw capture file -> packed payload arena.  The temporaries live in
 * ws (caller-allocated, kmp_extract_ws_bytes(n) bytes).  Two phases because the arena size is only known
 * after the scan: phase 1 leaves {arena bytes, payload count} in ws->totals. */
size_t kmp_extract_ws_bytes(uint64_t n_frames)
{
    const uint64_t nblk = (n_frames + KMP_SCAN_TILE - 1) / KMP_SCAN_TILE;
    return (size_t)(n_frames * (4 + 4 + 8 + 4) + nblk * (8 + 4) + 64 + 256);
}

hipError_t kmp_launch_extract_phase1(const uint8_t *file, const uint64_t *frame_off, const uint32_t *caplen, uint64_t n, int tcp,
                                     uint8_t *ws, unsigned long long *totals, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint64_t nblk = (n + KMP_SCAN_TILE - 1) / KMP_SCAN_TILE;
    uint64_t *loc_off = reinterpret_cast<uint64_t *>(ws);
    uint64_t *blk_bytes = loc_off + n;
    uint32_t *poff = reinterpret_cast<uint32_t *>(blk_bytes + nblk);
    uint32_t *plen = poff + n, *loc_idx = plen + n, *blk_cnt = loc_idx + n;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS, 4096);
    hipLaunchKernelGGL(kmp_extract_kernel, dim3(blocks), dim3(KMP_BLOCK_THREADS), 0, st, file, frame_off, caplen, n, tcp, poff, plen);
    hipLaunchKernelGGL(kmp_scan_local_kernel, dim3((uint32_t)nblk), dim3(KMP_BLOCK_THREADS), 0, st, plen, n, loc_off, loc_idx, blk_bytes, blk_cnt);
    hipLaunchKernelGGL(kmp_scan_totals_kernel, dim3(1), dim3(KMP_BLOCK_THREADS), 0, st, blk_bytes, blk_cnt, (uint32_t)nblk, totals);
    return hipGetLastError();
}

hipError_t kmp_launch_extract_phase2(const uint8_t *file, const uint64_t *frame_off, uint64_t n, uint8_t *ws, uint64_t n_pkts,
                                     uint8_t *arena, uint64_t *pkt_off, uint32_t *pkt_len, uint64_t *src_off, hipStream_t st)
{
    if (n == 0 || n_pkts == 0) return hipSuccess;
    const uint64_t nblk = (n + KMP_SCAN_TILE - 1) / KMP_SCAN_TILE;
    uint64_t *loc_off = reinterpret_cast<uint64_t *>(ws);
    uint64_t *blk_bytes = loc_off + n;
    uint32_t *poff = reinterpret_cast<uint32_t *>(blk_bytes + nblk);
    uint32_t *plen = poff + n, *loc_idx = plen + n, *blk_cnt = loc_idx + n;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS, 4096);
    hipLaunchKernelGGL(kmp_scatter_index_kernel, dim3(blocks), dim3(KMP_BLOCK_THREADS), 0, st, frame_off, poff, plen, loc_off, loc_idx,
                       blk_bytes, blk_cnt, n, pkt_off, pkt_len, src_off);
    uint32_t gblocks = (uint32_t)std::min<uint64_t>((n_pkts + KMP_BLOCK_WAVES - 1) / KMP_BLOCK_WAVES, 8192);
    hipLaunchKernelGGL(kmp_gather_kernel, dim3(gblocks), dim3(KMP_BLOCK_THREADS), 0, st, file, src_off, pkt_off, pkt_len, n_pkts, arena);
    return hipGetLastError();
}

/* Repack an arena whose slots are not back to back (gaps, shuffled order) into a packed one so that the
 * streaming kernels apply.  Phase 1: scan of the slot sizes (totals[0] = packed bytes).  Phase 2: new
 * offsets + copy.  ws: kmp_extract_ws_bytes(n) bytes. */
hipError_t kmp_launch_repack_phase1(const uint32_t *pkt_len, uint64_t n, uint8_t *ws, unsigned long long *totals, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint64_t nblk = (n + KMP_SCAN_TILE - 1) / KMP_SCAN_TILE;
    uint64_t *loc_off = reinterpret_cast<uint64_t *>(ws);
    uint64_t *blk_bytes = loc_off + n;
    uint32_t *loc_idx = reinterpret_cast<uint32_t *>(blk_bytes + nblk) + 2 * n, *blk_cnt = loc_idx + n;
    hipLaunchKernelGGL(kmp_scan_local_kernel, dim3((uint32_t)nblk), dim3(KMP_BLOCK_THREADS), 0, st, pkt_len, n, loc_off, loc_idx, blk_bytes, blk_cnt);
    hipLaunchKernelGGL(kmp_scan_totals_kernel, dim3(1), dim3(KMP_BLOCK_THREADS), 0, st, blk_bytes, blk_cnt, (uint32_t)nblk, totals);
    return hipGetLastError();
}

hipError_t kmp_launch_repack_phase2(const uint8_t *old_arena, const uint64_t *old_off, const uint32_t *pkt_len, uint64_t n, uint8_t *ws,
                                    uint8_t *new_arena, uint64_t *new_off, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint64_t *loc_off = reinterpret_cast<const uint64_t *>(ws);
    const uint64_t *blk_bytes = loc_off + n;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS, 4096);
    hipLaunchKernelGGL(kmp_repack_index_kernel, dim3(blocks), dim3(KMP_BLOCK_THREADS), 0, st, loc_off, blk_bytes, n, new_off);
    uint32_t gblocks = (uint32_t)std::min<uint64_t>((n + KMP_BLOCK_WAVES - 1) / KMP_BLOCK_WAVES, 8192);
    hipLaunchKernelGGL(kmp_gather_kernel, dim3(gblocks), dim3(KMP_BLOCK_THREADS), 0, st, old_arena, old_off, new_off, pkt_len, n, new_arena);
    return hipGetLastError();
}

hipError_t kmp_launch_reduce(const unsigned long long *partials, uint32_t blocks_x, const uint32_t *pat_ids,
                             uint32_t n_ids, unsigned long long *counts, hipStream_t st, const uint32_t *rows, int accumulate,
                             bool counts_zeroed)
{
    if (n_ids == 0) return hipSuccess;
    /* one slice per KMP_REDUCE_SLICE partials where the caller has seen to the counter's starting value (it accumulates, or the scan
     * kernel has put it to 0: kmp_scan_args::zero_counts) */
    uint32_t slices = 1u;
    if (kmp_reduce_is_sliced(blocks_x) && (accumulate || counts_zeroed)) slices = std::min<uint32_t>(256u, (blocks_x + KMP_REDUCE_SLICE - 1u) / KMP_REDUCE_SLICE);
    hipLaunchKernelGGL(kmp_reduce_kernel, dim3(n_ids, slices), dim3(blocks_x > 8192u ? KMP_REDUCE_THREADS : KMP_BLOCK_THREADS), 0, st, partials, blocks_x,
                       pat_ids, rows, counts, accumulate);
    return hipGetLastError();
}

/* Packed arenas: are the bytes between a payload's end and the end of its 16-byte-padded slot all 0x00?  The
 * host library, the device extraction and the repack write them so.  When they are, "the match window lies
 * inside the payload" equals "it lies inside the slot and holds no 0x00" (patterns are NUL-free), and the
 * packed kernel needs neither the payload's offset nor its length: the packet-start bitmap bounds the slot.
 * fix != 0 (arena owned by the context): clear what is not. */
namespace {
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_check_padding_kernel(uint8_t *__restrict__ arena, const uint64_t *__restrict__ pkt_off, const uint32_t *__restrict__ pkt_len,
                         uint64_t n, int fix, uint32_t *__restrict__ dirty)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t L = pkt_len[k], r = L & 15u;
        if (L != 0u && r == 0u) continue;                               /* the payload fills its slot */
        uint4 *g = reinterpret_cast<uint4 *>(arena + pkt_off[k] + (L - r));      /* last 16-byte group of the slot */
        uint4 v = *g;
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t bad = 0u;
#pragma unroll
        for (uint32_t d = 0; d < 4u; ++d) {
            const uint32_t lo = 4u * d;                                  /* payload bytes of this dword: [lo, r) */
            const uint32_t keep = (r >= lo + 4u) ? 0xFFFFFFFFu : (r <= lo) ? 0u : ((1u << (8u * (r - lo))) - 1u);
            bad |= w[d] & ~keep;
            w[d] &= keep;
        }
        if (bad) {
            atomicOr(dirty, 1u);
            if (fix) *g = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}
}  // namespace

hipError_t kmp_launch_check_padding(uint8_t *arena, const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n, int fix,
                                    uint32_t *dirty, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(kmp_check_padding_kernel, dim3((uint32_t)blocks), dim3(KMP_BLOCK_THREADS), 0, st, arena, pkt_off, pkt_len, n, fix, dirty);
    return hipGetLastError();
}

/* Sum over packets of min(len, first 0x00 + 1): the bytes a strlen()-bounded scan (serial.c:191) has to
 * touch.  One wavefront per packet, 1 KiB per step; SURVEY 8(d) asks for this figure beside the payload
 * bytes when the input carries NUL bytes.  Not on the hot path (one pass, on request). */
namespace {
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_effective_bytes_kernel(const uint8_t *__restrict__ arena, const uint64_t *__restrict__ pkt_off,
                           const uint32_t *__restrict__ pkt_len, uint64_t n, unsigned long long *__restrict__ out)
{
    const uint32_t lane = threadIdx.x & (KMP_WAVE - 1u);
    const uint64_t nw = (uint64_t)gridDim.x * KMP_BLOCK_WAVES;
    unsigned long long acc = 0ull;
    for (uint64_t k = (uint64_t)blockIdx.x * KMP_BLOCK_WAVES + (threadIdx.x >> 6); k < n; k += nw) {
        const uint32_t L = pkt_len[k];
        const uint8_t *p = arena + pkt_off[k];
        uint32_t eff = L;
        for (uint32_t base = 0u; base < L; base += KMP_CHUNK) {
            const uint32_t o = base + lane * KMP_LANE_BYTES;
            uint32_t first = 16u;                                  /* index of the lane's first 0x00 among its payload bytes */
            if (o < L) {                                           /* slots are padded to 16 bytes: the whole group is readable */
                const uint4 v = *reinterpret_cast<const uint4 *>(p + o);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int d = 3; d >= 0; --d) {
                    const uint32_t z = zero_byte_mask(w[d]);
                    if (z) first = 4u * (uint32_t)d + ((uint32_t)__builtin_ctz(z) >> 3);
                }
                if (o + first >= L) first = 16u;                   /* a zero in the slot padding is not payload */
            }
            const uint64_t hit = ballot64(first < 16u);
            if (hit) {
                const uint32_t src = (uint32_t)__builtin_ctzll(hit);
                const uint32_t f = (uint32_t)__shfl((int)first, (int)src);
                eff = base + src * KMP_LANE_BYTES + f + 1u;
                break;
            }
        }
        acc += eff;
    }
    if (lane == 0u && acc) atomicAdd(out, acc);
}
}  // namespace

hipError_t kmp_launch_effective_bytes(const uint8_t *arena, const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n,
                                      unsigned long long *out, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + KMP_BLOCK_WAVES - 1) / KMP_BLOCK_WAVES;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(kmp_effective_bytes_kernel, dim3((uint32_t)blocks), dim3(KMP_BLOCK_THREADS), 0, st, arena, pkt_off, pkt_len, n, out);
    return hipGetLastError();
}

hipError_t kmp_launch_validate(const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t n, uint64_t arena_bytes,
                               uint32_t *err, unsigned long long *payload_bytes, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS);
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(kmp_validate_index_kernel, dim3(blocks), dim3(KMP_BLOCK_THREADS), 0, st, pkt_off, pkt_len, n,
                       arena_bytes, err, payload_bytes);
    return hipGetLastError();
}

hipError_t kmp_launch_synth_fill(uint8_t *arena, const uint64_t *pkt_off, const uint32_t *pkt_len, uint64_t first_pkt_id,
                                 uint64_t n, const kmp_synth_params &sp, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + KMP_BLOCK_WAVES - 1) / KMP_BLOCK_WAVES;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(kmp_synth_fill_kernel, dim3((uint32_t)blocks), dim3(KMP_BLOCK_THREADS), 0, st, arena, pkt_off,
                       pkt_len, first_pkt_id, n, sp);
    return hipGetLastError();
}

hipError_t kmp_launch_fixed_index(uint64_t *pkt_off, uint32_t *pkt_len, uint64_t n, uint32_t len, uint64_t stride,
                                  hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS;
    if (blocks > 4096u) blocks = 4096u;
    hipLaunchKernelGGL(kmp_fixed_index_kernel, dim3((uint32_t)blocks), dim3(KMP_BLOCK_THREADS), 0, st, pkt_off, pkt_len, n,
                       len, stride);
    return hipGetLastError();
}

/* dst[i] += src[i]: the merge of two contexts' counters on one device (openmp_task.c:172-175, the omp atomic merge of a task's counts). */
__global__ void __launch_bounds__(KMP_BLOCK_THREADS)
kmp_add_counts_kernel(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

hipError_t kmp_launch_add_counts(unsigned long long *dst, const unsigned long long *src, uint32_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(kmp_add_counts_kernel, dim3((n + KMP_BLOCK_THREADS - 1) / KMP_BLOCK_THREADS), dim3(KMP_BLOCK_THREADS), 0, st, dst, src, n);
    return hipGetLastError();
}
