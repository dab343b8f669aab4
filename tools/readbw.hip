// Tuning only: what does a kernel that ONLY reads 1.5 GB reach on this MI355X?  Gives the attainable
// HBM read rate that the scan kernel's 80 % of the 8 TB/s data-sheet peak has to be judged against.
// Variants: access pattern (contiguous range per wavefront as the scan kernels do / chunk-interleaved
// grid-stride), load width per lane, loads in flight, blocks per CU, nt hint.
//   hipcc --offload-arch=gfx950 -O3 -o tools/readbw tools/readbw.hip && tools/readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ inline u32x4 ld16(const u32x4 *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}

// every wavefront reads its own contiguous range, `INFL` independent 1 KiB chunk loads in flight
template <int INFL, bool NT>
__global__ void __launch_bounds__(256) read_range(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nw = (uint64_t)gridDim.x * 4u, gw = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t chunks = n16 / 64u, per = (chunks + nw - 1) / nw;
    const uint64_t c0 = gw * per, c1 = std::min(chunks, c0 + per);
    u32x4 acc = {0, 0, 0, 0};
    uint64_t c = c0;
    for (; c + INFL <= c1; c += INFL) {
        u32x4 v[INFL];
#pragma unroll
        for (int i = 0; i < INFL; ++i) v[i] = ld16<NT>(src + (c + i) * 64u + lane);
#pragma unroll
        for (int i = 0; i < INFL; ++i) acc ^= v[i];
    }
    for (; c < c1; ++c) acc ^= ld16<NT>(src + c * 64u + lane);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;      // keeps the loads alive
}

// chunk-interleaved: chunk c goes to wavefront c % nwaves (every HBM channel busy at any instant)
template <int INFL, bool NT>
__global__ void __launch_bounds__(256) read_stride(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nw = (uint64_t)gridDim.x * 4u, gw = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t chunks = n16 / 64u;
    u32x4 acc = {0, 0, 0, 0};
    uint64_t c = gw;
    for (; c + (INFL - 1) * nw < chunks; c += INFL * nw) {
        u32x4 v[INFL];
#pragma unroll
        for (int i = 0; i < INFL; ++i) v[i] = ld16<NT>(src + (c + i * nw) * 64u + lane);
#pragma unroll
        for (int i = 0; i < INFL; ++i) acc ^= v[i];
    }
    for (; c < chunks; c += nw) acc ^= ld16<NT>(src + c * 64u + lane);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// block-contiguous: each block owns a contiguous range, its 4 wavefronts interleave 1 KiB chunks in it
template <int INFL, bool NT>
__global__ void __launch_bounds__(256) read_block(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t chunks = n16 / 64u, per = (chunks + gridDim.x - 1) / gridDim.x;
    const uint64_t c0 = (uint64_t)blockIdx.x * per, c1 = std::min(chunks, c0 + per);
    u32x4 acc = {0, 0, 0, 0};
    uint64_t c = c0 + wave;
    for (; c + (INFL - 1) * 4u < c1; c += INFL * 4u) {
        u32x4 v[INFL];
#pragma unroll
        for (int i = 0; i < INFL; ++i) v[i] = ld16<NT>(src + (c + i * 4u) * 64u + lane);
#pragma unroll
        for (int i = 0; i < INFL; ++i) acc ^= v[i];
    }
    for (; c < c1; c += 4u) acc ^= ld16<NT>(src + c * 64u + lane);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// packet-interleaved inside a block: the block owns a contiguous run of 1504-byte slots, wavefront w streams
// the slots with index % 4 == w as ONE logical byte stream cut into 1 KiB chunks (lane slot t -> packet t/94)
template <int INFL, bool NT>
__global__ void __launch_bounds__(256) read_pktil(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t pkts = n16 / 94u, per = ((pkts + gridDim.x - 1) / gridDim.x + 3u) & ~3ull;
    const uint64_t p0 = (uint64_t)blockIdx.x * per, p1 = std::min(pkts, p0 + per);
    const uint32_t mine = p1 > p0 ? (uint32_t)((p1 - p0 + 3u - wave) / 4u) : 0u;       // packets of this wavefront
    const uint32_t slots = mine * 94u, chunks = (slots + 63u) / 64u;
    const u32x4 *base = src + p0 * 94u;
    u32x4 acc = {0, 0, 0, 0};
    uint32_t c = 0;
    for (; c + INFL <= chunks; c += INFL) {
        u32x4 v[INFL];
#pragma unroll
        for (int i = 0; i < INFL; ++i) {
            uint32_t t = (c + i) * 64u + lane;
            t = t < slots ? t : slots - 1u;
            const uint32_t q = t / 94u, r = t - q * 94u;
            v[i] = ld16<NT>(base + (uint64_t)(q * 4u + wave) * 94u + r);
        }
#pragma unroll
        for (int i = 0; i < INFL; ++i) acc ^= v[i];
    }
    for (; c < chunks; ++c) {
        uint32_t t = c * 64u + lane;
        t = t < slots ? t : slots - 1u;
        const uint32_t q = t / 94u, r = t - q * 94u;
        acc ^= ld16<NT>(base + (uint64_t)(q * 4u + wave) * 94u + r);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// contiguous range per wavefront as read_range, block size as a parameter (does the number of workgroups to
// dispatch matter for the ramp-up of a 230 us launch?)
template <int INFL, bool NT, int BS>
__global__ void __launch_bounds__(BS) read_range_bs(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nw = (uint64_t)gridDim.x * (BS / 64), gw = (uint64_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6);
    const uint64_t chunks = n16 / 64u, per = (chunks + nw - 1) / nw;
    const uint64_t c0 = gw * per, c1 = std::min(chunks, c0 + per);
    u32x4 acc = {0, 0, 0, 0};
    uint64_t c = c0;
    for (; c + INFL <= c1; c += INFL) {
        u32x4 v[INFL];
#pragma unroll
        for (int i = 0; i < INFL; ++i) v[i] = ld16<NT>(src + (c + i) * 64u + lane);
#pragma unroll
        for (int i = 0; i < INFL; ++i) acc ^= v[i];
    }
    for (; c < c1; ++c) acc ^= ld16<NT>(src + c * 64u + lane);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// tile-interleaved: tile t (TILE consecutive 1 KiB chunks, all in flight at once) goes to wavefront t % nwaves
template <int TILE, bool NT>
__global__ void __launch_bounds__(256) read_tiles(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nw = (uint64_t)gridDim.x * 4u, gw = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t chunks = n16 / 64u, tiles = chunks / TILE;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t t = gw; t < tiles; t += nw) {
        u32x4 v[TILE];
#pragma unroll
        for (int i = 0; i < TILE; ++i) v[i] = ld16<NT>(src + (t * TILE + i) * 64u + lane);
#pragma unroll
        for (int i = 0; i < TILE; ++i) acc ^= v[i];
    }
    if (gw == 0) for (uint64_t c = tiles * TILE; c < chunks; ++c) acc ^= ld16<NT>(src + c * 64u + lane);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// the access pattern a tile-interleaved scan kernel would have: per 2 KiB tile two chunk loads + the first
// 128 bytes of the next tile (lanes 0..7), two tiles in flight per wavefront
template <bool NT>
__global__ void __launch_bounds__(256) read_tiles_halo(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nw = (uint64_t)gridDim.x * 4u, gw = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t tiles = n16 / 128u;
    u32x4 acc = {0, 0, 0, 0};
    const u32x4 z = {0, 0, 0, 0};
    uint64_t t = gw;
    if (t >= tiles) return;
    u32x4 a0 = ld16<NT>(src + t * 128u + lane), a1 = ld16<NT>(src + t * 128u + 64u + lane);
    u32x4 ah = (lane < 8u && t + 1 < tiles) ? ld16<NT>(src + (t + 1) * 128u + lane) : z;
    for (t += nw; t < tiles; t += nw) {
        const u32x4 b0 = ld16<NT>(src + t * 128u + lane), b1 = ld16<NT>(src + t * 128u + 64u + lane);
        const u32x4 bh = (lane < 8u && t + 1 < tiles) ? ld16<NT>(src + (t + 1) * 128u + lane) : z;
        acc ^= a0; acc ^= a1; acc ^= ah;
        a0 = b0; a1 = b1; ah = bh;
    }
    acc ^= a0; acc ^= a1; acc ^= ah;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// every wavefront reads contiguous ranges, but takes them from a global counter: ranges = PARTS x wavefronts
// (the first one statically), so a wavefront that is slowed down ends up with fewer ranges
template <int INFL, bool NT, int GROUPS>
__global__ void __launch_bounds__(256) read_steal(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *out, uint32_t *ticket, uint32_t parts)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nw = gridDim.x * 4u, gw = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t chunks = n16 / 64u;
    const uint32_t nr = nw * parts;
    const uint64_t per = (chunks + nr - 1) / nr;
    u32x4 acc = {0, 0, 0, 0};
    const uint32_t groups = GROUPS, g = gw % groups, per_group = nr / groups;      // nw and nr are multiples of GROUPS
    uint32_t k = gw / groups;                       // index inside the group: first range statically
    while (k < per_group) {
        const uint32_t r = k * groups + g;
        const uint64_t c0 = (uint64_t)r * per, c1 = std::min(chunks, c0 + per);
        uint64_t c = c0;
        for (; c + INFL <= c1; c += INFL) {
            u32x4 v[INFL];
#pragma unroll
            for (int i = 0; i < INFL; ++i) v[i] = ld16<NT>(src + (c + i) * 64u + lane);
#pragma unroll
            for (int i = 0; i < INFL; ++i) acc ^= v[i];
        }
        for (; c < c1; ++c) acc ^= ld16<NT>(src + c * 64u + lane);
        uint32_t nx = 0;
        if (lane == 0) nx = atomicAdd(ticket + g * 32u, 1u);      // one counter per group, 128 bytes apart
        k = nw / groups + (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// static ranges, every wavefront records when it finished (s_memtime, 100 MHz): how ragged is the end of a launch?
template <int INFL, bool NT>
__global__ void __launch_bounds__(256) read_range_t(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *out, unsigned long long *tend)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nw = (uint64_t)gridDim.x * 4u, gw = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t chunks = n16 / 64u, per = (chunks + nw - 1) / nw;
    const uint64_t c0 = gw * per, c1 = std::min(chunks, c0 + per);
    u32x4 acc = {0, 0, 0, 0};
    uint64_t c = c0;
    for (; c + INFL <= c1; c += INFL) {
        u32x4 v[INFL];
#pragma unroll
        for (int i = 0; i < INFL; ++i) v[i] = ld16<NT>(src + (c + i) * 64u + lane);
#pragma unroll
        for (int i = 0; i < INFL; ++i) acc ^= v[i];
    }
    for (; c < c1; ++c) acc ^= ld16<NT>(src + c * 64u + lane);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
    if (lane == 0) tend[gw] = __builtin_amdgcn_s_memrealtime();
}

// round 3: the shape the scan kernel has had since the end of round 2 -- SMALL contiguous ranges, one 256-thread block per four of
// them, as many blocks as that takes (62 500 for the benchmark arena), handed out in arena order by the hardware: what does a
// kernel reach that only READS with that shape?  per16 = range in 16-byte units (376 = four 1504-byte slots).
template <int INFL, bool NT>
__global__ void __launch_bounds__(256) read_small(const u32x4 *__restrict__ src, uint64_t n16, uint32_t per16, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t gw = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t b = gw * per16, e = std::min<uint64_t>(n16, b + per16);
    u32x4 acc = {0, 0, 0, 0};
    const u32x4 z = {0, 0, 0, 0};
    for (uint64_t c = b; c < e; c += 64u * INFL) {
        u32x4 v[INFL];
#pragma unroll
        for (int i = 0; i < INFL; ++i) { const uint64_t k = c + 64u * i + lane; v[i] = k < e ? ld16<NT>(src + k) : z; }
#pragma unroll
        for (int i = 0; i < INFL; ++i) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <typename F> double sustained_us(F launch)
{
    const int warm = 60, n = 100;
    for (int i = 0; i < warm; ++i) launch();
    std::vector<hipEvent_t> ev(2 * n);
    for (auto &e : ev) CK(hipEventCreate(&e));
    for (int i = 0; i < n; ++i) { CK(hipEventRecord(ev[2 * i])); launch(); CK(hipEventRecord(ev[2 * i + 1])); }
    CK(hipDeviceSynchronize());
    std::vector<float> t(n);
    for (int i = 0; i < n; ++i) CK(hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]));
    for (auto &e : ev) CK(hipEventDestroy(e));
    std::sort(t.begin(), t.end());
    double s = 0; for (float x : t) s += x;
    (void)s;
    return 1e3 * t[n / 2];
}

int main()
{
    const uint64_t bytes = 1504ull * 1000000ull;           // the benchmark arena
    const uint64_t n16 = bytes / 16;
    uint8_t *d; uint32_t *out;
    CK(hipMalloc(&d, bytes + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 0x61, bytes + 4096));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs; reading %.3f GB per launch, median of 100 launches after 60 warm-up\n", p.gcnArchName, cus, bytes / 1e9);
    const u32x4 *s = reinterpret_cast<const u32x4 *>(d);
#define RUN(name, kern, bpc) do { const int b_ = cus * (bpc); \
        double us = sustained_us([&] { hipLaunchKernelGGL(kern, dim3(b_), dim3(256), 0, 0, s, n16, out); }); \
        printf("%-44s blocks/CU=%d  %7.1f us  %6.0f GB/s\n", name, bpc, us, bytes / us / 1e3); fflush(stdout); } while (0)
    {   // how ragged is the end of a launch with static ranges?
        const int b_ = cus * 4; unsigned long long *tend; CK(hipMalloc(&tend, sizeof(unsigned long long) * b_ * 4));
        for (int i = 0; i < 80; ++i) hipLaunchKernelGGL((read_range_t<4, true>), dim3(b_), dim3(256), 0, 0, s, n16, out, tend);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(b_ * 4);
        CK(hipMemcpy(h.data(), tend, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double us = 1.0 / 100.0;     // 100 MHz
        printf("static ranges, finish time of the %zu wavefronts relative to the last one: p1 %.1f us, p10 %.1f, p50 %.1f, p90 %.1f, p99 %.1f\n", h.size(),
               (h.back() - h[h.size() / 100]) * us, (h.back() - h[h.size() / 10]) * us, (h.back() - h[h.size() / 2]) * us,
               (h.back() - h[h.size() * 9 / 10]) * us, (h.back() - h[h.size() * 99 / 100]) * us);
        uint32_t *ticket; CK(hipMalloc(&ticket, 4096 * 128));
#define STEAL(G, parts) do { double us2 = sustained_us([&] { hipMemsetAsync(ticket, 0, 4096 * 128, 0); \
            hipLaunchKernelGGL((read_steal<4, true, G>), dim3(b_), dim3(256), 0, 0, s, n16, out, ticket, (uint32_t)(parts)); }); \
        printf("ranges from %4d counters, %2d per wavefront   blocks/CU=4  %7.1f us  %6.0f GB/s\n", G, parts, us2, bytes / us2 / 1e3); fflush(stdout); } while (0)
        for (int parts : {2, 3, 4, 8}) { STEAL(64, parts); STEAL(256, parts); STEAL(1024, parts); }
    }
    // round 3: small ranges, non-persistent grid (the scan kernel's shape since the end of round 2)
    for (uint32_t per16 : {188u, 376u, 752u, 1504u, 3008u}) {
        const uint64_t waves = (n16 + per16 - 1) / per16;
        const int b_ = (int)((waves + 3) / 4);
#define RUNS(INFL) do { double us = sustained_us([&] { hipLaunchKernelGGL((read_small<INFL, true>), dim3(b_), dim3(256), 0, 0, s, n16, per16, out); }); \
        printf("small ranges of %5u B per wavefront, %6d blocks, %d x 1 KiB in flight nt   %7.1f us  %6.0f GB/s\n", per16 * 16u, b_, INFL, us, bytes / us / 1e3); fflush(stdout); } while (0)
        RUNS(2); RUNS(3); RUNS(6);
    }
    if (getenv("READBW_SMALL_ONLY")) return 0;
    // block size: the same 3072 / 4096 wavefronts as 64..1024-thread workgroups
    for (int waves : {3072, 4096}) {
#define RUNBS(BS) do { const int b_ = waves * 64 / BS; double us = sustained_us([&] { hipLaunchKernelGGL((read_range_bs<4, true, BS>), dim3(b_), dim3(BS), 0, 0, s, n16, out); }); \
        printf("range/wave 4 in flight nt, %4d wavefronts as %4d blocks of %4d threads  %7.1f us  %6.0f GB/s\n", waves, b_, BS, us, bytes / us / 1e3); fflush(stdout); } while (0)
        RUNBS(64); RUNBS(128); RUNBS(256); RUNBS(512); RUNBS(1024);
    }
    // tile-interleaved patterns against contiguous ranges, over the number of resident blocks
    for (int blocks : {cus * 3, cus * 4}) {
#define RUNB(name, kern) do { double us = sustained_us([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, s, n16, out); }); \
        printf("%-42s blocks=%4d  %7.1f us  %6.0f GB/s\n", name, blocks, us, bytes / us / 1e3); fflush(stdout); } while (0)
        RUNB("range/wave 4 in flight nt", (read_range<4, true>));
        RUNB("grid-stride 1 KiB x4 in flight nt", (read_stride<4, true>));
        RUNB("tiles of 2 KiB nt", (read_tiles<2, true>));
        RUNB("tiles of 4 KiB nt", (read_tiles<4, true>));
        RUNB("tiles of 2 KiB + 128 B halo, 2 in flight", (read_tiles_halo<true>));
    }
    return 0;
}
