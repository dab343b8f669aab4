// Tuning only (round 3): what do 64 MiB uploads from a ring of pinned buffers reach -- the shape of the streamed capture
// (bin/openmp_task) --, by allocation flags, with one or two streams, with and without a host thread that refills the buffers?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/h2d_batches tools/h2d_batches.hip -lpthread && /tmp/h2d_batches
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t B = 64ull << 20, NB = 24, RING = 5;
    void *d[2]; CK(hipMalloc(&d[0], B)); CK(hipMalloc(&d[1], B));
    hipStream_t st[2]; CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    char *src = (char *)malloc(B); memset(src, 7, B);
    struct { const char *name; unsigned flags; } kinds[] = {{"hipHostMallocDefault", hipHostMallocDefault}, {"NonCoherent", hipHostMallocNonCoherent},
        {"Coherent", hipHostMallocCoherent}, {"WriteCombined", hipHostMallocWriteCombined}, {"Portable", hipHostMallocPortable}, {"NumaUser", hipHostMallocNumaUser}};
    for (auto &k : kinds) {
        void *h[RING];
        bool ok = true;
        for (size_t i = 0; i < RING; i++) if (hipHostMalloc(&h[i], B, k.flags) != hipSuccess) { ok = false; (void)hipGetLastError(); break; }
        if (!ok) { printf("%-22s: allocation refused\n", k.name); continue; }
        for (size_t i = 0; i < RING; i++) memset(h[i], 1, B);
        for (int streams = 1; streams <= 2; streams++)
            for (int refill = 0; refill <= 1; refill++) {
                for (int rep = 0; rep < 2; rep++) {
                    std::atomic<bool> stop{false};
                    std::thread filler;
                    if (refill) filler = std::thread([&] { size_t i = 2; while (!stop) { memcpy(h[i % RING], src, B); i++; } });
                    CK(hipDeviceSynchronize());
                    const double t = now();
                    for (size_t b = 0; b < NB; b++) CK(hipMemcpyAsync(d[b % streams], h[b % RING], B, hipMemcpyHostToDevice, st[b % streams]));
                    CK(hipStreamSynchronize(st[0])); CK(hipStreamSynchronize(st[1]));
                    const double dt = now() - t;
                    stop = true;
                    if (refill) filler.join();
                    if (rep) printf("%-22s %d stream(s), host refilling the ring: %-3s  %zu x 64 MiB in %.1f ms = %.1f GB/s\n", k.name, streams, refill ? "yes" : "no", NB, dt * 1e3, NB * B / dt / 1e9);
                }
            }
        for (size_t i = 0; i < RING; i++) CK(hipHostFree(h[i]));
    }
    return 0;
}
