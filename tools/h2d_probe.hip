// Tuning only: what does it cost to get a 1.5 GB file image to the GPU?  pinned allocation, pageable copy,
// copy straight from a mapped file, hipHostRegister of the mapping.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t n = 1504ull * 1000000ull;
    const char *path = argc > 1 ? argv[1] : "/tmp/h2d_probe.bin";
    void *d; CK(hipMalloc(&d, n)); CK(hipMemset(d, 0, n)); CK(hipDeviceSynchronize());
    double t = now(); void *p; CK(hipHostMalloc(&p, n, hipHostMallocDefault)); printf("hipHostMalloc 1.5 GB            : %.3f s\n", now() - t);
    t = now(); memset(p, 1, n); printf("memset pinned (1 thread)         : %.3f s\n", now() - t);
    t = now(); CK(hipMemcpy(d, p, n, hipMemcpyHostToDevice)); printf("H2D from pinned                  : %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9);
    t = now(); CK(hipHostFree(p)); printf("hipHostFree                      : %.3f s\n", now() - t);
    t = now(); char *q = (char *)malloc(n); memset(q, 2, n); printf("malloc + first touch             : %.3f s\n", now() - t);
    t = now(); CK(hipMemcpy(d, q, n, hipMemcpyHostToDevice)); printf("H2D from pageable                : %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(d, q, n, hipMemcpyHostToDevice)); printf("H2D from pageable, again         : %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9);
    int fd = open(path, O_CREAT | O_TRUNC | O_RDWR, 0600);
    for (size_t o = 0; o < n; ) { ssize_t w = write(fd, q + o, n - o > (64u << 20) ? (64u << 20) : n - o); if (w <= 0) { perror("write"); return 1; } o += (size_t)w; }
    free(q);
    t = now(); void *m = mmap(NULL, n, PROT_READ, MAP_PRIVATE, fd, 0); madvise(m, n, MADV_WILLNEED);
    CK(hipMemcpy(d, m, n, hipMemcpyHostToDevice)); printf("H2D straight from the mapped file: %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(d, m, n, hipMemcpyHostToDevice)); printf("  again                          : %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9);
    t = now(); hipError_t e = hipHostRegister(m, n, hipHostRegisterReadOnly | hipHostRegisterMapped); printf("hipHostRegister(mapping)         : %.3f s (%s)\n", now() - t, hipGetErrorString(e));
    if (e == hipSuccess) { t = now(); CK(hipMemcpy(d, m, n, hipMemcpyHostToDevice)); printf("H2D from the registered mapping  : %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9); hipHostUnregister(m); }
    else (void)hipGetLastError();
    munmap(m, n); close(fd); unlink(path);
    return 0;
}
