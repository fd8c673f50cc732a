// dev probe (standalone, runs on the GPU box): how fast can 92 MB -- one 25-snapshot batch of the cylinder3D grid, float64 -- get from a
// page-locked host buffer (or from the device) into ONE regular file of the box's file system?  Decides VERDICT r5 item 3.
//   (1) pwrite, 1 / 4 / 8 threads on disjoint ranges of one fd            (what libs3h5's writer does today)
//   (2) the same into a range fallocate()d first
//   (3) mmap(MAP_SHARED) of the range + memcpy, 1 / 8 threads               (page fault per 4 KiB inside the copy)
//   (4) mmap + madvise(MADV_POPULATE_WRITE) by 1 / 8 threads, then memcpy by 8 threads   (faults taken ahead, in parallel)
//   (5) hipHostRegister of the mapping (does the driver take a file-backed mapping at all?) and a device-to-host copy into it
// hipcc --offload-arch=gfx950 -O2 -o tools/bin/sink_probe tools/sink_probe.hip -lpthread && tools/bin/sink_probe /tmp
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t N = (size_t)461130 * 25 * 8 / 4096 * 4096;        // 92 MB, whole pages
template <typename F> static void par(int n, F f) { std::vector<std::thread> t; for (int i = 0; i < n; ++i) t.emplace_back(f, i); for (auto &x : t) x.join(); }

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    char path[512]; snprintf(path, sizeof path, "%s/s3_sink_probe.bin", dir);
    char *src = nullptr;
    if (hipHostMalloc((void **)&src, N, hipHostMallocDefault) != hipSuccess) { printf("no pinned memory\n"); return 1; }
    memset(src, 1, N);
    const int reps = 6;
    auto fresh = [&](bool alloc) { unlink(path); int fd = open(path, O_RDWR | O_CREAT, 0644); if (alloc) { if (posix_fallocate(fd, 0, N * reps)) perror("fallocate"); } else { if (ftruncate(fd, N * reps)) perror("ftruncate"); } return fd; };
    for (int alloc = 0; alloc < 2; ++alloc)
        for (int nt : {1, 4, 8}) {
            int fd = fresh(alloc); double best = 1e9, sum = 0;
            for (int r = 0; r < reps; ++r) {
                const double t0 = now();
                par(nt, [&](int i) { size_t lo = N / nt * i, hi = i == nt - 1 ? N : N / nt * (i + 1); for (size_t o = lo; o < hi;) { ssize_t w = pwrite(fd, src + o, std::min<size_t>(hi - o, 8 << 20), (off_t)(r * N + o)); if (w <= 0) { perror("pwrite"); break; } o += w; } });
                const double dt = now() - t0; sum += dt; if (dt < best) best = dt;
            }
            printf("pwrite %s %d threads: mean %.2f ms, best %.2f ms (%.1f GB/s)\n", alloc ? "fallocated" : "sparse    ", nt, sum / reps * 1e3, best * 1e3, N / best / 1e9); fflush(stdout);
            close(fd);
        }
    for (int alloc = 0; alloc < 2; ++alloc)
        for (int mode = 0; mode < 4; ++mode) {       // 0: memcpy 1 thread, 1: memcpy 8 threads, 2: populate 1 + memcpy 8, 3: populate 8 + memcpy 8
            int fd = fresh(alloc); double best = 1e9, sum = 0, psum = 0;
            for (int r = 0; r < reps; ++r) {
                char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)(r * N));
                if (m == MAP_FAILED) { perror("mmap"); return 1; }
                const double t0 = now();
                double tp = 0;
                if (mode >= 2) {
                    const int np = mode == 2 ? 1 : 8;
                    par(np, [&](int i) { size_t lo = N / np / 4096 * 4096 * i, hi = i == np - 1 ? N : N / np / 4096 * 4096 * (i + 1); if (madvise(m + lo, hi - lo, MADV_POPULATE_WRITE)) perror("madvise"); });
                    tp = now() - t0;
                }
                const int nc = mode == 0 ? 1 : 8;
                par(nc, [&](int i) { size_t lo = N / nc / 4096 * 4096 * i, hi = i == nc - 1 ? N : N / nc / 4096 * 4096 * (i + 1); memcpy(m + lo, src + lo, hi - lo); });
                const double dt = now() - t0; sum += dt; psum += tp; if (dt < best) best = dt;
                munmap(m, N);
            }
            const char *names[] = {"memcpy 1 thread", "memcpy 8 threads", "populate 1 thread + memcpy 8", "populate 8 threads + memcpy 8"};
            printf("mmap %s %-30s: mean %.2f ms (populate %.2f), best %.2f ms (%.1f GB/s)\n", alloc ? "fallocated" : "sparse    ", names[mode], sum / reps * 1e3, psum / reps * 1e3, best * 1e3, N / best / 1e9); fflush(stdout);
            close(fd);
        }
    {   // does the driver take a file-backed mapping?
        int fd = fresh(1);
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double t0 = now();
        hipError_t e = hipHostRegister(m, N, hipHostRegisterDefault);
        printf("hipHostRegister(file mapping, %zu MB): %s in %.2f ms\n", N >> 20, hipGetErrorString(e), (now() - t0) * 1e3); fflush(stdout);
        if (e == hipSuccess) {
            char *d = nullptr; (void)hipMalloc((void **)&d, N); (void)hipMemset(d, 7, N); (void)hipDeviceSynchronize();
            for (int r = 0; r < 3; ++r) { t0 = now(); hipError_t c = hipMemcpy(m, d, N, hipMemcpyDeviceToHost); printf("  device -> mapping: %s, %.2f ms (%.1f GB/s)\n", hipGetErrorString(c), (now() - t0) * 1e3, N / (now() - t0) / 1e9); fflush(stdout); }
            printf("  first / last byte in the mapping: %d %d\n", m[0], m[N - 1]);
            t0 = now(); (void)hipHostUnregister(m); printf("  unregister %.2f ms\n", (now() - t0) * 1e3);
            (void)hipFree(d);
        }
        munmap(m, N); close(fd);
    }
    unlink(path);
    return 0;
}
