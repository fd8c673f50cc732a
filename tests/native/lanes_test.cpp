// CPU test driver of the host-side transfer helpers (sparsespatialsampling_amd/csrc/host_lanes.h), built by
// tests/test_sanitizers.py with -fsanitize=thread and with -fsanitize=address,undefined (there is no GPU sanitizer on the pool).
//   LanePool     -- jobs of changing width, shutdown and reuse, a forked child that starts lanes of its own
//   StreamPacker -- random pieces appended through the bounce buffer / streamed directly == memcpy of the concatenation
#include "host_lanes.h"

#include <atomic>
#include <cstdio>
#include <random>
#include <sys/wait.h>

static int fail(const char *what) {
    std::fprintf(stderr, "lanes_test: %s\n", what);
    return 1;
}

static int pool_jobs(s3::LanePool &pool, int rounds) {
    std::mt19937 g(7);
    for (int r = 0; r < rounds; ++r) {
        const int n = 1 + (int)(g() % 12);
        std::vector<int> hits((size_t)n, 0);
        std::atomic<long> sum{0};
        pool.run(n, [&](int t) {
            hits[(size_t)t] += 1;                      // every lane exactly once, no lane beyond n
            sum += t + 1;
        });
        for (int t = 0; t < n; ++t)
            if (hits[(size_t)t] != 1) return fail("a lane ran zero times or twice");
        if (sum.load() != (long)n * (n + 1) / 2) return fail("lane ids wrong");
    }
    return 0;
}

int main() {
    s3::LanePool &pool = *new s3::LanePool();
    if (pool_jobs(pool, 400)) return 1;
    if (pool.shutdown() < 1) return fail("shutdown joined no thread");
    if (pool.shutdown() != 0) return fail("second shutdown found threads");
    if (pool_jobs(pool, 100)) return 1;                 // usable again after a shutdown
    // a forked child inherits the bookkeeping of threads it does not have: it must start its own and leave the handles alone
#ifndef LANES_NO_FORK                                  /* (ThreadSanitizer refuses new threads in the child of a threaded process) */
    const pid_t child = fork();
    if (child == 0) {
        const int rc = pool_jobs(pool, 50);
        pool.shutdown();
        _exit(rc);
    }
    int status = 0;
    if (waitpid(child, &status, 0) != child || !WIFEXITED(status) || WEXITSTATUS(status) != 0) return fail("forked child failed");
#endif
    if (pool_jobs(pool, 50)) return 1;
    if (pool.shutdown() < 1) return fail("final shutdown joined no thread");

    std::mt19937 g(11);
    for (int round = 0; round < 60; ++round) {
        const size_t total = 1 + g() % (3u << 20);
        std::vector<char> src(total);
        for (auto &c : src) c = (char)g();
        char *dst = nullptr, *ref = nullptr;
        if (posix_memalign(reinterpret_cast<void **>(&dst), 64, total + 64) || posix_memalign(reinterpret_cast<void **>(&ref), 64, total + 64))
            return fail("posix_memalign");
        std::memset(dst, 0x5a, total + 64);
        s3::StreamPacker pack(dst);
        size_t off = 0;
        while (off < total) {                           // pieces from 1 byte to 64 KiB, as rows of any length come in
            size_t n = (g() % 4 == 0) ? 1 + g() % 65536 : 1 + g() % 400;
            n = std::min(n, total - off);
            pack.append(src.data() + off, n);
            off += n;
        }
        pack.finish();
        if (std::memcmp(dst, src.data(), total) != 0) return fail("StreamPacker: bytes differ from the concatenation");
        for (size_t i = total; i < total + 64; ++i)
            if (dst[i] != 0x5a) return fail("StreamPacker: wrote beyond the end");
        std::free(dst);
        std::free(ref);
    }
    std::puts("lanes_test ok");
    return 0;
}
