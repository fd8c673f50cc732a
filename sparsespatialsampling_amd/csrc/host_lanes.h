// Host-side helpers of the staged transfers (csrc/runtime.hip): plain C++, no HIP types -- so that the CPU test job can build
// them with -fsanitize=thread / address,undefined (tests/native/lanes_test.cpp, tests/test_sanitizers.py; there is no GPU
// sanitizer on the pool).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <unistd.h>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

namespace s3 {

// Rows appended back to back into a 64-byte aligned pinned buffer with NON-TEMPORAL stores: a plain memcpy into the staging
// buffer first reads every destination line into the cache (write-allocate) and writes it back later, next to the DMA engine
// that reads the same lines for the transfer -- four trips through the host's memory system per byte uploaded.  Streaming
// stores skip the read and leave the caches to the source rows.
struct StreamPacker {
    static constexpr int BLOCK = 4096;               // flushed at a time; pieces of up to BLOCK bytes go through the bounce buffer
    char *dst;
    alignas(64) char bounce[2 * BLOCK];
    int fill = 0;
    explicit StreamPacker(char *d) : dst(d) {}
    static void stream(char *d, const char *s, size_t n) {      // n = multiple of 64, d 64-byte aligned
#if !defined(__x86_64__)
        std::memcpy(d, s, n);                                    // (no streaming stores spelled out for this host: cached copy)
#else
        for (size_t o = 0; o < n; o += 64) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + o)), b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + o + 16)),
                          c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + o + 32)), e = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + o + 48));
            _mm_stream_si128(reinterpret_cast<__m128i *>(d + o), a);
            _mm_stream_si128(reinterpret_cast<__m128i *>(d + o + 16), b);
            _mm_stream_si128(reinterpret_cast<__m128i *>(d + o + 32), c);
            _mm_stream_si128(reinterpret_cast<__m128i *>(d + o + 48), e);
        }
#endif
    }
    void append(const char *src, size_t n) {
        while (n) {
            if (fill == 0 && n >= BLOCK) {                       // long pieces: whole lines straight from the source
                const size_t whole = n & ~(size_t)63;
                stream(dst, src, whole);
                dst += whole; src += whole; n -= whole;
                continue;
            }
            const size_t m = std::min(n, (size_t)(2 * BLOCK - fill));
            std::memcpy(bounce + fill, src, m);                  // short pieces gather in the (cache-resident) bounce buffer ...
            fill += (int)m; src += m; n -= m;
            if (fill >= BLOCK) {                                 // ... and leave it a block at a time
                const int whole = fill & ~63;
                stream(dst, bounce, (size_t)whole);
                dst += whole;
                std::memmove(bounce, bounce + whole, (size_t)(fill - whole));
                fill -= whole;
            }
        }
    }
    void finish() {
        const int whole = fill & ~63;
        stream(dst, bounce, (size_t)whole);
        if (fill > whole) std::memcpy(dst + whole, bounce + whole, (size_t)(fill - whole));
#if defined(__x86_64__)
        _mm_sfence();
#endif
    }
};

// The lanes' host threads, kept between calls (creating seven threads cost an upload or download ~0.2 ms: 4 % of a 25-snapshot
// batch).  run(n, fn): fn(0) on the caller, fn(1 .. n-1) on the pool's threads; returns when all are done.  One job at a time (the
// callers hold g_upload_mutex).  S3_LANE_POOL=0: a fresh thread per lane and call, as before (A/B runs).
class LanePool {
    // everything the lanes synchronise on lives in a Core on the heap.  A forked child inherits the parent's Core with its condition
    // variables in the state the parent's WAITING lanes left them in -- threads the child does not have: glibc's broadcast then waits
    // for those phantom waiters to leave their group and never returns (found by tests/native/lanes_test.cpp: the child's third job
    // hung).  The child therefore starts with a fresh Core and leaves the inherited one alone (never freed, never joined).
    struct Core {
        std::mutex m;
        std::condition_variable cv_go, cv_done;
        const std::function<void(int)> *job = nullptr;
        int n_threads = 0, n_active = 0, remaining = 0;
        uint64_t generation = 0;
        bool stop = false;
        std::vector<std::thread> threads;           // joinable: shutdown() ends them in an orderly way
    };
    Core *core = nullptr;
    pid_t owner = 0;                                // (callers serialise run() / shutdown(): the transfers hold one mutex)

    static void loop(Core *c, int t, uint64_t seen) {
        while (true) {
            const std::function<void(int)> *fn = nullptr;
            {
                std::unique_lock<std::mutex> lk(c->m);
                c->cv_go.wait(lk, [&] { return c->stop || c->generation != seen; });
                if (c->stop) return;
                seen = c->generation;
                if (t < c->n_active) fn = c->job;
            }
            if (!fn) continue;
            (*fn)(t);
            std::lock_guard<std::mutex> lk(c->m);
            if (--c->remaining == 0) c->cv_done.notify_all();
        }
    }

public:
    void run(int n, const std::function<void(int)> &fn) {
        static const bool pooled = [] { const char *e = getenv("S3_LANE_POOL"); return !(e && e[0] == '0'); }();
        if (!pooled) {
            std::vector<std::thread> fresh;
            for (int t = 1; t < n; ++t) fresh.emplace_back(fn, t);
            fn(0);
            for (auto &w : fresh) w.join();
            return;
        }
        if (owner != getpid()) {                             // first use, or the Core belongs to the process this one was forked from
            owner = getpid();
            core = new Core();
        }
        Core *c = core;
        {
            std::lock_guard<std::mutex> lk(c->m);
            while (c->n_threads < n - 1) {
                const int t = ++c->n_threads;
                const uint64_t start = c->generation;        // (a thread born now must not take a job that was finished before it)
                c->threads.emplace_back([c, t, start] { loop(c, t, start); });
            }
            c->job = &fn;
            c->n_active = n;
            c->remaining = n - 1;
            ++c->generation;
        }
        c->cv_go.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(c->m);
        c->cv_done.wait(lk, [&] { return c->remaining == 0; });
        c->job = nullptr;
    }

    // stop and join the lanes (idle between jobs: they wait on cv_go); returns how many.  The pool can be used again afterwards.
    int shutdown() {
        if (owner != getpid() || !core) return 0;
        Core *c = core;
        {
            std::lock_guard<std::mutex> lk(c->m);
            if (c->n_threads == 0) return 0;
            c->stop = true;
        }
        c->cv_go.notify_all();
        int joined = 0;
        for (auto &t : c->threads)
            if (t.joinable()) {
                t.join();
                ++joined;
            }
        std::lock_guard<std::mutex> lk(c->m);
        c->threads.clear();
        c->n_threads = 0;
        c->stop = false;
        return joined;
    }
};

}  // namespace s3
