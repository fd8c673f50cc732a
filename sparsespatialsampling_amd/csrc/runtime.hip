// Runtime plumbing of libs3hip.so: error string, device selection, raw memory helpers for hosts without torch.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "host_lanes.h"
#include <execinfo.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <unistd.h>
#include <sched.h>
#include <cstdio>
#include <string>

namespace s3 {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- staged upload of pageable host rows -------------------------------------------------------------------------
// A snapshot batch arrives as a pageable host tensor; handed to hipMemcpy as it is, it goes up at 14-26 GB/s on an
// MI355X host (the runtime stages it through one thread).  Here a few host threads copy row chunks into their own
// persistent pinned buffers and each chunk goes up asynchronously, so the host-side copies of all threads and the DMA
// transfers overlap.
namespace {
constexpr int UP_THREADS_MAX = 32;
constexpr int UP_BUFS = 2;                           // pinned buffers per thread
constexpr size_t UP_CHUNK_BYTES = (size_t)8 << 20;   // per buffer

struct UploadLane {
    void *pinned[UP_BUFS] = {nullptr, nullptr};
    hipEvent_t ev[UP_BUFS] = {nullptr, nullptr};
    hipStream_t down = nullptr;                      // the lane's own stream for downloads (kept: creating and destroying one per
                                                     // call and thread cost a download ~2.5 ms -- 100 MB took 10.9 ms)
    int next = 0;                                    // buffer the lane's next chunk goes through (rotates ACROSS calls: a small
                                                     // upload must not wait for the previous small upload's copy)
};
std::mutex g_upload_mutex;
UploadLane g_lanes[UP_THREADS_MAX];

// host threads that pack rows into the pinned lanes (S3_UPLOAD_THREADS overrides; at most UP_THREADS_MAX).  EIGHT: more of
// them pack faster than the link carries and then fight the link's DMA -- and the download of the previous batch, which
// runs at the same time -- for the host's memory system.  MI355X host with 128 hardware threads, 2.43 M referenced rows picked
// out of a pageable tensor, batches back to back (tools/e2e_probe.py): 200 snapshots (800-byte rows) 47 / 46 / 49 / 50 / 56 ms
// per batch with 6 / 8 / 12 / 16 / 24 threads (32: 67 -- slower than waiting for the download inside the call, 60);
// 25 snapshots (100-byte rows) 6.3 / 6.5 / 8.1 / 7.7 / 8.6 ms
int upload_threads() {
    static const int v = [] {
        const char *e = getenv("S3_UPLOAD_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        int n = e ? atoi(e) : std::min(8, std::max(1, hw / 2));
        return std::max(1, std::min(n, UP_THREADS_MAX));
    }();
    return v;
}

// ... for LARGE uploads only: a small batch's staging lines are still in the last-level cache when the DMA engine comes for them
// (16 buffers of 8 MB), and streaming them to DRAM first makes it slower.  MI355X host, batches back to back, download of the
// previous batch running (tools/e2e_probe.py; streamed / cached): 243 MB (25 snapshots) 9.7 / 6.5-7.2 ms, 1.94 GB (200) 44.2 /
// 45.7-53.0 ms, 9.7 GB (1000) 225 / 263-266 ms.  S3_UPLOAD_NT=0 / 1 forces one form.
bool upload_streaming_stores(int64_t total_bytes) {
    static const int v = [] { const char *e = getenv("S3_UPLOAD_NT"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    return v < 0 ? total_bytes >= ((int64_t)1 << 30) : v == 1;
}

// The host cores next to the GPU.  On a two-socket host a transfer thread on the far socket packs into pinned memory the DMA engine
// reads across the socket link, and the Linux scheduler puts a process's threads on either socket as it likes: 25-snapshot batches
// through ExportData took 7.6-8.2 ms per batch with the threads anywhere (or on the far socket) and 5.7-6.8 ms on the GPU's socket
// (MI355X box, 2 x EPYC 9575F; tools/e2e_probe.py under taskset).  So the packing / draining threads of an upload or a download
// -- and the calling thread for the duration of the call -- run on the CPUs of the GPU's NUMA node (PCI bus id ->
// /sys/bus/pci/devices/<id>/numa_node -> /sys/devices/system/node/node<N>/cpulist), within the affinity mask the process was
// given; the pinned staging buffers, allocated by such a thread, land on that node.  S3_NUMA_PIN=0 switches it off; anything
// unreadable (no sysfs, one node, an empty intersection) leaves the threads alone.
struct NodeCpus {
    bool known = false;
    cpu_set_t set;
};
const NodeCpus &gpu_node_cpus(int dev) {
    static NodeCpus table[16];
    static std::mutex m;
    static bool tried[16] = {};
    static const NodeCpus none{};
    if (dev < 0 || dev >= 16) return none;
    std::lock_guard<std::mutex> g(m);
    if (tried[dev]) return table[dev];
    tried[dev] = true;
    const char *sw = getenv("S3_NUMA_PIN");
    if (sw && sw[0] == '0') return table[dev];
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) != hipSuccess) return table[dev];
    for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
    int node = -1;
    if (FILE *f = fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return table[dev];
    char list[4096] = "";
    if (FILE *f = fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r")) {
        if (!fgets(list, (int)sizeof(list), f)) list[0] = 0;
        fclose(f);
    }
    cpu_set_t set;
    CPU_ZERO(&set);
    int n_set = 0;
    for (const char *c = list; *c;) {                        // "0-63,128-191"
        char *end = nullptr;
        const long a = strtol(c, &end, 10);
        if (end == c) break;
        long b = a;
        c = end;
        if (*c == '-') {
            b = strtol(c + 1, &end, 10);
            c = end;
        }
        for (long i = a; i <= b && i < CPU_SETSIZE; ++i) {
            CPU_SET((int)i, &set);
            ++n_set;
        }
        if (*c == ',') ++c;
        else break;
    }
    if (n_set > 0) {
        table[dev].set = set;
        table[dev].known = true;
    }
    return table[dev];
}
// the affinity mask the process was given, saved the first time anybody asks (before this library has moved any thread): the pool's
// threads are pinned PERMANENTLY, and a thread pinned to the node of the first GPU it served must be able to move to another
// GPU's node later -- the intersection is taken with this mask, not with the thread's present one
const cpu_set_t &process_mask() {
    static const cpu_set_t saved = [] {
        cpu_set_t m;
        CPU_ZERO(&m);
        if (sched_getaffinity(getpid(), sizeof(m), &m) != 0)
            for (int i = 0; i < CPU_SETSIZE; ++i) CPU_SET(i, &m);
        return m;
    }();
    return saved;
}
// this thread onto the GPU's node (within the process's mask); restores its previous mask when it goes out of scope if `restore`
struct OnGpuNode {
    cpu_set_t before;
    bool changed = false, restore;
    OnGpuNode(int dev, bool restore_) : restore(restore_) {
        const cpu_set_t &allowed = process_mask();
        const NodeCpus &nc = gpu_node_cpus(dev);
        if (!nc.known || pthread_getaffinity_np(pthread_self(), sizeof(before), &before) != 0) return;
        cpu_set_t want;
        CPU_AND(&want, &allowed, &nc.set);
        if (CPU_COUNT(&want) == 0 || CPU_EQUAL(&want, &before)) return;
        changed = pthread_setaffinity_np(pthread_self(), sizeof(want), &want) == 0;
    }
    ~OnGpuNode() {
        if (changed && restore) (void)pthread_setaffinity_np(pthread_self(), sizeof(before), &before);
    }
};

LanePool &g_pool = *new LanePool();      // (the object itself is never destroyed; its threads end in s3_shutdown)

hipError_t upload_lane_init(UploadLane &l) {
    for (int b = 0; b < UP_BUFS; ++b) {
        if (!l.pinned[b]) {
            hipError_t e = hipHostMalloc(&l.pinned[b], UP_CHUNK_BYTES, hipHostMallocPortable);
            if (e != hipSuccess) return e;
        }
        if (!l.ev[b]) {
            hipError_t e = hipEventCreateWithFlags(&l.ev[b], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}
}  // namespace

}  // namespace s3

extern "C" {

// h_rows == NULL: rows 0..n_rows-1 of h_src; otherwise the rows listed in h_rows[n_rows] (any order), packed on the device.
// Source row r starts at h_src + r * src_stride + src_off and contributes n_seg segments of seg_bytes, seg_stride apart
// (a snapshot piece [t0, t1) of a field [N, n_comp, T]: n_comp segments of (t1 - t0) values, T values apart); the device
// row holds them back to back.
static int upload_rows_impl(const void *h_src, const int32_t *h_rows, int64_t n_rows, int64_t src_stride, int64_t src_off,
                            int n_seg, int64_t seg_bytes, int64_t seg_stride, void *d_dst, int64_t dst_pitch_bytes,
                            s3_stream stream) try {
    using namespace s3;
    const int64_t row_bytes = (int64_t)n_seg * seg_bytes;
    S3_REQUIRE(n_rows >= 0 && n_seg >= 1 && seg_bytes >= 0 && dst_pitch_bytes >= row_bytes && src_off >= 0 &&
               (n_seg == 1 || seg_stride >= seg_bytes) && src_stride >= src_off + (n_seg - 1) * seg_stride + seg_bytes,
               "s3_upload_rows: bad shape");
    if (n_rows == 0 || row_bytes == 0) return S3_OK;
    S3_REQUIRE(h_src && d_dst, "s3_upload_rows: null array");
    S3_REQUIRE((size_t)dst_pitch_bytes <= UP_CHUNK_BYTES, "s3_upload_rows: rows longer than %zu bytes are not staged", UP_CHUNK_BYTES);
    hipStream_t st = as_stream(stream);
    std::lock_guard<std::mutex> guard(g_upload_mutex);
    int dev = 0;
    S3_HIP_CHECK(hipGetDevice(&dev));
    OnGpuNode caller_near(dev, true);                // (also while the lanes' pinned buffers are allocated: first touch)
    // chunk = what one thread packs before its transfer is queued.  No transfer can start before somebody's first chunk is
    // packed (~0.4 ms per MB and thread), so the first round of chunks is small and the rounds double up to 8 MB: the 243 MB of
    // a 25-snapshot batch used to be thirty 8-MB chunks packed side by side and THEN sent (7.5 ms where the link needs 4.3).
    // Small chunks throughout are no answer: 1-MB chunks made the 32 threads queue 240 copies on one stream and the runtime's
    // per-call cost, serialised by the stream's lock, doubled the time.
    const int n_lanes = upload_threads();
    std::vector<int64_t> chunk_begin;
    for (int64_t r = 0, c = 0; r < n_rows; ++c) {
        const int64_t bytes = std::min<int64_t>((int64_t)UP_CHUNK_BYTES, ((int64_t)512 << 10) << std::min<int64_t>(4, c / n_lanes));
        chunk_begin.push_back(r);
        r += std::max<int64_t>(1, bytes / dst_pitch_bytes);
    }
    const int64_t n_chunks = (int64_t)chunk_begin.size();
    chunk_begin.push_back(n_rows);
    const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)n_lanes, n_chunks));
    for (int t = 0; t < n_thr; ++t) S3_HIP_CHECK(upload_lane_init(g_lanes[t]));
    const bool whole_rows = n_seg == 1 && src_stride == row_bytes;       // the selection of rows is one dense block per run

    std::atomic<int64_t> next{0};
    std::atomic<int> first_error{(int)hipSuccess};
    auto work = [&](int t) {
        OnGpuNode near(dev, false);
        if (hipSetDevice(dev) != hipSuccess) { first_error = (int)hipErrorInvalidDevice; return; }
        UploadLane &l = g_lanes[t];
        int b = l.next;
        while (first_error.load() == (int)hipSuccess) {
            const int64_t c = next.fetch_add(1);
            if (c >= n_chunks) break;
            const int64_t r0 = chunk_begin[c], rows = chunk_begin[c + 1] - r0;
            hipError_t e = hipEventSynchronize(l.ev[b]);             // the buffer's previous transfer has left it
            if (e == hipSuccess) {
                const char *base = static_cast<const char *>(h_src) + src_off;
                char *stage = static_cast<char *>(l.pinned[b]);
                char *dst = static_cast<char *>(d_dst) + r0 * dst_pitch_bytes;
                // long rows are packed in the pinned buffer and the 2-D copy converts the pitch (4000-byte rows: 49 GB/s);
                // short rows are laid out with the device pitch on the host and go up as one contiguous copy (100-byte
                // rows: 32 GB/s, 21 GB/s as a 2-D copy); the padding of the chunk's last row is left alone
                const bool pitched = row_bytes != dst_pitch_bytes && row_bytes < 512;
                const int64_t step = pitched ? dst_pitch_bytes : row_bytes;
                if (!h_rows && whole_rows && !pitched) {
                    std::memcpy(stage, base + r0 * row_bytes, (size_t)(rows * row_bytes));
                } else {
                    // a scattered selection of rows (or a piece of every row) defeats the hardware prefetcher: every row would
                    // start with a full memory latency.  The lines of the row `ahead` rows further on are requested while this
                    // one is copied (ahead chosen so that ~4 KiB per thread are on their way)
                    const int64_t ahead = std::max<int64_t>(2, std::min<int64_t>(32, 4096 / std::max<int64_t>(64, row_bytes)));
                    auto row_of = [&](int64_t r) { return base + (int64_t)(h_rows ? h_rows[r0 + r] : r0 + r) * src_stride; };
                    if (!pitched && upload_streaming_stores(n_rows * row_bytes)) {             // rows back to back: streamed out line by line
                        StreamPacker pack(stage);
                        for (int64_t r = 0; r < rows; ++r) {
                            const char *src = row_of(r);
                            for (int sgm = 0; sgm < n_seg; ++sgm) pack.append(src + sgm * seg_stride, (size_t)seg_bytes);
                        }
                        pack.finish();
                    } else
                    for (int64_t r = 0; r < rows; ++r) {
                        if (r + ahead < rows) {
                            const char *nx = row_of(r + ahead);
                            for (int sgm = 0; sgm < n_seg; ++sgm)
                                for (int64_t o = 0; o < seg_bytes + 63; o += 64) __builtin_prefetch(nx + sgm * seg_stride + std::min(o, seg_bytes - 1), 0, 0);
                        }
                        const char *src = row_of(r);
                        if (n_seg == 1) {
                            std::memcpy(stage + r * step, src, (size_t)row_bytes);
                        } else {
                            for (int sgm = 0; sgm < n_seg; ++sgm)
                                std::memcpy(stage + r * step + sgm * seg_bytes, src + sgm * seg_stride, (size_t)seg_bytes);
                        }
                    }
                }
                if (row_bytes == dst_pitch_bytes)
                    e = hipMemcpyAsync(dst, stage, (size_t)(rows * row_bytes), hipMemcpyHostToDevice, st);
                else if (pitched)
                    e = hipMemcpyAsync(dst, stage, (size_t)((rows - 1) * dst_pitch_bytes + row_bytes), hipMemcpyHostToDevice, st);
                else
                    e = hipMemcpy2DAsync(dst, (size_t)dst_pitch_bytes, stage, (size_t)row_bytes, (size_t)row_bytes, (size_t)rows,
                                         hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipEventRecord(l.ev[b], st);
            }
            if (e != hipSuccess) { first_error = (int)e; break; }
            b = (b + 1) % UP_BUFS;
        }
        l.next = b;
    };
    const bool trace = getenv("S3_UPLOAD_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    const auto t_spawned = std::chrono::steady_clock::now();
    g_pool.run(n_thr, work);
    if (trace) {
        const auto t_end = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(st);
        const auto t_sync = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[s3_upload] %lld rows x %lld B, %lld chunks, %d threads: spawn %.2f ms, packed+queued %.2f ms, transfers done %.2f ms\n",
                (long long)n_rows, (long long)row_bytes, (long long)n_chunks, n_thr, ms(t_begin, t_spawned), ms(t_begin, t_end), ms(t_begin, t_sync));
    }
    S3_HIP_CHECK((hipError_t)first_error.load());
    return S3_OK;
} catch (const std::exception &e) {          // thread creation
    s3::set_error("s3_upload_rows: %s", e.what());
    return S3_ENOMEM;
}

int s3_upload_rows(const void *h_src, int64_t n_rows, int64_t row_bytes, void *d_dst, int64_t dst_pitch_bytes,
                   s3_stream stream) {
    return upload_rows_impl(h_src, nullptr, n_rows, row_bytes, 0, 1, row_bytes, row_bytes, d_dst, dst_pitch_bytes, stream);
}

int s3_upload_rows_indexed(const void *h_src, const int32_t *h_rows, int64_t n_sel, int64_t row_bytes, void *d_dst,
                           int64_t dst_pitch_bytes, s3_stream stream) {
    S3_REQUIRE(n_sel == 0 || h_rows != nullptr, "s3_upload_rows_indexed: null row list");
    return upload_rows_impl(h_src, h_rows, n_sel, row_bytes, 0, 1, row_bytes, row_bytes, d_dst, dst_pitch_bytes, stream);
}

int s3_upload_row_pieces(const void *h_src, const int32_t *h_rows, int64_t n_rows, int64_t src_row_stride_bytes,
                         int64_t src_offset_bytes, int n_segments, int64_t segment_bytes, int64_t segment_stride_bytes,
                         void *d_dst, int64_t dst_pitch_bytes, s3_stream stream) {
    return upload_rows_impl(h_src, h_rows, n_rows, src_row_stride_bytes, src_offset_bytes, n_segments, segment_bytes,
                            segment_stride_bytes, d_dst, dst_pitch_bytes, stream);
}

// Device -> pageable host memory through the same pinned lanes: every thread brings 8 MiB chunks down into its pinned
// buffers (asynchronous copies on `stream`) and moves the chunk before last into the caller's array meanwhile.  Returns
// when h_dst is complete.  A plain hipMemcpy into pageable memory runs at 12-15 GB/s on an MI355X host.
int s3_download(void *h_dst, const void *d_src, size_t bytes, s3_stream stream) try {
    using namespace s3;
    if (bytes == 0) return S3_OK;
    S3_REQUIRE(h_dst && d_src, "s3_download: null array");
    hipStream_t st = as_stream(stream);
    if (bytes < 4 * UP_CHUNK_BYTES) {
        // small: not worth the threads -- but through a page-locked buffer of the library all the same (r5): handed a pageable
        // destination the runtime pins the caller's pages on the fly, and that path produced rare GPU memory faults ("write access
        // to a read-only page" at a host heap address, HISTORY 9)
        std::lock_guard<std::mutex> guard(g_upload_mutex);
        S3_HIP_CHECK(upload_lane_init(g_lanes[0]));
        UploadLane &l = g_lanes[0];
        for (int b = 0; b < UP_BUFS; ++b) S3_HIP_CHECK(hipEventSynchronize(l.ev[b]));       // (an earlier upload may still read from them)
        for (size_t off = 0; off < bytes; off += UP_CHUNK_BYTES) {
            const size_t n = std::min(UP_CHUNK_BYTES, bytes - off);
            S3_HIP_CHECK(hipMemcpyAsync(l.pinned[0], static_cast<const char *>(d_src) + off, n, hipMemcpyDeviceToHost, st));
            S3_HIP_CHECK(hipStreamSynchronize(st));
            std::memcpy(static_cast<char *>(h_dst) + off, l.pinned[0], n);
        }
        return S3_OK;
    }
    std::lock_guard<std::mutex> guard(g_upload_mutex);
    int dev = 0;
    S3_HIP_CHECK(hipGetDevice(&dev));
    OnGpuNode caller_near(dev, true);
    S3_HIP_CHECK(hipStreamSynchronize(st));                              // the source is complete; lanes use their own order
    const int64_t n_chunks = (int64_t)((bytes + UP_CHUNK_BYTES - 1) / UP_CHUNK_BYTES);
    const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)8, (int64_t)upload_threads(), n_chunks / 2}));
    for (int t = 0; t < n_thr; ++t) S3_HIP_CHECK(upload_lane_init(g_lanes[t]));
    std::atomic<int64_t> next{0};
    std::atomic<int> first_error{(int)hipSuccess};
    auto work = [&](int t) {
        OnGpuNode near(dev, false);
        if (hipSetDevice(dev) != hipSuccess) { first_error = (int)hipErrorInvalidDevice; return; }
        UploadLane &l = g_lanes[t];
        static const bool keep = [] { const char *e = getenv("S3_DOWNLOAD_KEEP_STREAMS"); return !(e && e[0] == '0'); }();
        hipStream_t fresh = nullptr;
        if (!keep) {
            if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) { first_error = (int)hipErrorUnknown; return; }
        } else if (!l.down && hipStreamCreateWithFlags(&l.down, hipStreamNonBlocking) != hipSuccess) {
            l.down = nullptr;
            first_error = (int)hipErrorUnknown;
            return;
        }
        const hipStream_t own = keep ? l.down : fresh;
        int64_t pending[UP_BUFS] = {-1, -1};                              // chunk sitting in each pinned buffer
        auto drain = [&](int b) {
            if (pending[b] < 0) return hipSuccess;
            const hipError_t e = hipEventSynchronize(l.ev[b]);
            if (e != hipSuccess) return e;
            const size_t off = (size_t)pending[b] * UP_CHUNK_BYTES;
            std::memcpy(static_cast<char *>(h_dst) + off, l.pinned[b], std::min(UP_CHUNK_BYTES, bytes - off));
            pending[b] = -1;
            return hipSuccess;
        };
        int b = 0;
        hipError_t e = hipSuccess;
        while (e == hipSuccess && first_error.load() == (int)hipSuccess) {
            const int64_t c = next.fetch_add(1);
            if (c >= n_chunks) break;
            e = drain(b);                                                // the buffer's previous chunk goes to the caller first
            if (e != hipSuccess) break;
            const size_t off = (size_t)c * UP_CHUNK_BYTES;
            e = hipMemcpyAsync(l.pinned[b], static_cast<const char *>(d_src) + off, std::min(UP_CHUNK_BYTES, bytes - off),
                               hipMemcpyDeviceToHost, own);
            if (e == hipSuccess) e = hipEventRecord(l.ev[b], own);
            pending[b] = c;
            b = (b + 1) % UP_BUFS;
        }
        for (int k = 0; k < UP_BUFS && e == hipSuccess; ++k) e = drain((b + k) % UP_BUFS);
        (void)hipStreamSynchronize(own);
        if (fresh) (void)hipStreamDestroy(fresh);
        if (e != hipSuccess) first_error = (int)e;
    };
    g_pool.run(n_thr, work);
    S3_HIP_CHECK((hipError_t)first_error.load());
    return S3_OK;
} catch (const std::exception &e) {
    s3::set_error("s3_download: %s", e.what());
    return S3_ENOMEM;
}

const char *s3_last_error(void) { return s3::g_err; }

// Debugging aid (S3_ABORT_BACKTRACE=1 through the bindings; off otherwise): a SIGABRT handler that writes the native frames of the
// ABORTING thread to stderr (glibc backtrace_symbols_fd: async-signal-safe enough for a process that is ending anyway) and then
// lets the default action run.  Python's faulthandler shows the interpreter's frames only; an abort() inside a runtime library says
// nothing about itself.
static struct sigaction s3_previous_abort_action;
static int s3_abort_backtrace_fd = 2;                  // (a test runner may have redirected descriptor 2: S3_ABORT_BACKTRACE=<file>)
static void s3_abort_backtrace_handler(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    static const char head[] = "\n[s3] SIGABRT: native frames of the aborting thread:\n";
    if (write(s3_abort_backtrace_fd, head, sizeof(head) - 1) < 0) {}
    backtrace_symbols_fd(frames, n, s3_abort_backtrace_fd);
    (void)sigaction(sig, &s3_previous_abort_action, nullptr);      // whoever was there before (Python's faulthandler) goes next
    raise(sig);
}
int s3_debug_abort_backtrace(void) {
    // installed once: a second installation would save THIS handler as the "previous" one, and the handler would then restore
    // itself and raise again without end (SA_NODEFER) instead of reaching the default action
    static bool installed = false;
    if (installed) return S3_OK;
    {   // the first backtrace() of a process may load the unwinder (dlopen of libgcc): done here, not inside the signal handler
        void *warm[2];
        (void)backtrace(warm, 2);
    }
    if (const char *e = getenv("S3_ABORT_BACKTRACE"))
        if (e[0] == '/') {
            const int fd = open(e, O_WRONLY | O_CREAT | O_APPEND, 0644);
            if (fd >= 0) s3_abort_backtrace_fd = fd;
        }
    struct sigaction sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.sa_handler = s3_abort_backtrace_handler;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_NODEFER;
    if (sigaction(SIGABRT, &sa, &s3_previous_abort_action) != 0) return S3_EINVAL;
    installed = true;
    return S3_OK;
}

int s3_abi_version(void) { return S3_ABI_VERSION; }

// stops and joins the host threads the library keeps between calls (the transfer lanes); returns how many were joined.  Meant
// for the very end of a process: the bindings call it from an atexit hook that runs BEFORE the interpreter and the HIP runtime
// tear down (round 4 ended a suite run with a core dump in a joining destructor at static-destruction time and then detached
// the threads for good; joining them while everything they could touch is still alive is the orderly form).  Harmless at any
// other time: the next upload / download starts fresh lanes.
int s3_shutdown(void) try {
    // (never beside a transfer: the lanes' jobs run under this mutex.  A transfer still running two seconds into the process's
    // exit is left alone -- its threads then simply end with the process, as they did before round 5)
    std::unique_lock<std::mutex> guard(s3::g_upload_mutex, std::defer_lock);
    for (int i = 0; i < 200 && !guard.try_lock(); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(10));
    if (!guard.owns_lock()) return 0;
    return s3::g_pool.shutdown();
} catch (...) {                                      // (nothing may cross the C boundary: a failing join must not end the process)
    return -1;
}

int s3_device_count(int *h_count) {
    S3_REQUIRE(h_count != nullptr, "s3_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *h_count = 0;
        s3::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return S3_ENODEV;
    }
    *h_count = n;
    return S3_OK;
}

int s3_set_device(int device) {
    S3_HIP_CHECK(hipSetDevice(device));
    return S3_OK;
}

int s3_malloc(void **d_ptr, size_t bytes) {
    S3_REQUIRE(d_ptr != nullptr, "s3_malloc: null output");
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        s3::set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP;
    }
    return S3_OK;
}

int s3_free(void *d_ptr) {
    if (d_ptr) S3_HIP_CHECK(hipFree(d_ptr));
    return S3_OK;
}

// Host memory that several processes share (a POSIX shared-memory mapping: the sharded export's snapshot-major batch buffer)
// made visible to this process's device, so that kernels can write their results straight into it over this GPU's own PCIe link.
int s3_host_register(void *h_ptr, size_t bytes, void **d_ptr) {
    S3_REQUIRE(h_ptr != nullptr && bytes > 0 && d_ptr != nullptr, "s3_host_register: null argument");
    *d_ptr = nullptr;
    hipError_t e = hipHostRegister(h_ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        s3::set_error("hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP;
    }
    e = hipHostGetDevicePointer(d_ptr, h_ptr, 0);
    if (e != hipSuccess) {
        (void)hipHostUnregister(h_ptr);
        s3::set_error("hipHostGetDevicePointer: %s", hipGetErrorString(e));
        return S3_EHIP;
    }
    return S3_OK;
}

int s3_host_unregister(void *h_ptr) {
    if (h_ptr) S3_HIP_CHECK(hipHostUnregister(h_ptr));
    return S3_OK;
}

// page-locked (hipHostMalloc / hipHostRegister) host memory?  Only such a pointer is handed to the runtime's copy engine as it is:
// given a PAGEABLE pointer the runtime pins the caller's pages on the fly for copies of a megabyte and more, and that path ended
// rare test processes of round 5 with "Memory access fault by GPU ... write access to a read-only page" at a host heap address
// (DESIGN §8).  Pageable memory goes through the library's own page-locked lanes instead, whatever the size.
static bool host_memory_is_page_locked(const void *p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                       // (an unregistered pointer is an "error" for older runtimes: not ours)
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int s3_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, s3_stream stream) {
    if (bytes == 0) return S3_OK;
    S3_REQUIRE(d_dst && h_src, "s3_memcpy_h2d: null array");
    if (host_memory_is_page_locked(h_src)) {
        S3_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s3::as_stream(stream)));
        return S3_OK;
    }
    // pageable source: 1-MiB rows through the staged upload (+ the tail as one short row); complete on return
    const int64_t row = 1 << 20, n_full = (int64_t)(bytes / (size_t)row);
    const size_t tail = bytes - (size_t)n_full * (size_t)row;
    if (n_full) {
        const int rc = s3_upload_rows(h_src, n_full, row, d_dst, row, stream);
        if (rc != S3_OK) return rc;
    }
    if (tail) {
        const int rc = s3_upload_rows(static_cast<const char *>(h_src) + (size_t)n_full * row, 1, (int64_t)tail,
                                      static_cast<char *>(d_dst) + (size_t)n_full * row, (int64_t)tail, stream);
        if (rc != S3_OK) return rc;
    }
    S3_HIP_CHECK(hipStreamSynchronize(s3::as_stream(stream)));
    return S3_OK;
}

int s3_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, s3_stream stream) {
    if (bytes == 0) return S3_OK;
    S3_REQUIRE(h_dst && d_src, "s3_memcpy_d2h: null array");
    if (host_memory_is_page_locked(h_dst)) {
        S3_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s3::as_stream(stream)));
        return S3_OK;
    }
    return s3_download(h_dst, d_src, bytes, stream);          // pageable destination: staged, complete on return
}

int s3_stream_synchronize(s3_stream stream) {
    S3_HIP_CHECK(hipStreamSynchronize(s3::as_stream(stream)));
    return S3_OK;
}

}  // extern "C"
