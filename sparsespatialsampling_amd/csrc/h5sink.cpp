// libs3h5.so -- HDF5 sink / source of the S^3 export path on the HDF5 C library (include/s3h5.h).  Host code, no GPU.
//
// Reference behaviour covered: Datawriter.write_data (data.py:361-430: one dataset per call in grid/, constant/ or
// data/<time>/), the per-write-time loop of ExportData._write_data_to_hdf5 (export.py:283-299) and the reads of Dataloader /
// XDMFWriter (data.py:22-300, 504-777).  The on-disk layout is the reference's; what differs is how it gets there: a whole
// snapshot batch is queued with one call and written by a background thread from the snapshot-major host buffer while
// the GPU works on the next batch.  The HDF5 library does one thing at a time, and a single thread copies into the page
// cache at ~4 GB/s -- less than the GPU path delivers -- so the background thread only lets HDF5 create the datasets of a
// batch (contiguous layout, space allocated at creation, no fill values), asks for their file offsets (H5Dget_offset) and
// has a few threads pwrite() the raw values there through a second descriptor of the same file.
#include "s3h5.h"

#include <hdf5.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";
std::mutex g_hdf5;                       // the HDF5 library is not assumed to be thread safe: one call at a time

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

hid_t mem_type(int dtype) {
    switch (dtype) {
        case S3H5_F32: return H5T_NATIVE_FLOAT;
        case S3H5_F64: return H5T_NATIVE_DOUBLE;
        case S3H5_I32: return H5T_NATIVE_INT32;
        case S3H5_I64: return H5T_NATIVE_INT64;
        case S3H5_U8: return H5T_NATIVE_UINT8;
        default: return -1;
    }
}

struct Job {
    std::string path;
    int dtype, ndim;
    hsize_t dims[S3H5_MAX_DIMS];
    const void *data;
    size_t bytes;
    // (optional) the values are on their way into `data` -- a device-to-host copy the producer queued before this job: they
    // are complete once *ready >= ready_value (written by the producer's stream behind the copy)
    const volatile int32_t *ready = nullptr;
    int32_t ready_value = 0;
};

}  // namespace

struct Segment { off_t offset; const char *data; size_t bytes; };

struct s3h5_file {
    hid_t fid = -1;
    bool writable = false;
    std::string os_path;                 // for the second descriptor of the raw writes
    int raw_fd = -1;
    bool raw_refused = false;
    std::vector<std::pair<const char *, size_t>> busy_batch;      // buffers of the batch being written
    std::deque<Job> queue;
    std::mutex m;
    std::condition_variable cv_work, cv_idle;
    std::thread worker;
    bool stop = false;
    const void *busy_data = nullptr;     // buffer of the job being written
    size_t busy_bytes = 0;
    int async_error = 0;
    std::string async_message;
    int64_t skipped = 0;

    // creates missing intermediate groups (the caller has checked that the dataset does not exist and holds g_hdf5)
    int write_locked(const std::string &path, int dtype, int ndim, const hsize_t *dims, const void *data) {
        const hid_t mt = mem_type(dtype);
        if (mt < 0) { set_error("unknown element type %d", dtype); return S3H5_EINVAL; }
        hid_t lcpl = H5Pcreate(H5P_LINK_CREATE);
        H5Pset_create_intermediate_group(lcpl, 1);
        hid_t space = ndim == 0 ? H5Screate(H5S_SCALAR) : H5Screate_simple(ndim, dims, nullptr);
        hid_t ds = H5Dcreate2(fid, path.c_str(), mt, space, lcpl, H5P_DEFAULT, H5P_DEFAULT);
        int rc = S3H5_OK;
        if (ds < 0) {
            set_error("could not create dataset '%s'", path.c_str());
            rc = S3H5_EIO;
        } else {
            hsize_t n = 1;
            for (int i = 0; i < ndim; ++i) n *= dims[i];
            if (n > 0 && H5Dwrite(ds, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) {
                set_error("could not write dataset '%s'", path.c_str());
                rc = S3H5_EIO;
            }
            H5Dclose(ds);
        }
        H5Sclose(space);
        H5Pclose(lcpl);
        return rc;
    }

    // values of a dataset that exists already, through the library (the fallback of the raw path)
    int write_existing_locked(const std::string &path, int dtype, const void *data) {
        hid_t ds = H5Dopen2(fid, path.c_str(), H5P_DEFAULT);
        if (ds < 0) { set_error("could not open dataset '%s'", path.c_str()); return S3H5_EIO; }
        const int rc = H5Dwrite(ds, mem_type(dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0 ? S3H5_EIO : S3H5_OK;
        if (rc != S3H5_OK) set_error("could not write dataset '%s'", path.c_str());
        H5Dclose(ds);
        return rc;
    }

    // dataset with its space allocated now and no fill values written: returns its address in the file through *offset, or
    // S3H5_EIO with *offset = -1 when the library gives none (the caller then writes through H5Dwrite).  Holds g_hdf5.
    int create_raw_locked(const std::string &path, int dtype, int ndim, const hsize_t *dims, off_t *offset) {
        *offset = -1;
        const hid_t mt = mem_type(dtype);
        if (mt < 0) { set_error("unknown element type %d", dtype); return S3H5_EINVAL; }
        hid_t lcpl = H5Pcreate(H5P_LINK_CREATE), dcpl = H5Pcreate(H5P_DATASET_CREATE);
        H5Pset_create_intermediate_group(lcpl, 1);
        H5Pset_layout(dcpl, H5D_CONTIGUOUS);
        H5Pset_alloc_time(dcpl, H5D_ALLOC_TIME_EARLY);
        H5Pset_fill_time(dcpl, H5D_FILL_TIME_NEVER);
        hid_t space = H5Screate_simple(ndim, dims, nullptr);
        hid_t ds = H5Dcreate2(fid, path.c_str(), mt, space, lcpl, dcpl, H5P_DEFAULT);
        int rc = S3H5_OK;
        if (ds < 0) {
            set_error("could not create dataset '%s'", path.c_str());
            rc = S3H5_EIO;
        } else {
            const haddr_t a = H5Dget_offset(ds);
            if (a != HADDR_UNDEF) *offset = (off_t)a;
            H5Dclose(ds);
        }
        H5Sclose(space);
        H5Pclose(dcpl);
        H5Pclose(lcpl);
        return rc;
    }

    // the second descriptor of the file (without it every dataset goes through H5Dwrite)
    // The raw path (values written with pwrite() at the datasets' offsets through a second descriptor) assumes the file IS
    // the plain POSIX file HDF5 writes through: only with the sec2 driver, and not when S3_H5_RAW_WRITES=0 asks for
    // H5Dwrite throughout.  (Called with g_hdf5 held.)
    bool raw_ready() {
        if (raw_fd < 0 && !raw_refused) {
            const char *sw = getenv("S3_H5_RAW_WRITES");
            bool sec2 = false;
            const hid_t fapl = H5Fget_access_plist(fid);
            if (fapl >= 0) {
                sec2 = H5Pget_driver(fapl) == H5FD_SEC2;
                H5Pclose(fapl);
            }
            if (sec2 && !(sw && sw[0] == '0')) raw_fd = ::open(os_path.c_str(), O_WRONLY);
            raw_refused = raw_fd < 0;
        }
        return raw_fd >= 0;
    }

    // raw values of several datasets straight into the file, a few threads, 4-MiB pieces; 0 or an errno
    int write_segments(const std::vector<Segment> &segs) {
        if (segs.empty()) return 0;
        constexpr size_t PIECE = (size_t)4 << 20;
        struct Piece { off_t off; const char *p; size_t n; };
        std::vector<Piece> pieces;
        size_t total = 0;
        for (const Segment &g : segs) {
            for (size_t o = 0; o < g.bytes; o += PIECE) pieces.push_back({g.offset + (off_t)o, g.data + o, std::min(PIECE, g.bytes - o)});
            total += g.bytes;
        }
        // The space of the batch's datasets is handed to the file system in ONE request before the values are written: a buffered
        // pwrite that also has to allocate its blocks took 8.9-9.6 ms per 92-MB batch on the boxes' file system, into a preallocated
        // range 7.7-8.2 (tools/sink_probe.hip; S3H5_FALLOCATE=0 switches it off; a file system that refuses is simply written to).
        static const bool prealloc = [] { const char *e = getenv("S3H5_FALLOCATE"); return !(e && e[0] == '0'); }();
        if (prealloc) {
            off_t lo = segs.front().offset, hi = lo;
            for (const Segment &g : segs) {
                lo = std::min(lo, g.offset);
                hi = std::max(hi, g.offset + (off_t)g.bytes);
            }
            if ((size_t)(hi - lo) <= total + total / 8 + ((size_t)1 << 20)) (void)::posix_fallocate(raw_fd, lo, hi - lo);   // (contiguous batch: the usual case)
            else for (const Segment &g : segs) (void)::posix_fallocate(raw_fd, g.offset, (off_t)g.bytes);
        }
        std::atomic<size_t> next{0};
        std::atomic<int> err{0};
        auto work = [&] {
            while (err.load() == 0) {
                const size_t i = next.fetch_add(1);
                if (i >= pieces.size()) break;
                const Piece &pc = pieces[i];
                size_t done = 0;
                while (done < pc.n) {
                    const ssize_t w = ::pwrite(raw_fd, pc.p + done, pc.n - done, pc.off + (off_t)done);
                    if (w < 0) {
                        if (errno == EINTR) continue;
                        err = errno ? errno : EIO;
                        return;
                    }
                    done += (size_t)w;
                }
            }
        };
        // buffered writes to ONE inode are serialised by its lock: more writer threads do not write faster (92 MB: 1 / 4 / 8 threads
        // 8.9 / 9.1 / 9.6 ms, tools/sink_probe.hip) and take cores from the upload's packing threads.  S3H5_WRITE_THREADS overrides.
        static const int want_thr = [] { const char *e = getenv("S3H5_WRITE_THREADS"); return e ? std::max(1, atoi(e)) : 2; }();
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const int n_thr = (int)std::min<size_t>({(size_t)want_thr, (size_t)std::max(1u, hw / 2), std::max<size_t>(1, total / ((size_t)8 << 20))});
        std::vector<std::thread> threads;
        for (int t = 1; t < n_thr; ++t) threads.emplace_back(work);
        work();
        for (auto &t : threads) t.join();
        return err.load();
    }

    // an intermediate path component that exists as a link makes H5Lexists on the full path safe only step by step
    bool exists_locked(const std::string &path) {
        size_t pos = 0;
        while (true) {
            pos = path.find('/', pos + 1);
            const std::string part = path.substr(0, pos);
            if (!part.empty() && H5Lexists(fid, part.c_str(), H5P_DEFAULT) <= 0) return false;
            if (pos == std::string::npos) return true;
        }
    }

    void loop() {
        {
            std::lock_guard<std::mutex> h(g_hdf5);
            H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);      // the error stack (and its printing) is per thread
        }
        while (true) {
            std::vector<Job> batch;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [this] { return stop || !queue.empty(); });
                if (queue.empty()) return;
                // everything queued so far (one snapshot batch, normally) is written together
                while (!queue.empty()) {
                    batch.push_back(std::move(queue.front()));
                    queue.pop_front();
                }
                busy_batch.clear();
                for (const Job &j : batch) busy_batch.emplace_back(static_cast<const char *>(j.data), j.bytes);
                busy_data = batch.front().data;
                busy_bytes = batch.front().bytes;
            }
            int rc = S3H5_OK, n_skipped = 0;
            std::string msg;
            // the batch's values may still be coming down from the device: wait for the producer's word (a poll every 50 us;
            // S3H5_READY_TIMEOUT_S, default 600, bounds it -- a device fault upstream must not hang the writer for ever)
            for (const Job &j : batch) {
                if (!j.ready) continue;
                static const double limit_s = [] { const char *e = getenv("S3H5_READY_TIMEOUT_S"); return e ? atof(e) : 600.0; }();
                const auto t0 = std::chrono::steady_clock::now();
                while (*j.ready < j.ready_value) {
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
                        rc = S3H5_EIO;
                        msg = "the values of '" + j.path + "' never arrived in the host buffer (device-to-host copy not completed)";
                        break;
                    }
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                if (rc != S3H5_OK) break;
            }
            if (rc != S3H5_OK) {
                std::lock_guard<std::mutex> lk(m);
                if (async_error == 0) { async_error = rc; async_message = msg; }
                busy_data = nullptr;
                busy_bytes = 0;
                busy_batch.clear();
                cv_idle.notify_all();
                continue;
            }
            std::vector<Segment> segs;
            std::vector<const Job *> raw_jobs;                // datasets that exist in the file but hold no values yet
            {
                std::lock_guard<std::mutex> h(g_hdf5);
                for (const Job &j : batch) {
                    int r;
                    if (exists_locked(j.path)) {
                        r = S3H5_EEXIST;
                    } else if (j.ndim >= 1 && j.bytes >= ((size_t)1 << 20) && raw_ready()) {
                        off_t off = -1;
                        r = create_raw_locked(j.path, j.dtype, j.ndim, j.dims, &off);
                        if (r == S3H5_OK && off >= 0) {
                            segs.push_back({off, static_cast<const char *>(j.data), j.bytes});
                            raw_jobs.push_back(&j);
                        } else if (r == S3H5_OK) r = write_existing_locked(j.path, j.dtype, j.data);
                    } else {
                        r = write_locked(j.path, j.dtype, j.ndim, j.dims, j.data);
                    }
                    if (r == S3H5_EEXIST) ++n_skipped;
                    else if (r != S3H5_OK && rc == S3H5_OK) { rc = r; msg = g_err; }
                }
                if (!segs.empty()) H5Fflush(fid, H5F_SCOPE_LOCAL);      // the datasets' space exists in the file before it is written to
            }
            int raw_errno = 0;
            if (rc == S3H5_OK) raw_errno = write_segments(segs);
            if ((rc != S3H5_OK || raw_errno != 0) && !raw_jobs.empty()) {
                // A dataset created for the raw path (space allocated, no fill values) whose values did not arrive -- another
                // dataset of the batch failed, or pwrite did (a full disk) -- must not stay in the file as a hole that a re-run with
                // append_existing would take for data: its values go through H5Dwrite now, and if that fails too the dataset
                // is unlinked.
                std::lock_guard<std::mutex> h(g_hdf5);
                for (const Job *j : raw_jobs) {
                    if (write_existing_locked(j->path, j->dtype, j->data) != S3H5_OK) {
                        H5Ldelete(fid, j->path.c_str(), H5P_DEFAULT);
                        if (rc == S3H5_OK) {
                            rc = S3H5_EIO;
                            msg = std::string("raw write failed (") + (raw_errno ? strerror(raw_errno) : "batch aborted") +
                                  ") and so did H5Dwrite: " + g_err + "; the dataset was removed";
                        }
                    }
                }
                raw_refused = true;                           // no further raw writes to this file
                if (raw_fd >= 0) { ::close(raw_fd); raw_fd = -1; }
            }
            {
                std::lock_guard<std::mutex> lk(m);
                skipped += n_skipped;
                if (rc != S3H5_OK && async_error == 0) { async_error = rc; async_message = msg; }
                busy_data = nullptr;
                busy_bytes = 0;
                busy_batch.clear();
            }
            cv_idle.notify_all();
        }
    }
};

extern "C" {

const char *s3h5_last_error(void) { return g_err; }

int s3h5_version(unsigned *major, unsigned *minor, unsigned *release) {
    std::lock_guard<std::mutex> h(g_hdf5);
    return H5get_libversion(major, minor, release) < 0 ? S3H5_EIO : S3H5_OK;
}

int s3h5_open(const char *path, const char *mode, s3h5_file **out) {
    if (!path || !mode || !out) { set_error("s3h5_open: null argument"); return S3H5_EINVAL; }
    *out = nullptr;
    std::lock_guard<std::mutex> h(g_hdf5);
    H5open();
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);          // errors are reported through return codes, not stderr
    hid_t fid = -1;
    bool writable = true;
    if (!strcmp(mode, "w")) {
        fid = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    } else if (!strcmp(mode, "a") || !strcmp(mode, "r+")) {
        fid = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
        if (fid < 0 && !strcmp(mode, "a")) fid = H5Fcreate(path, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT);
    } else if (!strcmp(mode, "r")) {
        fid = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
        writable = false;
    } else {
        set_error("s3h5_open: unknown mode '%s'", mode);
        return S3H5_EINVAL;
    }
    if (fid < 0) { set_error("could not open '%s' (mode %s)", path, mode); return S3H5_ENOENT; }
    s3h5_file *f = new s3h5_file();
    f->fid = fid;
    f->writable = writable;
    f->os_path = path;
    *out = f;
    return S3H5_OK;
}

int s3h5_flush(s3h5_file *f, int64_t *n_skipped) {
    if (!f) { set_error("s3h5_flush: null file"); return S3H5_EINVAL; }
    std::unique_lock<std::mutex> lk(f->m);
    f->cv_idle.wait(lk, [f] { return f->queue.empty() && f->busy_data == nullptr; });
    if (n_skipped) *n_skipped = f->skipped;
    f->skipped = 0;
    if (f->async_error) {
        const int rc = f->async_error;
        set_error("%s", f->async_message.c_str());
        f->async_error = 0;
        return rc;
    }
    return S3H5_OK;
}

int s3h5_wait_buffer(s3h5_file *f, const void *h_base, size_t bytes) {
    if (!f) { set_error("s3h5_wait_buffer: null file"); return S3H5_EINVAL; }
    const char *lo = static_cast<const char *>(h_base), *hi = lo + bytes;
    auto overlaps = [&](const void *p, size_t n) {
        const char *a = static_cast<const char *>(p);
        return p != nullptr && a < hi && a + n > lo;
    };
    std::unique_lock<std::mutex> lk(f->m);
    f->cv_idle.wait(lk, [&] {
        if (overlaps(f->busy_data, f->busy_bytes)) return false;
        for (const auto &b : f->busy_batch)
            if (overlaps(b.first, b.second)) return false;
        for (const Job &j : f->queue)
            if (overlaps(j.data, j.bytes)) return false;
        return true;
    });
    return S3H5_OK;
}

int s3h5_close(s3h5_file *f) {
    if (!f) return S3H5_OK;
    int rc = s3h5_flush(f, nullptr);
    {
        std::lock_guard<std::mutex> lk(f->m);
        f->stop = true;
    }
    f->cv_work.notify_all();
    if (f->worker.joinable()) f->worker.join();
    if (f->raw_fd >= 0) ::close(f->raw_fd);
    {
        std::lock_guard<std::mutex> h(g_hdf5);
        if (f->fid >= 0 && H5Fclose(f->fid) < 0 && rc == S3H5_OK) { set_error("H5Fclose failed"); rc = S3H5_EIO; }
    }
    delete f;
    return rc;
}

int s3h5_write(s3h5_file *f, const char *path, int dtype, int ndim, const int64_t *dims, const void *data) {
    if (!f || !path || ndim < 0 || ndim > S3H5_MAX_DIMS || (ndim > 0 && !dims)) { set_error("s3h5_write: bad arguments"); return S3H5_EINVAL; }
    if (!f->writable) { set_error("s3h5_write: file opened read-only"); return S3H5_EINVAL; }
    int rc = s3h5_flush(f, nullptr);                      // keep the order of writes
    if (rc != S3H5_OK) return rc;
    hsize_t hd[S3H5_MAX_DIMS] = {0};
    for (int i = 0; i < ndim; ++i) {
        if (dims[i] < 0) { set_error("s3h5_write: negative dimension"); return S3H5_EINVAL; }
        hd[i] = (hsize_t)dims[i];
    }
    std::lock_guard<std::mutex> h(g_hdf5);
    if (f->exists_locked(path)) { set_error("dataset '%s' exists already", path); return S3H5_EEXIST; }
    return f->write_locked(path, dtype, ndim, hd, data);
}

int s3h5_write_snapshots_async(s3h5_file *f, const char *group, const char *const *times, int64_t n_snapshots, const char *name,
                               int dtype, int ndim, const int64_t *dims, const void *h_base, int64_t stride_bytes) {
    return s3h5_write_snapshots_async_when(f, group, times, n_snapshots, name, dtype, ndim, dims, h_base, stride_bytes, nullptr, 0);
}

int s3h5_write_snapshots_async_when(s3h5_file *f, const char *group, const char *const *times, int64_t n_snapshots, const char *name,
                                    int dtype, int ndim, const int64_t *dims, const void *h_base, int64_t stride_bytes,
                                    const int32_t *h_ready, int32_t ready_value) {
    if (!f || !group || !times || !name || n_snapshots < 0 || ndim < 0 || ndim > S3H5_MAX_DIMS || (ndim > 0 && !dims) || !h_base ||
        mem_type(dtype) < 0) {
        set_error("s3h5_write_snapshots_async: bad arguments");
        return S3H5_EINVAL;
    }
    if (!f->writable) { set_error("s3h5_write_snapshots_async: file opened read-only"); return S3H5_EINVAL; }
    static const size_t item[] = {4, 8, 4, 8, 1};
    size_t bytes = item[dtype];
    Job proto{};
    proto.dtype = dtype;
    proto.ndim = ndim;
    for (int i = 0; i < ndim; ++i) {
        if (dims[i] < 0) { set_error("s3h5_write_snapshots_async: negative dimension"); return S3H5_EINVAL; }
        proto.dims[i] = (hsize_t)dims[i];
        bytes *= (size_t)dims[i];
    }
    proto.bytes = bytes;
    proto.ready = h_ready;
    proto.ready_value = ready_value;
    if (stride_bytes < (int64_t)bytes) { set_error("s3h5_write_snapshots_async: snapshots overlap"); return S3H5_EINVAL; }
    {
        std::lock_guard<std::mutex> lk(f->m);
        for (int64_t i = 0; i < n_snapshots; ++i) {
            Job j = proto;
            j.path = std::string(group) + "/" + times[i] + "/" + name;
            j.data = static_cast<const char *>(h_base) + i * stride_bytes;
            f->queue.push_back(std::move(j));
        }
        if (!f->worker.joinable()) f->worker = std::thread([f] { f->loop(); });
    }
    f->cv_work.notify_one();
    return S3H5_OK;
}

int s3h5_exists(s3h5_file *f, const char *path) {
    if (!f || !path) { set_error("s3h5_exists: null argument"); return S3H5_EINVAL; }
    int rc = s3h5_flush(f, nullptr);
    if (rc != S3H5_OK) return rc;
    std::lock_guard<std::mutex> h(g_hdf5);
    return f->exists_locked(path) ? 1 : 0;
}

int s3h5_shape(s3h5_file *f, const char *path, int *dtype, int *ndim, int64_t *dims) {
    if (!f || !path || !ndim || !dims) { set_error("s3h5_shape: null argument"); return S3H5_EINVAL; }
    int rc = s3h5_flush(f, nullptr);
    if (rc != S3H5_OK) return rc;
    std::lock_guard<std::mutex> h(g_hdf5);
    if (!f->exists_locked(path)) { set_error("no dataset '%s'", path); return S3H5_ENOENT; }
    hid_t ds = H5Dopen2(f->fid, path, H5P_DEFAULT);
    if (ds < 0) { set_error("'%s' is not a dataset", path); return S3H5_ENOENT; }
    hid_t space = H5Dget_space(ds);
    const int nd = H5Sget_simple_extent_ndims(space);
    hsize_t hd[S3H5_MAX_DIMS] = {0};
    if (nd < 0 || nd > S3H5_MAX_DIMS) { H5Sclose(space); H5Dclose(ds); set_error("'%s': unsupported rank", path); return S3H5_EIO; }
    if (nd > 0) H5Sget_simple_extent_dims(space, hd, nullptr);
    *ndim = nd;
    for (int i = 0; i < nd; ++i) dims[i] = (int64_t)hd[i];
    if (dtype) {
        hid_t t = H5Dget_type(ds);
        const H5T_class_t cls = H5Tget_class(t);
        const size_t sz = H5Tget_size(t);
        *dtype = -1;
        if (cls == H5T_FLOAT) *dtype = sz == 4 ? S3H5_F32 : S3H5_F64;
        else if (cls == H5T_INTEGER) *dtype = sz == 1 ? S3H5_U8 : (sz <= 4 ? S3H5_I32 : S3H5_I64);
        H5Tclose(t);
    }
    H5Sclose(space);
    H5Dclose(ds);
    return S3H5_OK;
}

int s3h5_read(s3h5_file *f, const char *path, int dtype, void *out, int64_t n_elements) {
    if (!f || !path || (!out && n_elements > 0)) { set_error("s3h5_read: null argument"); return S3H5_EINVAL; }
    const hid_t mt = mem_type(dtype);
    if (mt < 0) { set_error("s3h5_read: unknown element type %d", dtype); return S3H5_EINVAL; }
    int rc = s3h5_flush(f, nullptr);
    if (rc != S3H5_OK) return rc;
    std::lock_guard<std::mutex> h(g_hdf5);
    if (!f->exists_locked(path)) { set_error("no dataset '%s'", path); return S3H5_ENOENT; }
    hid_t ds = H5Dopen2(f->fid, path, H5P_DEFAULT);
    if (ds < 0) { set_error("'%s' is not a dataset", path); return S3H5_ENOENT; }
    hid_t space = H5Dget_space(ds);
    const hssize_t n = H5Sget_simple_extent_npoints(space);
    H5Sclose(space);
    if (n != n_elements) {
        H5Dclose(ds);
        set_error("'%s' holds %lld elements, the buffer %lld", path, (long long)n, (long long)n_elements);
        return S3H5_EINVAL;
    }
    rc = S3H5_OK;
    if (n > 0 && H5Dread(ds, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out) < 0) { set_error("could not read '%s'", path); rc = S3H5_EIO; }
    H5Dclose(ds);
    return rc;
}

static herr_t collect_names(hid_t, const char *name, const H5L_info_t *, void *op) {
    static_cast<std::vector<std::string> *>(op)->push_back(name);
    return 0;
}

int s3h5_list(s3h5_file *f, const char *group, char *buf, size_t buf_bytes, size_t *needed, int64_t *n_members) {
    if (!f || !group) { set_error("s3h5_list: null argument"); return S3H5_EINVAL; }
    int rc = s3h5_flush(f, nullptr);
    if (rc != S3H5_OK) return rc;
    std::lock_guard<std::mutex> h(g_hdf5);
    const bool root = !strcmp(group, "/") || !strcmp(group, "");
    if (!root && !f->exists_locked(group)) { set_error("no group '%s'", group); return S3H5_ENOENT; }
    hid_t g = H5Gopen2(f->fid, root ? "/" : group, H5P_DEFAULT);
    if (g < 0) { set_error("'%s' is not a group", group); return S3H5_ENOENT; }
    std::vector<std::string> names;
    hsize_t idx = 0;
    const herr_t e = H5Literate(g, H5_INDEX_NAME, H5_ITER_INC, &idx, collect_names, &names);
    H5Gclose(g);
    if (e < 0) { set_error("could not list '%s'", group); return S3H5_EIO; }
    size_t need = 1;
    for (const std::string &s : names) need += s.size() + 1;
    if (needed) *needed = need;
    if (n_members) *n_members = (int64_t)names.size();
    if (buf && buf_bytes >= need) {
        char *p = buf;
        for (const std::string &s : names) {
            memcpy(p, s.data(), s.size());
            p += s.size();
            *p++ = '\n';
        }
        *p = 0;
    }
    return S3H5_OK;
}

}  // extern "C"
