// CPython's `set` of small non-negative ints, restated natively (plain C++, part of libs3topo.so).
//
// Why: the reference keeps its cell bookkeeping in Python sets (`_leaf_cells`, `to_refine`, `all_parents`, ...;
// s_cube.py:531-555, 601-621, 865-902) and the ids of new cells follow the ITERATION ORDER of those sets (SURVEY.md
// section 7, hard part 1).  Round 1 reproduced that order by using real CPython sets on the host; a 10^7-entry Python
// set is the memory / time ceiling the reference's README describes.  This file reproduces the same table states with
// 8-byte slots and bulk operations, so the order is identical and the interpreter only sees whole-array calls.
//
// Restated behaviour (CPython 3.8 - 3.12, Objects/setobject.c; hash(i) == i for 0 <= i < 2^61 - 1):
//   * open addressing, table size a power of two >= 8, slot i = hash & mask, then LINEAR_PROBES = 9 consecutive slots
//     (only when i + 9 <= mask), then i = (i * 5 + 1 + perturb) & mask with perturb >>= 5            (set_add_entry)
//   * deletion leaves a dummy; an insertion that passes dummies before it reaches an unused slot takes the LAST dummy
//     it saw (fill unchanged); otherwise the unused slot (fill + 1) and, when fill * 5 >= mask * 3, the table is
//     rebuilt with room for used * 4 (used * 2 above 50 000 entries)                                   (set_add_entry)
//   * rebuild: smallest power of two > the request (>= 8), entries re-inserted in slot order without comparisons
//                                                                                  (set_table_resize, set_insert_clean)
//   * s |= t: one rebuild up front when (fill + len(t)) * 5 >= mask * 3 (room for (used + len(t)) * 2), then slot-wise
//     copy (empty s, same mask, no dummies in t), clean insertion (empty s) or normal insertion, in t's slot order
//                                                                                                            (set_merge)
//   * s -= t: discard t's entries in t's slot order, then rebuild when more than mask / 4 slots are dummies
//                                                                                    (set_difference_update_internal)
//   * iteration = slot order.
// Deferred updates (s3set_*_async): the tree's leaf set takes a batch of 10^4 .. 10^6 new ids per iteration into a table of
// 10^5 .. 10^7 entries -- cache misses and the occasional rebuild, 0.3 - 5 ms -- and nothing the refine loop does next (the
// geometry kernels, the download of their flags) needs the result.  A set can therefore carry a worker thread that applies
// queued bulk updates in the order they were issued; every other entry point waits for the queue to drain first, so the table
// states are the same as without it.
// Pinned against the interpreter's own set objects slot by slot (tests/test_pyset.py reads PySetObject through
// ctypes), on random add / discard / update / difference traces.
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr int64_t EMPTY = -1, DUMMY = -2;
constexpr int LINEAR_PROBES = 9, PERTURB_SHIFT = 5;
constexpr int64_t MINSIZE = 8;

struct PySet;

struct Job {
    int kind;                      // 0: |= set(range(a, b)); 1: -= set(ids)
    int64_t a, b;
    std::vector<int64_t> ids;
};

struct Worker {
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<Job> jobs;
    int64_t pending = 0;           // queued + running
    int error = 0;                 // first failure of a deferred update (-1: out of memory)
    bool stop = false;
};

struct PySet {
    int64_t *table = nullptr;
    int64_t mask = 0, fill = 0, used = 0;
    Worker *worker = nullptr;

    void wait() {
        if (!worker) return;
        std::unique_lock<std::mutex> lock(worker->mu);
        worker->cv_done.wait(lock, [&] { return worker->pending == 0; });
    }
    void submit(Job &&job);
    void run(const Job &job);
    void end_worker() {
        if (!worker) return;
        {
            std::unique_lock<std::mutex> lock(worker->mu);
            worker->stop = true;
        }
        worker->cv_job.notify_all();
        worker->thread.join();
        delete worker;
        worker = nullptr;
    }

    static int64_t *alloc(int64_t size) {
        int64_t *t = static_cast<int64_t *>(std::malloc(sizeof(int64_t) * (size_t)size));
        if (!t) throw std::bad_alloc();
        for (int64_t i = 0; i < size; ++i) t[i] = EMPTY;
        return t;
    }

    PySet() : table(alloc(MINSIZE)), mask(MINSIZE - 1) {}
    ~PySet() {
        end_worker();
        std::free(table);
    }
    PySet(const PySet &) = delete;
    PySet &operator=(const PySet &) = delete;

    static void insert_clean(int64_t *t, int64_t m, int64_t key) {
        uint64_t perturb = (uint64_t)key, i = (uint64_t)key & (uint64_t)m;
        while (true) {
            int64_t *e = &t[i];
            if (*e == EMPTY) { *e = key; return; }
            if (i + LINEAR_PROBES <= (uint64_t)m)
                for (int j = 0; j < LINEAR_PROBES; ++j) {
                    ++e;
                    if (*e == EMPTY) { *e = key; return; }
                }
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & (uint64_t)m;
        }
    }

    void resize(int64_t minused) {
        int64_t newsize = MINSIZE;
        while (newsize <= minused) newsize <<= 1;
        int64_t *nt = alloc(newsize);
        const int64_t nm = newsize - 1;
        for (int64_t i = 0; i <= mask; ++i) {
            if (i + 16 <= mask && table[i + 16] >= 0) __builtin_prefetch(&nt[(uint64_t)table[i + 16] & (uint64_t)nm]);
            if (table[i] >= 0) insert_clean(nt, nm, table[i]);
        }
        std::free(table);
        table = nt;
        mask = nm;
        fill = used;
    }

    int64_t *lookup(int64_t key) const {           // slot holding key, or the unused slot that ends its probe sequence
        uint64_t perturb = (uint64_t)key, i = (uint64_t)key & (uint64_t)mask;
        while (true) {
            int64_t *e = &table[i];
            int probes = (i + LINEAR_PROBES <= (uint64_t)mask) ? LINEAR_PROBES : 0;
            do {
                if (*e == EMPTY || *e == key) return e;
                ++e;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & (uint64_t)mask;
        }
    }

    bool contains(int64_t key) const { return *lookup(key) == key; }

    void add(int64_t key) {
        int64_t *freeslot = nullptr;
        uint64_t perturb = (uint64_t)key, i = (uint64_t)key & (uint64_t)mask;
        int64_t *e;
        while (true) {
            e = &table[i];
            int probes = (i + LINEAR_PROBES <= (uint64_t)mask) ? LINEAR_PROBES : 0;
            do {
                if (*e == EMPTY) goto found_unused_or_dummy;
                if (*e == key) return;
                if (*e == DUMMY) freeslot = e;
                ++e;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & (uint64_t)mask;
        }
    found_unused_or_dummy:
        if (freeslot) {
            ++used;
            *freeslot = key;
            return;
        }
        ++fill;
        ++used;
        *e = key;
        if (fill * 5 < mask * 3) return;
        resize(used > 50000 ? used * 2 : used * 4);
    }

    void discard(int64_t key) {
        int64_t *e = lookup(key);
        if (*e != key) return;
        *e = DUMMY;
        --used;
    }

    void prefetch(int64_t key) const { __builtin_prefetch(&table[(uint64_t)key & (uint64_t)mask]); }

    void merge(const PySet &o) {
        if (&o == this || o.used == 0) return;
        if ((fill + o.used) * 5 >= mask * 3) resize((used + o.used) * 2);
        if (fill == 0 && mask == o.mask && o.fill == o.used) {
            std::memcpy(table, o.table, sizeof(int64_t) * (size_t)(mask + 1));
            fill = o.fill;
            used = o.used;
            return;
        }
        // (the first probe of an element some slots ahead is prefetched: a 10^7-entry table does not fit any cache, and a
        // rebuild in the middle of the loop only makes the prefetches useless, not wrong)
        constexpr int64_t AHEAD = 16;
        if (fill == 0) {
            fill = used = o.used;
            for (int64_t i = 0; i <= o.mask; ++i) {
                if (i + AHEAD <= o.mask && o.table[i + AHEAD] >= 0) prefetch(o.table[i + AHEAD]);
                if (o.table[i] >= 0) insert_clean(table, mask, o.table[i]);
            }
            return;
        }
        for (int64_t i = 0; i <= o.mask; ++i) {
            if (i + AHEAD <= o.mask && o.table[i + AHEAD] >= 0) prefetch(o.table[i + AHEAD]);
            if (o.table[i] >= 0) add(o.table[i]);
        }
    }

    // mask of set(range(a, a + n)) built by n single insertions into an empty set: only the rebuilds are replayed
    static int64_t range_mask(int64_t n) {
        int64_t m = MINSIZE - 1, cnt = 0;
        while (true) {
            const int64_t trigger = (m * 3 + 4) / 5;          // first fill with fill * 5 >= mask * 3
            if (n < trigger) return m;
            cnt = trigger;
            const int64_t minused = cnt > 50000 ? cnt * 2 : cnt * 4;
            int64_t newsize = MINSIZE;
            while (newsize <= minused) newsize <<= 1;
            m = newsize - 1;
        }
    }

    // consecutive ids never collide in a table that holds them all: id i sits in slot i & mask, the iteration order is the
    // part of the range behind the last multiple of the table size, then the part before it
    template <typename F>
    static void for_range_in_slot_order(int64_t a, int64_t b, int64_t omask, F f) {
        const int64_t wrap = (a | omask) + 1;                // first id > a whose slot is 0 (ids a .. wrap-1: slots a&mask ..)
        if (wrap < b) {
            for (int64_t i = wrap; i < b; ++i) f(i);
            for (int64_t i = a; i < wrap; ++i) f(i);
        } else {
            for (int64_t i = a; i < b; ++i) f(i);
        }
    }

    // s.update(range(a, b)) on an EMPTY set: the final table directly
    void fill_range(int64_t a, int64_t b) {
        const int64_t n = b - a, m = range_mask(n);
        if (m != mask) {
            std::free(table);
            table = nullptr;
            table = alloc(m + 1);
            mask = m;
        }
        for (int64_t i = a; i < b; ++i) table[(uint64_t)i & (uint64_t)mask] = i;
        fill = used = n;
    }

    // s |= set(range(a, b)) without building that set (merge() above with the source's table replayed)
    void merge_range(int64_t a, int64_t b) {
        const int64_t n = b - a;
        if (n <= 0) return;
        const int64_t omask = range_mask(n);
        if ((fill + n) * 5 >= mask * 3) resize((used + n) * 2);
        if (fill == 0 && mask == omask) {
            for (int64_t i = a; i < b; ++i) table[(uint64_t)i & (uint64_t)mask] = i;
            fill = used = n;
            return;
        }
        constexpr int64_t AHEAD = 16;
        if (fill == 0) {
            fill = used = n;
            for_range_in_slot_order(a, b, omask, [&](int64_t i) {
                prefetch(i + AHEAD);
                insert_clean(table, mask, i);
            });
            return;
        }
        // consecutive ids go to consecutive slots; where those are unused (new cell ids are larger than anything in the table:
        // nearly always) add() reduces to storing the id, so a run of unused slots is filled directly -- up to the insertion
        // that triggers a rebuild or meets an occupied slot, which goes through add()
        auto add_run = [&](int64_t x, int64_t y) {
            while (x < y) {
                const int64_t room = (mask * 3 + 4) / 5 - fill - 1;          // insertions that stay below the rebuild trigger
                const int64_t slot = (int64_t)((uint64_t)x & (uint64_t)mask);
                int64_t c = y - x;
                if (room < c) c = room;
                if (mask + 1 - slot < c) c = mask + 1 - slot;
                int64_t e = 0;
                if (c > 0) {
                    int64_t *t = table + slot;
                    while (e < c && t[e] == EMPTY) ++e;
                    for (int64_t j = 0; j < e; ++j) t[j] = x + j;
                    fill += e;
                    used += e;
                    x += e;
                }
                if (e < c || c <= 0) {
                    prefetch(x + AHEAD);
                    add(x++);
                }
            }
        };
        const int64_t wrap = (a | omask) + 1;
        if (wrap < b) {
            add_run(wrap, b);
            add_run(a, wrap);
        } else {
            add_run(a, b);
        }
    }

    // s -= set(ids) for distinct ids: discards do not depend on their order, the rebuild rule is checked once at the end
    void difference_update_ids(const int64_t *ids, int64_t n) {
        for (int64_t i = 0; i < n; ++i) {
            if (i + 16 < n && ids[i + 16] >= 0) prefetch(ids[i + 16]);
            if (ids[i] >= 0) discard(ids[i]);
        }
        if (fill - used <= mask / 4) return;
        resize(used > 50000 ? used * 2 : used * 4);
    }

    void difference_update(const PySet &o) {
        if (&o == this) {                              // set_clear_internal
            std::free(table);
            table = alloc(MINSIZE);
            mask = MINSIZE - 1;
            fill = used = 0;
            return;
        }
        for (int64_t i = 0; i <= o.mask; ++i) {
            if (i + 16 <= o.mask && o.table[i + 16] >= 0) prefetch(o.table[i + 16]);
            if (o.table[i] >= 0) discard(o.table[i]);
        }
        if (fill - used <= mask / 4) return;
        resize(used > 50000 ? used * 2 : used * 4);
    }
};

void PySet::run(const Job &job) {
    if (job.kind == 0) merge_range(job.a, job.b);
    else difference_update_ids(job.ids.data(), (int64_t)job.ids.size());
}

void PySet::submit(Job &&job) {
    if (!worker) {
        worker = new Worker();
        worker->thread = std::thread([this] {
            Worker *w = worker;
            std::unique_lock<std::mutex> lock(w->mu);
            while (true) {
                w->cv_job.wait(lock, [&] { return w->stop || !w->jobs.empty(); });
                if (w->jobs.empty()) return;               // (stop is only set once the queue has drained: end_worker after wait)
                Job job = std::move(w->jobs.front());
                w->jobs.pop_front();
                lock.unlock();
                int err = 0;
                try { run(job); } catch (...) { err = -1; }
                lock.lock();
                if (err && !w->error) w->error = err;
                if (--w->pending == 0) w->cv_done.notify_all();
            }
        });
    }
    {
        std::unique_lock<std::mutex> lock(worker->mu);
        worker->jobs.push_back(std::move(job));
        ++worker->pending;
    }
    worker->cv_job.notify_one();
}

inline PySet *settled(void *h) {               // the set behind a handle, its deferred updates applied
    PySet *s = static_cast<PySet *>(h);
    s->wait();
    return s;
}

}  // namespace

extern "C" {

void *s3set_create() try { return new PySet(); } catch (...) { return nullptr; }
void s3set_destroy(void *h) { delete settled(h); }
int64_t s3set_len(void *h) { return settled(h)->used; }
int64_t s3set_mask(void *h) { return settled(h)->mask; }
int64_t s3set_fill(void *h) { return settled(h)->fill; }
const int64_t *s3set_table(void *h) { return settled(h)->table; }      // -1 unused, -2 dummy
int s3set_contains(void *h, int64_t key) { return key >= 0 && settled(h)->contains(key); }

// every entry point that can allocate returns 0 / -1 (out of memory) / -2 (negative key)
int s3set_add(void *h, int64_t key) try {
    if (key < 0) return -2;
    settled(h)->add(key);
    return 0;
} catch (...) { return -1; }

void s3set_discard(void *h, int64_t key) { if (key >= 0) settled(h)->discard(key); }

// s.update(list): one insertion per element, in order
int s3set_update_ids(void *h, const int64_t *ids, int64_t n) try {
    PySet *s = settled(h);
    for (int64_t i = 0; i < n; ++i) {
        if (ids[i] < 0) return -2;
        if (i + 16 < n && ids[i + 16] >= 0) s->prefetch(ids[i + 16]);
        s->add(ids[i]);
    }
    return 0;
} catch (...) { return -1; }

// s.update(range(a, b))
int s3set_update_range(void *h, int64_t a, int64_t b) try {
    if (a < 0) return -2;
    PySet *s = settled(h);
    if (b <= a) return 0;
    if (s->fill == 0 && s->mask == MINSIZE - 1) s->fill_range(a, b);       // a new set: the final table directly
    else for (int64_t i = a; i < b; ++i) s->add(i);
    return 0;
} catch (...) { return -1; }

// s.update(t) with t == set(range(a, b)) freshly built (t itself is never materialised)
int s3set_update_rangeset(void *h, int64_t a, int64_t b) try {
    if (a < 0) return -2;
    settled(h)->merge_range(a, b);
    return 0;
} catch (...) { return -1; }

// mask of set(range(0, n))
int64_t s3set_range_mask(int64_t n) { return PySet::range_mask(n); }

// s -= set(ids), ids distinct
int s3set_difference_update_ids(void *h, const int64_t *ids, int64_t n) try {
    settled(h)->difference_update_ids(ids, n);
    return 0;
} catch (...) { return -1; }

// s |= t  /  s.update(t) with t a set
int s3set_update_set(void *h, void *other) try {
    settled(h)->merge(*settled(other));
    return 0;
} catch (...) { return -1; }

// s -= t
int s3set_difference_update(void *h, void *other) try {
    settled(h)->difference_update(*settled(other));
    return 0;
} catch (...) { return -1; }

// iteration order -> out[0..len)
void s3set_to_array(void *h, int64_t *out) {
    PySet *s = settled(h);
    const int64_t size = s->mask + 1;
    // a 10^7-entry set is a 256 MB table: its slices are counted and then copied by a few threads
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_thr = size >= ((int64_t)1 << 22) ? (int)std::min<unsigned>(8u, std::max(1u, hw / 2)) : 1;
    if (n_thr == 1) {
        int64_t n = 0;
        for (int64_t i = 0; i < size; ++i)
            if (s->table[i] >= 0) out[n++] = s->table[i];
        return;
    }
    try {
        std::vector<int64_t> cnt(n_thr + 1, 0);
        const int64_t per = size / n_thr;
        auto range = [&](int t, int64_t &a, int64_t &b) { a = t * per; b = t == n_thr - 1 ? size : a + per; };
        {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; ++t)
                th.emplace_back([&, t] {
                    int64_t a, b, c = 0;
                    range(t, a, b);
                    for (int64_t i = a; i < b; ++i) c += s->table[i] >= 0;
                    cnt[t + 1] = c;
                });
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < n_thr; ++t) cnt[t + 1] += cnt[t];
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; ++t)
            th.emplace_back([&, t] {
                int64_t a, b, n = cnt[t];
                range(t, a, b);
                for (int64_t i = a; i < b; ++i)
                    if (s->table[i] >= 0) out[n++] = s->table[i];
            });
        for (auto &x : th) x.join();
    } catch (...) {                                         // no threads to be had: one pass on this one
        int64_t n = 0;
        for (int64_t i = 0; i < size; ++i)
            if (s->table[i] >= 0) out[n++] = s->table[i];
    }
}

// {i for i in ids[flags] if i}: insertion of the flagged, non-zero ids in order (s_cube.py:709)
int s3set_update_flagged(void *h, const int64_t *ids, const uint8_t *flags, int64_t n) try {
    PySet *s = settled(h);
    for (int64_t i = 0; i < n; ++i)
        if (flags[i] && ids[i] > 0) s->add(ids[i]);
    return 0;
} catch (...) { return -1; }

// ---- deferred bulk updates (applied by the set's worker thread in the order issued; every other entry point waits) ----------
int s3set_update_rangeset_async(void *h, int64_t a, int64_t b) try {
    if (a < 0) return -2;
    if (b <= a) return 0;
    static_cast<PySet *>(h)->submit(Job{0, a, b, {}});
    return 0;
} catch (...) { return -1; }

int s3set_difference_update_ids_async(void *h, const int64_t *ids, int64_t n) try {
    if (n <= 0) return 0;
    static_cast<PySet *>(h)->submit(Job{1, 0, 0, std::vector<int64_t>(ids, ids + n)});
    return 0;
} catch (...) { return -1; }

// waits for the deferred updates; 0, or the first failure among them (-1: out of memory), reported once
int s3set_wait(void *h) {
    PySet *s = settled(h);
    if (!s->worker) return 0;
    std::unique_lock<std::mutex> lock(s->worker->mu);
    const int e = s->worker->error;
    s->worker->error = 0;
    return e;
}

}  // extern "C"
