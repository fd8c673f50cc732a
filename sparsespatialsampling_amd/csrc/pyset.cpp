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
// Pinned against the interpreter's own set objects slot by slot (tests/test_pyset.py reads PySetObject through
// ctypes), on random add / discard / update / difference traces.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

constexpr int64_t EMPTY = -1, DUMMY = -2;
constexpr int LINEAR_PROBES = 9, PERTURB_SHIFT = 5;
constexpr int64_t MINSIZE = 8;

struct PySet {
    int64_t *table = nullptr;
    int64_t mask = 0, fill = 0, used = 0;

    static int64_t *alloc(int64_t size) {
        int64_t *t = static_cast<int64_t *>(std::malloc(sizeof(int64_t) * (size_t)size));
        if (!t) throw std::bad_alloc();
        for (int64_t i = 0; i < size; ++i) t[i] = EMPTY;
        return t;
    }

    PySet() : table(alloc(MINSIZE)), mask(MINSIZE - 1) {}
    ~PySet() { std::free(table); }
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

}  // namespace

extern "C" {

void *s3set_create() try { return new PySet(); } catch (...) { return nullptr; }
void s3set_destroy(void *h) { delete static_cast<PySet *>(h); }
int64_t s3set_len(void *h) { return static_cast<PySet *>(h)->used; }
int64_t s3set_mask(void *h) { return static_cast<PySet *>(h)->mask; }
int64_t s3set_fill(void *h) { return static_cast<PySet *>(h)->fill; }
const int64_t *s3set_table(void *h) { return static_cast<PySet *>(h)->table; }      // -1 unused, -2 dummy
int s3set_contains(void *h, int64_t key) { return key >= 0 && static_cast<PySet *>(h)->contains(key); }

// every entry point that can allocate returns 0 / -1 (out of memory) / -2 (negative key)
int s3set_add(void *h, int64_t key) try {
    if (key < 0) return -2;
    static_cast<PySet *>(h)->add(key);
    return 0;
} catch (...) { return -1; }

void s3set_discard(void *h, int64_t key) { if (key >= 0) static_cast<PySet *>(h)->discard(key); }

// s.update(list): one insertion per element, in order
int s3set_update_ids(void *h, const int64_t *ids, int64_t n) try {
    PySet *s = static_cast<PySet *>(h);
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
    PySet *s = static_cast<PySet *>(h);
    for (int64_t i = a; i < b; ++i) s->add(i);
    return 0;
} catch (...) { return -1; }

// s |= t  /  s.update(t) with t a set
int s3set_update_set(void *h, void *other) try {
    static_cast<PySet *>(h)->merge(*static_cast<PySet *>(other));
    return 0;
} catch (...) { return -1; }

// s -= t
int s3set_difference_update(void *h, void *other) try {
    static_cast<PySet *>(h)->difference_update(*static_cast<PySet *>(other));
    return 0;
} catch (...) { return -1; }

// iteration order -> out[0..len)
void s3set_to_array(void *h, int64_t *out) {
    PySet *s = static_cast<PySet *>(h);
    int64_t n = 0;
    for (int64_t i = 0; i <= s->mask; ++i)
        if (s->table[i] >= 0) out[n++] = s->table[i];
}

// {i for i in ids[flags] if i}: insertion of the flagged, non-zero ids in order (s_cube.py:709)
int s3set_update_flagged(void *h, const int64_t *ids, const uint8_t *flags, int64_t n) try {
    PySet *s = static_cast<PySet *>(h);
    for (int64_t i = 0; i < n; ++i)
        if (flags[i] && ids[i] > 0) s->add(ids[i]);
    return 0;
} catch (...) { return -1; }

}  // extern "C"
