// Host-side topology engine of the S^3 sampling tree (plain C++, no GPU): neighbour links, shared-node numbering and
// the final renumbering, with the reference's *sequential* semantics.
//
// Reference behaviour restated here (file:line relative to the reference checkout):
//   Cell                         s_cube.py:32-83      -> structure-of-arrays (level, parent, first_child, nb, node_idx)
//   _assign_neighbors            s_cube.py:904-1186   -> generated from the lattice rule (see build_nb_table)
//   _assign_indices              s_cube.py:1188-1536  -> NODE_RULES_2D / NODE_RULES_3D decision tables
//   check_nb_node                s_cube.py:1739-1755
//   _remove_invalid_cells (nb)   s_cube.py:721-728
//   _check_nb                    s_cube.py:447-464
//   _resort_nodes_and_indices_of_grid + renumber_node_indices_parallel   s_cube.py:734-772, 1695-1736
//
// Which cells are refined, and in which order, is decided by the Python host (real CPython sets give the
// reference's iteration order); this engine receives ordered id arrays.  Neighbour links are deliberately NOT kept
// current: like the reference they are written when children are created and refreshed only where the reference
// refreshes them, because the shared-node numbering depends on that staleness.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "topo_tables.h"

namespace {

using s3topo::INVALID;
using s3topo::LEAF;
using s3topo::NbEntry;
using s3topo::NodeRule;
using s3topo::REF_BASE;

// child / node directions, s_cube.py:188-194
const int DIR2[4][3] = {{-1, -1, 0}, {-1, 1, 0}, {1, 1, 0}, {1, -1, 0}};
const int DIR3[8][3] = {{-1, -1, 1}, {-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, -1}, {-1, 1, -1}, {1, 1, -1}, {1, -1, -1}};

// neighbour slots, s_cube.py:22-26: in-plane order w, nw, n, ne, e, se, s, sw; 0-7 same plane, 8-15 lower plane,
// 16 = directly below, 17-24 upper plane, 25 = directly above
const int PLANE[8][2] = {{-1, 0}, {-1, 1}, {0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}};

void slot_offset(int dim, int slot, int o[3]) {
    o[2] = 0;
    if (slot < 8) { o[0] = PLANE[slot][0]; o[1] = PLANE[slot][1]; return; }
    (void)dim;
    if (slot < 16) { o[0] = PLANE[slot - 8][0]; o[1] = PLANE[slot - 8][1]; o[2] = -1; return; }
    if (slot == 16) { o[0] = 0; o[1] = 0; o[2] = -1; return; }
    if (slot < 25) { o[0] = PLANE[slot - 17][0]; o[1] = PLANE[slot - 17][1]; o[2] = 1; return; }
    o[0] = 0; o[1] = 0; o[2] = 1;
}

int offset_slot(int dim, const int o[3]) {
    int nnb = dim == 2 ? 8 : 26;
    for (int s = 0; s < nnb; ++s) {
        int t[3];
        slot_offset(dim, s, t);
        if (t[0] == o[0] && t[1] == o[1] && (dim == 2 || t[2] == o[2])) return s;
    }
    return -1;
}

int dir_child(int dim, const int d[3]) {
    int nch = 1 << dim;
    for (int c = 0; c < nch; ++c) {
        const int *t = dim == 2 ? DIR2[c] : DIR3[c];
        if (t[0] == d[0] && t[1] == d[1] && (dim == 2 || t[2] == d[2])) return c;
    }
    return -1;
}

// minimal fork-join pool: run(n, grain, fn) calls fn(begin, end) over [0, n) in chunks of `grain` on the pool's threads
// and the caller; returns when every chunk is done
struct Pool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_start, cv_done;
    const std::function<void(int64_t, int64_t)> *fn = nullptr;
    int64_t n = 0, grain = 1;
    std::atomic<int64_t> next{0};
    int working = 0;
    uint64_t epoch = 0;
    bool stop = false;

    explicit Pool(int n_threads) {
        for (int i = 1; i < n_threads; ++i) threads.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_start.notify_all();
        for (auto &t : threads) t.join();
    }
    void drain() {
        while (true) {
            const int64_t b = next.fetch_add(grain);
            if (b >= n) return;
            (*fn)(b, std::min(n, b + grain));
        }
    }
    void loop() {
        uint64_t seen = 0;
        while (true) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv_start.wait(lk, [&] { return stop || epoch != seen; });
                if (stop) return;
                seen = epoch;
            }
            drain();
            {
                std::lock_guard<std::mutex> lk(m);
                --working;
            }
            cv_done.notify_one();
        }
    }
    void run(int64_t n_, int64_t grain_, const std::function<void(int64_t, int64_t)> &f) {
        if (threads.empty() || n_ <= grain_) {
            if (n_ > 0) f(0, n_);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m);
            fn = &f;
            n = n_;
            grain = grain_;
            next.store(0);
            working = (int)threads.size();
            ++epoch;
        }
        cv_start.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return working == 0; });
    }
};

// transient encodings of a node id while a batch is assembled in parallel (final ids are >= 0)
inline int64_t enc_new(int local) { return -(int64_t)(1 + local); }                      // l-th new node of this parent
inline bool is_new(int64_t v) { return v < 0 && v > -REF_BASE; }
inline int dec_new(int64_t v) { return (int)(-v - 1); }
inline int64_t enc_ref(int64_t entry) { return -(REF_BASE + entry); }                     // entry = cell * nch + node
inline bool is_ref(int64_t v) { return v <= -REF_BASE; }
inline int64_t dec_ref(int64_t v) { return -v - REF_BASE; }

// growable table without value initialisation: growth is a realloc (for tables of this size an address-space remap, no
// copy) and the new part is first touched by whichever thread fills it
template <typename T>
struct Buf {
    T *p = nullptr;
    size_t cap = 0;
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    ~Buf() { std::free(p); }
    void grow(size_t want) {
        if (want <= cap) return;
        const size_t nc = std::max<size_t>(std::max(want, cap * 2), 1024);
        T *q = static_cast<T *>(std::realloc(p, nc * sizeof(T)));
        if (!q) throw std::bad_alloc();
        p = q;
        cap = nc;
    }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *data() { return p; }
    const T *data() const { return p; }
};

const std::vector<NodeRule> NODE_RULES_2D[4] = S3_NODE_RULES_2D_INIT;
const std::vector<NodeRule> NODE_RULES_3D[8] = S3_NODE_RULES_3D_INIT;

struct Topo {
    int dim, nch, nnb;
    double width;
    Buf<int32_t> level, parent, first_child;
    Buf<int32_t> nb;                  // [n_cells][nnb]
    Buf<int64_t> node_idx;            // [n_cells][nch]
    Buf<double> center;               // [n_cells][dim]
    Buf<double> nodes;                // [n_nodes][dim]
    int64_t n_nodes_used = 0;
    size_t cell_cap = 0;              // cells the cell tables have room for
    std::vector<NbEntry> nb_table;    // [nch][nnb]
    // finalize() results
    std::vector<int64_t> face_ids;    // after finalize(): old node id -> new node id (-1 = dropped)
    int64_t n_leaf = 0, n_unique = 0;
    std::vector<int64_t> chunk_first_row;   // after finalize(): output row of the first leaf of every 2^16-cell chunk

    int64_t n_used = 0;               // cells created so far; the cell tables are sized to their capacity
    double half_width[64], quarter_width[64];   // (0.5 * width) / 2^level, (0.25 * width) / 2^level

    int64_t n_cells() const { return n_used; }
    int64_t n_nodes() const { return n_nodes_used; }
    const int *dir(int c) const { return dim == 2 ? DIR2[c] : DIR3[c]; }

    // lattice rule behind the reference's hand-written neighbour table (verified against the reference's tables,
    // SURVEY.md 8(a) a10): child direction dc, slot offset o, p = dc + 2o; |p_j| == 3 -> crosses into the parent's
    // neighbour in that direction
    void build_nb_table() {
        nb_table.assign((size_t)nch * nnb, NbEntry{-1, -1});
        for (int c = 0; c < nch; ++c)
            for (int s = 0; s < nnb; ++s) {
                int o[3], p[3] = {0, 0, 0}, big[3] = {0, 0, 0}, t[3] = {0, 0, 0};
                slot_offset(dim, s, o);
                bool crosses = false;
                for (int j = 0; j < dim; ++j) {
                    p[j] = dir(c)[j] + 2 * o[j];
                    big[j] = p[j] == 3 ? 1 : (p[j] == -3 ? -1 : 0);
                    t[j] = p[j] - 4 * big[j];
                    crosses |= big[j] != 0;
                }
                NbEntry e;
                e.pslot = crosses ? (int8_t)offset_slot(dim, big) : (int8_t)-1;
                e.target = (int8_t)dir_child(dim, t);
                nb_table[(size_t)c * nnb + s] = e;
            }
    }

    void push_cell(int32_t lvl, int32_t par, const double *c) {
        const size_t i = (size_t)n_used++;
        level[i] = lvl;
        parent[i] = par;
        first_child[i] = LEAF;
        for (int s = 0; s < nnb; ++s) nb[i * nnb + s] = -1;
        for (int s = 0; s < nch; ++s) node_idx[i * nch + s] = 0;
        for (int j = 0; j < dim; ++j) center[i * dim + j] = c[j];
    }

    // room for `extra` more cells (and as many new nodes at most: every new cell brings at most 2^d - 1 of them... the
    // node table is grown again where a batch needs more): geometric growth, no initialisation
    void reserve_cells(int64_t extra) {
        const size_t want = (size_t)(n_cells() + extra);
        nodes.grow((size_t)(n_nodes_used + extra * nch) * dim);
        if (want <= cell_cap) return;
        const size_t cap = std::max<size_t>(std::max(want, cell_cap * 2), 1024);
        level.grow(cap);
        parent.grow(cap);
        first_child.grow(cap);
        nb.grow(cap * nnb);
        node_idx.grow(cap * nch);
        center.grow(cap * dim);
        const size_t old = batch_pos.cap;
        batch_pos.grow(cap);
        for (size_t i = old; i < batch_pos.cap; ++i) batch_pos[i] = -1;
        cell_cap = cap;
    }

    // _assign_neighbors(cell, children=existing children)
    void assign_neighbors(int32_t P) {
        const int32_t fc = first_child[P];
        if (fc < 0) return;
        const int32_t *pnb = &nb[(size_t)P * nnb];
        int32_t q_of[26], fc_of[26];                                   // the parent's neighbours and their first child
        for (int s = 0; s < nnb; ++s) {
            q_of[s] = pnb[s];
            fc_of[s] = q_of[s] >= 0 ? first_child[q_of[s]] : -1;
        }
        const NbEntry *tab = nb_table.data();
        for (int c = 0; c < nch; ++c) {
            int32_t *cnb = &nb[(size_t)(fc + c) * nnb];
            for (int s = 0; s < nnb; ++s) {
                const NbEntry e = tab[c * nnb + s];
                if (e.pslot < 0) { cnb[s] = fc + e.target; continue; }
                cnb[s] = fc_of[e.pslot] >= 0 ? fc_of[e.pslot] + e.target : q_of[e.pslot];     // parent_or_child
            }
        }
    }

    bool check_nb_node(int32_t cell, int slot) const {
        int32_t q = nb[(size_t)cell * nnb + slot];
        return q >= 0 && first_child[q] == LEAF && level[q] == level[cell];
    }

    int64_t new_node(int32_t cell, int node) {
        const double off = half_width[level[cell]];
        nodes.grow((size_t)(n_nodes_used + 1) * dim);
        for (int j = 0; j < dim; ++j) nodes[(size_t)n_nodes_used * dim + j] = center[(size_t)cell * dim + j] + dir(node)[j] * off;
        return n_nodes_used++;
    }

    // _assign_indices(children of P)
    void assign_indices(int32_t P) {
        const int32_t fc = first_child[P];
        for (int i = 0; i < nch; ++i) {
            const int32_t cell = fc + i;
            int64_t *ni = &node_idx[(size_t)cell * nch];
            ni[i] = node_idx[(size_t)P * nch + i];
            const std::vector<NodeRule> &rules = dim == 2 ? NODE_RULES_2D[i] : NODE_RULES_3D[i];
            for (const NodeRule &r : rules) {
                if (r.n_cand < 0) {
                    ni[r.node] = node_idx[(size_t)(fc + r.cand[0][0]) * nch + r.cand[0][1]];
                    continue;
                }
                bool found = false;
                for (int a = 0; a < r.n_cand && !found; ++a)
                    if (check_nb_node(cell, r.cand[a][0])) {
                        ni[r.node] = node_idx[(size_t)nb[(size_t)cell * nnb + r.cand[a][0]] * nch + r.cand[a][1]];
                        found = true;
                    }
                if (!found) ni[r.node] = new_node(cell, r.node);
            }
        }
    }

    // ---- a whole batch at once, on several threads, with the result of the sequential procedure ------------------
    // Sequentially, parent i sees the parents before it in the batch as refined and their children as existing leaves,
    // and new node ids are handed out in processing order.  Because the ids of the children are known up front
    // (first + 2^d * position) all of that can be evaluated per parent from the state before the batch plus the
    // position table: pass A builds the children of every parent independently (links; node entries as final id /
    // l-th new node of this parent / reference to an entry of an earlier parent's child), an exclusive scan of the
    // new-node counts gives every parent its id range, pass C turns "l-th new node" into ids and writes the
    // coordinates, pass D follows the references (a lattice point is shared by at most 2^d cells: short chains).
    Buf<int32_t> batch_pos;              // cell -> position in the current batch, -1 otherwise (sized with the cell tables)
    std::vector<int32_t> new_count;      // per parent of the batch: nodes it creates
    std::vector<int64_t> new_base;       // exclusive scan of new_count
    Pool *pool = nullptr;
    int n_threads = 1;
    // seconds spent per phase (s3t_stats): 0 validate, 1 pass A, 2 scan + node table growth, 3 pass C, 4 pass D,
    // 5 finish, 6 relink pass of uniform levels, 7 relink_parent_of, 8 mark_invalid, 9 table growth, 10 sequential batches
    double phase_s[12] = {0};
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    int64_t par_min = 64;                // smaller batches take the sequential procedure (S3_TOPO_PAR_MIN)

    const std::vector<NodeRule> &rules_of(int child) const { return dim == 2 ? NODE_RULES_2D[child] : NODE_RULES_3D[child]; }

    void build_children(int64_t i, const int64_t *parents, int64_t first) {
        const int32_t P = (int32_t)parents[i];
        const int32_t fc = (int32_t)(first + i * nch);
        const int32_t lvl = level[P] + 1;
        const double off = quarter_width[level[P]];
        for (int c = 0; c < nch; ++c) {
            const size_t cell = (size_t)fc + c;
            level[cell] = lvl;
            parent[cell] = P;
            first_child[cell] = LEAF;
            for (int j = 0; j < dim; ++j) center[cell * dim + j] = center[(size_t)P * dim + j] + dir(c)[j] * off;
            for (int m = 0; m < nch; ++m) node_idx[cell * nch + m] = 0;
        }
        // links: a neighbour of the parent counts as refined when it was before the batch or comes earlier in it
        const int32_t *pnb = &nb[(size_t)P * nnb];
        int32_t q_of[26], fc_of[26];
        for (int s = 0; s < nnb; ++s) {
            const int32_t q = pnb[s];
            int32_t f = -1;
            if (q >= 0) {
                f = first_child[q];
                if (f == LEAF && batch_pos[q] >= 0 && batch_pos[q] < i) f = (int32_t)(first + (int64_t)batch_pos[q] * nch);
            }
            q_of[s] = q;
            fc_of[s] = f;
        }
        const NbEntry *tab = nb_table.data();
        for (int c = 0; c < nch; ++c) {
            int32_t *cnb = &nb[(size_t)(fc + c) * nnb];
            for (int s = 0; s < nnb; ++s) {
                const NbEntry e = tab[c * nnb + s];
                if (e.pslot < 0) { cnb[s] = fc + e.target; continue; }
                cnb[s] = fc_of[e.pslot] >= 0 ? fc_of[e.pslot] + e.target : q_of[e.pslot];
            }
        }
        // node entries
        int local = 0;
        for (int k = 0; k < nch; ++k) {
            const int32_t cell = fc + k;
            int64_t *ni = &node_idx[(size_t)cell * nch];
            const int32_t *cnb = &nb[(size_t)cell * nnb];
            ni[k] = node_idx[(size_t)P * nch + k];
            for (const NodeRule &r : rules_of(k)) {
                if (r.n_cand < 0) {
                    ni[r.node] = node_idx[(size_t)(fc + r.cand[0][0]) * nch + r.cand[0][1]];
                    continue;
                }
                bool found = false;
                for (int a = 0; a < r.n_cand && !found; ++a) {
                    const int32_t q = cnb[r.cand[a][0]];
                    if (q < 0) continue;
                    const int64_t entry = (int64_t)q * nch + r.cand[a][1];
                    if (q >= first) {
                        // a cell of this batch: a leaf by construction; its level is its parent's + 1
                        const int64_t j = (q - first) / nch;
                        if (level[parents[j]] + 1 != lvl) continue;
                        ni[r.node] = j == i ? node_idx[entry] : enc_ref(entry);      // own sibling: entry as it stands
                        found = true;
                    } else if (first_child[q] == LEAF && !(batch_pos[q] >= 0 && batch_pos[q] < i) && level[q] == lvl) {
                        ni[r.node] = node_idx[entry];
                        found = true;
                    }
                }
                if (!found) ni[r.node] = enc_new(local++);
            }
        }
        new_count[i] = local;
    }

    // pass C: the parent's new nodes get their ids (in the order pass A met them) and their coordinates
    void number_new_nodes(int64_t i, int64_t first, int64_t nodes_before) {
        const int32_t fc = (int32_t)(first + i * nch);
        const int64_t base = nodes_before + new_base[i];
        int seen = 0;
        for (int k = 0; k < nch; ++k) {
            const int32_t cell = fc + k;
            int64_t *ni = &node_idx[(size_t)cell * nch];
            auto fix = [&](int node) {
                const int64_t v = ni[node];
                if (!is_new(v)) return;
                const int l = dec_new(v);
                if (l == seen) {
                    const double off = half_width[level[cell]];
                    for (int j = 0; j < dim; ++j)
                        nodes[(size_t)(base + l) * dim + j] = center[(size_t)cell * dim + j] + dir(node)[j] * off;
                    ++seen;
                }
                ni[node] = base + l;
            };
            fix(k);
            for (const NodeRule &r : rules_of(k)) fix(r.node);
        }
    }

    // pass D: follow references into earlier parents' children until a final id is met
    void resolve_refs(int64_t i, int64_t first) {
        int64_t *ni = &node_idx[(size_t)(first + i * nch) * nch];
        for (int e = 0; e < nch * nch; ++e) {
            int64_t v = ni[e];
            if (!is_ref(v)) continue;
            while (is_ref(v)) v = __atomic_load_n(&node_idx[(size_t)dec_ref(v)], __ATOMIC_RELAXED);
            __atomic_store_n(&ni[e], v, __ATOMIC_RELAXED);
        }
    }

    void prefetch_parent(int32_t P) const {
        const int32_t *row = &nb[(size_t)P * nnb];
        __builtin_prefetch(row);
        __builtin_prefetch(row + 16);
        __builtin_prefetch(&node_idx[(size_t)P * nch]);
        __builtin_prefetch(&center[(size_t)P * dim]);
        __builtin_prefetch(&level[P]);
    }
    void prefetch_neighbours(int32_t P) const {
        const int32_t *row = &nb[(size_t)P * nnb];
        for (int s = 0; s < nnb; ++s) {
            const int32_t q = row[s];
            if (q < 0) continue;
            __builtin_prefetch(&first_child[q]);
            __builtin_prefetch(&level[q]);
            __builtin_prefetch(&batch_pos[q]);
            __builtin_prefetch(&node_idx[(size_t)q * nch]);
        }
    }
    void prefetch_neighbour_children(int32_t P) const {
        const int32_t *row = &nb[(size_t)P * nnb];
        for (int s = 0; s < nnb; ++s) {
            const int32_t q = row[s];
            if (q < 0) continue;
            const int32_t fc = first_child[q];
            if (fc < 0) continue;
            __builtin_prefetch(&first_child[fc]);
            __builtin_prefetch(&level[fc]);
            __builtin_prefetch(&batch_pos[fc]);
            for (int c = 0; c < nch; ++c) __builtin_prefetch(&node_idx[(size_t)(fc + c) * nch]);
        }
    }

    // returns the id of the first new cell, -1 if a parent is not a leaf (or listed twice)
    int64_t refine_batch_parallel(const int64_t *parents, int64_t n, int relink) {
        const int64_t first = n_cells(), nodes_before = n_nodes();
        double t0 = now(), t1;
        reserve_cells(n * nch);
        t1 = now(); phase_s[9] += t1 - t0; t0 = t1;
        int64_t bad = -1;
        for (int64_t i = 0; i < n; ++i) {
            const int64_t P = parents[i];
            if (P < 0 || P >= first || first_child[P] != LEAF || batch_pos[P] >= 0) { bad = i; break; }
            batch_pos[P] = (int32_t)i;
        }
        if (bad >= 0) {
            for (int64_t i = 0; i < bad; ++i) batch_pos[parents[i]] = -1;
            return -1;
        }
        new_count.assign((size_t)n, 0);
        new_base.assign((size_t)n, 0);
        t1 = now(); phase_s[0] += t1 - t0; t0 = t1;
        const int64_t grain = 128;
        pool->run(n, grain, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) {
                if (i + 4 < e) prefetch_parent((int32_t)parents[i + 4]);
                if (i + 2 < e) prefetch_neighbours((int32_t)parents[i + 2]);
                if (i + 1 < e) prefetch_neighbour_children((int32_t)parents[i + 1]);
                build_children(i, parents, first);
            }
        });
        t1 = now(); phase_s[1] += t1 - t0; t0 = t1;
        int64_t total = 0;
        for (int64_t i = 0; i < n; ++i) {
            new_base[i] = total;
            total += new_count[i];
        }
        nodes.grow((size_t)(nodes_before + total) * dim);
        n_nodes_used = nodes_before + total;
        t1 = now(); phase_s[2] += t1 - t0; t0 = t1;
        pool->run(n, 512, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) number_new_nodes(i, first, nodes_before);
        });
        t1 = now(); phase_s[3] += t1 - t0; t0 = t1;
        pool->run(n, 512, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) resolve_refs(i, first);
        });
        t1 = now(); phase_s[4] += t1 - t0; t0 = t1;
        for (int64_t i = 0; i < n; ++i) {
            first_child[parents[i]] = (int32_t)(first + i * nch);
            batch_pos[parents[i]] = -1;
        }
        n_used = first + n * nch;
        t1 = now(); phase_s[5] += t1 - t0; t0 = t1;
        if (relink)
            pool->run(n, 256, [&](int64_t b, int64_t e) {
                for (int64_t i = b; i < e; ++i) assign_neighbors((int32_t)parents[i]);
            });
        phase_s[6] += now() - t0;
        return first;
    }

    // ---- command queue: with max_delta_level=False nothing the refine loop decides depends on the links or node ids
    // (SURVEY.md section 7.4), so the host submits its batches and carries on; one worker thread applies them in order
    struct Cmd { int kind; int relink; std::vector<int64_t> ids; };     // kind 0 refine, 1 relink parents, 2 mark invalid
    std::deque<Cmd> queue;
    std::mutex mtx;
    std::condition_variable cv_work, cv_idle;
    std::thread worker;
    bool busy = false, stop = false;
    int error = 0;                       // first failure of a queued command: -1 not a leaf, -2 out of memory

    void worker_loop();
    void submit(Cmd &&c) {
        {
            std::lock_guard<std::mutex> lk(mtx);
            if (!worker.joinable()) worker = std::thread([this] { worker_loop(); });
            queue.push_back(std::move(c));
        }
        cv_work.notify_one();
    }
    int wait_idle() {
        std::unique_lock<std::mutex> lk(mtx);
        cv_idle.wait(lk, [this] { return queue.empty() && !busy; });
        return error;
    }
    ~Topo() {
        {
            std::lock_guard<std::mutex> lk(mtx);
            stop = true;
        }
        cv_work.notify_one();
        if (worker.joinable()) worker.join();
        delete pool;
    }

    // one parent of _refine_cells / the uniform loop: children, neighbour links, node ids
    void refine_one(int32_t P) {
        const int32_t fc = (int32_t)n_cells();
        const double off = quarter_width[level[P]];
        for (int c = 0; c < nch; ++c) {
            double x[3];
            for (int j = 0; j < dim; ++j) x[j] = center[(size_t)P * dim + j] + dir(c)[j] * off;
            push_cell(level[P] + 1, P, x);
        }
        first_child[P] = fc;
        assign_neighbors(P);
        assign_indices(P);
    }
};

}  // namespace

extern "C" {

// C++ exceptions (std::bad_alloc of the growing tables) must not cross the C boundary: the entry points that allocate
// report them as an error value instead (nullptr / -2 / -1)
void *s3t_create(int dim, double width, const double *root_center) try {
    if (dim != 2 && dim != 3) return nullptr;
    Topo *t = new Topo();
    t->dim = dim;
    t->nch = 1 << dim;
    t->nnb = dim == 2 ? 8 : 26;
    t->width = width;
    t->build_nb_table();
    for (int l = 0; l < 64; ++l) {
        t->half_width[l] = (0.5 * width) / std::ldexp(1.0, l);
        t->quarter_width[l] = (0.25 * width) / std::ldexp(1.0, l);
    }
    t->reserve_cells(1);
    t->push_cell(0, -1, root_center);
    {
        // worker threads of a batch: S3_TOPO_THREADS, else up to 8 of the cores this process may use
        const char *env = std::getenv("S3_TOPO_THREADS");
        int nt = env ? std::atoi(env) : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
        if (nt < 1) nt = 1;
        if (nt > 64) nt = 64;
        t->n_threads = nt;
        if (nt > 1) t->pool = new Pool(nt);
        if (const char *pm = std::getenv("S3_TOPO_PAR_MIN")) t->par_min = std::max(1, std::atoi(pm));
    }
    // root nodes, s_cube.py:368,386-394: centre + dir * 0.5 * width, ids 0..2^d-1
    for (int c = 0; c < t->nch; ++c) {
        for (int j = 0; j < dim; ++j) t->nodes[(size_t)c * dim + j] = root_center[j] + t->dir(c)[j] * 0.5 * width;
        t->node_idx[c] = c;
    }
    t->n_nodes_used = t->nch;
    return t;
} catch (...) {
    return nullptr;
}

void s3t_destroy(void *h) { delete static_cast<Topo *>(h); }

int64_t s3t_n_cells(void *h) { static_cast<Topo *>(h)->wait_idle(); return static_cast<Topo *>(h)->n_cells(); }
int64_t s3t_n_nodes(void *h) { static_cast<Topo *>(h)->wait_idle(); return static_cast<Topo *>(h)->n_nodes(); }

// raw views for the Python host (valid until the next mutating call)
int32_t *s3t_level(void *h) { return static_cast<Topo *>(h)->level.data(); }
int32_t *s3t_parent(void *h) { return static_cast<Topo *>(h)->parent.data(); }
int32_t *s3t_first_child(void *h) { return static_cast<Topo *>(h)->first_child.data(); }
int32_t *s3t_nb(void *h) { return static_cast<Topo *>(h)->nb.data(); }
int64_t *s3t_node_idx(void *h) { return static_cast<Topo *>(h)->node_idx.data(); }
double *s3t_center(void *h) { return static_cast<Topo *>(h)->center.data(); }
double *s3t_nodes(void *h) { return static_cast<Topo *>(h)->nodes.data(); }

// refine the listed parents in order (s_cube.py:879-895 / 531-544).  relink != 0 additionally re-runs the neighbour
// assignment of every parent of the batch afterwards (the "update all nb" pass of the uniform loop, s_cube.py:547-549).
// Returns the id of the first new cell, -1 if a parent is not a leaf, -2 if the tables could not grow.
static int64_t refine_batch_sequential(Topo *t, const int64_t *parents, int64_t n, int relink);
static int64_t refine_batch(Topo *t, const int64_t *parents, int64_t n, int relink) {
    if (t->pool && n >= t->par_min) return t->refine_batch_parallel(parents, n, relink);
    const double t0 = Topo::now();
    const int64_t r = refine_batch_sequential(t, parents, n, relink);
    t->phase_s[10] += Topo::now() - t0;
    return r;
}

static int64_t refine_batch_sequential(Topo *t, const int64_t *parents, int64_t n, int relink) {
    const int64_t first = t->n_cells();
    t->reserve_cells(n * t->nch);
    // the work per parent is a few dozen dependent look-ups in tables far larger than the caches; two software
    // prefetch stages run ahead of it: the neighbour row of the parent four positions ahead, and -- once that row is
    // there -- the entries of those neighbours two positions ahead
    const int64_t n_before = t->n_cells();
    auto valid = [&](int64_t i) { return i < n && parents[i] >= 0 && parents[i] < n_before; };
    for (int64_t i = 0; i < n; ++i) {
        if (valid(i + 4)) {
            const int32_t *row = &t->nb[(size_t)parents[i + 4] * t->nnb];
            __builtin_prefetch(row);
            __builtin_prefetch(row + 16);
            __builtin_prefetch(&t->node_idx[(size_t)parents[i + 4] * t->nch]);
            __builtin_prefetch(&t->center[(size_t)parents[i + 4] * t->dim]);
        }
        if (valid(i + 2)) {
            const int32_t *row = &t->nb[(size_t)parents[i + 2] * t->nnb];
            for (int s_ = 0; s_ < t->nnb; ++s_) {
                const int32_t q = row[s_];
                if (q < 0) continue;
                __builtin_prefetch(&t->first_child[q]);
                __builtin_prefetch(&t->level[q]);
                __builtin_prefetch(&t->node_idx[(size_t)q * t->nch]);
            }
        }
        if (valid(i + 1)) {
            // third stage: where a neighbour is refined already, its children are what the new cells link to and share
            // nodes with
            const int32_t *row = &t->nb[(size_t)parents[i + 1] * t->nnb];
            for (int s_ = 0; s_ < t->nnb; ++s_) {
                const int32_t q = row[s_];
                if (q < 0) continue;
                const int32_t fc = t->first_child[q];
                if (fc < 0) continue;
                __builtin_prefetch(&t->first_child[fc]);
                __builtin_prefetch(&t->level[fc]);
                for (int c = 0; c < t->nch; ++c) __builtin_prefetch(&t->node_idx[(size_t)(fc + c) * t->nch]);
            }
        }
        if (parents[i] < 0 || parents[i] >= t->n_cells() || t->first_child[parents[i]] != LEAF) return -1;
        t->refine_one((int32_t)parents[i]);
    }
    if (relink)
        for (int64_t i = 0; i < n; ++i) t->assign_neighbors((int32_t)parents[i]);
    return first;
}

int64_t s3t_refine(void *h, const int64_t *parents, int64_t n, int relink) try {
    Topo *t = static_cast<Topo *>(h);
    if (t->wait_idle() != 0) return t->error;
    return refine_batch(t, parents, n, relink);
} catch (...) {
    return -2;
}

// cell.parent.children = _assign_neighbors(cell.parent, children=cell.parent.children)   (s_cube.py:609, 834, 494)
static void relink_parents(Topo *t, const int64_t *cells, int64_t n) {
    if (t->pool && n >= 4 * t->par_min) {
        // The refresh of a parent p reads p's own neighbour row and rewrites the rows of p's children.  Two listed parents
        // interact only when one is the parent of the other (the refresh of g rewrites the row that the refresh of its
        // child p reads): those few are replayed sequentially in list order, occurrence by occurrence; all the others
        // are independent of everything in the list and are refreshed once each, in parallel.
        std::vector<int32_t> par((size_t)n);
        std::vector<uint8_t> dep((size_t)n, 0);
        int32_t *stamp = t->batch_pos.data();                      // -1 outside batches; holds the winning occurrence here
        t->pool->run(n, 2048, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) {
                if (i + 8 < e) __builtin_prefetch(&t->parent[cells[i + 8]]);
                const int32_t p = t->parent[cells[i]];
                par[i] = p;
                if (p < 0) continue;
                int32_t expected = -1;
                __atomic_compare_exchange_n(&stamp[p], &expected, (int32_t)i, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
            }
        });
        t->pool->run(n, 2048, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) {
                const int32_t p = par[i];
                if (p < 0) continue;
                const int32_t g = t->parent[p];
                if (g >= 0 && stamp[g] >= 0) {
                    __atomic_store_n(&dep[stamp[p]], (uint8_t)1, __ATOMIC_RELAXED);
                    __atomic_store_n(&dep[stamp[g]], (uint8_t)1, __ATOMIC_RELAXED);
                }
            }
        });
        for (int64_t i = 0; i < n; ++i)                            // the interacting ones, exactly as the list says
            if (par[i] >= 0 && dep[stamp[par[i]]]) t->assign_neighbors(par[i]);
        t->pool->run(n, 256, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) {
                if (i + 2 < e && par[i + 2] >= 0) __builtin_prefetch(&t->nb[(size_t)par[i + 2] * t->nnb]);
                if (par[i] >= 0 && stamp[par[i]] == i && !dep[i]) t->assign_neighbors(par[i]);
            }
        });
        t->pool->run(n, 2048, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i)
                if (par[i] >= 0) __atomic_store_n(&stamp[par[i]], (int32_t)-1, __ATOMIC_RELAXED);
        });
        return;
    }
    // stays sequential: a listed cell's parent may itself be a child of another listed cell's parent, whose refresh
    // rewrites the row this one reads -- the result depends on the order of the list.  Software prefetch runs ahead:
    // the parent id eight positions ahead, its neighbour row four ahead, the neighbours' first-child entries two ahead
    for (int64_t i = 0; i < n; ++i) {
        if (i + 8 < n) __builtin_prefetch(&t->parent[cells[i + 8]]);
        if (i + 4 < n) {
            const int32_t p4 = t->parent[cells[i + 4]];
            if (p4 >= 0) {
                __builtin_prefetch(&t->nb[(size_t)p4 * t->nnb]);
                __builtin_prefetch(&t->nb[(size_t)p4 * t->nnb] + 16);
                __builtin_prefetch(&t->first_child[p4]);
            }
        }
        if (i + 2 < n) {
            const int32_t p2 = t->parent[cells[i + 2]];
            if (p2 >= 0) {
                const int32_t *row = &t->nb[(size_t)p2 * t->nnb];
                for (int s_ = 0; s_ < t->nnb; ++s_)
                    if (row[s_] >= 0) __builtin_prefetch(&t->first_child[row[s_]]);
            }
        }
        int32_t p = t->parent[cells[i]];
        if (p >= 0) t->assign_neighbors(p);
    }
}

void s3t_relink_parent_of(void *h, const int64_t *cells, int64_t n) {
    Topo *t = static_cast<Topo *>(h);
    t->wait_idle();
    relink_parents(t, cells, n);
}

// s_cube.py:721-728: children = [], and the cell disappears from its neighbours' nb lists
static void mark_invalid_cells(Topo *t, const int64_t *cells, int64_t n) {
    // sequential on purpose (two removed cells that are neighbours: the first removes itself from the second's row, the
    // second then no longer visits the first); the rows ahead are prefetched
    for (int64_t i = 0; i < n; ++i) {
        if (i + 4 < n) {
            __builtin_prefetch(&t->nb[(size_t)cells[i + 4] * t->nnb]);
            __builtin_prefetch(&t->nb[(size_t)cells[i + 4] * t->nnb] + 16);
        }
        if (i + 2 < n) {
            const int32_t *row = &t->nb[(size_t)cells[i + 2] * t->nnb];
            for (int s_ = 0; s_ < t->nnb; ++s_)
                if (row[s_] >= 0) {
                    __builtin_prefetch(&t->nb[(size_t)row[s_] * t->nnb]);
                    __builtin_prefetch(&t->nb[(size_t)row[s_] * t->nnb] + 16);
                }
        }
        const int32_t x = (int32_t)cells[i];
        t->first_child[x] = INVALID;
        for (int s = 0; s < t->nnb; ++s) {
            int32_t y = t->nb[(size_t)x * t->nnb + s];
            if (y < 0) continue;
            int32_t *ynb = &t->nb[(size_t)y * t->nnb];
            for (int u = 0; u < t->nnb; ++u)
                if (ynb[u] == x) ynb[u] = -1;
        }
    }
}

void s3t_mark_invalid(void *h, const int64_t *cells, int64_t n) {
    Topo *t = static_cast<Topo *>(h);
    t->wait_idle();
    mark_invalid_cells(t, cells, n);
}

// ---- asynchronous forms: the ids are copied, the call returns at once, one worker applies the commands in order.
// s3t_sync waits for the queue to drain and reports the first failure (0 ok, -1 a parent was not a leaf, -2 out of
// memory); every synchronous entry point and every table view drains the queue first.
int s3t_submit(void *h, int kind, const int64_t *ids, int64_t n, int relink) try {
    Topo *t = static_cast<Topo *>(h);
    if (kind < 0 || kind > 2) return -3;
    Topo::Cmd c{kind, relink, std::vector<int64_t>(ids, ids + n)};
    t->submit(std::move(c));
    return 0;
} catch (...) {
    return -2;
}

int s3t_sync(void *h) { return static_cast<Topo *>(h)->wait_idle(); }

// seconds per phase of the engine so far (development aid; see Topo::phase_s), out[12]
void s3t_stats(void *h, double *out) {
    Topo *t = static_cast<Topo *>(h);
    t->wait_idle();
    for (int i = 0; i < 12; ++i) out[i] = t->phase_s[i];
}

}  // extern "C"

void Topo::worker_loop() {
    while (true) {
        Cmd c;
        {
            std::unique_lock<std::mutex> lk(mtx);
            cv_work.wait(lk, [this] { return stop || !queue.empty(); });
            if (queue.empty()) return;                       // stop requested and nothing left
            c = std::move(queue.front());
            queue.pop_front();
            busy = true;
        }
        int rc = 0;
        if (error == 0) {
            try {
                if (c.kind == 0) {
                    const int64_t first = refine_batch(this, c.ids.data(), (int64_t)c.ids.size(), c.relink);
                    if (first < 0) rc = (int)first;
                } else if (c.kind == 1) {
                    const double t0 = now();
                    relink_parents(this, c.ids.data(), (int64_t)c.ids.size());
                    phase_s[7] += now() - t0;
                } else {
                    const double t0 = now();
                    mark_invalid_cells(this, c.ids.data(), (int64_t)c.ids.size());
                    phase_s[8] += now() - t0;
                }
            } catch (...) {
                rc = -2;
            }
        }
        {
            std::lock_guard<std::mutex> lk(mtx);
            if (rc != 0 && error == 0) error = rc;
            busy = false;
        }
        cv_idle.notify_all();
    }
}

extern "C" {

// _check_nb, s_cube.py:463-464: leaf neighbours with a lower level; returns the count, ids in slot order
int s3t_check_nb(void *h, int64_t cell, int64_t *out) {
    Topo *t = static_cast<Topo *>(h);
    t->wait_idle();
    int cnt = 0;
    for (int s = 0; s < t->nnb; ++s) {
        int32_t q = t->nb[(size_t)cell * t->nnb + s];
        if (q >= 0 && t->first_child[q] == LEAF && t->level[q] < t->level[cell]) out[cnt++] = q;
    }
    return cnt;
}

// _resort_nodes_and_indices_of_grid, s_cube.py:734-772 (+ 1695-1736).  Returns the number of leaf cells; results
// are read through s3t_face_ids / s3t_unique_nodes.
// run fn(begin, end) over [0, n) on the engine's pool (or inline without one)
static void for_chunks(Topo *t, int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)> &fn) {
    if (t->pool) t->pool->run(n, grain, fn);
    else if (n > 0) fn(0, n);
}

int64_t s3t_finalize(void *h, int64_t *n_unique_nodes) try {
    Topo *t = static_cast<Topo *>(h);
    if (t->wait_idle() != 0) return -1;
    const int nch = t->nch;
    const int64_t nc = t->n_cells(), nn = t->n_nodes();
    constexpr int64_t CH = 1 << 16;                                 // cells per chunk
    const int64_t n_chunks = (nc + CH - 1) / CH;
    std::vector<uint8_t> used(nn, 0);
    std::vector<int64_t> chunk_leaf((size_t)n_chunks + 1, 0), chunk_lo((size_t)n_chunks, INT64_MAX), chunk_hi((size_t)n_chunks, -1);
    for_chunks(t, n_chunks, 1, [&](int64_t cb, int64_t ce) {
        for (int64_t k = cb; k < ce; ++k) {
            int64_t lo = INT64_MAX, hi = -1, leaves = 0;
            for (int64_t c = k * CH; c < std::min(nc, (k + 1) * CH); ++c) {
                if (t->first_child[c] != LEAF) continue;
                ++leaves;
                const int64_t *ni = &t->node_idx[(size_t)c * nch];
                for (int s = 0; s < nch; ++s) {
                    const int64_t v = ni[s];
                    __atomic_store_n(&used[v], (uint8_t)1, __ATOMIC_RELAXED);    // several chunks may mark the same node
                    if (v < lo) lo = v;
                    if (v > hi) hi = v;
                }
            }
            chunk_leaf[k + 1] = leaves;
            chunk_lo[k] = lo;
            chunk_hi[k] = hi;
        }
    });
    int64_t lo = INT64_MAX, hi = -1;
    for (int64_t k = 0; k < n_chunks; ++k) {
        chunk_leaf[k + 1] += chunk_leaf[k];                            // first output row of every chunk
        lo = std::min(lo, chunk_lo[k]);
        hi = std::max(hi, chunk_hi[k]);
    }
    t->n_leaf = chunk_leaf[n_chunks];
    t->chunk_first_row = chunk_leaf;
    // unused = ids in {0..2^d-1} U [min, max] that no leaf references; everything else keeps a slot (reference quirk)
    t->face_ids.assign((size_t)nn, -1);                 // old node id -> new node id
    int64_t counter = 0;
    for (int64_t i = 0; i < nn; ++i) {
        const bool available = i < nch || (i >= lo && i <= hi);
        if (!(available && !used[i])) t->face_ids[i] = counter++;
    }
    t->n_unique = counter;
    *n_unique_nodes = counter;
    return t->n_leaf;
} catch (...) {
    return -1;
}

// the assembled grid after s3t_finalize: faces [n_leaf][2^d] (int32 when as32, else int64; leaves in ascending cell id)
// and nodes [n_unique][dim], written into the caller's arrays
void s3t_export_grid(void *h, void *faces_out, int as32, double *nodes_out) {
    Topo *t = static_cast<Topo *>(h);
    const int nch = t->nch, dim = t->dim;
    const int64_t nc = t->n_cells(), nn = t->n_nodes();
    const int64_t *map = t->face_ids.data();
    constexpr int64_t CH = 1 << 16;
    const int64_t n_chunks = (nc + CH - 1) / CH;
    for_chunks(t, n_chunks, 1, [&](int64_t cb, int64_t ce) {
        for (int64_t k = cb; k < ce; ++k) {
            int64_t row = t->chunk_first_row[k];
            for (int64_t c = k * CH; c < std::min(nc, (k + 1) * CH); ++c) {
                if (t->first_child[c] != LEAF) continue;
                const int64_t *ni = &t->node_idx[(size_t)c * nch];
                if (as32) {
                    int32_t *o = static_cast<int32_t *>(faces_out) + row * nch;
                    for (int s = 0; s < nch; ++s) o[s] = (int32_t)map[ni[s]];
                } else {
                    int64_t *o = static_cast<int64_t *>(faces_out) + row * nch;
                    for (int s = 0; s < nch; ++s) o[s] = map[ni[s]];
                }
                ++row;
            }
        }
    });
    for_chunks(t, nn, 1 << 16, [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i)
            if (map[i] >= 0)
                for (int j = 0; j < dim; ++j) nodes_out[(size_t)map[i] * dim + j] = t->nodes[(size_t)i * dim + j];
    });
}

// centres / levels of the listed cells (the leaves in the host's set order), gathered natively
void s3t_gather_cells(void *h, const int64_t *ids, int64_t n, double *centers_out, int64_t *levels_out) {
    Topo *t = static_cast<Topo *>(h);
    t->wait_idle();
    for_chunks(t, n, 1 << 15, [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            if (i + 8 < e) {
                __builtin_prefetch(&t->center[(size_t)ids[i + 8] * t->dim]);
                __builtin_prefetch(&t->level[ids[i + 8]]);
            }
            for (int j = 0; j < t->dim; ++j) centers_out[i * t->dim + j] = t->center[(size_t)ids[i] * t->dim + j];
            levels_out[i] = t->level[ids[i]];
        }
    });
}

// geometric self-check of the node rule tables: every (slot, nb_node) candidate and every sibling copy must name the
// same lattice point as the node it supplies.  Returns the number of inconsistent entries (0 expected).
int s3t_selfcheck(int dim) {
    int bad = 0;
    const int nch = 1 << dim;
    for (int i = 0; i < nch; ++i) {
        const int *di = dim == 2 ? DIR2[i] : DIR3[i];
        const std::vector<NodeRule> &rules = dim == 2 ? NODE_RULES_2D[i] : NODE_RULES_3D[i];
        std::vector<int> seen(nch, 0);
        seen[i] = 1;
        for (const NodeRule &r : rules) {
            const int *dn = dim == 2 ? DIR2[r.node] : DIR3[r.node];
            seen[r.node] += 1;
            if (r.n_cand < 0) {
                const int sib = r.cand[0][0], sn = r.cand[0][1];
                const int *ds = dim == 2 ? DIR2[sib] : DIR3[sib];
                const int *dsn = dim == 2 ? DIR2[sn] : DIR3[sn];
                if (sib >= i) ++bad;
                for (int j = 0; j < dim; ++j)
                    if (di[j] + dn[j] != ds[j] + dsn[j]) { ++bad; break; }           // positions in quarter parent widths
            } else {
                for (int a = 0; a < r.n_cand; ++a) {
                    int o[3];
                    slot_offset(dim, r.cand[a][0], o);
                    const int *dq = dim == 2 ? DIR2[r.cand[a][1]] : DIR3[r.cand[a][1]];
                    for (int j = 0; j < dim; ++j)
                        if (dn[j] != 2 * o[j] + dq[j]) { ++bad; break; }              // same-level neighbour at offset o
                }
            }
        }
        for (int c = 0; c < nch; ++c)
            if (seen[c] != 1) ++bad;                                                   // every node assigned exactly once
    }
    return bad;
}

}  // extern "C"
