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
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr int LEAF = -1;      // Cell.children is None
constexpr int INVALID = -2;   // Cell.children == []

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

struct NbEntry { int8_t pslot; int8_t target; };   // pslot < 0: sibling `target`; else parent's neighbour slot + its child

// node-sharing rules of _assign_indices.  For child i the entries are processed in order; an entry either looks the
// node up in same-level leaf neighbours (first hit wins, else a new node is appended) or copies it from an earlier
// sibling.  {node, n_cand, {slot, nb_node}...} / {node, -1, {sibling, sibling_node}}
struct NodeRule { int8_t node; int8_t n_cand; int8_t cand[3][2]; };

// slots: w0 nw1 n2 ne3 e4 se5 s6 sw7 | wl8 nwl9 nl10 nel11 el12 sel13 sl14 swl15 cl16 | wu17 nwu18 nu19 neu20 eu21 seu22
// su23 swu24 cu25 ; nodes: swu0 nwu1 neu2 seu3 swl4 nwl5 nel6 sel7
const std::vector<NodeRule> NODE_RULES_2D[4] = {
    /* child 0 */ {{1, 1, {{0, 2}}}, {2, 0, {}}, {3, 1, {{6, 2}}}},
    /* child 1 */ {{2, 1, {{2, 3}}}, {0, -1, {{0, 1}}}, {3, -1, {{0, 2}}}},
    /* child 2 */ {{3, 1, {{4, 0}}}, {0, -1, {{0, 2}}}, {1, -1, {{1, 2}}}},
    /* child 3 */ {{0, -1, {{0, 3}}}, {1, -1, {{0, 2}}}, {2, -1, {{2, 3}}}},
};

const std::vector<NodeRule> NODE_RULES_3D[8] = {
    /* child 0 */ {{1, 3, {{0, 2}, {17, 6}, {25, 5}}}, {2, 1, {{25, 6}}}, {3, 3, {{6, 2}, {23, 6}, {25, 7}}},
                   {4, 3, {{0, 7}, {7, 6}, {6, 5}}}, {5, 1, {{0, 6}}}, {6, 0, {}}, {7, 1, {{6, 6}}}},
    /* child 1 */ {{2, 3, {{2, 3}, {19, 7}, {25, 6}}}, {5, 3, {{0, 6}, {1, 7}, {2, 4}}}, {6, 1, {{2, 7}}},
                   {0, -1, {{0, 1}}}, {3, -1, {{0, 2}}}, {4, -1, {{0, 5}}}, {7, -1, {{0, 6}}}},
    /* child 2 */ {{3, 3, {{4, 0}, {21, 4}, {25, 7}}}, {6, 3, {{4, 5}, {3, 4}, {2, 7}}}, {7, 1, {{4, 4}}},
                   {0, -1, {{0, 2}}}, {1, -1, {{1, 2}}}, {4, -1, {{0, 6}}}, {5, -1, {{1, 6}}}},
    /* child 3 */ {{7, 3, {{4, 4}, {5, 5}, {6, 6}}}, {0, -1, {{0, 3}}}, {1, -1, {{0, 2}}}, {2, -1, {{2, 3}}},
                   {4, -1, {{0, 7}}}, {5, -1, {{0, 6}}}, {6, -1, {{2, 7}}}},
    /* child 4 */ {{5, 3, {{0, 6}, {8, 2}, {16, 1}}}, {6, 1, {{16, 2}}}, {7, 3, {{6, 6}, {14, 2}, {16, 3}}},
                   {0, -1, {{0, 4}}}, {1, -1, {{0, 5}}}, {2, -1, {{0, 6}}}, {3, -1, {{0, 7}}}},
    /* child 5 */ {{6, 3, {{2, 7}, {10, 3}, {16, 2}}}, {0, -1, {{1, 4}}}, {1, -1, {{1, 5}}}, {2, -1, {{1, 6}}},
                   {3, -1, {{1, 7}}}, {4, -1, {{4, 5}}}, {7, -1, {{4, 6}}}},
    /* child 6 */ {{7, 3, {{4, 4}, {12, 0}, {16, 3}}}, {0, -1, {{2, 4}}}, {1, -1, {{2, 5}}}, {2, -1, {{2, 6}}},
                   {3, -1, {{2, 7}}}, {4, -1, {{5, 7}}}, {5, -1, {{5, 6}}}},
    /* child 7 */ {{0, -1, {{3, 4}}}, {1, -1, {{3, 5}}}, {2, -1, {{3, 6}}}, {3, -1, {{3, 7}}}, {4, -1, {{4, 7}}},
                   {5, -1, {{4, 6}}}, {6, -1, {{6, 7}}}},
};

struct Topo {
    int dim, nch, nnb;
    double width;
    std::vector<int32_t> level, parent, first_child;
    std::vector<int32_t> nb;          // [n_cells][nnb]
    std::vector<int64_t> node_idx;    // [n_cells][nch]
    std::vector<double> center;       // [n_cells][dim]
    std::vector<double> nodes;        // [n_nodes][dim]
    std::vector<NbEntry> nb_table;    // [nch][nnb]
    // finalize() results
    std::vector<int64_t> face_ids;
    std::vector<double> unique_nodes;
    int64_t n_leaf = 0;

    int64_t n_cells() const { return (int64_t)level.size(); }
    int64_t n_nodes() const { return (int64_t)(nodes.size() / dim); }
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
        level.push_back(lvl);
        parent.push_back(par);
        first_child.push_back(LEAF);
        nb.insert(nb.end(), nnb, -1);
        node_idx.insert(node_idx.end(), nch, 0);
        center.insert(center.end(), c, c + dim);
    }

    // room for `extra` more cells (one geometric growth step instead of one reallocation check per vector per cell)
    void reserve_cells(int64_t extra) {
        const size_t want = (size_t)(n_cells() + extra);
        if (want <= level.capacity()) return;
        const size_t cap = std::max(want, level.capacity() * 2);
        level.reserve(cap);
        parent.reserve(cap);
        first_child.reserve(cap);
        nb.reserve(cap * nnb);
        node_idx.reserve(cap * nch);
        center.reserve(cap * dim);
        nodes.reserve(std::max(nodes.capacity(), (size_t)(nodes.size() + (size_t)extra * dim)));
    }

    // _assign_neighbors(cell, children=existing children)
    void assign_neighbors(int32_t P) {
        const int32_t fc = first_child[P];
        if (fc < 0) return;
        const int32_t *pnb = &nb[(size_t)P * nnb];
        for (int c = 0; c < nch; ++c) {
            int32_t *cnb = &nb[(size_t)(fc + c) * nnb];
            for (int s = 0; s < nnb; ++s) {
                const NbEntry e = nb_table[(size_t)c * nnb + s];
                if (e.pslot < 0) { cnb[s] = fc + e.target; continue; }
                int32_t q = pnb[e.pslot];
                if (q >= 0 && first_child[q] >= 0) q = first_child[q] + e.target;     // parent_or_child
                cnb[s] = q;
            }
        }
    }

    bool check_nb_node(int32_t cell, int slot) const {
        int32_t q = nb[(size_t)cell * nnb + slot];
        return q >= 0 && first_child[q] == LEAF && level[q] == level[cell];
    }

    int64_t new_node(int32_t cell, int node) {
        const double off = (0.5 * width) / std::ldexp(1.0, level[cell]);
        for (int j = 0; j < dim; ++j) nodes.push_back(center[(size_t)cell * dim + j] + dir(node)[j] * off);
        return n_nodes() - 1;
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
    }

    // one parent of _refine_cells / the uniform loop: children, neighbour links, node ids
    void refine_one(int32_t P) {
        const int32_t fc = (int32_t)n_cells();
        const double off = (0.25 * width) / std::ldexp(1.0, level[P]);
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
    t->push_cell(0, -1, root_center);
    // root nodes, s_cube.py:368,386-394: centre + dir * 0.5 * width, ids 0..2^d-1
    for (int c = 0; c < t->nch; ++c) {
        for (int j = 0; j < dim; ++j) t->nodes.push_back(root_center[j] + t->dir(c)[j] * 0.5 * width);
        t->node_idx[c] = c;
    }
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
static int64_t refine_batch(Topo *t, const int64_t *parents, int64_t n, int relink) {
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
    for (int64_t i = 0; i < n; ++i) {
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
    for (int64_t i = 0; i < n; ++i) {
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
                    relink_parents(this, c.ids.data(), (int64_t)c.ids.size());
                } else {
                    mark_invalid_cells(this, c.ids.data(), (int64_t)c.ids.size());
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
int64_t s3t_finalize(void *h, int64_t *n_unique_nodes) try {
    Topo *t = static_cast<Topo *>(h);
    if (t->wait_idle() != 0) return -1;
    const int nch = t->nch;
    t->face_ids.clear();
    for (int64_t c = 0; c < t->n_cells(); ++c)
        if (t->first_child[c] == LEAF)
            t->face_ids.insert(t->face_ids.end(), &t->node_idx[(size_t)c * nch], &t->node_idx[(size_t)c * nch] + nch);
    t->n_leaf = (int64_t)t->face_ids.size() / nch;
    const int64_t nn = t->n_nodes();
    std::vector<uint8_t> used(nn, 0);
    int64_t lo = INT64_MAX, hi = -1;
    for (int64_t v : t->face_ids) {
        used[v] = 1;
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    // unused = ids in {0..2^d-1} U [min, max] that no leaf references; everything else keeps a slot (reference quirk)
    std::vector<int64_t> mapping(nn, -1);
    int64_t counter = 0;
    for (int64_t i = 0; i < nn; ++i) {
        bool available = i < nch || (i >= lo && i <= hi);
        bool unused = available && !used[i];
        if (!unused) mapping[i] = counter++;
    }
    t->unique_nodes.assign((size_t)counter * t->dim, 0.0);
    for (int64_t i = 0; i < nn; ++i)
        if (mapping[i] >= 0)
            for (int j = 0; j < t->dim; ++j) t->unique_nodes[(size_t)mapping[i] * t->dim + j] = t->nodes[(size_t)i * t->dim + j];
    for (int64_t &v : t->face_ids) v = mapping[v];
    *n_unique_nodes = counter;
    return t->n_leaf;
} catch (...) {
    return -1;
}

int64_t *s3t_face_ids(void *h) { return static_cast<Topo *>(h)->face_ids.data(); }
double *s3t_unique_nodes(void *h) { return static_cast<Topo *>(h)->unique_nodes.data(); }

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
