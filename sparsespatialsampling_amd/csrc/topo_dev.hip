// Device-resident topology engine of the S^3 sampling tree: neighbour links, shared-node numbering, invalid-cell
// bookkeeping and the final renumbering as HIP kernels over tables that live in HBM.  gfx950 only.
//
// Reference behaviour restated here (file:line relative to the reference checkout):
//   Cell                         s_cube.py:32-83      -> structure-of-arrays (level, parent, first_child, nb, node_idx)
//   _assign_neighbors            s_cube.py:904-1186   -> lattice rule table (built on the host as in topology.cpp)
//   _assign_indices              s_cube.py:1188-1536  -> topo_tables.h decision tables
//   check_nb_node                s_cube.py:1739-1755
//   _remove_invalid_cells (nb)   s_cube.py:721-728
//   _resort_nodes_and_indices_of_grid + renumber_node_indices_parallel   s_cube.py:734-772, 1695-1736
//
// The reference applies all of this cell by cell, in the iteration order of Python sets, and the node numbering and the
// (deliberately stale) neighbour links depend on that order.  Every operation here receives the ordered id list the
// host decided on and reproduces the result of the sequential procedure with data-parallel passes:
//   * refine: children ids are known up front (first + 2^d * position), so every parent evaluates "was my neighbour
//     refined before me" from the state before the batch plus its position (pass A: links and node entries, new nodes
//     counted per parent; scan; pass C: ids + coordinates of the new nodes; pass D: references into earlier parents'
//     children are followed to a final id) -- the four-pass form of csrc/topology.cpp;
//   * relink (cell.parent.children = _assign_neighbors(...), s_cube.py:609, 834): a refresh of parent p rewrites the rows
//     of p's children from p's own row, so the result of the ordered list depends only on, per listed parent, the time
//     of its last refresh and the last refresh of each ancestor before that: every listed parent walks up that chain,
//     recomputes its row from the first ancestor whose row is still the old one, and writes its children's rows in a
//     second pass (no kernel reads a row another one is writing);
//   * mark invalid: x removes itself from the rows of the neighbours listed in its own row *as that row is when x's turn
//     comes*; an entry y of x's row has been wiped by then exactly if y comes earlier in the list and had x in its row,
//     which is decided from the rows before the call (pass 1), the wipes are applied in pass 2.
// The host engine (libs3topo.so) remains the executable specification: tests/test_gpu_topology.py compares whole tables.
#include "common.h"
#include "scan_sort.h"
#include "topo_tables.h"


#include <algorithm>
#include <cmath>
#include <cstring>
#include <exception>
#include <vector>

using s3topo::INVALID;
using s3topo::LEAF;
using s3topo::NbEntry;
using s3topo::NodeRule;
using s3topo::REF_BASE;

namespace {

// host copies of the decision tables (fixed shape: every child has N_RULES entries)
const NodeRule H_RULES_2D[4][s3topo::N_RULES_2D] = S3_NODE_RULES_2D_INIT;
const NodeRule H_RULES_3D[8][s3topo::N_RULES_3D] = S3_NODE_RULES_3D_INIT;
const int H_DIR2[4][3] = {{-1, -1, 0}, {-1, 1, 0}, {1, 1, 0}, {1, -1, 0}};
const int H_DIR3[8][3] = {{-1, -1, 1}, {-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, -1}, {-1, 1, -1}, {1, 1, -1}, {1, -1, -1}};
const int H_PLANE[8][2] = {{-1, 0}, {-1, 1}, {0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}};

void slot_offset(int slot, int o[3]) {
    o[2] = 0;
    if (slot < 8) { o[0] = H_PLANE[slot][0]; o[1] = H_PLANE[slot][1]; return; }
    if (slot < 16) { o[0] = H_PLANE[slot - 8][0]; o[1] = H_PLANE[slot - 8][1]; o[2] = -1; return; }
    if (slot == 16) { o[0] = 0; o[1] = 0; o[2] = -1; return; }
    if (slot < 25) { o[0] = H_PLANE[slot - 17][0]; o[1] = H_PLANE[slot - 17][1]; o[2] = 1; return; }
    o[0] = 0; o[1] = 0; o[2] = 1;
}

// lattice rule behind the reference's neighbour table (same construction as topology.cpp)
std::vector<NbEntry> build_nb_table(int dim) {
    const int nch = 1 << dim, nnb = dim == 2 ? 8 : 26;
    std::vector<NbEntry> tab((size_t)nch * nnb);
    for (int c = 0; c < nch; ++c)
        for (int s = 0; s < nnb; ++s) {
            int o[3], p[3] = {0, 0, 0}, big[3] = {0, 0, 0}, t[3] = {0, 0, 0};
            slot_offset(s, o);
            const int *dc = dim == 2 ? H_DIR2[c] : H_DIR3[c];
            bool crosses = false;
            for (int j = 0; j < dim; ++j) {
                p[j] = dc[j] + 2 * o[j];
                big[j] = p[j] == 3 ? 1 : (p[j] == -3 ? -1 : 0);
                t[j] = p[j] - 4 * big[j];
                crosses |= big[j] != 0;
            }
            int pslot = -1, target = -1;
            if (crosses)
                for (int q = 0; q < nnb; ++q) {
                    int u[3];
                    slot_offset(q, u);
                    if (u[0] == big[0] && u[1] == big[1] && (dim == 2 || u[2] == big[2])) pslot = q;
                }
            for (int q = 0; q < nch; ++q) {
                const int *d = dim == 2 ? H_DIR2[q] : H_DIR3[q];
                if (d[0] == t[0] && d[1] == t[1] && (dim == 2 || d[2] == t[2])) target = q;
            }
            tab[(size_t)c * nnb + s] = NbEntry{(int8_t)pslot, (int8_t)target};
        }
    return tab;
}

}  // namespace

// what the kernels see (by value)
struct TopoView {
    int dim, nch, nnb, n_rules;
    int32_t *level, *parent, *first_child, *batch_pos, *nb;
    int64_t *node_idx;
    double *center, *nodes;
    const NbEntry *nb_table;
    const NodeRule *rules;           // [nch][n_rules]
    const double *half_width, *quarter_width;
    int64_t *counters;               // [0] nodes in use, [1] error flag, [2] min referenced node, [3] max, [4] scratch total
};

struct s3_topo : TopoView {
    double width = 0;
    int64_t cell_cap = 0, node_cap = 0;
    int64_t n_used = 0;               // cells created (host count: the ids of a batch are known up front)
    int64_t n_nodes_bound = 0;        // upper bound of the nodes in use (exact after a sync)
    hipStream_t st = nullptr;         // the engine's own stream: its kernels overlap the KNN kernels of the refine loop
    // scratch, grown on demand
    int64_t *ids = nullptr, *cnt = nullptr, *base = nullptr;
    int64_t ids_cap = 0;
    void *scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    int32_t *rows_tmp = nullptr;
    int64_t rows_tmp_cap = 0;
    uint8_t *flags = nullptr;
    int64_t flags_cap = 0;
    // finalize results
    int64_t *map = nullptr, *leaf_row = nullptr;
    int64_t map_cap = 0, leaf_row_cap = 0;
    int64_t n_leaf = 0, n_unique = 0;
    NbEntry *d_nb_table = nullptr;
    NodeRule *d_rules = nullptr;
    double *d_widths = nullptr;
    // id lists go up through two pinned staging buffers (a pageable source would make every update wait for the stream)
    int64_t *h_stage[2] = {nullptr, nullptr};
    int64_t h_stage_cap[2] = {0, 0};
    hipEvent_t h_stage_free[2] = {nullptr, nullptr};
    int h_turn = 0;
    // tables replaced by larger ones: released at the next sync (hipFree waits for the whole device, i.e. for the KNN
    // kernel of the refine loop that runs beside the engine)
    std::vector<void *> retired;
};

namespace s3 {

constexpr int TB = 128;
static unsigned blocks_for(int64_t n) { return (unsigned)((n + TB - 1) / TB); }

__device__ __forceinline__ int64_t enc_new(int local) { return -(int64_t)(1 + local); }
__device__ __forceinline__ bool is_new(int64_t v) { return v < 0 && v > -REF_BASE; }
__device__ __forceinline__ int dec_new(int64_t v) { return (int)(-v - 1); }
__device__ __forceinline__ int64_t enc_ref(int64_t entry) { return -(REF_BASE + entry); }
__device__ __forceinline__ bool is_ref(int64_t v) { return v <= -REF_BASE; }
__device__ __forceinline__ int64_t dec_ref(int64_t v) { return -v - REF_BASE; }

__global__ void topo_validate_kernel(TopoView t, const int64_t *__restrict__ parents, int64_t n, int64_t first) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i >= n) return;
    const int64_t P = parents[i];
    if (P < 0 || P >= first || t.first_child[P] != LEAF || atomicCAS(&t.batch_pos[P], -1, (int32_t)i) != -1)
        atomicExch(reinterpret_cast<unsigned long long *>(&t.counters[1]), 1ull);
}

// row of child c of a cell whose row is `prow` and whose children start at fc (_assign_neighbors for one child);
// fc_of(q) = first child of q as the caller sees it
template <typename FcOf>
__device__ __forceinline__ void child_row(const TopoView &t, const int32_t *prow, int32_t fc, int c, int32_t *out, FcOf fc_of) {
    for (int s = 0; s < t.nnb; ++s) {
        const NbEntry e = t.nb_table[c * t.nnb + s];
        if (e.pslot < 0) { out[s] = fc + e.target; continue; }
        const int32_t q = prow[e.pslot];
        const int32_t f = q >= 0 ? fc_of(q) : -1;
        out[s] = f >= 0 ? f + e.target : q;                    // parent_or_child, s_cube.py:1758-1775
    }
}

// pass A of a refine batch: the children of parent i -- levels, centres, links, node entries (final id / l-th new node of
// this parent / reference to an entry of an earlier parent's child); one thread per parent, as the sequential numbering
// of a parent's new nodes asks for
__global__ void __launch_bounds__(TB)
topo_build_kernel(TopoView t, const int64_t *__restrict__ parents, int64_t n, int64_t first, int64_t *__restrict__ new_count) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i >= n || t.counters[1] != 0) return;
    const int nch = t.nch, nnb = t.nnb, dim = t.dim;
    const int32_t P = (int32_t)parents[i];
    const int32_t fc = (int32_t)(first + i * nch);
    const int32_t lvl = t.level[P] + 1;
    const double off = t.quarter_width[lvl - 1];
    for (int c = 0; c < nch; ++c) {
        const size_t cell = (size_t)fc + c;
        t.level[cell] = lvl;
        t.parent[cell] = P;
        t.first_child[cell] = LEAF;
        t.batch_pos[cell] = -1;
        for (int j = 0; j < dim; ++j) t.center[cell * dim + j] = t.center[(size_t)P * dim + j] + dir_comp(dim, c, j) * off;
    }
    // links: a neighbour of the parent counts as refined when it was before the batch or comes earlier in it
    int32_t prow[26];
    for (int s = 0; s < nnb; ++s) prow[s] = t.nb[(size_t)P * nnb + s];
    auto fc_of = [&](int32_t q) {
        int32_t f = t.first_child[q];
        if (f == LEAF) {
            const int32_t bp = t.batch_pos[q];
            if (bp >= 0 && bp < i) f = (int32_t)(first + (int64_t)bp * nch);
        }
        return f;
    };
    for (int c = 0; c < nch; ++c) child_row(t, prow, fc, c, &t.nb[(size_t)(fc + c) * nnb], fc_of);
    // node entries
    int local = 0;
    for (int k = 0; k < nch; ++k) {
        const int32_t cell = fc + k;
        int64_t *ni = &t.node_idx[(size_t)cell * nch];
        const int32_t *cnb = &t.nb[(size_t)cell * nnb];
        ni[k] = t.node_idx[(size_t)P * nch + k];
        for (int ri = 0; ri < t.n_rules; ++ri) {
            const NodeRule r = t.rules[k * t.n_rules + ri];
            if (r.n_cand < 0) {
                ni[r.node] = t.node_idx[(size_t)(fc + r.cand[0][0]) * nch + r.cand[0][1]];
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
                    if (t.level[parents[j]] + 1 != lvl) continue;
                    ni[r.node] = j == i ? t.node_idx[entry] : enc_ref(entry);      // own sibling: entry as it stands
                    found = true;
                } else if (t.first_child[q] == LEAF && !(t.batch_pos[q] >= 0 && t.batch_pos[q] < i) && t.level[q] == lvl) {
                    ni[r.node] = t.node_idx[entry];
                    found = true;
                }
            }
            if (!found) ni[r.node] = enc_new(local++);
        }
    }
    new_count[i] = local;
}

// pass C: the parent's new nodes get their ids (in the order pass A met them) and their coordinates
__global__ void __launch_bounds__(TB)
topo_number_kernel(TopoView t, int64_t n, int64_t first, const int64_t *__restrict__ new_base) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i >= n || t.counters[1] != 0) return;
    const int nch = t.nch, dim = t.dim;
    const int32_t fc = (int32_t)(first + i * nch);
    const int64_t base = t.counters[0] + new_base[i];
    int seen = 0;
    for (int k = 0; k < nch; ++k) {
        const int32_t cell = fc + k;
        int64_t *ni = &t.node_idx[(size_t)cell * nch];
        auto fix = [&](int node) {
            const int64_t v = ni[node];
            if (!is_new(v)) return;
            const int l = dec_new(v);
            if (l == seen) {
                const double off = t.half_width[t.level[cell]];
                for (int j = 0; j < dim; ++j)
                    t.nodes[(size_t)(base + l) * dim + j] = t.center[(size_t)cell * dim + j] + dir_comp(dim, node, j) * off;
                ++seen;
            }
            ni[node] = base + l;
        };
        fix(k);
        for (int ri = 0; ri < t.n_rules; ++ri) fix(t.rules[k * t.n_rules + ri].node);
    }
}

// pass D: follow references into earlier parents' children until a final id is met (a stale read sees an older link of
// the same chain, never a wrong id)
__global__ void __launch_bounds__(TB)
topo_resolve_kernel(TopoView t, int64_t n, int64_t first) {
    const int64_t e = blockIdx.x * (int64_t)TB + threadIdx.x;
    const int per = t.nch * t.nch;
    if (e >= n * per || t.counters[1] != 0) return;
    volatile int64_t *tab = t.node_idx;
    const size_t at = (size_t)first * t.nch + (size_t)e;
    int64_t v = tab[at];
    if (!is_ref(v)) return;
    // a lattice point is shared by at most 2^d cells, so a chain has at most 2^d - 1 links; the bound is the exit every
    // wave reaches whatever the tables hold
    for (int hop = 0; hop < 64 && is_ref(v); ++hop) v = tab[(size_t)dec_ref(v)];
    if (is_ref(v)) {
        atomicExch(reinterpret_cast<unsigned long long *>(&t.counters[1]), 2ull);
        return;
    }
    tab[at] = v;
}

__global__ void __launch_bounds__(TB)
topo_finish_kernel(TopoView t, const int64_t *__restrict__ parents, int64_t n, int64_t first, const int64_t *__restrict__ new_base,
                   const int64_t *__restrict__ new_count) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i >= n || t.counters[1] != 0) return;
    t.first_child[parents[i]] = (int32_t)(first + i * t.nch);
    t.batch_pos[parents[i]] = -1;
    if (i == n - 1) t.counters[0] += new_base[i] + new_count[i];
}

// _assign_neighbors(P) for every listed P, one thread per (parent, child): the "update all nb" pass of a uniform level
// (s_cube.py:547-549); the parents of one batch are leaves before it, none is a child of another
__global__ void __launch_bounds__(TB)
topo_relink_batch_kernel(TopoView t, const int64_t *__restrict__ parents, int64_t n) {
    const int64_t w = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (w >= n * t.nch || t.counters[1] != 0) return;
    const int64_t i = w / t.nch;
    const int c = (int)(w - i * t.nch);
    const int32_t P = (int32_t)parents[i];
    const int32_t fc = t.first_child[P];
    if (fc < 0) return;
    int32_t prow[26];
    for (int s = 0; s < t.nnb; ++s) prow[s] = t.nb[(size_t)P * t.nnb + s];
    child_row(t, prow, fc, c, &t.nb[(size_t)(fc + c) * t.nnb], [&](int32_t q) { return t.first_child[q]; });
}

// ---- relink_parent_of(cells): ordered refreshes of the listed cells' parents -----------------------------------------
__global__ void topo_set_pos_kernel(TopoView t, const int64_t *__restrict__ cells, int64_t n, int set) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i < n) t.batch_pos[cells[i]] = set ? (int32_t)i : -1;
}

// last position < limit at which parent g is refreshed (= one of its children is listed), -1 if none
__device__ __forceinline__ int32_t last_refresh_before(const TopoView &t, int32_t g, int32_t limit) {
    const int32_t fc = t.first_child[g];
    int32_t best = -1;
    for (int c = 0; c < t.nch; ++c) {
        const int32_t p = t.batch_pos[fc + c];
        if (p >= 0 && p < limit && p > best) best = p;
    }
    return best;
}

// pass 1: the occurrence that is the LAST refresh of its parent p computes the rows of p's children as the sequential
// list leaves them, into rows_tmp[i]; nothing is written to the tables
__global__ void __launch_bounds__(TB)
topo_relink_compute_kernel(TopoView t, const int64_t *__restrict__ cells, int64_t n, int32_t *__restrict__ rows_tmp) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i >= n) return;
    const int nnb = t.nnb, nch = t.nch;
    int32_t *mine = rows_tmp + (size_t)i * nch * nnb;
    mine[0] = INT32_MIN;                                   // "nothing to write for this occurrence"
    const int32_t p = t.parent[cells[i]];
    if (p < 0) return;
    if (last_refresh_before(t, p, INT32_MAX) != (int32_t)i) return;
    // walk up: ancestors whose refresh precedes (transitively) this one
    int8_t path[64];                                       // child number of each step down, deepest last
    int depth = 0;
    int32_t a = p, when = (int32_t)i;
    while (true) {
        const int32_t g = t.parent[a];
        if (g < 0) break;
        const int32_t tg = last_refresh_before(t, g, when);
        if (tg < 0) break;
        if (depth == 63) break;                            // (deeper than any tree: level <= 63)
        path[depth++] = (int8_t)(a - t.first_child[g]);
        a = g;
        when = tg;
    }
    // the row of `a` is still the one in the table; come back down
    int32_t row[26], next[26];
    for (int s = 0; s < nnb; ++s) row[s] = t.nb[(size_t)a * nnb + s];
    auto fc_of = [&](int32_t q) { return t.first_child[q]; };
    while (depth > 0) {
        const int c = path[--depth];
        child_row(t, row, t.first_child[a], c, next, fc_of);
        a = t.first_child[a] + c;
        for (int s = 0; s < nnb; ++s) row[s] = next[s];
    }
    const int32_t fc = t.first_child[p];
    for (int c = 0; c < nch; ++c) child_row(t, row, fc, c, mine + (size_t)c * nnb, fc_of);
}

// pass 2: the computed rows go into the table
__global__ void __launch_bounds__(TB)
topo_relink_write_kernel(TopoView t, const int64_t *__restrict__ cells, int64_t n, const int32_t *__restrict__ rows_tmp) {
    const int64_t w = blockIdx.x * (int64_t)TB + threadIdx.x;
    const int per = t.nch * t.nnb;
    if (w >= n * per) return;
    const int64_t i = w / per;
    const int e = (int)(w - i * per);
    const int32_t *mine = rows_tmp + (size_t)i * per;
    if (mine[0] == INT32_MIN) return;
    const int32_t fc = t.first_child[t.parent[cells[i]]];
    t.nb[(size_t)fc * t.nnb + e] = mine[e];
}

// ---- mark_invalid(cells) -------------------------------------------------------------------------------------------------
// pass 1: does x_i, when its turn comes, still find y = row(x_i)[s] in its row?  Not if y comes earlier in the list and
// had x_i in its own row (it then wiped itself from x_i's row)
__global__ void __launch_bounds__(TB)
topo_invalid_decide_kernel(TopoView t, const int64_t *__restrict__ cells, int64_t n, uint8_t *__restrict__ visit) {
    const int64_t w = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (w >= n * t.nnb) return;
    const int64_t i = w / t.nnb;
    const int s = (int)(w - i * t.nnb);
    const int32_t x = (int32_t)cells[i];
    const int32_t y = t.nb[(size_t)x * t.nnb + s];
    uint8_t v = 0;
    if (y >= 0) {
        v = 1;
        const int32_t h = t.batch_pos[y];
        if (h >= 0 && h < i)
            for (int u = 0; u < t.nnb; ++u)
                if (t.nb[(size_t)y * t.nnb + u] == x) { v = 0; break; }
    }
    visit[w] = v;
}

__global__ void __launch_bounds__(TB)
topo_invalid_apply_kernel(TopoView t, const int64_t *__restrict__ cells, int64_t n, const uint8_t *__restrict__ visit,
                          const int32_t *__restrict__ nb_before) {
    const int64_t w = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (w >= n * t.nnb) return;
    const int64_t i = w / t.nnb;
    const int s = (int)(w - i * t.nnb);
    const int32_t x = (int32_t)cells[i];
    if (s == 0) t.first_child[x] = INVALID;
    if (!visit[w]) return;
    const int32_t y = nb_before[w];
    int32_t *ynb = &t.nb[(size_t)y * t.nnb];
    for (int u = 0; u < t.nnb; ++u)
        if (ynb[u] == x) ynb[u] = -1;
}

__global__ void topo_copy_rows_kernel(TopoView t, const int64_t *__restrict__ cells, int64_t n, int32_t *__restrict__ out) {
    const int64_t w = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (w >= n * t.nnb) return;
    const int64_t i = w / t.nnb;
    out[w] = t.nb[(size_t)cells[i] * t.nnb + (w - i * t.nnb)];
}

// ---- finalize ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TB)
topo_mark_used_kernel(TopoView t, int64_t nc, int64_t *__restrict__ leaf_flag, uint8_t *__restrict__ used) {
    const int64_t c = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (c >= nc) return;
    const bool leaf = t.first_child[c] == LEAF;
    leaf_flag[c] = leaf ? 1 : 0;
    if (!leaf) return;
    long long lo = INT64_MAX, hi = -1;
    for (int s = 0; s < t.nch; ++s) {
        const int64_t v = t.node_idx[(size_t)c * t.nch + s];
        used[v] = 1;
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    atomicMin(reinterpret_cast<long long *>(&t.counters[2]), lo);
    atomicMax(reinterpret_cast<long long *>(&t.counters[3]), hi);
}

// unused = ids in {0..2^d-1} U [min, max] that no leaf references; everything else keeps a slot (reference quirk)
__global__ void topo_keep_kernel(TopoView t, int64_t nn, const uint8_t *__restrict__ used, int64_t *__restrict__ keep) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i >= nn) return;
    const bool available = i < t.nch || (i >= t.counters[2] && i <= t.counters[3]);
    keep[i] = (available && !used[i]) ? 0 : 1;
}

// exclusive scans -> row of every leaf / new id of every kept node (-1 = dropped)
__global__ void topo_map_kernel(int64_t nn, const uint8_t *__restrict__ used, const int64_t *__restrict__ keep_scan,
                                const int64_t *__restrict__ keep, int64_t *__restrict__ map) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i < nn) map[i] = keep[i] ? keep_scan[i] : -1;
}

template <typename F>
__global__ void topo_faces_kernel(TopoView t, int64_t nc, const int64_t *__restrict__ leaf_row, const int64_t *__restrict__ map,
                                  F *__restrict__ faces) {
    const int64_t w = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (w >= nc * t.nch) return;
    const int64_t c = w / t.nch;
    if (t.first_child[c] != LEAF) return;
    const int s = (int)(w - c * t.nch);
    faces[leaf_row[c] * t.nch + s] = (F)map[t.node_idx[w]];
}

__global__ void topo_nodes_out_kernel(TopoView t, int64_t nn, const int64_t *__restrict__ map, double *__restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i >= nn || map[i] < 0) return;
    for (int j = 0; j < t.dim; ++j) out[(size_t)map[i] * t.dim + j] = t.nodes[(size_t)i * t.dim + j];
}

__global__ void topo_gather_cells_kernel(TopoView t, const int64_t *__restrict__ ids, int64_t n, double *__restrict__ centers,
                                         int64_t *__restrict__ levels) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < t.dim; ++j) centers[i * t.dim + j] = t.center[(size_t)ids[i] * t.dim + j];
    levels[i] = t.level[ids[i]];
}

__global__ void topo_fill_i32_kernel(int32_t *p, int64_t n, int32_t v) {
    const int64_t i = blockIdx.x * (int64_t)TB + threadIdx.x;
    if (i < n) p[i] = v;
}

template <typename T>
static int grow(T *&p, int64_t &cap, int64_t want, int64_t keep, hipStream_t st, std::vector<void *> *retired = nullptr) {
    if (want <= cap) return S3_OK;
    const int64_t nc = std::max<int64_t>(std::max(want, cap * 2), 1024);
    T *q = nullptr;
    const hipError_t e = hipMalloc(reinterpret_cast<void **>(&q), sizeof(T) * (size_t)nc);
    if (e != hipSuccess) {
        s3::set_error("topology engine: hipMalloc of %zu bytes failed: %s", sizeof(T) * (size_t)nc, hipGetErrorString(e));
        return S3_ENOMEM;
    }
    if (p && keep > 0) S3_HIP_CHECK(hipMemcpyAsync(q, p, sizeof(T) * (size_t)keep, hipMemcpyDeviceToDevice, st));
    if (p) {
        if (retired) {
            retired->push_back(p);                       // kernels in flight may still read it: freed at the next sync
        } else {
            S3_HIP_CHECK(hipStreamSynchronize(st));
            (void)hipFree(p);
        }
    }
    p = q;
    cap = nc;
    return S3_OK;
}

static int reserve_cells(s3_topo *t, int64_t extra) {
    const int64_t want = t->n_used + extra;
    if (want > t->cell_cap) {
        const int64_t cap = std::max<int64_t>(std::max(want, t->cell_cap * 2), 4096);
        int64_t c;
#define S3_GROW_TABLE(PTR, TYPE, PER)                                                        \
    do {                                                                                     \
        c = t->cell_cap * (PER);                                                             \
        const int rc_ = grow<TYPE>(t->PTR, c, cap * (PER), t->n_used * (PER), t->st, &t->retired);        \
        if (rc_ != S3_OK) return rc_;                                                        \
    } while (0)
        S3_GROW_TABLE(level, int32_t, 1);
        S3_GROW_TABLE(parent, int32_t, 1);
        S3_GROW_TABLE(first_child, int32_t, 1);
        S3_GROW_TABLE(nb, int32_t, t->nnb);
        S3_GROW_TABLE(node_idx, int64_t, t->nch);
        S3_GROW_TABLE(center, double, t->dim);
        // batch_pos: -1 everywhere outside a running operation; the new part is initialised by pass A for new cells, so
        // only the table itself has to be carried over
        S3_GROW_TABLE(batch_pos, int32_t, 1);
#undef S3_GROW_TABLE
        t->cell_cap = cap;
    }
    return S3_OK;
}

static int upload_ids(s3_topo *t, const int64_t *h_ids, int64_t n) {
    if (n > t->ids_cap) {
        int64_t c0 = t->ids_cap, c1 = t->ids_cap, c2 = t->ids_cap;
        int rc = grow<int64_t>(t->ids, c0, n, 0, t->st);
        if (rc == S3_OK) rc = grow<int64_t>(t->cnt, c1, n, 0, t->st);
        if (rc == S3_OK) rc = grow<int64_t>(t->base, c2, n, 0, t->st);
        if (rc != S3_OK) return rc;
        t->ids_cap = c0;
    }
    // through a pinned buffer: the caller's array is copied at once (it may be reused), the transfer itself is asynchronous
    const int b = t->h_turn;
    t->h_turn ^= 1;
    if (!t->h_stage_free[b]) S3_HIP_CHECK(hipEventCreateWithFlags(&t->h_stage_free[b], hipEventDisableTiming));
    S3_HIP_CHECK(hipEventSynchronize(t->h_stage_free[b]));                 // the transfer before last has left the buffer
    if (n > t->h_stage_cap[b]) {
        if (t->h_stage[b]) (void)hipHostFree(t->h_stage[b]);
        t->h_stage[b] = nullptr;
        t->h_stage_cap[b] = 0;
        const int64_t cap = std::max<int64_t>(n + n / 2, 1 << 16);
        if (hipHostMalloc(reinterpret_cast<void **>(&t->h_stage[b]), sizeof(int64_t) * (size_t)cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            t->h_stage[b] = nullptr;                                       // no page-locked memory: plain copy
            S3_HIP_CHECK(hipMemcpyAsync(t->ids, h_ids, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, t->st));
            return S3_OK;
        }
        t->h_stage_cap[b] = cap;
    }
    std::memcpy(t->h_stage[b], h_ids, sizeof(int64_t) * (size_t)n);
    S3_HIP_CHECK(hipMemcpyAsync(t->ids, t->h_stage[b], sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, t->st));
    S3_HIP_CHECK(hipEventRecord(t->h_stage_free[b], t->st));
    return S3_OK;
}

static int exclusive_scan(s3_topo *t, const int64_t *in, int64_t *out, int64_t n) {
    const size_t need = sizeof(int64_t) * scan_tmp_items(n);
    if (need > t->scan_tmp_bytes) {
        if (t->scan_tmp) {
            S3_HIP_CHECK(hipStreamSynchronize(t->st));
            (void)hipFree(t->scan_tmp);
            t->scan_tmp = nullptr;
        }
        S3_HIP_CHECK(hipMalloc(&t->scan_tmp, need));
        t->scan_tmp_bytes = need;
    }
    S3_HIP_CHECK(s3::exclusive_scan<int64_t>(in, out, n, static_cast<int64_t *>(t->scan_tmp), t->st));   // csrc/scan_sort.h
    return S3_OK;
}

}  // namespace s3

using namespace s3;

extern "C" {

void s3_topo_destroy(s3_topo *t) {
    if (!t) return;
    if (t->st) (void)hipStreamSynchronize(t->st);
    void *ptrs[] = {t->level, t->parent, t->first_child, t->batch_pos, t->nb, t->node_idx, t->center, t->nodes, t->counters,
                    t->ids, t->cnt, t->base, t->scan_tmp, t->rows_tmp, t->flags, t->map, t->leaf_row, t->d_nb_table,
                    t->d_rules, t->d_widths};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (void *p : t->retired) (void)hipFree(p);
    for (int b = 0; b < 2; ++b) {
        if (t->h_stage[b]) (void)hipHostFree(t->h_stage[b]);
        if (t->h_stage_free[b]) (void)hipEventDestroy(t->h_stage_free[b]);
    }
    if (t->st) (void)hipStreamDestroy(t->st);
    delete t;
}

int s3_topo_create(int dim, double width, const double *h_root_center, s3_topo **out) try {
    S3_REQUIRE(out != nullptr && h_root_center != nullptr, "s3_topo_create: null argument");
    *out = nullptr;
    S3_REQUIRE(dim == 2 || dim == 3, "s3_topo_create: dim must be 2 or 3, got %d", dim);
    s3_topo *t = new s3_topo();
    *static_cast<TopoView *>(t) = TopoView{};
    t->dim = dim;
    t->nch = 1 << dim;
    t->nnb = dim == 2 ? 8 : 26;
    t->n_rules = dim == 2 ? s3topo::N_RULES_2D : s3topo::N_RULES_3D;
    t->width = width;
    auto fail = [&](int rc) {
        s3_topo_destroy(t);
        return rc;
    };
#define S3_TOPO_TRY(EXPR)                                                                              \
    do {                                                                                               \
        const hipError_t e_ = (EXPR);                                                                  \
        if (e_ != hipSuccess) {                                                                        \
            s3::set_error("s3_topo_create: %s failed: %s", #EXPR, hipGetErrorString(e_));              \
            return fail(e_ == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP);                              \
        }                                                                                              \
    } while (0)
    S3_TOPO_TRY(hipStreamCreateWithFlags(&t->st, hipStreamNonBlocking));
    const std::vector<NbEntry> tab = build_nb_table(dim);
    S3_TOPO_TRY(hipMalloc(reinterpret_cast<void **>(&t->d_nb_table), tab.size() * sizeof(NbEntry)));
    S3_TOPO_TRY(hipMemcpy(t->d_nb_table, tab.data(), tab.size() * sizeof(NbEntry), hipMemcpyHostToDevice));
    const size_t rule_bytes = sizeof(NodeRule) * (size_t)t->nch * t->n_rules;
    S3_TOPO_TRY(hipMalloc(reinterpret_cast<void **>(&t->d_rules), rule_bytes));
    S3_TOPO_TRY(hipMemcpy(t->d_rules, dim == 2 ? (const void *)H_RULES_2D : (const void *)H_RULES_3D, rule_bytes, hipMemcpyHostToDevice));
    double widths[128];
    for (int l = 0; l < 64; ++l) {
        widths[l] = (0.5 * width) / std::ldexp(1.0, l);            // half_width, as topology.cpp
        widths[64 + l] = (0.25 * width) / std::ldexp(1.0, l);      // quarter_width
    }
    S3_TOPO_TRY(hipMalloc(reinterpret_cast<void **>(&t->d_widths), sizeof(widths)));
    S3_TOPO_TRY(hipMemcpy(t->d_widths, widths, sizeof(widths), hipMemcpyHostToDevice));
    S3_TOPO_TRY(hipMalloc(reinterpret_cast<void **>(&t->counters), 8 * sizeof(int64_t)));
    t->nb_table = t->d_nb_table;
    t->rules = t->d_rules;
    t->half_width = t->d_widths;
    t->quarter_width = t->d_widths + 64;
    int rc = reserve_cells(t, 1);
    if (rc != S3_OK) return fail(rc);
    rc = grow<double>(t->nodes, t->node_cap, (int64_t)t->nch * dim * 1024, 0, t->st);
    if (rc != S3_OK) return fail(rc);
    t->node_cap /= dim;
    // root cell and its nodes (s_cube.py:368, 386-394: centre + dir * 0.5 * width, ids 0..2^d-1)
    std::vector<int32_t> nbrow(t->nnb, -1);
    std::vector<int64_t> ni(t->nch);
    std::vector<double> nodes((size_t)t->nch * dim);
    for (int c = 0; c < t->nch; ++c) {
        ni[c] = c;
        const int *d = dim == 2 ? H_DIR2[c] : H_DIR3[c];
        for (int j = 0; j < dim; ++j) nodes[(size_t)c * dim + j] = h_root_center[j] + d[j] * 0.5 * width;
    }
    const int32_t zero = 0, minus1 = -1, leaf = LEAF;
    const int64_t counters[8] = {t->nch, 0, INT64_MAX, -1, 0, 0, 0, 0};
    S3_TOPO_TRY(hipMemcpy(t->level, &zero, 4, hipMemcpyHostToDevice));
    S3_TOPO_TRY(hipMemcpy(t->parent, &minus1, 4, hipMemcpyHostToDevice));
    S3_TOPO_TRY(hipMemcpy(t->first_child, &leaf, 4, hipMemcpyHostToDevice));
    S3_TOPO_TRY(hipMemcpy(t->batch_pos, &minus1, 4, hipMemcpyHostToDevice));
    S3_TOPO_TRY(hipMemcpy(t->nb, nbrow.data(), 4 * (size_t)t->nnb, hipMemcpyHostToDevice));
    S3_TOPO_TRY(hipMemcpy(t->node_idx, ni.data(), 8 * (size_t)t->nch, hipMemcpyHostToDevice));
    S3_TOPO_TRY(hipMemcpy(t->center, h_root_center, 8 * (size_t)dim, hipMemcpyHostToDevice));
    S3_TOPO_TRY(hipMemcpy(t->nodes, nodes.data(), 8 * nodes.size(), hipMemcpyHostToDevice));
    S3_TOPO_TRY(hipMemcpy(t->counters, counters, sizeof(counters), hipMemcpyHostToDevice));
#undef S3_TOPO_TRY
    t->n_used = 1;
    t->n_nodes_bound = t->nch;
    *out = t;
    return S3_OK;
} catch (const std::exception &e) {
    s3::set_error("s3_topo_create: %s", e.what());
    return S3_ENOMEM;
}

// children of the listed parents, in list order (s_cube.py:879-895 / 531-544); relink != 0: the "update all nb" pass of
// the uniform loop afterwards (s_cube.py:547-549).  Asynchronous on the engine's stream; a parent that is not a leaf (or
// listed twice) raises the engine's error flag, reported by s3_topo_sync.  *first_out = id of the first new cell.
int s3_topo_refine(s3_topo *t, const int64_t *h_parents, int64_t n, int relink, int64_t *first_out) {
    S3_REQUIRE(t != nullptr && (n == 0 || h_parents != nullptr), "s3_topo_refine: null argument");
    const int64_t first = t->n_used;
    if (first_out) *first_out = first;
    if (n == 0) return S3_OK;
    S3_REQUIRE((first + n * t->nch) < ((int64_t)1 << 31), "s3_topo_refine: more than 2^31 cells");
    int rc = reserve_cells(t, n * t->nch);
    if (rc != S3_OK) return rc;
    // every parent brings at most 3^d - 2^d new nodes (the lattice points of its children that are not its own corners)
    const int64_t max_new = (t->dim == 2 ? 5 : 19) * n;
    int64_t cap = t->node_cap * t->dim;
    rc = grow<double>(t->nodes, cap, (t->n_nodes_bound + max_new) * t->dim, t->n_nodes_bound * t->dim, t->st, &t->retired);
    if (rc != S3_OK) return rc;
    t->node_cap = cap / t->dim;
    t->n_nodes_bound += max_new;
    rc = upload_ids(t, h_parents, n);
    if (rc != S3_OK) return rc;
    const TopoView v = *t;
    topo_validate_kernel<<<blocks_for(n), TB, 0, t->st>>>(v, t->ids, n, first);
    topo_build_kernel<<<blocks_for(n), TB, 0, t->st>>>(v, t->ids, n, first, t->cnt);
    S3_LAUNCH_CHECK();
    rc = exclusive_scan(t, t->cnt, t->base, n);
    if (rc != S3_OK) return rc;
    topo_number_kernel<<<blocks_for(n), TB, 0, t->st>>>(v, n, first, t->base);
    topo_resolve_kernel<<<blocks_for(n * t->nch * t->nch), TB, 0, t->st>>>(v, n, first);
    topo_finish_kernel<<<blocks_for(n), TB, 0, t->st>>>(v, t->ids, n, first, t->base, t->cnt);
    if (relink) topo_relink_batch_kernel<<<blocks_for(n * t->nch), TB, 0, t->st>>>(v, t->ids, n);
    S3_LAUNCH_CHECK();
    t->n_used = first + n * t->nch;
    return S3_OK;
}

// cell.parent.children = _assign_neighbors(cell.parent, ...) for every listed cell, in list order (s_cube.py:609, 834)
int s3_topo_relink_parent_of(s3_topo *t, const int64_t *h_cells, int64_t n) {
    S3_REQUIRE(t != nullptr && (n == 0 || h_cells != nullptr), "s3_topo_relink_parent_of: null argument");
    if (n == 0) return S3_OK;
    int rc = upload_ids(t, h_cells, n);
    if (rc != S3_OK) return rc;
    rc = grow<int32_t>(t->rows_tmp, t->rows_tmp_cap, n * t->nch * t->nnb, 0, t->st);
    if (rc != S3_OK) return rc;
    const TopoView v = *t;
    topo_set_pos_kernel<<<blocks_for(n), TB, 0, t->st>>>(v, t->ids, n, 1);
    topo_relink_compute_kernel<<<blocks_for(n), TB, 0, t->st>>>(v, t->ids, n, t->rows_tmp);
    topo_relink_write_kernel<<<blocks_for(n * t->nch * t->nnb), TB, 0, t->st>>>(v, t->ids, n, t->rows_tmp);
    topo_set_pos_kernel<<<blocks_for(n), TB, 0, t->st>>>(v, t->ids, n, 0);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

// s_cube.py:721-728: children = [], and the cell disappears from the rows of the neighbours it still lists
int s3_topo_mark_invalid(s3_topo *t, const int64_t *h_cells, int64_t n) {
    S3_REQUIRE(t != nullptr && (n == 0 || h_cells != nullptr), "s3_topo_mark_invalid: null argument");
    if (n == 0) return S3_OK;
    int rc = upload_ids(t, h_cells, n);
    if (rc != S3_OK) return rc;
    rc = grow<uint8_t>(t->flags, t->flags_cap, n * t->nnb, 0, t->st);
    if (rc != S3_OK) return rc;
    rc = grow<int32_t>(t->rows_tmp, t->rows_tmp_cap, n * t->nnb, 0, t->st);
    if (rc != S3_OK) return rc;
    const TopoView v = *t;
    topo_set_pos_kernel<<<blocks_for(n), TB, 0, t->st>>>(v, t->ids, n, 1);
    topo_copy_rows_kernel<<<blocks_for(n * t->nnb), TB, 0, t->st>>>(v, t->ids, n, t->rows_tmp);
    topo_invalid_decide_kernel<<<blocks_for(n * t->nnb), TB, 0, t->st>>>(v, t->ids, n, t->flags);
    topo_invalid_apply_kernel<<<blocks_for(n * t->nnb), TB, 0, t->st>>>(v, t->ids, n, t->flags, t->rows_tmp);
    topo_set_pos_kernel<<<blocks_for(n), TB, 0, t->st>>>(v, t->ids, n, 0);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

// waits for everything submitted; *h_error: 0 ok, 1 a listed parent was not a leaf (or listed twice)
int s3_topo_sync(s3_topo *t, int64_t *h_n_cells, int64_t *h_n_nodes, int *h_error) {
    S3_REQUIRE(t != nullptr, "s3_topo_sync: null engine");
    int64_t c[2];
    S3_HIP_CHECK(hipMemcpyAsync(c, t->counters, sizeof(c), hipMemcpyDeviceToHost, t->st));
    S3_HIP_CHECK(hipStreamSynchronize(t->st));
    for (void *p : t->retired) (void)hipFree(p);
    t->retired.clear();
    t->n_nodes_bound = c[0];
    if (h_n_cells) *h_n_cells = t->n_used;
    if (h_n_nodes) *h_n_nodes = c[0];
    if (h_error) *h_error = (int)c[1];
    return S3_OK;
}

// device pointers of the tables (valid until the next call that may grow them): 0 level i32, 1 parent i32, 2 first_child
// i32, 3 nb i32 [n][nnb], 4 node_idx i64 [n][nch], 5 center f64 [n][dim], 6 nodes f64 [n_nodes][dim]
int s3_topo_table(s3_topo *t, int which, const void **d_ptr) {
    S3_REQUIRE(t != nullptr && d_ptr != nullptr, "s3_topo_table: null argument");
    const void *p[] = {t->level, t->parent, t->first_child, t->nb, t->node_idx, t->center, t->nodes};
    S3_REQUIRE(which >= 0 && which < 7, "s3_topo_table: unknown table %d", which);
    *d_ptr = p[which];
    return S3_OK;
}

// _resort_nodes_and_indices_of_grid (s_cube.py:734-772, 1695-1736): number of leaves and of nodes that keep a slot
int s3_topo_finalize(s3_topo *t, int64_t *h_n_leaf, int64_t *h_n_unique_nodes) {
    S3_REQUIRE(t != nullptr && h_n_leaf && h_n_unique_nodes, "s3_topo_finalize: null argument");
    int64_t nn = 0;
    int err = 0;
    int rc = s3_topo_sync(t, nullptr, &nn, &err);
    if (rc != S3_OK) return rc;
    S3_REQUIRE(err == 0, "s3_topo_finalize: the engine is in an error state (a refined cell was not a leaf)");
    const int64_t nc = t->n_used;
    // scratch: leaf flags / keep flags share `cnt`/`base`-like arrays sized for max(nc, nn)
    const int64_t m = std::max(nc, nn);
    int64_t c0 = t->leaf_row_cap, c1 = t->map_cap;
    rc = grow<int64_t>(t->leaf_row, c0, m, 0, t->st);
    if (rc == S3_OK) rc = grow<int64_t>(t->map, c1, m, 0, t->st);
    if (rc == S3_OK) rc = grow<uint8_t>(t->flags, t->flags_cap, nn, 0, t->st);
    if (rc != S3_OK) return rc;
    t->leaf_row_cap = c0;
    t->map_cap = c1;
    int64_t *flag = nullptr, *scan = nullptr;               // temporaries of this call
    S3_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&flag), sizeof(int64_t) * (size_t)m));
    if (hipMalloc(reinterpret_cast<void **>(&scan), sizeof(int64_t) * (size_t)m) != hipSuccess) {
        (void)hipFree(flag);
        s3::set_error("s3_topo_finalize: out of device memory");
        return S3_ENOMEM;
    }
    auto done = [&](int code) {
        (void)hipStreamSynchronize(t->st);
        (void)hipFree(flag);
        (void)hipFree(scan);
        return code;
    };
    const int64_t reset[2] = {INT64_MAX, -1};
    if (hipMemcpyAsync(t->counters + 2, reset, sizeof(reset), hipMemcpyHostToDevice, t->st) != hipSuccess ||
        hipMemsetAsync(t->flags, 0, (size_t)nn, t->st) != hipSuccess)
        return done(S3_EHIP);
    const TopoView v = *t;
    topo_mark_used_kernel<<<blocks_for(nc), TB, 0, t->st>>>(v, nc, flag, t->flags);
    rc = exclusive_scan(t, flag, t->leaf_row, nc);
    if (rc != S3_OK) return done(rc);
    int64_t last[2] = {0, 0}, lastk[2] = {0, 0};
    if (hipMemcpyAsync(&last[0], flag + (nc - 1), 8, hipMemcpyDeviceToHost, t->st) != hipSuccess ||
        hipMemcpyAsync(&last[1], t->leaf_row + (nc - 1), 8, hipMemcpyDeviceToHost, t->st) != hipSuccess)
        return done(S3_EHIP);
    topo_keep_kernel<<<blocks_for(nn), TB, 0, t->st>>>(v, nn, t->flags, flag);
    rc = exclusive_scan(t, flag, scan, nn);
    if (rc != S3_OK) return done(rc);
    topo_map_kernel<<<blocks_for(nn), TB, 0, t->st>>>(nn, t->flags, scan, flag, t->map);
    if (hipMemcpyAsync(&lastk[0], flag + (nn - 1), 8, hipMemcpyDeviceToHost, t->st) != hipSuccess ||
        hipMemcpyAsync(&lastk[1], scan + (nn - 1), 8, hipMemcpyDeviceToHost, t->st) != hipSuccess)
        return done(S3_EHIP);
    if (hipStreamSynchronize(t->st) != hipSuccess || hipGetLastError() != hipSuccess) return done(S3_EHIP);
    t->n_leaf = last[0] + last[1];
    t->n_unique = lastk[0] + lastk[1];
    *h_n_leaf = t->n_leaf;
    *h_n_unique_nodes = t->n_unique;
    return done(S3_OK);
}

// the assembled grid after s3_topo_finalize, into device arrays of the caller: faces [n_leaf][2^d] (int32 when as32, else
// int64; leaves in ascending cell id), nodes [n_unique][dim]; returns when they are complete
int s3_topo_export_grid(s3_topo *t, void *d_faces, int as32, double *d_nodes) {
    S3_REQUIRE(t != nullptr && d_faces != nullptr && d_nodes != nullptr, "s3_topo_export_grid: null argument");
    int64_t nn = 0;
    int rc = s3_topo_sync(t, nullptr, &nn, nullptr);
    if (rc != S3_OK) return rc;
    const TopoView v = *t;
    const int64_t nc = t->n_used;
    if (as32)
        topo_faces_kernel<int32_t><<<blocks_for(nc * t->nch), TB, 0, t->st>>>(v, nc, t->leaf_row, t->map, static_cast<int32_t *>(d_faces));
    else
        topo_faces_kernel<int64_t><<<blocks_for(nc * t->nch), TB, 0, t->st>>>(v, nc, t->leaf_row, t->map, static_cast<int64_t *>(d_faces));
    topo_nodes_out_kernel<<<blocks_for(nn), TB, 0, t->st>>>(v, nn, t->map, d_nodes);
    S3_LAUNCH_CHECK();
    S3_HIP_CHECK(hipStreamSynchronize(t->st));
    return S3_OK;
}

// centres [n][dim] / levels [n] (int64) of the listed cells (the leaves in the host's set order), into device arrays
int s3_topo_gather_cells(s3_topo *t, const int64_t *h_ids, int64_t n, double *d_centers, int64_t *d_levels) {
    S3_REQUIRE(t != nullptr && (n == 0 || (h_ids && d_centers && d_levels)), "s3_topo_gather_cells: null argument");
    if (n == 0) return S3_OK;
    const int rc = upload_ids(t, h_ids, n);
    if (rc != S3_OK) return rc;
    topo_gather_cells_kernel<<<blocks_for(n), TB, 0, t->st>>>(*t, t->ids, n, d_centers, d_levels);
    S3_LAUNCH_CHECK();
    S3_HIP_CHECK(hipStreamSynchronize(t->st));
    return S3_OK;
}

}  // extern "C"
