// Planned (LDS-tiled) KNN inverse-distance interpolation: the production form of the roofline kernel.  gfx950 only.
//
// Reference behaviour: interpolate_data, export.py:446-468, driven with the cached neighbour table of
// ExportData._build_knn_cache (export.py:403-444).  Same arithmetic as s3_interp (f64 FMA in neighbour order).
//
// Why a plan: the direct kernel (export.hip) issues one row read per (cell, neighbour) -- k = 26 reads per output row
// although spatially adjacent cells share most of their neighbours.  On MI355X that redundant gather traffic is served
// by the Infinity Cache at ~7-8 TB/s and bounds the kernel.  The neighbour table is static for a whole export (it is
// computed once and reused for every snapshot batch and field), so it pays to de-duplicate it once:
//
//   plan (host, once):  cells are put in Hilbert order of their centres (radix sort); consecutive cells are packed
//                       greedily (host threads, one hash table per tile) into
//                       tiles of <= 64 cells whose neighbour sets contain <= ucap (~500) distinct source rows; per tile the
//                       distinct row ids and, per (cell, neighbour), the 16-bit position in that list are stored.
//   kernel (per batch): one workgroup (256 threads) per tile.  For every 128-byte column chunk of the row it stages the
//                       tile's distinct source rows ONCE into LDS (coalesced 128-B segments; the loads of the next two
//                       chunks are in flight in 2 x 16 registers per lane), then every thread accumulates its cell's k
//                       neighbours from LDS
//                       (2 x ds_read_b128 + 8 cvt + 8 FMA per neighbour; the tile's weights and LDS positions are
//                       staged in LDS once per tile), and writes 2 x 32 B of the f64 output row.
//
// HBM/L2 traffic per tile-chunk drops from n_cells*k segments to n_distinct segments (2.5-3x fewer on the cylinder3D
// workload), which moves the kernel from the Infinity-Cache gather bound towards the HBM bound.
#include "common.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

struct s3_interp_plan {
    int64_t nc = 0, n_src = 0, n_tiles = 0, total_rows = 0;
    int k = 0, ucap = 0, tc = 64;
    int32_t *perm = nullptr;             // [nc] processing position -> cell id
    int32_t *tile_cell_begin = nullptr;  // [n_tiles+1]
    int32_t *tile_row_begin = nullptr;   // [n_tiles+1]
    int32_t *rows = nullptr;             // [total_rows] distinct source rows, tile after tile
    uint16_t *loc = nullptr;             // [nc*k] per tile: [m][cell in tile] -> position in the tile's row list
};

namespace s3 {

constexpr int PL_SEG = 128;        // bytes of one row staged per chunk
constexpr int PL_NP = 16;          // staging passes of (tile cells)/2 rows -> at most 8 * (tile cells) distinct rows

template <typename T>
struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };
template <typename T, int TC>
__global__ void __launch_bounds__(TC * 4, 2)  // 226 VGPRs: two waves per SIMD
interp_planned_kernel(const int32_t *__restrict__ perm, const int32_t *__restrict__ tile_cell_begin,
                      const int32_t *__restrict__ tile_row_begin, const int32_t *__restrict__ rows,
                      const uint16_t *__restrict__ loc, const double *__restrict__ w, int k, int ucap,
                      const T *__restrict__ data, int64_t row_len, int64_t in_stride, double *__restrict__ out,
                      int64_t n_tiles, int64_t tiles_per_xcd, int chunks_per_block, int n_chunks) {
    using V = typename Vec16<T>::type;
    constexpr int EPV = Vec16<T>::N;                 // elements per 16-byte vector
    constexpr int EPC = PL_SEG / (int)sizeof(T);     // elements per chunk
    constexpr int BLOCK = TC * 4;                    // 4 lanes per cell
    constexpr int RPP = BLOCK / 8;                   // rows staged per pass (8 lanes per 128-B segment)
    extern __shared__ float4 lds_raw[];
    V *s_data = reinterpret_cast<V *>(lds_raw);                                  // [ucap][8] 16-byte vectors
    double *s_w = reinterpret_cast<double *>(lds_raw + (size_t)ucap * 8);        // [k][TC]
    uint16_t *s_loc = reinterpret_cast<uint16_t *>(s_w + (size_t)k * TC);        // [k][TC]

    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);     // XCD-aware (speed only)
    if (tile >= n_tiles) return;
    const int c_begin = tile_cell_begin[tile], n_c = tile_cell_begin[tile + 1] - c_begin;
    const int r_begin = tile_row_begin[tile], n_r = tile_row_begin[tile + 1] - r_begin;

    // the tile's weights and LDS row positions, [neighbour][cell] so that a wavefront reads consecutive words
    for (int i = threadIdx.x; i < k * TC; i += BLOCK) {
        const int m = i / TC, j = i - m * TC;
        const bool ok = j < n_c;
        s_w[i] = ok ? w[(int64_t)perm[c_begin + j] * k + m] : 0.0;
        s_loc[i] = ok ? loc[(int64_t)c_begin * k + (int64_t)m * n_c + j] : (uint16_t)0;
    }

    // this thread's cell: 4 lanes per cell, each lane two 16-byte vectors of the chunk (v0 and v0+4)
    const int cl = threadIdx.x >> 2, v0 = threadIdx.x & 3;
    const bool has_cell = cl < n_c;
    const int64_t cell = has_cell ? perm[c_begin + cl] : 0;

    const int chunk0 = blockIdx.y * chunks_per_block;
    const int chunk1 = min(n_chunks, chunk0 + chunks_per_block);

    // staging role: 8 lanes per 128-B row segment, 32 rows per pass, <= PL_NP passes.  The row ids of this lane's passes
    // are loaded once per tile; the segment loads of chunk c+1 are issued (into registers) before chunk c is accumulated
    // from LDS, so HBM latency overlaps the LDS/FMA phase.
    const int srow = threadIdx.x >> 3, svec = threadIdx.x & 7;
    // 2 x sixteen named registers per lane (sets A and B) instead of arrays: a loop-carried local array ends up in
    // scratch memory.  Two chunks are kept in flight: while chunk c is accumulated from LDS, the segments of chunk c+1
    // (other set) and c+2 (the set just emptied into LDS) are on their way.
#define S3_REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
    static_assert(PL_NP == 16, "S3_REP16 expands PL_NP staging passes");
#define S3_DECL(P) const int64_t rbase##P = (int64_t)rows[r_begin + min(P * RPP + srow, n_r - 1)] * in_stride; V preA##P, preB##P;
    S3_REP16(S3_DECL)
    // (row ids are clamped to the tile's last row and the column to the row: every load is in bounds, unconditional)
#define S3_LOAD_A(P) preA##P = *reinterpret_cast<const V *>(seg_ + rbase##P);
#define S3_LOAD_B(P) preB##P = *reinterpret_cast<const V *>(seg_ + rbase##P);
#define S3_ISSUE(SET, CH)                                            \
    do {                                                             \
        const int64_t c0_ = (int64_t)(CH) * EPC;                     \
        const bool ok_ = c0_ + (int64_t)(svec + 1) * EPV <= row_len; \
        const T *seg_ = data + c0_ + (ok_ ? svec : 0) * EPV;         \
        S3_REP16(S3_LOAD_##SET)                                      \
    } while (0)
#define S3_STORE_A(P) if (P * RPP + srow < n_r) s_data[(P * RPP + srow) * 8 + svec] = preA##P;
#define S3_STORE_B(P) if (P * RPP + srow < n_r) s_data[(P * RPP + srow) * 8 + svec] = preB##P;

    // accumulate the k neighbours of this thread's cell from LDS and write its 2 x 32 B of the output row
    auto accumulate = [&](int chunk) {
        if (!has_cell) return;
        const int64_t col0 = (int64_t)chunk * EPC;
        double acc0[EPV], acc1[EPV];
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc0[i] = acc1[i] = 0.0;
#pragma unroll 4
        for (int m = 0; m < k; ++m) {
            const int pos = s_loc[m * TC + cl];
            const double wm = s_w[m * TC + cl];
            const V a = s_data[pos * 8 + v0];
            const V c = s_data[pos * 8 + v0 + 4];
            const T *ae = reinterpret_cast<const T *>(&a);
            const T *ce = reinterpret_cast<const T *>(&c);
#pragma unroll
            for (int i = 0; i < EPV; ++i) {
                acc0[i] = fma(wm, (double)ae[i], acc0[i]);
                acc1[i] = fma(wm, (double)ce[i], acc1[i]);
            }
        }
        double *o = out + cell * row_len + col0;
        if (col0 + (int64_t)(v0 + 1) * EPV <= row_len) {
#pragma unroll
            for (int i = 0; i < EPV; i += 2)
                *reinterpret_cast<double2 *>(o + v0 * EPV + i) = make_double2(acc0[i], acc0[i + 1]);
        }
        if (col0 + (int64_t)(v0 + 5) * EPV <= row_len) {
#pragma unroll
            for (int i = 0; i < EPV; i += 2)
                *reinterpret_cast<double2 *>(o + (v0 + 4) * EPV + i) = make_double2(acc1[i], acc1[i + 1]);
        }
    };

    if (chunk0 < chunk1) S3_ISSUE(A, chunk0);
    if (chunk0 + 1 < chunk1) S3_ISSUE(B, chunk0 + 1);
    for (int chunk = chunk0; chunk < chunk1; chunk += 2) {
        S3_REP16(S3_STORE_A)
        __syncthreads();
        if (chunk + 2 < chunk1) S3_ISSUE(A, chunk + 2);
        accumulate(chunk);
        __syncthreads();
        if (chunk + 1 >= chunk1) break;
        S3_REP16(S3_STORE_B)
        __syncthreads();
        if (chunk + 3 < chunk1) S3_ISSUE(B, chunk + 3);
        accumulate(chunk + 1);
        __syncthreads();
    }
}

#undef S3_ISSUE
#undef S3_LOAD_A
#undef S3_LOAD_B
#undef S3_STORE_A
#undef S3_STORE_B
#undef S3_DECL
#undef S3_REP16

// distinct rows a tile of `tc` cells may hold: what is left of the LDS budget (80 KiB -> two 256-thread workgroups per CU
// for tc = 64; 160 KiB -> one 512-thread workgroup per CU for tc = 128) after the tile's weights and positions
static int plan_ucap(int k, int tc) {
    const int budget = (tc == 128 ? 160 : 80) * 1024;
    int u = (budget - k * tc * (int)(sizeof(double) + sizeof(uint16_t))) / PL_SEG;
    u = (u / 16) * 16;
    const int cap = PL_NP * tc / 2;
    return u > cap ? cap : u;
}

// Hilbert index of a quantised point (dim axes, b bits each; Skilling's transpose algorithm).  Consecutive cells of the
// curve are always face neighbours, so runs of the curve make more compact tiles than Z-order runs (3 % fewer staged
// rows on the cylinder3D workload) and consecutive tiles always touch.
static inline uint64_t hilbert_key(const uint32_t *q, int dim, int b) {
    uint32_t X[3] = {q[0], q[1], dim == 3 ? q[2] : 0};
    const uint32_t M = 1u << (b - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        for (int i = 0; i < dim; ++i) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const uint32_t t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    for (int i = 1; i < dim; ++i) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[dim - 1] & Q) t ^= Q - 1;
    for (int i = 0; i < dim; ++i) X[i] ^= t;
    uint64_t h = 0;
    for (int bit = b - 1; bit >= 0; --bit)
        for (int i = 0; i < dim; ++i) h = (h << 1) | ((X[i] >> bit) & 1u);
    return h;
}

}  // namespace s3

using namespace s3;

template <typename T>
static int launch_planned(const s3_interp_plan *p, const double *w, const void *data, int64_t row_len,
                          int64_t in_stride, double *out, hipStream_t st) {
    constexpr int EPC = PL_SEG / (int)sizeof(T);
    const int n_chunks = (int)((row_len + EPC - 1) / EPC);
    const int64_t tiles_per_xcd = (p->n_tiles + 7) / 8;
    const int64_t gx = tiles_per_xcd * 8;
    // the column chunks are split over blockIdx.y (x = tile runs fastest in dispatch order) only when there are too few
    // tiles to fill the chip: one workgroup per tile over ALL chunks is fastest (MI355X, cylinder3D workload: 3.7 ms vs
    // 4.4 ms with 4 chunks per workgroup)
    int gy = 1;
    while (gx * gy < 2048 && gy < n_chunks) gy *= 2;
    if (gy > n_chunks) gy = n_chunks;
    if (gy < 1) gy = 1;
    const int chunks_per_block = (n_chunks + gy - 1) / gy;
    gy = (n_chunks + chunks_per_block - 1) / chunks_per_block;
    S3_REQUIRE(gx < ((int64_t)1 << 31), "s3_interp_planned: too many tiles");
    const size_t lds = (size_t)p->ucap * PL_SEG + (size_t)p->k * p->tc * (sizeof(double) + sizeof(uint16_t));
    dim3 grid((unsigned)gx, (unsigned)gy);
    if (p->tc == 128) {
        auto kern = interp_planned_kernel<T, 128>;
        S3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, 512, lds, st>>>(p->perm, p->tile_cell_begin, p->tile_row_begin, p->rows, p->loc, w, p->k, p->ucap,
                                     static_cast<const T *>(data), row_len, in_stride, out, p->n_tiles, tiles_per_xcd,
                                     chunks_per_block, n_chunks);
    } else {
        auto kern = interp_planned_kernel<T, 64>;
        S3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, 256, lds, st>>>(p->perm, p->tile_cell_begin, p->tile_row_begin, p->rows, p->loc, w, p->k, p->ucap,
                                     static_cast<const T *>(data), row_len, in_stride, out, p->n_tiles, tiles_per_xcd,
                                     chunks_per_block, n_chunks);
    }
    S3_LAUNCH_CHECK();
    return S3_OK;
}

extern "C" {

void s3_interp_plan_destroy(s3_interp_plan *p) {
    if (!p) return;
    if (p->perm) (void)hipFree(p->perm);
    if (p->tile_cell_begin) (void)hipFree(p->tile_cell_begin);
    if (p->tile_row_begin) (void)hipFree(p->tile_row_begin);
    if (p->rows) (void)hipFree(p->rows);
    if (p->loc) (void)hipFree(p->loc);
    delete p;
}

int s3_interp_plan_create(const int32_t *d_idx, int64_t nc, int k, int64_t n_src, const double *d_centers, int dim,
                          int tile_cells, s3_stream stream, s3_interp_plan **out) try {
    S3_REQUIRE(out != nullptr, "s3_interp_plan_create: null output");
    *out = nullptr;
    S3_REQUIRE(nc >= 1 && nc < ((int64_t)1 << 31) && n_src >= 1 && n_src < ((int64_t)1 << 31),
               "s3_interp_plan_create: bad sizes nc=%lld n_src=%lld", (long long)nc, (long long)n_src);
    S3_REQUIRE(k >= 1 && k <= S3_MAX_K, "s3_interp_plan_create: k=%d outside [1,%d]", k, S3_MAX_K);
    S3_REQUIRE(d_idx != nullptr, "s3_interp_plan_create: null index table");
    if (tile_cells == 0) tile_cells = 64;
    S3_REQUIRE(tile_cells == 64 || tile_cells == 128, "s3_interp_plan_create: tile_cells must be 64 or 128");
    const int PL_TC = tile_cells;
    S3_REQUIRE(d_centers == nullptr || dim == 2 || dim == 3, "s3_interp_plan_create: dim must be 2 or 3");
    S3_REQUIRE(nc * (int64_t)k < ((int64_t)1 << 31), "s3_interp_plan_create: table too large");
    hipStream_t st = as_stream(stream);

    std::vector<int32_t> idx((size_t)nc * k);
    S3_HIP_CHECK(hipMemcpyAsync(idx.data(), d_idx, sizeof(int32_t) * idx.size(), hipMemcpyDeviceToHost, st));
    std::vector<double> ctr;
    if (d_centers) {
        ctr.resize((size_t)nc * dim);
        S3_HIP_CHECK(hipMemcpyAsync(ctr.data(), d_centers, sizeof(double) * ctr.size(), hipMemcpyDeviceToHost, st));
    }
    S3_HIP_CHECK(hipStreamSynchronize(st));
    for (int32_t v : idx)
        S3_REQUIRE(v >= 0 && v < n_src, "s3_interp_plan_create: neighbour index %d outside [0, %lld)", v, (long long)n_src);

    // processing order: Hilbert order of the cell centres (spatially adjacent cells share neighbours); LSD radix sort
    // of (key, cell) pairs, stable, so equal keys keep the caller's order
    std::vector<int32_t> perm(nc);
    std::iota(perm.begin(), perm.end(), 0);
    if (d_centers) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int64_t c = 0; c < nc; ++c)
            for (int j = 0; j < dim; ++j) {
                lo[j] = std::min(lo[j], ctr[c * dim + j]);
                hi[j] = std::max(hi[j], ctr[c * dim + j]);
            }
        double ext = 0;
        for (int j = 0; j < dim; ++j) ext = std::max(ext, hi[j] - lo[j]);
        const int bits = dim == 3 ? 16 : 24;                // 48-bit keys: three 16-bit sorting passes
        const double scale = ext > 0 ? ((double)((1u << bits) - 1) / ext) : 0.0;
        std::vector<uint64_t> key(nc), key2(nc);
        std::vector<int32_t> perm2(nc);
        {
            const int kt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 32, nc / 16384 + 1}));
            auto keys = [&](int t) {
                for (int64_t c = nc * t / kt; c < nc * (t + 1) / kt; ++c) {
                    uint32_t q[3] = {0, 0, 0};
                    for (int j = 0; j < dim; ++j) q[j] = (uint32_t)((ctr[c * dim + j] - lo[j]) * scale);
                    key[c] = hilbert_key(q, dim, bits);
                }
            };
            std::vector<std::thread> workers;
            for (int t = 1; t < kt; ++t) workers.emplace_back(keys, t);
            keys(0);
            for (auto &w : workers) w.join();
        }
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = 16 * pass;
            std::vector<int64_t> hist(65537, 0);
            for (int64_t c = 0; c < nc; ++c) ++hist[((key[c] >> shift) & 0xffff) + 1];
            for (int v = 0; v < 65536; ++v) hist[v + 1] += hist[v];
            for (int64_t c = 0; c < nc; ++c) {
                const int64_t dst = hist[(key[c] >> shift) & 0xffff]++;
                key2[dst] = key[c];
                perm2[dst] = perm[c];
            }
            key.swap(key2);
            perm.swap(perm2);
        }
    }

    // greedy packing into tiles: consecutive cells join a tile while it has room (<= tile_cells cells, <= ucap distinct
    // rows).  The cell sequence is cut into independent chunks packed by separate host threads; each tile keeps its
    // distinct rows in a small open-addressing table, so no O(n_src) scratch is needed.
    const int ucap = plan_ucap(k, PL_TC);
    struct Chunk {
        std::vector<int32_t> cell_end, row_end, rows;      // per tile: end position (in perm) / end of its row list
    };
    const int n_threads = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 32, nc / 4096 + 1}));
    std::vector<Chunk> chunks(n_threads);
    std::vector<uint16_t> loc((size_t)nc * k);
    auto pack = [&](int t) {
        const int64_t pos0 = nc * t / n_threads, pos1 = nc * (t + 1) / n_threads;
        Chunk &ch = chunks[t];
        constexpr int HS = 2048;                            // table slots (power of two, > 2 * max ucap is not needed: ucap <= 1024)
        std::vector<int32_t> hkey(HS, -1), used;
        std::vector<uint16_t> hval(HS);
        used.reserve(1100);
        auto find = [&](int32_t r) -> int {                 // slot of r, or the empty slot where it would go
            uint32_t s = ((uint32_t)r * 2654435761u) >> 21;  // 11 bits
            while (hkey[s] != -1 && hkey[s] != r) s = (s + 1) & (HS - 1);
            return (int)s;
        };
        int64_t tile_pos0 = pos0;
        int rows_in_tile = 0;
        auto close_tile = [&](int64_t pos_end) {
            // positions of the tile's (cell, neighbour) pairs, stored [m][cell]
            const int32_t n_c = (int32_t)(pos_end - tile_pos0);
            for (int32_t j = 0; j < n_c; ++j) {
                const int32_t cell = perm[tile_pos0 + j];
                for (int m = 0; m < k; ++m)
                    loc[(size_t)tile_pos0 * k + (size_t)m * n_c + j] = hval[find(idx[(size_t)cell * k + m])];
            }
            ch.cell_end.push_back((int32_t)pos_end);
            ch.row_end.push_back((int32_t)ch.rows.size());
            for (int32_t s : used) hkey[s] = -1;
            used.clear();
            rows_in_tile = 0;
            tile_pos0 = pos_end;
        };
        int32_t fresh_ids[S3_MAX_K];
        for (int64_t pos = pos0; pos < pos1; ++pos) {
            const int32_t *ci = &idx[(size_t)perm[pos] * k];
            int fresh = 0;                                   // rows this cell would add (repeats inside a row count once)
            for (int m = 0; m < k; ++m) {
                const int32_t r = ci[m];
                if (hkey[find(r)] == r) continue;
                bool dup = false;
                for (int u = 0; u < fresh; ++u) dup |= fresh_ids[u] == r;
                if (!dup) fresh_ids[fresh++] = r;
            }
            if (pos - tile_pos0 == PL_TC || rows_in_tile + fresh > ucap) {
                close_tile(pos);
                fresh = 0;                                   // recount against the empty table
                for (int m = 0; m < k; ++m) {
                    bool dup = false;
                    for (int u = 0; u < fresh; ++u) dup |= fresh_ids[u] == ci[m];
                    if (!dup) fresh_ids[fresh++] = ci[m];
                }
            }
            for (int u = 0; u < fresh; ++u) {
                const int s_ = find(fresh_ids[u]);
                hkey[s_] = fresh_ids[u];
                hval[s_] = (uint16_t)rows_in_tile++;
                used.push_back(s_);
                ch.rows.push_back(fresh_ids[u]);
            }
        }
        if (pos1 > tile_pos0) close_tile(pos1);
    };
    {
        std::vector<std::thread> workers;
        for (int t = 1; t < n_threads; ++t) workers.emplace_back(pack, t);
        pack(0);
        for (auto &w : workers) w.join();
    }
    std::vector<int32_t> tile_cell_begin{0}, tile_row_begin{0}, rows;
    {
        size_t total = 0;
        for (const Chunk &ch : chunks) total += ch.rows.size();
        S3_REQUIRE(total < ((size_t)1 << 31), "s3_interp_plan_create: row lists too large");
        rows.reserve(total);
        for (const Chunk &ch : chunks) {
            const int32_t row_base = (int32_t)rows.size();
            for (size_t i = 0; i < ch.cell_end.size(); ++i) {
                tile_cell_begin.push_back(ch.cell_end[i]);
                tile_row_begin.push_back(row_base + ch.row_end[i]);
            }
            rows.insert(rows.end(), ch.rows.begin(), ch.rows.end());
        }
    }
    const int32_t tile = (int32_t)tile_cell_begin.size() - 1;

    s3_interp_plan *p = new s3_interp_plan();
    p->nc = nc; p->k = k; p->ucap = ucap; p->tc = PL_TC; p->n_src = n_src; p->n_tiles = tile; p->total_rows = (int64_t)rows.size();
    auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 4);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st);
    };
    hipError_t e = upload((void **)&p->perm, perm.data(), sizeof(int32_t) * perm.size());
    if (e == hipSuccess) e = upload((void **)&p->tile_cell_begin, tile_cell_begin.data(), sizeof(int32_t) * tile_cell_begin.size());
    if (e == hipSuccess) e = upload((void **)&p->tile_row_begin, tile_row_begin.data(), sizeof(int32_t) * tile_row_begin.size());
    if (e == hipSuccess) e = upload((void **)&p->rows, rows.data(), sizeof(int32_t) * rows.size());
    if (e == hipSuccess) e = upload((void **)&p->loc, loc.data(), sizeof(uint16_t) * loc.size());
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        s3::set_error("s3_interp_plan_create: %s", hipGetErrorString(e));
        s3_interp_plan_destroy(p);
        return e == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP;
    }
    *out = p;
    return S3_OK;
} catch (const std::exception &e) {      // host tables of the builder (std::bad_alloc, std::system_error of a thread)
    s3::set_error("s3_interp_plan_create: %s", e.what());
    return S3_ENOMEM;
}

int s3_interp_plan_info(const s3_interp_plan *p, int64_t *n_tiles, int64_t *total_rows) {
    S3_REQUIRE(p != nullptr, "s3_interp_plan_info: null plan");
    if (n_tiles) *n_tiles = p->n_tiles;
    if (total_rows) *total_rows = p->total_rows;
    return S3_OK;
}

int s3_interp_planned(const s3_interp_plan *p, const double *d_w, const void *d_data, int dtype, int64_t row_len,
                      int64_t in_stride, double *d_out, s3_stream stream) {
    S3_REQUIRE(p != nullptr, "s3_interp_planned: null plan");
    S3_REQUIRE(dtype == S3_DTYPE_F32 || dtype == S3_DTYPE_F64, "s3_interp_planned: unknown dtype %d", dtype);
    S3_REQUIRE(row_len >= 0, "s3_interp_planned: bad row_len");
    if (row_len == 0) return S3_OK;
    S3_REQUIRE(d_w && d_data && d_out, "s3_interp_planned: null array");
    const int epv = dtype == S3_DTYPE_F32 ? 4 : 2;
    const uintptr_t a_in = reinterpret_cast<uintptr_t>(d_data), a_out = reinterpret_cast<uintptr_t>(d_out);
    if (in_stride <= 0) in_stride = row_len;
    S3_REQUIRE(in_stride >= row_len, "s3_interp_planned: in_stride %lld < row_len %lld", (long long)in_stride, (long long)row_len);
    S3_REQUIRE(row_len % epv == 0 && in_stride % epv == 0 && a_in % 16 == 0 && a_out % 16 == 0,
               "s3_interp_planned: rows must be 16-byte aligned (row_len %% %d == 0); use s3_interp for ragged rows", epv);
    if (dtype == S3_DTYPE_F32) return launch_planned<float>(p, d_w, d_data, row_len, in_stride, d_out, as_stream(stream));
    return launch_planned<double>(p, d_w, d_data, row_len, in_stride, d_out, as_stream(stream));
}

}  // extern "C"
