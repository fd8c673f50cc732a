// Device-side construction of an interpolation plan (the tables interp_planned_kernel consumes).  gfx950 only.
//
// The neighbour table never leaves HBM:
//   1. validate the indices, bounding box of the cell centres, Hilbert key per cell, radix sort (scan_sort.h: hand-written) -> processing
//      order `perm` (stable: equal keys keep the caller's order; without centres the caller's order is kept);
//   2. the ordered cells are cut into blocks of 8 * `tc` consecutive cells, one wavefront per block packs its cells
//      greedily into tiles (a tile is closed when it holds `tc` cells or the next cell would push it past `ucap` distinct
//      source rows): lane m owns neighbour m of the current cell, the tile's distinct rows live in an LDS hash table
//      (open addressing, compare-and-swap insertion), positions are handed out by ballot rank;
//   3. exclusive scans (scan_sort.h) of the per-block tile / row counts, then a gather kernel writes the compact tables.
#include "common.h"
#include "plan_build.h"

#include "scan_sort.h"

#include <cmath>
#include <vector>

namespace s3 {

namespace {

constexpr int HT_SLOTS = 2048;          // > 2 * max ucap (1024): load factor <= 0.5
constexpr int BLOCK_TILES = 8;          // cells per packing block = BLOCK_TILES * tc (a tile never spans two blocks)
constexpr int32_t HT_EMPTY = -1;

__global__ void validate_kernel(const int32_t *__restrict__ idx, int64_t n, int32_t n_src, int32_t *__restrict__ bad) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n && (idx[i] < 0 || idx[i] >= n_src)) atomicExch(bad, 1);
}

__global__ void __launch_bounds__(256)
plan_bbox_kernel(const double *__restrict__ c, int64_t n, int dim, double *__restrict__ partial) {
    __shared__ double smin[3][256], smax[3][256];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int j = 0; j < dim; ++j) {
            const double v = c[i * dim + j];
            mn[j] = fmin(mn[j], v);
            mx[j] = fmax(mx[j], v);
        }
    for (int j = 0; j < 3; ++j) {
        smin[j][threadIdx.x] = mn[j];
        smax[j][threadIdx.x] = mx[j];
    }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int j = 0; j < 3; ++j) {
                smin[j][threadIdx.x] = fmin(smin[j][threadIdx.x], smin[j][threadIdx.x + s]);
                smax[j][threadIdx.x] = fmax(smax[j][threadIdx.x], smax[j][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int j = 0; j < 3; ++j) {
            partial[blockIdx.x * 6 + j] = smin[j][0];
            partial[blockIdx.x * 6 + 3 + j] = smax[j][0];
        }
}

// Hilbert index of a quantised point (dim axes, b bits each; Skilling's transpose algorithm).  Consecutive cells of the
// curve are always face neighbours, so runs of the curve make compact tiles and consecutive tiles always touch.
__device__ __forceinline__ uint64_t hilbert_key(const uint32_t (&q)[3], int dim, int b) {
    uint32_t X[3] = {q[0], q[1], dim == 3 ? q[2] : 0u};
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

struct KeyParams { double lo[3], scale; int dim, bits; };

__global__ void key_kernel(const double *__restrict__ c, int64_t n, KeyParams kp, uint64_t *__restrict__ key,
                           int32_t *__restrict__ val) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3] = {0, 0, 0};
    for (int j = 0; j < kp.dim; ++j) q[j] = (uint32_t)((c[i * kp.dim + j] - kp.lo[j]) * kp.scale);
    key[i] = hilbert_key(q, kp.dim, kp.bits);
    val[i] = (int32_t)i;
}

__global__ void iota_kernel(int32_t *__restrict__ v, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

// one wavefront per block of `cb` consecutive cells: greedy packing into tiles of <= tc cells.
//   blk_rows   [n_blocks][cb*k]  distinct rows of the block's tiles, tile after tile
//   blk_tiles  [n_blocks][cb][2] per tile of the block: cells, rows
//   cnt_tiles / cnt_rows [n_blocks]  tiles, rows of the block
//   loc        final layout: per tile [m][cell in tile], at (first position of the tile) * k
__global__ void __launch_bounds__(64)
pack_kernel(const int32_t *__restrict__ idx, const int32_t *__restrict__ perm, int64_t nc, int k, int tc, int cb, int ucap,
            int32_t *__restrict__ blk_rows, int32_t *__restrict__ blk_tiles, int32_t *__restrict__ cnt_tiles,
            int32_t *__restrict__ cnt_rows, uint16_t *__restrict__ loc) {
    extern __shared__ int32_t lds[];
    int32_t *ht_key = lds;                                                  // [HT_SLOTS]
    uint16_t *ht_val = reinterpret_cast<uint16_t *>(ht_key + HT_SLOTS);      // [HT_SLOTS]
    uint16_t *pos_buf = ht_val + HT_SLOTS;                                   // [tc][k] positions of the open tile's cells
    const int lane = threadIdx.x;
    const int64_t blk = blockIdx.x;
    const int64_t p0 = blk * cb;
    const int n_cells = (int)min((int64_t)cb, nc - p0);
    int32_t *rows_out = blk_rows + blk * (int64_t)cb * k;

    for (int s = lane; s < HT_SLOTS; s += 64) ht_key[s] = HT_EMPTY;
    __syncthreads();

    int tile_first = 0;        // first cell (within the block) of the open tile
    int rows_in_tile = 0;      // distinct rows of the open tile
    int rows_done = 0;         // rows of the closed tiles of this block
    int n_tiles = 0;

    auto close_tile = [&](int end_cell) {
        const int n_c = end_cell - tile_first;
        // positions of the tile's (cell, neighbour) pairs in the kernel's layout [m][cell]
        for (int i = lane; i < n_c * k; i += 64) {
            const int m = i / n_c, j = i - m * n_c;
            loc[(p0 + tile_first) * k + i] = pos_buf[j * k + m];
        }
        if (lane == 0) {
            blk_tiles[(blk * cb + n_tiles) * 2 + 0] = n_c;
            blk_tiles[(blk * cb + n_tiles) * 2 + 1] = rows_in_tile;
        }
        __syncthreads();
        for (int s = lane; s < HT_SLOTS; s += 64) ht_key[s] = HT_EMPTY;
        __syncthreads();
        rows_done += rows_in_tile;
        rows_in_tile = 0;
        tile_first = end_cell;
        ++n_tiles;
    };

    for (int cell = 0; cell < n_cells; ++cell) {
        const bool active = lane < k;
        const int32_t r = active ? idx[(int64_t)perm[p0 + cell] * k + lane] : 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            // insert this cell's neighbours; `won` = this lane put a new row into the table
            bool won = false;
            uint32_t s = ((uint32_t)r * 2654435761u) >> 21;
            if (active) {
                while (true) {
                    const int32_t seen = atomicCAS(&ht_key[s], HT_EMPTY, r);
                    if (seen == HT_EMPTY) { won = true; break; }
                    if (seen == r) break;
                    s = (s + 1) & (HT_SLOTS - 1);
                }
            }
            const uint64_t winners = __ballot(won);
            const int fresh = __popcll(winners);
            const bool overflow = (cell - tile_first == tc) || (rows_in_tile + fresh > ucap);
            if (overflow && attempt == 0 && cell > tile_first) {
                close_tile(cell);                                           // empties the table, this step's insertions included
                continue;                                                   // second attempt: first cell of a fresh tile
            }
            if (won) {
                const int pos = rows_in_tile + __popcll(winners & ((1ull << lane) - 1ull));
                ht_val[s] = (uint16_t)pos;
                rows_out[rows_done + pos] = r;
            }
            rows_in_tile += fresh;
            __syncthreads();
            // every lane stopped on the slot of its row: its own insertion, an equal value inserted by another lane of
            // this step, or a row the tile held already
            if (active) pos_buf[(cell - tile_first) * k + lane] = ht_val[s];
            break;
        }
    }
    __syncthreads();
    if (n_cells > tile_first) close_tile(n_cells);
    if (lane == 0) {
        cnt_tiles[blk] = n_tiles;
        cnt_rows[blk] = rows_done;
    }
}

// compact tables from the per-block results: tile_cell_begin / tile_row_begin (exclusive prefix form) and rows
__global__ void __launch_bounds__(64)
gather_kernel(const int32_t *__restrict__ blk_rows, const int32_t *__restrict__ blk_tiles, const int32_t *__restrict__ cnt_tiles,
              const int32_t *__restrict__ cnt_rows,
              const int32_t *__restrict__ tile_off, const int32_t *__restrict__ row_off, int cb, int k, int64_t n_blocks,
              int32_t *__restrict__ tile_cell_begin, int32_t *__restrict__ tile_row_begin, int32_t *__restrict__ rows) {
    const int64_t blk = blockIdx.x;
    const int lane = threadIdx.x;
    const int nt = cnt_tiles[blk], nr = cnt_rows[blk];
    const int32_t t0 = tile_off[blk], r0 = row_off[blk];
    for (int i = lane; i < nr; i += 64) rows[r0 + i] = blk_rows[blk * (int64_t)cb * k + i];
    if (lane == 0) {
        int32_t c = (int32_t)(blk * cb), r = r0;
        for (int t = 0; t < nt; ++t) {
            tile_cell_begin[t0 + t] = c;
            tile_row_begin[t0 + t] = r;
            c += blk_tiles[(blk * cb + t) * 2 + 0];
            r += blk_tiles[(blk * cb + t) * 2 + 1];
        }
        if (blk == n_blocks - 1) {
            tile_cell_begin[t0 + nt] = c;
            tile_row_begin[t0 + nt] = r;
        }
    }
}

}  // namespace

struct Scratch {                           // frees whatever was allocated when it goes out of scope
    std::vector<void *> ptrs;
    ~Scratch() { for (void *p : ptrs) (void)hipFree(p); }
    template <typename T> hipError_t alloc(T **p, size_t n) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(p), sizeof(T) * (n ? n : 1));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

#define S3_PB_CHECK(expr)                                                                             \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            s3::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP;                                   \
        }                                                                                             \
    } while (0)

// Hilbert-curve order of n points (dim 2 | 3): bounding box, 48-bit keys, stable radix sort -> perm[position] = point
static int spatial_perm(const double *d_points, int64_t n, int dim, hipStream_t st, int32_t *d_perm, Scratch &tmp) {
    const int nb = 256;
    double *d_part = nullptr;
    S3_PB_CHECK(tmp.alloc(&d_part, (size_t)nb * 6));
    plan_bbox_kernel<<<nb, 256, 0, st>>>(d_points, n, dim, d_part);
    S3_PB_CHECK(hipGetLastError());
    std::vector<double> part((size_t)nb * 6);
    S3_PB_CHECK(hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st));
    S3_PB_CHECK(hipStreamSynchronize(st));
    KeyParams kp{};
    double hi[3] = {-1e300, -1e300, -1e300};
    for (int j = 0; j < 3; ++j) kp.lo[j] = 1e300;
    for (int b = 0; b < nb; ++b)
        for (int j = 0; j < dim; ++j) {
            kp.lo[j] = std::fmin(kp.lo[j], part[(size_t)b * 6 + j]);
            hi[j] = std::fmax(hi[j], part[(size_t)b * 6 + 3 + j]);
        }
    double ext = 0;
    for (int j = 0; j < dim; ++j) ext = std::fmax(ext, hi[j] - kp.lo[j]);
    kp.dim = dim;
    kp.bits = dim == 3 ? 16 : 24;                       // 48-bit keys
    kp.scale = ext > 0 && std::isfinite(ext) ? ((double)((1u << kp.bits) - 1) / ext) : 0.0;
    uint64_t *key_in = nullptr, *key_out = nullptr;
    int32_t *val_in = nullptr;
    S3_PB_CHECK(tmp.alloc(&key_in, (size_t)n));
    S3_PB_CHECK(tmp.alloc(&key_out, (size_t)n));
    S3_PB_CHECK(tmp.alloc(&val_in, (size_t)n));
    key_kernel<<<grid_for(n, 256), 256, 0, st>>>(d_points, n, kp, key_in, val_in);
    S3_PB_CHECK(hipGetLastError());
    // stable LSD radix sort of (key, position) pairs (csrc/scan_sort.h); 48-bit keys: six passes, the result ends in the
    // buffers it started from
    int32_t *d_hist = nullptr;
    const size_t hist_items = sort_hist_items(n);
    S3_PB_CHECK(tmp.alloc(&d_hist, hist_items + scan_tmp_items((int64_t)hist_items)));
    bool in_alt = false;
    S3_PB_CHECK(radix_sort_pairs(key_in, key_out, val_in, d_perm, n, dim * kp.bits, d_hist, st, &in_alt));
    if (!in_alt) S3_PB_CHECK(hipMemcpyAsync(d_perm, val_in, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    return S3_OK;
}

int build_plan_tables(const int32_t *d_idx, int64_t nc, int k, int64_t n_src, const double *d_centers, int dim, int tc,
                      int ucap, hipStream_t st, PlanTables *out) {
    Scratch tmp;
    // ---- 1. validation + processing order ----------------------------------------------------------------------
    int32_t *d_bad = nullptr;
    S3_PB_CHECK(tmp.alloc(&d_bad, 1));
    S3_PB_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int32_t), st));
    validate_kernel<<<grid_for(nc * k, 256), 256, 0, st>>>(d_idx, nc * k, (int32_t)n_src, d_bad);
    S3_PB_CHECK(hipGetLastError());

    S3_PB_CHECK(hipMalloc(reinterpret_cast<void **>(&out->perm), sizeof(int32_t) * nc));
    if (d_centers) {
        const int rc = spatial_perm(d_centers, nc, dim, st, out->perm, tmp);
        if (rc != S3_OK) return rc;
    } else {
        iota_kernel<<<grid_for(nc, 256), 256, 0, st>>>(out->perm, nc);
        S3_PB_CHECK(hipGetLastError());
    }

    // ---- 2. per-block greedy packing ---------------------------------------------------------------------------
    const int cb = BLOCK_TILES * tc;
    const int64_t n_blocks = (nc + cb - 1) / cb;
    S3_REQUIRE(n_blocks < ((int64_t)1 << 31), "s3_interp_plan_create: too many cells");
    int32_t *blk_rows = nullptr, *blk_tiles = nullptr, *cnt_t = nullptr, *cnt_r = nullptr, *off_t = nullptr, *off_r = nullptr;
    S3_PB_CHECK(tmp.alloc(&blk_rows, (size_t)n_blocks * cb * k));
    S3_PB_CHECK(tmp.alloc(&blk_tiles, (size_t)n_blocks * cb * 2));
    S3_PB_CHECK(tmp.alloc(&cnt_t, (size_t)n_blocks));
    S3_PB_CHECK(tmp.alloc(&cnt_r, (size_t)n_blocks));
    S3_PB_CHECK(tmp.alloc(&off_t, (size_t)n_blocks));
    S3_PB_CHECK(tmp.alloc(&off_r, (size_t)n_blocks));
    S3_PB_CHECK(hipMalloc(reinterpret_cast<void **>(&out->loc), sizeof(uint16_t) * (size_t)nc * k));
    const size_t lds = sizeof(int32_t) * HT_SLOTS + sizeof(uint16_t) * HT_SLOTS + sizeof(uint16_t) * (size_t)tc * k;
    pack_kernel<<<(unsigned)n_blocks, 64, lds, st>>>(d_idx, out->perm, nc, k, tc, cb, ucap, blk_rows, blk_tiles, cnt_t, cnt_r,
                                                     out->loc);
    S3_PB_CHECK(hipGetLastError());

    // ---- 3. offsets + compact tables ---------------------------------------------------------------------------
    int32_t *d_scan = nullptr;
    S3_PB_CHECK(tmp.alloc(&d_scan, scan_tmp_items(n_blocks)));
    S3_PB_CHECK(exclusive_scan<int32_t>(cnt_t, off_t, n_blocks, d_scan, st));
    S3_PB_CHECK(exclusive_scan<int32_t>(cnt_r, off_r, n_blocks, d_scan, st));
    int32_t last[4] = {0, 0, 0, 0}, bad = 0;                      // offsets and counts of the last block -> totals
    S3_PB_CHECK(hipMemcpyAsync(&last[0], off_t + n_blocks - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    S3_PB_CHECK(hipMemcpyAsync(&last[1], off_r + n_blocks - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    S3_PB_CHECK(hipMemcpyAsync(&last[2], cnt_t + n_blocks - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    S3_PB_CHECK(hipMemcpyAsync(&last[3], cnt_r + n_blocks - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    S3_PB_CHECK(hipMemcpyAsync(&bad, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    S3_PB_CHECK(hipStreamSynchronize(st));
    S3_REQUIRE(bad == 0, "s3_interp_plan_create: neighbour index outside [0, %lld) in the table", (long long)n_src);
    out->n_tiles = (int64_t)last[0] + last[2];
    out->total_rows = (int64_t)last[1] + last[3];
    S3_PB_CHECK(hipMalloc(reinterpret_cast<void **>(&out->tile_cell_begin), sizeof(int32_t) * (size_t)(out->n_tiles + 1)));
    S3_PB_CHECK(hipMalloc(reinterpret_cast<void **>(&out->tile_row_begin), sizeof(int32_t) * (size_t)(out->n_tiles + 1)));
    S3_PB_CHECK(hipMalloc(reinterpret_cast<void **>(&out->rows), sizeof(int32_t) * (size_t)std::max<int64_t>(out->total_rows, 1)));
    gather_kernel<<<(unsigned)n_blocks, 64, 0, st>>>(blk_rows, blk_tiles, cnt_t, cnt_r, off_t, off_r, cb, k, n_blocks,
                                                     out->tile_cell_begin, out->tile_row_begin, out->rows);
    S3_PB_CHECK(hipGetLastError());
    S3_PB_CHECK(hipStreamSynchronize(st));
    return S3_OK;
}

}  // namespace s3

// ---- referenced-row bookkeeping of the KNN cache (include/s3hip.h) ---------------------------------------------
namespace {

__global__ void mark_rows_kernel(const int32_t *__restrict__ idx, int64_t n, int32_t n_src, int32_t *__restrict__ flag,
                                 int32_t *__restrict__ bad) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = idx[i];
    if (r < 0 || r >= n_src) { atomicExch(bad, 1); return; }
    if (flag[r] == 0) flag[r] = 1;              // benign race: every writer stores the same value
}

__global__ void compact_rows_kernel(int32_t *__restrict__ flag_remap, const int32_t *__restrict__ pos, int64_t n_src,
                                    int32_t *__restrict__ used) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_src) return;
    if (flag_remap[i]) {
        used[pos[i]] = (int32_t)i;
        flag_remap[i] = pos[i];
    } else {
        flag_remap[i] = -1;
    }
}

__global__ void remap_kernel(int32_t *__restrict__ idx, int64_t n, const int32_t *__restrict__ remap, int32_t n_src,
                             int32_t *__restrict__ bad) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = idx[i];
    if (r < 0 || r >= n_src || remap[r] < 0) { atomicExch(bad, 1); return; }
    idx[i] = remap[r];
}

// one wavefront per row: 16-byte pieces when source and destination rows are 16-byte aligned, 4-byte pieces otherwise
template <typename P>
__global__ void __launch_bounds__(256)
gather_rows_kernel(const char *__restrict__ src, int64_t src_pitch, const int32_t *__restrict__ ids, int64_t n,
                   int64_t row_bytes, char *__restrict__ dst, int64_t dst_pitch) {
    const int lane = threadIdx.x & 63;
    const int64_t pieces = row_bytes / (int64_t)sizeof(P);
    for (int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
        const P *s = reinterpret_cast<const P *>(src + (ids ? (int64_t)ids[row] : row) * src_pitch);
        P *d = reinterpret_cast<P *>(dst + row * dst_pitch);
        for (int64_t i = lane; i < pieces; i += 64) d[i] = s[i];
    }
}

// short rows (a few dozen pieces): one piece per thread, consecutive threads walk consecutive pieces of consecutive rows --
// every lane busy whatever the row length (the kernel above gives a 100-byte row to a whole wavefront: 25 of 64 lanes)
template <typename P>
__global__ void __launch_bounds__(256)
gather_short_rows_kernel(const char *__restrict__ src, int64_t src_pitch, const int32_t *__restrict__ ids, int64_t n,
                         int pieces, char *__restrict__ dst, int64_t dst_pitch) {
    const int64_t total = n * pieces;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / pieces;
        const int j = (int)(i - row * pieces);
        reinterpret_cast<P *>(dst + row * dst_pitch)[j] = reinterpret_cast<const P *>(src + (ids ? (int64_t)ids[row] : row) * src_pitch)[j];
    }
}

__global__ void check_ids_kernel(const int32_t *__restrict__ ids, int64_t n, int32_t n_rows, int32_t *__restrict__ bad) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n && (ids[i] < 0 || ids[i] >= n_rows)) atomicExch(bad, 1);
}

int read_flag(int32_t *d_bad, hipStream_t st, int32_t *h) {
    S3_HIP_CHECK(hipMemcpyAsync(h, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    S3_HIP_CHECK(hipStreamSynchronize(st));
    return S3_OK;
}

}  // namespace

extern "C" {

int s3_mark_rows(const int32_t *d_idx, int64_t n, int64_t n_src, int32_t *d_flag, s3_stream stream) {
    S3_REQUIRE(n >= 0 && n_src >= 1 && n_src < ((int64_t)1 << 31), "s3_mark_rows: bad sizes");
    if (n == 0) return S3_OK;
    S3_REQUIRE(d_idx && d_flag, "s3_mark_rows: null array");
    hipStream_t st = s3::as_stream(stream);
    s3::Scratch tmp;
    int32_t *d_bad = nullptr, bad = 0;
    S3_HIP_CHECK(tmp.alloc(&d_bad, 1));
    S3_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int32_t), st));
    mark_rows_kernel<<<s3::grid_for(n, 256), 256, 0, st>>>(d_idx, n, (int32_t)n_src, d_flag, d_bad);
    S3_LAUNCH_CHECK();
    const int rc = read_flag(d_bad, st, &bad);
    if (rc != S3_OK) return rc;
    S3_REQUIRE(bad == 0, "s3_mark_rows: neighbour index outside [0, %lld)", (long long)n_src);
    return S3_OK;
}

int s3_compact_rows(int32_t *d_flag_remap, int64_t n_src, int32_t *d_used, int64_t *h_n_used, s3_stream stream) {
    S3_REQUIRE(n_src >= 1 && n_src < ((int64_t)1 << 31) && d_flag_remap && d_used && h_n_used, "s3_compact_rows: bad arguments");
    hipStream_t st = s3::as_stream(stream);
    s3::Scratch tmp;
    int32_t *pos = nullptr;
    S3_HIP_CHECK(tmp.alloc(&pos, (size_t)n_src));
    int32_t *d_scan = nullptr;
    S3_HIP_CHECK(tmp.alloc(&d_scan, s3::scan_tmp_items(n_src)));
    S3_HIP_CHECK(s3::exclusive_scan<int32_t>(d_flag_remap, pos, n_src, d_scan, st));
    int32_t last_pos = 0, last_flag = 0;
    S3_HIP_CHECK(hipMemcpyAsync(&last_pos, pos + n_src - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    S3_HIP_CHECK(hipMemcpyAsync(&last_flag, d_flag_remap + n_src - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    compact_rows_kernel<<<s3::grid_for(n_src, 256), 256, 0, st>>>(d_flag_remap, pos, n_src, d_used);
    S3_LAUNCH_CHECK();
    S3_HIP_CHECK(hipStreamSynchronize(st));
    *h_n_used = (int64_t)last_pos + (last_flag ? 1 : 0);
    return S3_OK;
}

int s3_remap_indices(int32_t *d_idx, int64_t n, const int32_t *d_remap, int64_t n_src, s3_stream stream) {
    S3_REQUIRE(n >= 0 && n_src >= 1 && n_src < ((int64_t)1 << 31), "s3_remap_indices: bad sizes");
    if (n == 0) return S3_OK;
    S3_REQUIRE(d_idx && d_remap, "s3_remap_indices: null array");
    hipStream_t st = s3::as_stream(stream);
    s3::Scratch tmp;
    int32_t *d_bad = nullptr, bad = 0;
    S3_HIP_CHECK(tmp.alloc(&d_bad, 1));
    S3_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int32_t), st));
    remap_kernel<<<s3::grid_for(n, 256), 256, 0, st>>>(d_idx, n, d_remap, (int32_t)n_src, d_bad);
    S3_LAUNCH_CHECK();
    const int rc = read_flag(d_bad, st, &bad);
    if (rc != S3_OK) return rc;
    S3_REQUIRE(bad == 0, "s3_remap_indices: index without a position in the remap table");
    return S3_OK;
}

int s3_gather_rows(const void *d_src, int64_t n_src_rows, int64_t row_bytes, int64_t src_pitch_bytes, const int32_t *d_ids,
                   int64_t n, void *d_dst, int64_t dst_pitch_bytes, s3_stream stream) {
    S3_REQUIRE(n >= 0 && row_bytes >= 0 && n_src_rows >= 0 && n_src_rows < ((int64_t)1 << 31), "s3_gather_rows: bad sizes");
    if (n == 0 || row_bytes == 0) return S3_OK;
    S3_REQUIRE(d_src && d_dst, "s3_gather_rows: null array");
    S3_REQUIRE(row_bytes % 4 == 0 && src_pitch_bytes >= row_bytes && dst_pitch_bytes >= row_bytes &&
               src_pitch_bytes % 4 == 0 && dst_pitch_bytes % 4 == 0 &&
               reinterpret_cast<uintptr_t>(d_src) % 4 == 0 && reinterpret_cast<uintptr_t>(d_dst) % 4 == 0,
               "s3_gather_rows: rows must be multiples of 4 bytes on 4-byte boundaries");
    S3_REQUIRE(d_ids != nullptr || n <= n_src_rows, "s3_gather_rows: more rows requested than the source holds");
    hipStream_t st = s3::as_stream(stream);
    if (d_ids) {
        s3::Scratch tmp;
        int32_t *d_bad = nullptr, bad = 0;
        S3_HIP_CHECK(tmp.alloc(&d_bad, 1));
        S3_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int32_t), st));
        check_ids_kernel<<<s3::grid_for(n, 256), 256, 0, st>>>(d_ids, n, (int32_t)n_src_rows, d_bad);
        S3_LAUNCH_CHECK();
        const int rc = read_flag(d_bad, st, &bad);
        if (rc != S3_OK) return rc;
        S3_REQUIRE(bad == 0, "s3_gather_rows: row id outside [0, %lld)", (long long)n_src_rows);
    }
    const bool wide = row_bytes % 16 == 0 && src_pitch_bytes % 16 == 0 && dst_pitch_bytes % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(d_src) % 16 == 0 && reinterpret_cast<uintptr_t>(d_dst) % 16 == 0;
    const unsigned grid = s3::grid_for((n + 3) / 4, 1, 1 << 20);
    if (row_bytes <= 1024) {                     // fewer pieces than lanes of a wavefront (or a few times as many)
        if (wide) {
            const int pieces = (int)(row_bytes / 16);
            gather_short_rows_kernel<float4><<<s3::grid_for(n * pieces, 256, 1 << 16), 256, 0, st>>>(
                static_cast<const char *>(d_src), src_pitch_bytes, d_ids, n, pieces, static_cast<char *>(d_dst), dst_pitch_bytes);
        } else {
            const int pieces = (int)(row_bytes / 4);
            gather_short_rows_kernel<float><<<s3::grid_for(n * pieces, 256, 1 << 16), 256, 0, st>>>(
                static_cast<const char *>(d_src), src_pitch_bytes, d_ids, n, pieces, static_cast<char *>(d_dst), dst_pitch_bytes);
        }
        S3_LAUNCH_CHECK();
        return S3_OK;
    }
    if (wide)
        gather_rows_kernel<float4><<<grid, 256, 0, st>>>(static_cast<const char *>(d_src), src_pitch_bytes, d_ids, n, row_bytes,
                                                         static_cast<char *>(d_dst), dst_pitch_bytes);
    else
        gather_rows_kernel<float><<<grid, 256, 0, st>>>(static_cast<const char *>(d_src), src_pitch_bytes, d_ids, n, row_bytes,
                                                        static_cast<char *>(d_dst), dst_pitch_bytes);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

__global__ void fill_i32_kernel(int32_t *__restrict__ v, int64_t n, int32_t value) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) v[i] = value;
}

__global__ void scatter_positions_kernel(const int32_t *__restrict__ ids, int64_t n, int32_t n_src, int32_t *__restrict__ remap,
                                         int32_t *__restrict__ bad) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = ids[i];
    if (r < 0 || r >= n_src) { atomicExch(bad, 1); return; }
    remap[r] = (int32_t)i;
}

// d_remap[d_ids[i]] = i, every other entry -1 (the inverse of a list of distinct row ids)
int s3_positions_of(const int32_t *d_ids, int64_t n, int32_t *d_remap, int64_t n_src, s3_stream stream) {
    S3_REQUIRE(d_remap && n >= 0 && n_src >= 1 && n_src < ((int64_t)1 << 31) && (n == 0 || d_ids), "s3_positions_of: bad arguments");
    hipStream_t st = s3::as_stream(stream);
    s3::Scratch tmp;
    int32_t *d_bad = nullptr, bad = 0;
    S3_HIP_CHECK(tmp.alloc(&d_bad, 1));
    S3_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int32_t), st));
    fill_i32_kernel<<<s3::grid_for(n_src, 256), 256, 0, st>>>(d_remap, n_src, -1);
    if (n > 0) scatter_positions_kernel<<<s3::grid_for(n, 256), 256, 0, st>>>(d_ids, n, (int32_t)n_src, d_remap, d_bad);
    S3_LAUNCH_CHECK();
    S3_HIP_CHECK(hipMemcpyAsync(&bad, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    S3_HIP_CHECK(hipStreamSynchronize(st));
    S3_REQUIRE(bad == 0, "s3_positions_of: row id outside [0, %lld)", (long long)n_src);
    return S3_OK;
}

int s3_spatial_order(const double *d_points, int64_t n, int dim, int32_t *d_perm, s3_stream stream) {
    S3_REQUIRE(d_points && d_perm && n >= 1 && n < ((int64_t)1 << 31) && (dim == 2 || dim == 3), "s3_spatial_order: bad arguments");
    s3::Scratch tmp;
    const int rc = s3::spatial_perm(d_points, n, dim, s3::as_stream(stream), d_perm, tmp);
    if (rc != S3_OK) return rc;
    S3_HIP_CHECK(hipStreamSynchronize(s3::as_stream(stream)));       // the scratch arrays go away with `tmp`
    return S3_OK;
}

// the two primitives of csrc/scan_sort.h on their own (the planner and the device topology use them internally)
int s3_exclusive_scan(const void *d_in, void *d_out, int64_t n, int elem_bytes, s3_stream stream) {
    S3_REQUIRE(n >= 0 && (n == 0 || (d_in && d_out)) && (elem_bytes == 4 || elem_bytes == 8), "s3_exclusive_scan: int32 / int64 arrays");
    if (n == 0) return S3_OK;
    hipStream_t st = s3::as_stream(stream);
    s3::Scratch tmp;
    if (elem_bytes == 4) {
        int32_t *t = nullptr;
        S3_HIP_CHECK(tmp.alloc(&t, s3::scan_tmp_items(n)));
        S3_HIP_CHECK(s3::exclusive_scan<int32_t>(static_cast<const int32_t *>(d_in), static_cast<int32_t *>(d_out), n, t, st));
    } else {
        int64_t *t = nullptr;
        S3_HIP_CHECK(tmp.alloc(&t, s3::scan_tmp_items(n)));
        S3_HIP_CHECK(s3::exclusive_scan<int64_t>(static_cast<const int64_t *>(d_in), static_cast<int64_t *>(d_out), n, t, st));
    }
    S3_HIP_CHECK(hipStreamSynchronize(st));                           // the scratch array goes away with `tmp`
    return S3_OK;
}

int s3_sort_pairs(uint64_t *d_keys, int32_t *d_vals, int64_t n, int bits, s3_stream stream) {
    S3_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && (n == 0 || (d_keys && d_vals)) && bits >= 1 && bits <= 64, "s3_sort_pairs: bad arguments");
    if (n == 0) return S3_OK;
    hipStream_t st = s3::as_stream(stream);
    s3::Scratch tmp;
    uint64_t *k2 = nullptr;
    int32_t *v2 = nullptr, *hist = nullptr;
    const size_t items = s3::sort_hist_items(n);
    S3_HIP_CHECK(tmp.alloc(&k2, (size_t)n));
    S3_HIP_CHECK(tmp.alloc(&v2, (size_t)n));
    S3_HIP_CHECK(tmp.alloc(&hist, items + s3::scan_tmp_items((int64_t)items)));
    bool in_alt = false;
    S3_HIP_CHECK(s3::radix_sort_pairs(d_keys, k2, d_vals, v2, n, bits, hist, st, &in_alt));
    if (in_alt) {
        S3_HIP_CHECK(hipMemcpyAsync(d_keys, k2, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
        S3_HIP_CHECK(hipMemcpyAsync(d_vals, v2, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    S3_HIP_CHECK(hipStreamSynchronize(st));
    return S3_OK;
}

}  // extern "C"
