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
//   plan (device, once): cells are put in Hilbert order of their centres (radix sort); consecutive cells are packed
//                       greedily (plan_build.hip: one wavefront per 64 cells, LDS hash table of the tile's rows) into
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
#ifdef S3_PROBE_STAMPS
__device__ long long s3_probe_stamps[1024 * 4 * 8];
#define S3_STAMP(V) const long long V = __builtin_amdgcn_s_memrealtime();
#else
#define S3_STAMP(V)
#endif
#include "plan_build.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <type_traits>
#include <vector>

struct s3_interp_plan : s3::PlanTables {
    int64_t nc = 0, n_src = 0;
    int k = 0, ucap = 0, tc = 64;
    double *wp = nullptr;                // [nc*k] weights in the layout of `loc`: per tile [m][cell in tile]
    bool has_weights = false;
    double *dump = nullptr;              // per-lane slots for the stores the persistent kernel must not make
    size_t dump_doubles = 0;
    int32_t *rows_src = nullptr;         // [total_rows] `rows` as row ids of the caller's full table (s3_interp_plan_set_source_ids)
    int64_t n_table = 0;                 // rows of that table
    int32_t *sched_begin = nullptr;      // [sched_wgs + 1] tile lists of the persistent workgroups (plan_schedule)
    int32_t *sched_tiles = nullptr;      // [n_tiles]
    int4 *sched_desc = nullptr;          // [n_tiles] in list order: {first row, rows, first cell, cells} of each scheduled tile
    double *wl = nullptr;                // the weights and
    uint16_t *pl = nullptr;              // positions again, in the order the persistent kernel's LANES take them (lane_tables_*)
    int sched_wgs = 0;
};

namespace s3 {

constexpr int PL_SEG = 128;        // bytes of one row staged per chunk
constexpr int PL_NP = 16;          // staging passes of (tile cells)/2 rows -> at most 8 * (tile cells) distinct rows
#ifndef S3_PL_LP
#define S3_PL_LP 8
#endif
constexpr int PL_LP = S3_PL_LP;    // LDS row pitch of the persistent kernel in 16-byte vectors.  9 (144 bytes: the rows of the cells a
                                   // wavefront accumulates together then start in different banks) was measured against 8 in
                                   // alternating processes on one box: 0.167 / 0.152 / 0.457 ms against 0.167 / 0.149 / 0.466 ms at
                                   // 25 / 32 / 128 snapshots -- the bank conflicts are not on the critical path

template <typename T>
struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

// `left` = elements of the output row at and after p (may be <= 0 or < N for the ragged tail of a row); full pieces of
// rows with an even length are written as double2 (the row starts are then 16-byte aligned), everything else per element
template <int N>
__device__ __forceinline__ void store_piece(double *__restrict__ p, const double (&a)[N], int64_t left, bool even_rows) {
    if (left >= N && even_rows) {
#pragma unroll
        for (int i = 0; i < N; i += 2) *reinterpret_cast<double2 *>(p + i) = make_double2(a[i], a[i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < left) p[i] = a[i];
    }
}

// the tile's weights and LDS row positions: `n` consecutive entries of the plan-ordered streams ([m][cell in tile]) ->
// LDS, same layout; eight loads per lane are in flight before the first LDS store
__device__ __forceinline__ void stage_tile_tables(const double *__restrict__ wp, const uint16_t *__restrict__ loc, int n,
                                                  double *__restrict__ s_w, uint16_t *__restrict__ s_loc, int block) {
    constexpr int UN = 8;
    for (int base = 0; base < n; base += block * UN) {
        double wr[UN];
        uint16_t lr[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = min(base + u * block + (int)threadIdx.x, n - 1);
            wr[u] = wp[i];
            lr[u] = loc[i];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = base + u * block + (int)threadIdx.x;
            if (i < n) {
                s_w[i] = wr[u];
                s_loc[i] = lr[u];
            }
        }
    }
}

// the same with the LDS image laid out [m][TC] (row pitch = the tile capacity, not the tile's cell count): the accumulate
// loop then addresses weight / position m of its cell at a CONSTANT distance m * TC from those of neighbour 0 -- immediate offsets
// of the LDS instructions instead of a vector add per access (two of the ~20 vector instructions per neighbour).  Four neighbours
// per pass of the 256 threads; eight loads per lane in flight.
template <int TC>
__device__ __forceinline__ void stage_tile_tables_strided(const double *__restrict__ wp, const uint16_t *__restrict__ loc, int n_c, int k,
                                                          double *__restrict__ s_w, uint16_t *__restrict__ s_loc) {
    static_assert(TC == 64, "four rows of 64 cells per pass of 256 threads");
    const int cl = threadIdx.x & 63, m0 = threadIdx.x >> 6;
    const bool has = cl < n_c;
    constexpr int UN = 8;
    for (int base = 0; base < k; base += 4 * UN) {
        double wr[UN];
        uint16_t lr[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int m = base + 4 * u + m0;
            const int i = has && m < k ? m * n_c + cl : 0;
            wr[u] = wp[i];
            lr[u] = loc[i];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int m = base + 4 * u + m0;
            if (has && m < k) {
                s_w[m * TC + cl] = wr[u];
                s_loc[m * TC + cl] = lr[u];
            }
        }
    }
}

// weights of the caller's table [nc][k] -> plan order (one workgroup per tile)
__global__ void __launch_bounds__(256)
permute_weights_kernel(const int32_t *__restrict__ perm, const int32_t *__restrict__ tile_cell_begin,
                       const double *__restrict__ w, int k, double *__restrict__ wp) {
    const int c_begin = tile_cell_begin[blockIdx.x], n_c = tile_cell_begin[blockIdx.x + 1] - c_begin;
    for (int i = threadIdx.x; i < n_c * k; i += 256) {
        const int m = i / n_c, j = i - m * n_c;
        wp[(int64_t)c_begin * k + i] = w[(int64_t)perm[c_begin + j] * k + m];
    }
}

// The persistent kernel keeps a tile's tables in registers: lane (cell, v0) of a cell's DPP quad holds the entries m = 4 i + v0.  Read
// from the [m][cell] layout above that is 2 * ceil(k / 4) loads of 8 and 2 bytes per lane and tile -- fourteen vector-memory
// instructions at k = 26, as many as the tile's row segments take when a row is one chunk long, and the number of those a CU has
// accepted is what paces a step (HISTORY 5.2b).  Second copy in LANE order: per tile and lane the weights as ceil(KQ / 2) 16-byte
// vectors ([vector][lane]: a wavefront's load is one contiguous KiB) and the positions as ONE vector of eight 16-bit entries.
//   wl: tile offset c_begin * 4 * NV * 2 doubles, then [j < NV][lane < 4 n_c][2];  pl: c_begin * 32 entries, then [lane][8]
__global__ void __launch_bounds__(256)
lane_weights_kernel(const int32_t *__restrict__ tile_cell_begin, const double *__restrict__ wp, int k, double *__restrict__ wl) {
    const int c_begin = tile_cell_begin[blockIdx.x], n_c = tile_cell_begin[blockIdx.x + 1] - c_begin;
    const int kq = (k + 3) / 4, nv = (kq + 1) / 2, lanes = n_c * 4;
    const double *src = wp + (int64_t)c_begin * k;
    double *dst = wl + (int64_t)c_begin * 4 * nv * 2;
    for (int e = threadIdx.x; e < lanes * nv * 2; e += 256) {
        const int j = e / (lanes * 2), r = e - j * lanes * 2, lane = r >> 1, i = 2 * j + (r & 1);
        const int m = i * 4 + (lane & 3);
        dst[e] = i < kq ? src[(m < k ? m : 0) * n_c + (lane >> 2)] : 0.0;          // (entries beyond k: entry 0, as the kernel clamps)
    }
}
__global__ void __launch_bounds__(256)
lane_positions_kernel(const int32_t *__restrict__ tile_cell_begin, const uint16_t *__restrict__ loc, int k, uint16_t *__restrict__ pl) {
    const int c_begin = tile_cell_begin[blockIdx.x], n_c = tile_cell_begin[blockIdx.x + 1] - c_begin;
    const int kq = (k + 3) / 4;
    const uint16_t *src = loc + (int64_t)c_begin * k;
    uint16_t *dst = pl + (int64_t)c_begin * 32;
    for (int e = threadIdx.x; e < n_c * 32; e += 256) {
        const int lane = e >> 3, i = e & 7, m = i * 4 + (lane & 3);
        dst[e] = i < kq ? src[(m < k ? m : 0) * n_c + (lane >> 2)] : (uint16_t)0;
    }
}

// Short rows (a few 128-byte lines: small snapshot batches such as the 16- or 25-snapshot batches of the reference's
// cylinder3D script, examples/s3_for_cylinder3D_Re3900.py:28-69).  The per-tile start-up (weights, positions, row ids) is
// most of the traffic here and there is no chunk sweep to pipeline, so this variant keeps its register count low enough
// for three or four workgroups per CU, stages whole row pieces of `vc` <= 8 vectors with `vc` lanes per row and gives
// every (cell, vector) pair of the tile to one lane.  Same plan tables, same arithmetic (f64 FMA in neighbour order).
template <typename T, int TC>
__global__ void __launch_bounds__(256, 3)
interp_planned_short_kernel(const int32_t *__restrict__ perm, const int32_t *__restrict__ tile_cell_begin,
                            const int32_t *__restrict__ tile_row_begin, const int32_t *__restrict__ rows,
                            const uint16_t *__restrict__ loc, const double *__restrict__ w /*plan order*/, int k, int ucap,
                            const T *__restrict__ data, int64_t row_len, int64_t in_stride, double *__restrict__ out,
                            int64_t n_tiles, int64_t tiles_per_xcd, int vpr, int pitch) {
    using V = typename Vec16<T>::type;
    constexpr int EPV = Vec16<T>::N;
    constexpr int BLOCK = 256;
    constexpr int UN = 8;                                        // row pieces in flight per lane while staging
    extern __shared__ float4 lds_raw[];
    V *s_data = reinterpret_cast<V *>(lds_raw);                                  // [ucap][pitch] 16-byte vectors
    double *s_w = reinterpret_cast<double *>(lds_raw + (size_t)ucap * pitch);    // [k][TC]
    uint16_t *s_loc = reinterpret_cast<uint16_t *>(s_w + (size_t)k * TC);        // [k][TC]
    int32_t *s_cell = reinterpret_cast<int32_t *>(s_loc + (size_t)k * TC);       // [TC] output rows of the tile's cells

    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);     // XCD-aware (speed only)
    if (tile >= n_tiles) return;
    const int c_begin = tile_cell_begin[tile], n_c = tile_cell_begin[tile + 1] - c_begin;
    const int r_begin = tile_row_begin[tile], n_r = tile_row_begin[tile + 1] - r_begin;
    const bool even_rows = (row_len & 1) == 0;
    if ((int)threadIdx.x < n_c) s_cell[threadIdx.x] = perm[c_begin + threadIdx.x];

    for (int v_begin = 0; v_begin < vpr; v_begin += pitch) {
        const int vc = min(pitch, vpr - v_begin);                // vectors of this piece
        if (v_begin) __syncthreads();                            // the previous piece has been consumed
        // stage the piece of every distinct row: UN independent 16-byte loads per lane before the LDS stores
        const int n_items = n_r * vc;
        for (int base = 0; base < n_items; base += BLOCK * UN) {
            V reg[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int item = min(base + u * BLOCK + (int)threadIdx.x, n_items - 1);
                const int r = item / vc, v = item - r * vc;
                reg[u] = *reinterpret_cast<const V *>(data + (int64_t)rows[r_begin + r] * in_stride + (int64_t)(v_begin + v) * EPV);
            }
            if (base == 0 && v_begin == 0)     // issued behind the first row loads
                stage_tile_tables(w + (int64_t)c_begin * k, loc + (int64_t)c_begin * k, n_c * k, s_w, s_loc, BLOCK);
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int item = base + u * BLOCK + (int)threadIdx.x;
                if (item < n_items) {
                    const int r = item / vc, v = item - r * vc;
                    s_data[r * pitch + v] = reg[u];
                }
            }
        }
        __syncthreads();
        for (int item = threadIdx.x; item < n_c * vc; item += BLOCK) {
            const int cl = item / vc, v = item - cl * vc;
            double acc[EPV];
#pragma unroll
            for (int i = 0; i < EPV; ++i) acc[i] = 0.0;
#pragma unroll 8
            for (int m = 0; m < k; ++m) {
                const int pos = s_loc[m * n_c + cl];
                const double wm = s_w[m * n_c + cl];
                const V a = s_data[pos * pitch + v];
                const T *ae = reinterpret_cast<const T *>(&a);
#pragma unroll
                for (int i = 0; i < EPV; ++i) acc[i] = fma(wm, (double)ae[i], acc[i]);
            }
            const int64_t col = (int64_t)(v_begin + v) * EPV;
            store_piece<EPV>(out + (int64_t)s_cell[cl] * row_len + col, acc, row_len - col, even_rows);
        }
    }
}

// Short rows, weights in registers (k <= KMAX): with at most four vectors per row every (cell, vector) pair of a tile has its
// own lane, so the lane reads its cell's k weights / LDS positions straight from the plan-ordered streams into registers
// (coalesced: consecutive cells are consecutive in the stream) while the row pieces are on their way -- no LDS round trip
// for them, and LDS holds row data only (31 KiB: five workgroups per CU fit, registers allow four).
template <typename T, int KMAX>
__global__ void __launch_bounds__(256, KMAX <= 26 ? 4 : 3)      // 126 VGPRs at KMAX = 26; KMAX = 32 would spill at four waves
interp_planned_short_reg_kernel(const int32_t *__restrict__ perm, const int32_t *__restrict__ tile_cell_begin,
                                const int32_t *__restrict__ tile_row_begin, const int32_t *__restrict__ rows,
                                const uint16_t *__restrict__ loc, const double *__restrict__ w /*plan order*/, int k,
                                const T *__restrict__ data, int64_t row_len, int64_t in_stride, double *__restrict__ out,
                                int64_t n_tiles, int64_t tiles_per_xcd, int vc) {
    using V = typename Vec16<T>::type;
    constexpr int EPV = Vec16<T>::N;
    constexpr int BLOCK = 256;
    constexpr int UN = 8;
    extern __shared__ float4 lds_raw[];
    V *s_data = reinterpret_cast<V *>(lds_raw);                  // [n_r][vc]

    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);
    if (tile >= n_tiles) return;
    const int c_begin = tile_cell_begin[tile], n_c = tile_cell_begin[tile + 1] - c_begin;
    const int r_begin = tile_row_begin[tile], n_r = tile_row_begin[tile + 1] - r_begin;
    const bool even_rows = (row_len & 1) == 0;

    // stage the rows: up to UN pieces per lane in flight (n_r * vc <= 496 * 4 < 256 * UN)
    const int n_items = n_r * vc;
    V reg[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int item = min(u * BLOCK + (int)threadIdx.x, n_items - 1);
        const int r = item / vc, v = item - r * vc;
        reg[u] = *reinterpret_cast<const V *>(data + (int64_t)rows[r_begin + r] * in_stride + (int64_t)v * EPV);
    }
    // this lane's (cell, vector) pair and the cell's weights / positions
    const int cl = min((int)threadIdx.x / vc, n_c - 1), v = (int)threadIdx.x - ((int)threadIdx.x / vc) * vc;
    const bool has_item = (int)threadIdx.x < n_c * vc;
    const double *wt = w + (int64_t)c_begin * k + cl;
    const uint16_t *lt = loc + (int64_t)c_begin * k + cl;
    double wr[KMAX];
    int pr[KMAX];
#pragma unroll
    for (int m = 0; m < KMAX; ++m) {
        const int mm = m < k ? m : 0;
        wr[m] = wt[(int64_t)mm * n_c];
        pr[m] = lt[(int64_t)mm * n_c];
    }
    const int64_t cell = perm[c_begin + cl];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int item = u * BLOCK + (int)threadIdx.x;
        if (item < n_items) s_data[item] = reg[u];               // [r][v] with pitch vc == item
    }
    __syncthreads();
    if (!has_item) return;
    double acc[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[i] = 0.0;
#pragma unroll
    for (int m = 0; m < KMAX; ++m) {
        if (m < k) {
            const V a = s_data[pr[m] * vc + v];
            const T *ae = reinterpret_cast<const T *>(&a);
#pragma unroll
            for (int i = 0; i < EPV; ++i) acc[i] = fma(wr[m], (double)ae[i], acc[i]);
        }
    }
    const int64_t col = (int64_t)v * EPV;
    store_piece<EPV>(out + cell * row_len + col, acc, row_len - col, even_rows);
}

// Short rows of exactly four vectors (16 fp32 snapshots -- SURVEY C4 -- or 8 f64 values): the four lanes of a cell are one
// DPP quad, so each lane loads only every fourth weight / position of its cell (7 + 7 loads instead of 26 + 26 one-line
// transactions per lane: the memory pipeline of the kernel above spends more instructions on the tile's tables than on
// its rows) and the quad broadcasts them when the neighbour's turn comes.  45 fewer registers per lane: five workgroups
// per CU.  Same arithmetic, same results.
template <int J>
__device__ __forceinline__ int quad_bcast_i32(int v) {
    // (old = 0 with bound_ctrl: every source lane of a quad_perm exists, so `old` is never used and the compiler need not
    // copy the source into the destination first -- one v_mov_b32_dpp instead of two moves)
    return __builtin_amdgcn_update_dpp(0, v, J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, true);
}
template <int J>
__device__ __forceinline__ double quad_bcast_f64(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = quad_bcast_i32<J>((int)(b & 0xffffffffll)), hi = quad_bcast_i32<J>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <typename T, int KQ>        // KQ = ceil(k / 4) values of each table per lane
__global__ void __launch_bounds__(256, 5)
interp_planned_short_quad_kernel(const int32_t *__restrict__ perm, const int32_t *__restrict__ tile_cell_begin,
                                 const int32_t *__restrict__ tile_row_begin, const int32_t *__restrict__ rows,
                                 const uint16_t *__restrict__ loc, const double *__restrict__ w /*plan order*/, int k,
                                 const T *__restrict__ data, int64_t row_len, int64_t in_stride, double *__restrict__ out,
                                 int64_t n_tiles, int64_t tiles_per_xcd) {
    using V = typename Vec16<T>::type;
    constexpr int EPV = Vec16<T>::N;
    constexpr int BLOCK = 256, UN = 8, VC = 4;
    extern __shared__ float4 lds_raw[];
    V *s_data = reinterpret_cast<V *>(lds_raw);                  // [n_r][4]

    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);
    if (tile >= n_tiles) return;
    const int c_begin = tile_cell_begin[tile], n_c = tile_cell_begin[tile + 1] - c_begin;
    const int r_begin = tile_row_begin[tile], n_r = tile_row_begin[tile + 1] - r_begin;
    const bool even_rows = (row_len & 1) == 0;

    const int n_items = n_r * VC;
    V reg[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int item = min(u * BLOCK + (int)threadIdx.x, n_items - 1);
        reg[u] = *reinterpret_cast<const V *>(data + (int64_t)rows[r_begin + (item >> 2)] * in_stride + (int64_t)(item & 3) * EPV);
    }
    const int cl = min((int)threadIdx.x >> 2, n_c - 1), v = (int)threadIdx.x & 3;
    const bool has_item = ((int)threadIdx.x >> 2) < n_c;
    const double *wt = w + (int64_t)c_begin * k + cl;
    const uint16_t *lt = loc + (int64_t)c_begin * k + cl;
    double wq[KQ];
    int pq[KQ];
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
        const int m = i * 4 + v, mm = m < k ? m : 0;
        wq[i] = wt[(int64_t)mm * n_c];
        pq[i] = lt[(int64_t)mm * n_c];
    }
    const int64_t cell = perm[c_begin + cl];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int item = u * BLOCK + (int)threadIdx.x;
        if (item < n_items) s_data[item] = reg[u];
    }
    __syncthreads();
    double acc[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[i] = 0.0;
    auto step = [&](double wm, int pos) {
        const V a = s_data[pos * VC + v];
        const T *ae = reinterpret_cast<const T *>(&a);
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] = fma(wm, (double)ae[i], acc[i]);
    };
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
        // (every lane of the wavefront takes part in the broadcasts; lanes without a cell hold a clamped copy)
        const double w0 = quad_bcast_f64<0>(wq[i]), w1 = quad_bcast_f64<1>(wq[i]), w2 = quad_bcast_f64<2>(wq[i]), w3 = quad_bcast_f64<3>(wq[i]);
        const int p0 = quad_bcast_i32<0>(pq[i]), p1 = quad_bcast_i32<1>(pq[i]), p2 = quad_bcast_i32<2>(pq[i]), p3 = quad_bcast_i32<3>(pq[i]);
        if (i * 4 + 0 < k) step(w0, p0);
        if (i * 4 + 1 < k) step(w1, p1);
        if (i * 4 + 2 < k) step(w2, p2);
        if (i * 4 + 3 < k) step(w3, p3);
    }
    if (!has_item) return;
    const int64_t col = (int64_t)v * EPV;
    store_piece<EPV>(out + cell * row_len + col, acc, row_len - col, even_rows);
}

// Dispatch order of the chunk kernels (one workgroup per (tile, run of column chunks); a 1-D grid, workgroup b on XCD b % 8).
// XCD x owns the x-th eighth of the Hilbert-ordered tiles; its share is walked in BRICKS of `brick` consecutive tiles, and inside
// a brick run-major: all tiles of the brick for chunk run 0, then for run 1, ... -- the workgroups resident on an XCD at the same
// time then stage the same columns of rows that neighbouring tiles share (the halo) within a few steps of each other.
// brick >= tiles_per_xcd: run-major over the whole share (the order of a 2-D grid); n_split = 1: tile order.
__device__ __forceinline__ void brick_map(int64_t b, int64_t tiles_per_xcd, int brick, int n_split, int64_t &tile, int &run) {
    const int64_t i = b >> 3;
    const int64_t per = (int64_t)brick * n_split;
    const int64_t bk = i / per, rem = i - bk * per;
    const int64_t left = tiles_per_xcd - bk * brick;
    const int64_t bl = left < brick ? left : (int64_t)brick;
    run = (int)(rem / bl);
    tile = (b & 7) * tiles_per_xcd + bk * brick + (rem - (int64_t)run * bl);
}

// The same with a finer grain at the END of every XCD's share: the last `tail` tiles are cut into `tail_split` runs of column
// chunks each (run-major among them), all tiles before them are swept whole.  While the launch drains -- the last tile of a slot
// starts up to one tile lifetime (~240 us of a 3.6-ms launch) before the end, and the slots empty one by one -- the pieces still
// waiting are short, so the chip stays full for longer.  -> tile, [chunk0, chunk1)
__device__ __forceinline__ void tail_map(int64_t b, int64_t tiles_per_xcd, int tail, int tail_split, int n_chunks, int64_t &tile,
                                         int &chunk0, int &chunk1) {
    const int64_t i = b >> 3, head = tiles_per_xcd - tail;
    if (i < head) {
        tile = (b & 7) * tiles_per_xcd + i;
        chunk0 = 0;
        chunk1 = n_chunks;
        return;
    }
    const int64_t j = i - head;
    const int run = (int)(j / tail);
    const int per = (n_chunks + tail_split - 1) / tail_split;
    tile = (b & 7) * tiles_per_xcd + head + (j - (int64_t)run * tail);
    chunk0 = run * per;
    chunk1 = chunk0 + per < n_chunks ? chunk0 + per : n_chunks;
}

template <typename T, int TC>
__global__ void __launch_bounds__(TC * 4, 2)  // 226 VGPRs: two waves per SIMD
interp_planned_kernel(const int32_t *__restrict__ perm, const int32_t *__restrict__ tile_cell_begin,
                      const int32_t *__restrict__ tile_row_begin, const int32_t *__restrict__ rows,
                      const uint16_t *__restrict__ loc, const double *__restrict__ w /*plan order*/, int k, int ucap,
                      const T *__restrict__ data, int64_t row_len, int64_t in_stride, double *__restrict__ out,
                      int64_t n_tiles, int64_t tiles_per_xcd, int chunks_per_block, int n_chunks, int brick, int n_split, int tail, int tail_split) {
    using V = typename Vec16<T>::type;
    constexpr int EPV = Vec16<T>::N;                 // elements per 16-byte vector
    constexpr int EPC = PL_SEG / (int)sizeof(T);     // elements per chunk
    constexpr int BLOCK = TC * 4;                    // 4 lanes per cell
    constexpr int RPP = BLOCK / 8;                   // rows staged per pass (8 lanes per 128-B segment)
    extern __shared__ float4 lds_raw[];
    V *s_data = reinterpret_cast<V *>(lds_raw);                                  // [ucap][8] 16-byte vectors
    double *s_w = reinterpret_cast<double *>(lds_raw + (size_t)ucap * 8);        // [k][TC]
    uint16_t *s_loc = reinterpret_cast<uint16_t *>(s_w + (size_t)k * TC);        // [k][TC]

    int64_t tile;
    int run = 0, chunk0, chunk1;
    if (tail > 0) {
        tail_map(blockIdx.x, tiles_per_xcd, tail, tail_split, n_chunks, tile, chunk0, chunk1);
    } else {
        brick_map(blockIdx.x, tiles_per_xcd, brick, n_split, tile, run);     // XCD-aware (speed only)
        chunk0 = run * chunks_per_block;
        chunk1 = min(n_chunks, chunk0 + chunks_per_block);
    }
    if (tile >= n_tiles) return;
    // (functions of blockIdx: kept in scalar registers -- left to itself the compiler carried them in vector registers and spilled)
    chunk0 = __builtin_amdgcn_readfirstlane(chunk0);
    chunk1 = __builtin_amdgcn_readfirstlane(chunk1);
    const int c_begin = tile_cell_begin[tile], n_c = tile_cell_begin[tile + 1] - c_begin;
    const int r_begin = tile_row_begin[tile], n_r = tile_row_begin[tile + 1] - r_begin;
    const bool even_rows = (row_len & 1) == 0;                   // output rows 16-byte aligned -> double2 stores

    // the tile's weights and LDS row positions, [neighbour][cell] so that a wavefront reads consecutive words
    // the tile's weights and LDS row positions, [neighbour][cell] so that a wavefront reads consecutive words
    stage_tile_tables(w + (int64_t)c_begin * k, loc + (int64_t)c_begin * k, n_c * k, s_w, s_loc, BLOCK);

    // this thread's cell: 4 lanes per cell, each lane two 16-byte vectors of the chunk (v0 and v0+4)
    const int cl = threadIdx.x >> 2, v0 = threadIdx.x & 3;
    const bool has_cell = cl < n_c;
    const int64_t cell = has_cell ? perm[c_begin + cl] : 0;


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
        const bool ok_ = c0_ + (int64_t)svec * EPV < row_len;        \
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
            const int pos = s_loc[m * n_c + cl];
            const double wm = s_w[m * n_c + cl];
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
        store_piece<EPV>(o + v0 * EPV, acc0, row_len - col0 - (int64_t)v0 * EPV, even_rows);
        store_piece<EPV>(o + (v0 + 4) * EPV, acc1, row_len - col0 - (int64_t)(v0 + 4) * EPV, even_rows);
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

// Long rows that start on 16-byte but not on 128-byte boundaries: a dense [N, n_comp * T] batch read where it lies
// (interpolate_data consumes the caller's table as it stands, export.py:446-468; 1000 fp32 snapshots = 4000-byte rows whose
// starts sit 0 / 32 / 64 / 96 bytes into a cache line).  The kernel above would stage the 128 bytes [c * 128, c * 128 + 128) of
// every row per step: for three rows in four that segment straddles two lines, each line is asked for by two consecutive
// steps (7 us apart -- the L2 has turned over by then), and the launch moves 15-25 % more bytes (4.1 against 3.5 ms).
// Here every load instruction fetches whole ALIGNED lines: line j of a row = the 128 bytes at (row start rounded down to
// 128) + j * 128, lane v of the row's eight staging lanes holding its v-th vector.  With ph = (row start mod 128) / 16 the
// vectors v >= ph of line c and the vectors v < ph of line c + 1 together are the row's segment c, so the two register sets
// hold two CONSECUTIVE lines and the LDS image of step c takes, per lane, the older set's vector (v >= ph) or the newer
// one's (v < ph), at slot (v - ph) mod 8 -- one ds_write_b128 per lane and row as before, behind four v_cndmask.  The set
// that held line c is refilled with line c + 2 right away.  Every line is fetched once, by one full-line request; the LDS
// image, the accumulate phase and the arithmetic are those of the kernel above (same results bit for bit).
// Lines that reach beyond the last vector of a row (the row's last one or two, depending on ph) are loaded with the
// lane's offset clamped to the row's last vector: nothing outside the 128-byte lines that hold the row's own bytes is read.
// (component-wise: a ternary on the vector STRUCTS selects between their addresses and sends both register sets to scratch)
__device__ __forceinline__ float4 select16(bool c, const float4 &a, const float4 &b) {
    return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
__device__ __forceinline__ double2 select16(bool c, const double2 &a, const double2 &b) {
    return make_double2(c ? a.x : b.x, c ? a.y : b.y);
}
template <typename T, int HOLD>      // HOLD: 0 every step writes its own 256 bytes (default); 1 whole-line output stores (below)
__global__ void __launch_bounds__(256, 2)
interp_planned_shift_kernel(const int32_t *__restrict__ perm, const int32_t *__restrict__ tile_cell_begin,
                            const int32_t *__restrict__ tile_row_begin, const int32_t *__restrict__ rows,
                            const uint16_t *__restrict__ loc, const double *__restrict__ w /*plan order*/, int k, int ucap,
                            const T *__restrict__ data, int64_t row_len, int64_t in_stride, double *__restrict__ out,
                            int64_t n_tiles, int64_t tiles_per_xcd, int chunks_per_block, int n_chunks, int brick, int n_split, int tail, int tail_split) {
    using V = typename Vec16<T>::type;
    constexpr int TC = 64;
    constexpr int EPV = Vec16<T>::N;
    constexpr int EPC = PL_SEG / (int)sizeof(T);
    constexpr int BLOCK = TC * 4;
    constexpr int RPP = BLOCK / 8;
    extern __shared__ float4 lds_raw[];
    V *s_data = reinterpret_cast<V *>(lds_raw);                                  // [ucap][8] 16-byte vectors
    double *s_w = reinterpret_cast<double *>(lds_raw + (size_t)ucap * 8);        // [k][TC]
    uint16_t *s_loc = reinterpret_cast<uint16_t *>(s_w + (size_t)k * TC);        // [k][TC]

    int64_t tile;
    int run = 0, chunk0, chunk1;
    if (tail > 0) {
        tail_map(blockIdx.x, tiles_per_xcd, tail, tail_split, n_chunks, tile, chunk0, chunk1);
    } else {
        brick_map(blockIdx.x, tiles_per_xcd, brick, n_split, tile, run);     // XCD-aware (speed only)
        chunk0 = run * chunks_per_block;
        chunk1 = min(n_chunks, chunk0 + chunks_per_block);
    }
    if (tile >= n_tiles) return;
    // (functions of blockIdx: kept in scalar registers -- left to itself the compiler carried them in vector registers and spilled)
    chunk0 = __builtin_amdgcn_readfirstlane(chunk0);
    chunk1 = __builtin_amdgcn_readfirstlane(chunk1);
    const int c_begin = tile_cell_begin[tile], n_c = tile_cell_begin[tile + 1] - c_begin;
    const int r_begin = tile_row_begin[tile], n_r = tile_row_begin[tile + 1] - r_begin;
    const bool even_rows = (row_len & 1) == 0;

    // (tables at a constant pitch of TC entries: in THIS kernel -3 % per launch against the [m][n_c] layout of the kernel above,
    // 3.478 / 3.584 ms interleaved in one process; in the kernel above the same change measured +0.4 % and was not kept)
    stage_tile_tables_strided<TC>(w + (int64_t)c_begin * k, loc + (int64_t)c_begin * k, n_c, k, s_w, s_loc);

    const int cl = threadIdx.x >> 2, v0 = threadIdx.x & 3;
    const bool has_cell = cl < n_c;
    const int64_t cell = has_cell ? perm[c_begin + cl] : 0;


    const int srow = threadIdx.x >> 3, svec = threadIdx.x & 7;
    const uintptr_t base = reinterpret_cast<uintptr_t>(data);
    const uint64_t stride_bytes = (uint64_t)in_stride * sizeof(T);
    const int n_vec_row = (int)((row_len + EPV - 1) / EPV);      // 16-byte vectors that hold a row's own bytes
    const int last_vec = n_vec_row - 1;                          // the last of them, counted from the row start
    const uintptr_t base128 = base & ~(uintptr_t)127;
    char *const lds_raw_bytes = reinterpret_cast<char *>(lds_raw);
    // (pointer arithmetic on the kernel argument, so that the loads stay global_load: an integer cast to a pointer gives flat_load)
    const char *const data128 = reinterpret_cast<const char *>(data) - (base & 127);
    const int j_safe = n_vec_row / 8 - 1;                        // lines 0 .. j_safe lie inside the row whatever its phase
    // per staging pass: address of this lane's vector of the row's line 0, its LDS slot, and whether the OLDER of the two
    // lines in the registers is the one this lane contributes to an image (v >= ph)
#define S3H_DECL(P)                                                                                              \
    const char *ptr##P; /* this lane's vector of the row's line `chunk` (the loop's current image): advanced by two lines per   \
                           iteration, so that the lines an iteration asks for sit at IMMEDIATE offsets 256 / 384 from it */  \
    uint32_t lds##P;  /* LDS byte address of this lane's slot of the row image */                                \
    bool old##P;      /* this lane takes its vector from the OLDER of the two lines in the registers */          \
    V preA##P, preB##P;                                                                                          \
    {                                                                                                            \
        const uintptr_t a_ = base + (uint64_t)(uint32_t)rows[r_begin + min(P * RPP + srow, n_r - 1)] * stride_bytes; \
        const int ph_ = (int)(a_ >> 4) & 7;                                                                      \
        ptr##P = data128 + (((a_ & ~(uintptr_t)127) - base128) + 16u * (unsigned)svec) + (int64_t)chunk0 * 128;   \
        old##P = svec >= ph_;                                                                                    \
        /* (passes beyond the tile's last row hold a copy of it and store that copy where the row itself goes) */  \
        lds##P = (uint32_t)(min(P * RPP + srow, n_r - 1) * 8 + ((svec - ph_) & 7)) * 16u;                         \
    }
    S3_REP16(S3H_DECL)
    // line J of every row of this lane's passes -> register set SET.  Past j_safe the lane's vector index counted from the row
    // start (J * 8 + svec - ph) is clamped to the row's last vector.
#define S3H_LOAD_FAST(SET, P) pre##SET##P = *reinterpret_cast<const V *>(ptr##P + off_);
#define S3H_LOAD_SAFE(SET, P)                                                                                    \
    {                                                                                                            \
        const int s_ = (int)(lds##P >> 4) & 7;           /* (svec - ph) mod 8 */                                  \
        const int d_ = s_ <= svec ? s_ : s_ - 8;         /* svec - ph */                                          \
        const int over_ = max(0, (J_) * 8 + d_ - last_vec);                                                      \
        pre##SET##P = *reinterpret_cast<const V *>(ptr##P + off_ - 16 * (int64_t)over_);                         \
    }
#define S3H_LOAD_FAST_A(P) S3H_LOAD_FAST(A, P)
#define S3H_LOAD_FAST_B(P) S3H_LOAD_FAST(B, P)
#define S3H_LOAD_SAFE_A(P) S3H_LOAD_SAFE(A, P)
#define S3H_LOAD_SAFE_B(P) S3H_LOAD_SAFE(B, P)
    // line J of every row of this lane's passes -> register set SET; AHEAD = J - (the loop's current image), a compile-time 0 .. 3.
    // Past j_safe the lane's vector index counted from the row start (J * 8 + svec - ph) is clamped to the row's last vector.
#define S3H_ISSUE(SET, J, AHEAD)                                 \
    do {                                                         \
        constexpr int off_ = (AHEAD) * 128;                      \
        const int J_ = (J);                                      \
        if (J_ <= j_safe) {                                      \
            S3_REP16(S3H_LOAD_FAST_##SET)                        \
        } else {                                                 \
            S3_REP16(S3H_LOAD_SAFE_##SET)                        \
        }                                                        \
    } while (0)
#define S3H_ADVANCE(P) ptr##P += 256;
    // image c from (older set = line c, newer set = line c + 1)
    // One LDS store per row and lane behind a select of the older / newer line's vector.  (Two stores under complementary
    // lane masks instead -- no v_cndmask -- were built and measured: 3.85 against 3.51 ms; the LDS write port costs more than
    // the four vector-ALU instructions.)
#define S3H_STORE2(P, OLD, NEW) *reinterpret_cast<V *>(lds_raw_bytes + lds##P) = select16(old##P, OLD, NEW);
#define S3H_STORE_AB(P) S3H_STORE2(P, preA##P, preB##P)
#define S3H_STORE_BA(P) S3H_STORE2(P, preB##P, preA##P)
#define S3H_STORES_DONE()

    // Output rows of 8 * row_len bytes start 64 bytes into a 128-byte line whenever row_len is an odd multiple of 8 (1000 snapshots:
    // 8000-byte rows, every other cell): the 256 bytes a cell writes per step then end in HALF a line whose other half follows a
    // step (7 us) later -- the L2 has turned over by then and the line goes out as two partial writes (WRITE_SIZE 7 % over the
    // output's size).  Such a cell holds its last 64 bytes (blocks 6 and 7 of the chunk: the second vector of its lanes
    // 2 and 3) back for one step, so that every step writes the whole aligned lines [c * 256 - 64, c * 256 + 192) of the row.
    bool defer = false;
    if constexpr (std::is_same<T, float>::value && HOLD != 0)
        defer = has_cell && even_rows && v0 >= 2 && (reinterpret_cast<uintptr_t>(out + cell * row_len) & 127) == 64;
    double held0 = 0.0, held1 = 0.0, held2 = 0.0, held3 = 0.0;      // (named: a loop-carried local array ends up in scratch memory)

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
        store_piece<EPV>(o + v0 * EPV, acc0, row_len - col0 - (int64_t)v0 * EPV, even_rows);
        if (EPV == 4 && HOLD == 1 && defer) {
            // (the chunk before this one is never the row's last: its piece is whole)
            if constexpr (EPV == 4) {
                if (chunk > chunk0) {
                    double *h = o - EPC + (v0 + 4) * EPV;
                    *reinterpret_cast<double2 *>(h) = make_double2(held0, held1);
                    *reinterpret_cast<double2 *>(h + 2) = make_double2(held2, held3);
                }
                held0 = acc1[0], held1 = acc1[1], held2 = acc1[2], held3 = acc1[3];
            }
        } else {
            store_piece<EPV>(o + (v0 + 4) * EPV, acc1, row_len - col0 - (int64_t)(v0 + 4) * EPV, even_rows);
        }
    };

    // Built, measured and NOT kept (r5): chunk phases tied to a chip-wide clock (time cut into slots of the 100-MHz counter, slot n
    // belongs to chunk n mod n_chunks on every workgroup, the loads of a step issued no earlier than its slot begins, a tile sweeping
    // [c_start, n) and then [0, c_start)).  Tiles that share rows then ask for the same lines within the few microseconds the L2 keeps
    // them: FETCH_SIZE -11.6 % with 10-us slots, -6.3 % at 6.0 us, -3.7 % at 5.4 us (a step takes 5.8 us on its own) -- but the launch
    // took 4.07 / 3.86 ms against 3.65: in lock-step all workgroups load at once and compute at once, and the overlap between
    // workgroups that the memory system lives on is gone.  (tools/ab_order.py, tools/pmc_order.sh; HISTORY 5.1b.)
    if (chunk0 < chunk1) {
        S3H_ISSUE(A, chunk0, 0);
        S3H_ISSUE(B, chunk0 + 1, 1);
    }
    for (int chunk = chunk0; chunk < chunk1; chunk += 2) {
        S3_REP16(S3H_STORE_AB)
        S3H_STORES_DONE();
        __syncthreads();
        if (chunk + 1 < chunk1) S3H_ISSUE(A, chunk + 2, 2);
        accumulate(chunk);
        __syncthreads();
        if (chunk + 1 >= chunk1) break;
        S3_REP16(S3H_STORE_BA)
        S3H_STORES_DONE();
        __syncthreads();
        if (chunk + 2 < chunk1) S3H_ISSUE(B, chunk + 3, 3);
        accumulate(chunk + 1);
        __syncthreads();
        S3_REP16(S3H_ADVANCE)
    }
    if (defer && chunk1 > chunk0) {              // the held piece of the run's last chunk (ragged tails: nothing beyond the row)
        const int64_t col0 = (int64_t)(chunk1 - 1) * EPC;
        if constexpr (EPV == 4) {
            const double held[EPV] = {held0, held1, held2, held3};
            store_piece<EPV>(out + cell * row_len + col0 + (v0 + 4) * EPV, held, row_len - col0 - (int64_t)(v0 + 4) * EPV, even_rows);
        }
    }
#undef S3H_ADVANCE
#undef S3H_DECL
#undef S3H_LOAD_FAST
#undef S3H_LOAD_SAFE
#undef S3H_LOAD_FAST_A
#undef S3H_LOAD_FAST_B
#undef S3H_LOAD_SAFE_A
#undef S3H_LOAD_SAFE_B
#undef S3H_ISSUE
#undef S3H_STORE2
#undef S3H_STORE_AB
#undef S3H_STORE_BA
#undef S3H_STORES_DONE
}

// Rows that start on element boundaries only (a dense [N, n_comp * T] batch read where it lies: 25 fp32 snapshots make
// 100-byte rows).  A 16-byte load that is not 16-byte aligned runs at a quarter of the rate on this chip (measured: 0.69 ms
// against 0.17 ms for the pitched copy of the same batch), so the lanes load ALIGNED vectors -- the eight that cover the
// 7-vector chunk [a, a + 112) of the row, starting at a rounded down to 16 bytes -- and the shift by (a mod 16) / 4 dwords
// is undone on the way into LDS: four ds_write_b32 per lane at shifted positions (as cheap as the one ds_write_b128 they
// replace) into a row image of 9 vectors, one vector of slack in front.  Nothing outside the 16-byte blocks that hold the
// row's own bytes is read (aligned blocks never straddle a page).
// LPR lanes stage a row (one 16-byte vector each), NPASS passes of 256 / LPR rows cover a tile's <= 512 rows
struct StreamLayoutAligned   { static constexpr int CHUNK_VECS = 8, LP = PL_LP, VOFF = 0, LPR = 8, NPASS = 16; };
struct StreamLayoutUnaligned { static constexpr int CHUNK_VECS = 7, LP = 9, VOFF = 1, LPR = 8, NPASS = 16; };
// rows of at most four vectors (16 fp32 snapshots: SURVEY C4's batches): four lanes per row, eight passes -- half the gathers per
// lane and step of the eight-vector layout, whose upper four lanes would fetch a dummy vector each
struct StreamLayoutNarrow    { static constexpr int CHUNK_VECS = 4, LP = 4, VOFF = 0, LPR = 4, NPASS = 8; };

// Short and medium batches (the reference exports cylinder3D in batches of 25 snapshots, examples/s3_for_cylinder3D_Re3900.py:
// 28-69 -> utils.py:204: rows of 100 or 300 bytes): a tile has one to a few column chunks, so the start-up of a tile --
// its row ids, its weights, the first row segments: three dependent round trips to HBM -- is most of its life, and the
// chunk kernel above, two workgroups per CU each waiting for its own start-up, moves 3.1 TB/s.  Here the workgroups are
// PERSISTENT (two per CU) and walk a flat sequence of (tile, chunk) steps with everything the next step needs already
// on its way while the current one is accumulated:
//   * row segments of step s+1: sixteen named registers per lane, issued BETWEEN the neighbour pairs of step s's accumulate phase
//     (r4; before: all sixteen right after the barrier.  A CU accepts only so many line requests at a time: the sixteen gathers of a
//     lane took 2.4-3.7 us to issue, during which the wavefront -- in order -- could not start its conversions and FMAs; s_memrealtime
//     stamps, tools/stream_phases.sh.  Spread over the phase the requests drain while the wavefront computes);
//   * weights / positions of the next tile: in REGISTERS, shared by the four lanes of a cell (each lane loads every fourth
//     entry of its cell, the DPP quad broadcasts them when the neighbour's turn comes) -- LDS holds row data only, so there
//     is nothing to double-buffer there;
//   * row ids of the tile after next: two coalesced loads per lane, handed to the staging lanes through 2 KiB of LDS when
//     the issue pointer enters that tile;
//   * (r4) what the kernel needs to know of the tiles ahead -- first row, rows, first cell, cells -- comes from a descriptor list
//     in the workgroup's own order (plan_schedule), fetched a tile early with a VECTOR load: looked up through the tile id it was
//     a chain of dependent scalar loads in front of the accumulate phase, and a scalar load in flight turns every LDS wait of
//     that phase into lgkmcnt(0).
// s_waitcnt vmcnt counts loads AND stores in issue order, and the compiler takes the smallest count over all paths that
// reach a use: a step must therefore issue the same number of vector-memory instructions on every path, or the wait for
// the next step's first row segment silently becomes a wait for this step's output stores (that version ran no faster
// than the chunk kernel).  Hence no branch around a load or a store: lanes without a cell accumulate a clamped copy,
// elements beyond the end of a row and lanes without a cell store to a per-lane dump slot, EVEN (row length) is a
// template parameter, LDS has room for 16 x 32 rows so that the staging stores need no bounds check.
// The workgroups of an XCD take that XCD's run of the Hilbert-ordered tiles round robin (plan_schedule: explicit tile lists,
// fixed at plan creation): tiles in flight at the same time are neighbours in space, so most of a tile's halo is served by
// the XCD's L2.  Same arithmetic (f64 FMA in neighbour order), same results as every other variant.
// neighbours M, M+1, ... of this lane's cell, two per step.  K is a compile-time constant so that the whole accumulate phase is
// ONE basic block (with a run-time k every neighbour sits behind its own branch), and the LDS reads of the next pair are
// issued before the current pair is accumulated -- left to itself the compiler reads a neighbour's two vectors into the same
// registers every time and waits for them at once: 26 exposed LDS round trips per step.
template <int M, int K, bool BOTH, typename T, typename LAY>
__device__ __forceinline__ void stream_read_pair(const int (&pq)[(K + 3) / 4], const typename Vec16<T>::type *__restrict__ s_data,
                                                 int v0, typename Vec16<T>::type (&buf)[4]) {
    if constexpr (M < K) {
        const int pos = quad_bcast_i32<M % 4>(pq[M / 4]);
        buf[0] = s_data[pos * LAY::LP + LAY::VOFF + v0];
        if constexpr (BOTH) buf[1] = s_data[pos * LAY::LP + LAY::VOFF + v0 + 4];
    }
    if constexpr (M + 1 < K) {
        const int pos = quad_bcast_i32<(M + 1) % 4>(pq[(M + 1) / 4]);
        buf[2] = s_data[pos * LAY::LP + LAY::VOFF + v0];
        if constexpr (BOTH) buf[3] = s_data[pos * LAY::LP + LAY::VOFF + v0 + 4];
    }
}

// acc0 += wm * a, acc1 += wm * b (element-wise, f64 FMA).  fp32 data: the conversions run four instructions ahead of the
// FMAs that consume them (written as volatile asm, which the compiler keeps in this order: scheduled freely it converts
// into one register pair and multiplies from it in the very next instruction -- a dependent f64 pair every other
// instruction, 51 % of the wave's cycles stalled at issue).
__device__ __forceinline__ double cvt_f64_asm(float x) {
    double d;
    asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(d) : "v"(x));
    return d;
}
__device__ __forceinline__ void fmac_f64_asm(double &acc, double wm, double x) {
    asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(wm), "v"(x));
}
template <bool BOTH>
__device__ __forceinline__ void stream_fma_row(double wm, const float4 &a, const float4 &b, double (&acc0)[4], double (&acc1)[4]) {
    if constexpr (BOTH) {
        double t0 = cvt_f64_asm(a.x), t1 = cvt_f64_asm(b.x), t2 = cvt_f64_asm(a.y), t3 = cvt_f64_asm(b.y);
        fmac_f64_asm(acc0[0], wm, t0); t0 = cvt_f64_asm(a.z);
        fmac_f64_asm(acc1[0], wm, t1); t1 = cvt_f64_asm(b.z);
        fmac_f64_asm(acc0[1], wm, t2); t2 = cvt_f64_asm(a.w);
        fmac_f64_asm(acc1[1], wm, t3); t3 = cvt_f64_asm(b.w);
        fmac_f64_asm(acc0[2], wm, t0);
        fmac_f64_asm(acc1[2], wm, t1);
        fmac_f64_asm(acc0[3], wm, t2);
        fmac_f64_asm(acc1[3], wm, t3);
    } else {
        const double t0 = cvt_f64_asm(a.x), t1 = cvt_f64_asm(a.y), t2 = cvt_f64_asm(a.z), t3 = cvt_f64_asm(a.w);
        fmac_f64_asm(acc0[0], wm, t0);
        fmac_f64_asm(acc0[1], wm, t1);
        fmac_f64_asm(acc0[2], wm, t2);
        fmac_f64_asm(acc0[3], wm, t3);
    }
}
template <bool BOTH>
__device__ __forceinline__ void stream_fma_row(double wm, const double2 &a, const double2 &b, double (&acc0)[2], double (&acc1)[2]) {
    acc0[0] = fma(wm, a.x, acc0[0]);
    acc0[1] = fma(wm, a.y, acc0[1]);
    if constexpr (BOTH) {
        acc1[0] = fma(wm, b.x, acc1[0]);
        acc1[1] = fma(wm, b.y, acc1[1]);
    }
}

// BOTH = false: the upper four vectors of the chunk lie beyond the end of the row (the last chunk of a ragged row) -- nobody
// reads or accumulates them
// `issue(slot)`: the row-segment loads of the NEXT step that belong to neighbour pair `slot` -- the sixteen loads of a lane are spread
// over the pairs of the accumulate phase instead of standing in front of it: a CU accepts only so many line requests at a time, the
// issue of sixteen gathers per lane blocks for 2-4 us (s_memrealtime stamps, tools/stream_phases.sh) and the wavefront, in-order,
// cannot start its conversions and FMAs behind it; between the pairs the requests drain while the wavefront computes.
template <int M, int K, bool BOTH, typename T, typename LAY, typename ISSUE>
__device__ __forceinline__ void stream_accumulate(const double (&wq)[(K + 3) / 4], const int (&pq)[(K + 3) / 4],
                                                  const typename Vec16<T>::type *__restrict__ s_data, int v0,
                                                  typename Vec16<T>::type (&cur)[4], typename Vec16<T>::type (&nxt)[4],
                                                  double (&acc0)[Vec16<T>::N], double (&acc1)[Vec16<T>::N], ISSUE &issue) {
    if constexpr (M < K) {
        stream_read_pair<M + 2, K, BOTH, T, LAY>(pq, s_data, v0, nxt);
        issue(std::integral_constant<int, M / 2>{});
        __builtin_amdgcn_sched_barrier(0);
        stream_fma_row<BOTH>(quad_bcast_f64<M % 4>(wq[M / 4]), cur[0], cur[1], acc0, acc1);
        if constexpr (M + 1 < K) stream_fma_row<BOTH>(quad_bcast_f64<(M + 1) % 4>(wq[(M + 1) / 4]), cur[2], cur[3], acc0, acc1);
        __builtin_amdgcn_sched_barrier(0);
        stream_accumulate<M + 2, K, BOTH, T, LAY>(wq, pq, s_data, v0, nxt, cur, acc0, acc1, issue);
    }
}

template <typename T, int K, bool ALIGNED, bool EVEN, bool NARROW = false>
__global__ void __launch_bounds__(256, 2)
interp_planned_stream_kernel(const int32_t *__restrict__ perm, const int32_t *__restrict__ rows,
                             const uint16_t *__restrict__ pl, const double *__restrict__ wl /*lane order: lane_weights_kernel*/,
                             const T *__restrict__ data, int64_t row_len, int64_t in_stride, double *__restrict__ out,
                             double *__restrict__ dump, const int32_t *__restrict__ sched_begin,
                             const int4 *__restrict__ sched_desc, int n_chunks) {
    using V = typename Vec16<T>::type;
    static_assert(!NARROW || ALIGNED, "the narrow layout takes rows on 16-byte boundaries");
    using LAY = std::conditional_t<NARROW, StreamLayoutNarrow, std::conditional_t<ALIGNED, StreamLayoutAligned, StreamLayoutUnaligned>>;
    constexpr int EPV = Vec16<T>::N;
    constexpr int EPC = LAY::CHUNK_VECS * EPV;       // elements of a row per step
    constexpr int BLOCK = 256, RPP = BLOCK / LAY::LPR, NPASS = LAY::NPASS;
    constexpr int KQ = (K + 3) / 4, k = K;           // every lane of a quad holds KQ entries of its cell's tables
    extern __shared__ float4 lds_raw[];
    V *s_data = reinterpret_cast<V *>(lds_raw);                                           // [NPASS * RPP][LP] 16-byte vectors
    int32_t *s_ids = reinterpret_cast<int32_t *>(lds_raw + (size_t)NPASS * RPP * LAY::LP);  // [2 * BLOCK] row ids of the issue tile

    const int tid = threadIdx.x;
    // this workgroup's tiles: the descriptors sched_desc[my_begin .. my_begin + n_my) = {first row, rows, first cell, cells}
    // (plan_schedule below)
    const int my_begin = sched_begin[blockIdx.x], n_my = sched_begin[blockIdx.x + 1] - my_begin;
    if (n_my <= 0) return;
    const int4 *const my_desc = sched_desc + my_begin;
    // descriptor j of the list (clamped), fetched with a VECTOR load: it is counted by vmcnt, in order, behind loads that are waited
    // for anyway -- a scalar load would have every LDS wait of the accumulate phase wait for it (lgkmcnt, out of order)
    int lane_zero;                                   // (a zero the compiler cannot see through: keeps the address in a VGPR)
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    auto desc_at = [&](int j) { return my_desc[min(j, n_my - 1) + lane_zero]; };
    const int srow = tid / LAY::LPR, svec = tid % LAY::LPR;      // staging role: LPR lanes per row segment
    const int qcl = tid >> 2, v0 = tid & 3;          // accumulate role: 4 lanes per cell, vectors v0 and v0 + 4
    double *const dump_lane = dump + ((int64_t)blockIdx.x * BLOCK + tid) * 2;
    const uint32_t stride32 = (uint32_t)in_stride;   // (row pitch in elements < 2^31: one v_mad_u64_u32 per row address)

#define S3S_DECL(P) int rid##P = 0; V pre##P;
    S3_REP16(S3S_DECL)
    static_assert(PL_NP == 16, "S3_REP16 expands PL_NP staging passes");
    int ida = 0, idb = 0;                            // row ids of the tile after the issue tile (coalesced image)
    double wq[KQ], wqn[KQ];
    int pq[KQ];
    uint4 pqn_raw;                                   // the next tile's positions as loaded: eight 16-bit entries
    int64_t cell = 0;
    int celln = 0;                                   // (kept as loaded: a conversion here would wait for the load)
    int nc_c = 0, nc_n = 0, nr_n = 0;                // cells of the compute tile; cells / rows of the issue tile

    auto load_ids = [&](int rb, int nr) {
        ida = rows[rb + min(tid, nr - 1)];
        idb = rows[rb + min(BLOCK + tid, nr - 1)];
    };
    constexpr int NV = (KQ + 1) / 2;                 // 16-byte vectors of weights per lane
    typedef double wpair_t __attribute__((ext_vector_type(2)));
    auto load_tables = [&](int cb, int nc) {
        nc_n = nc;
        const int cl = min(qcl, nc_n - 1), lane4 = cl * 4 + v0, lanes = nc_n * 4;
        const wpair_t *wv = reinterpret_cast<const wpair_t *>(wl + (int64_t)cb * (4 * NV * 2)) + lane4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const wpair_t two = wv[j * lanes];
            wqn[2 * j] = two.x;
            if (2 * j + 1 < KQ) wqn[2 * j + 1] = two.y;
        }
        pqn_raw = reinterpret_cast<const uint4 *>(pl + (int64_t)cb * 32)[lane4];
        celln = perm[cb + cl];
    };
#define S3S_RID(P) if constexpr (P < NPASS) rid##P = s_ids[min(P * RPP + srow, nr_n - 1)];
    // ALIGNED: lane svec loads vector svec of the chunk (a dummy -- the chunk's first vector -- beyond the end of the row).
    // Otherwise: the aligned 16-byte block number svec counted from the block that holds the chunk's first byte (a dummy -- that
    // first block -- where the block lies wholly behind the chunk's last valid byte).
#define S3S_LOAD(P)                                                                                                          \
    if constexpr (P < NPASS) {                                                                                               \
        int r_ = rid##P;                                                                                                     \
        asm volatile("" : "+v"(r_));      /* the address is formed HERE: hoisted in front of the accumulate phase the sixteen \
                                             64-bit addresses would cost 32 registers */                                     \
        const T *a_ = data + (uint64_t)(uint32_t)r_ * stride32 + c0_;                                                        \
        if constexpr (ALIGNED) {                                                                                             \
            pre##P = *reinterpret_cast<const V *>(a_ + (ok_ ? svec : 0) * EPV);                                              \
        } else {                                                                                                             \
            /* (offsets from the kernel argument's pointer: an address made from an integer would be a flat_load, which  \
               counts against lgkmcnt too -- every LDS wait of the accumulate phase would wait for the prefetch) */         \
            const int mis_ = (int)(reinterpret_cast<uintptr_t>(a_) & 15);                                                    \
            const int off_ = 16 * svec - mis_;                                                                               \
            pre##P = *reinterpret_cast<const V *>(reinterpret_cast<const char *>(a_) + (off_ < (int)valid_bytes_ ? off_ : -mis_)); \
        }                                                                                                                    \
    }
#define S3S_LOAD_IF(P) if constexpr (P < NPASS && (P * NS) / NPASS == S) { S3S_LOAD(P) }
#define S3S_ISSUE(CH)                                                                    \
    do {                                                                                 \
        const int64_t c0_ = (int64_t)(CH) * EPC;                                         \
        const bool ok_ = c0_ + (int64_t)svec * EPV < row_len;                            \
        const uintptr_t valid_bytes_ = (uintptr_t)min((int64_t)EPC, row_len - c0_) * sizeof(T);   \
        (void)ok_; (void)valid_bytes_;                                                   \
        S3_REP16(S3S_LOAD)                                                               \
    } while (0)
    // (the row ids in rid* are those the segments in pre* were loaded with: ids and segments are renewed together)
#define S3S_STORE(P)                                                                                                         \
    if constexpr (P < NPASS) {                                                                                               \
        if constexpr (ALIGNED) {                                                                                             \
            s_data[(P * RPP + srow) * LAY::LP + svec] = pre##P;                                                              \
        } else {                                                                                                             \
            const T *a_ = data + (uint64_t)(uint32_t)rid##P * stride32 + store_c0_;                                          \
            const int ph_ = (int)((reinterpret_cast<uintptr_t>(a_) >> 2) & 3);                                               \
            uint32_t *row_ = reinterpret_cast<uint32_t *>(s_data) + (P * RPP + srow) * (LAY::LP * 4) + 4 + 4 * svec - ph_;   \
            const uint32_t *w_ = reinterpret_cast<const uint32_t *>(&pre##P);                                                \
            row_[0] = w_[0]; row_[1] = w_[1]; row_[2] = w_[2]; row_[3] = w_[3];                                              \
        }                                                                                                                    \
    }

    // prologue: the first tile's descriptor and ids are the exposed round trips of the workgroup
    int4 d_next;                                     // descriptor of the tile the issue pointer enters next
    int2 d_after;                                    // rows of the tile after that one (its ids are fetched a tile early)
    {
        const int4 d0 = desc_at(0);
        d_next = desc_at(1);
        const int4 d2 = desc_at(2);
        d_after = make_int2(d2.x, d2.y);
        load_ids(d0.x, d0.y);
        s_ids[tid] = ida;
        s_ids[BLOCK + tid] = idb;
        nr_n = d0.y;
        __syncthreads();
        S3_REP16(S3S_RID)
        __syncthreads();                             // (the first step may already refill s_ids)
        load_tables(d0.z, d0.w);
        if (n_my > 1) load_ids(d_next.x, d_next.y);
        S3S_ISSUE(0);
    }
    // as many stores behind these loads as behind the loads of a step of the loop, or the count of the loop's first wait
    // (the smallest over all ways into the loop) would be that of this prologue: every step would wait for its own stores
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<volatile double *>(dump_lane) = 0.0;

#ifdef S3_PROBE_STAMPS
    long long pr_sum[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    for (int j = 0; j < n_my; ++j) {
        // the tile whose steps are accumulated now: its tables arrived behind its first row segments
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            wq[i] = wqn[i];
            const uint32_t two = i < 2 ? pqn_raw.x : i < 4 ? pqn_raw.y : i < 6 ? pqn_raw.z : pqn_raw.w;
            pq[i] = (int)((i & 1) ? two >> 16 : two & 0xffffu);
        }
        cell = celln;
        nc_c = nc_n;
        const bool live = qcl < nc_c;                // lanes without a cell work on a clamped copy and store to the dump slot
        for (int c = 0; c < n_chunks; ++c) {
            const bool last_chunk = c + 1 == n_chunks;
            const bool enter_next = last_chunk && j + 1 < n_my;      // the issue pointer moves on to this workgroup's next tile
            const int64_t store_c0_ = (int64_t)c * EPC;
            (void)store_c0_;
            S3_STAMP(ts0)
            S3_REP16(S3S_STORE)
            if (enter_next) {
                s_ids[tid] = ida;
                s_ids[BLOCK + tid] = idb;
            }
            S3_STAMP(ts1)
            __syncthreads();
            S3_STAMP(ts2)
            // ONE issue site for the row segments (two would look to the compiler as if the second could overwrite registers
            // the first has loads pending for: it then waits for vmcnt(0), i.e. for the previous step's stores)
            if (enter_next) {                        // (older than the row segments: waiting for a table never waits for a segment)
                nr_n = d_next.y;
                S3_REP16(S3S_RID)
                load_tables(d_next.z, d_next.w);
                if (j + 2 < n_my) load_ids(d_after.x, d_after.y);
                d_next = desc_at(j + 2);             // consumed when the issue pointer moves on again: a tile's worth of steps away
                const int4 d3 = desc_at(j + 3);
                d_after = make_int2(d3.x, d3.y);
            }
            // the next step's row segments: issued between the neighbour pairs of the accumulate phase below
            // (unconditional: a branch per pair would cut the phase into basic blocks; the workgroup's very last step has no next
            // step and fetches chunk 0 of its own rows once more -- one step's lines in nineteen or more, fresh in the L2)
            const int64_t c0_ = (int64_t)(last_chunk ? 0 : c + 1) * EPC;
            const bool ok_ = c0_ + (int64_t)svec * EPV < row_len;
            const uintptr_t valid_bytes_ = (uintptr_t)min((int64_t)EPC, row_len - c0_) * sizeof(T);
            (void)ok_; (void)valid_bytes_;
            auto issue = [&](auto slot) __attribute__((always_inline)) {
                // (f64 input: the accumulate phase has no registers to spare -- all sixteen loads go with the first pair)
                constexpr int S = decltype(slot)::value, NS = sizeof(T) == 4 ? (K + 1) / 2 : 1;
                S3_REP16(S3S_LOAD_IF)
            };
            S3_STAMP(ts3)
            const int64_t col0 = (int64_t)c * EPC;
            double acc0[EPV], acc1[EPV];
#pragma unroll
            for (int i = 0; i < EPV; ++i) acc0[i] = acc1[i] = 0.0;
            V buf_a[4], buf_b[4];
            bool upper = false;                      // (uniform) vectors 4 .. of the chunk exist
            if constexpr (LAY::CHUNK_VECS > 4) upper = col0 + 4 * EPV < row_len;
            if (upper) {
                if constexpr (LAY::CHUNK_VECS > 4) {
                    stream_read_pair<0, K, true, T, LAY>(pq, s_data, v0, buf_a);
                    stream_accumulate<0, K, true, T, LAY>(wq, pq, s_data, v0, buf_a, buf_b, acc0, acc1, issue);
                }
            } else {
                stream_read_pair<0, K, false, T, LAY>(pq, s_data, v0, buf_a);
                stream_accumulate<0, K, false, T, LAY>(wq, pq, s_data, v0, buf_a, buf_b, acc0, acc1, issue);
            }
            __builtin_amdgcn_sched_barrier(0);
            S3_STAMP(ts4)
            // the same number of stores on every path: what must not be written goes to this lane's dump slot.  Pairs of
            // doubles; rows of odd length start on 8-byte boundaries only (element-aligned 16-byte stores) and end in a
            // single element, which one lane of the cell stores separately.
            double *const orow = out + cell * row_len;
            const int64_t e0 = col0 + (int64_t)v0 * EPV, e1 = col0 + (int64_t)(v0 + 4) * EPV;
            const bool second = v0 + 4 < LAY::CHUNK_VECS;    // (the chunk of an element-aligned batch has seven vectors)
            typedef double pair_t __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
            for (int i = 0; i < EPV; i += 2) {
                double *p0 = live && e0 + i + 1 < row_len ? orow + e0 + i : dump_lane;
                pair_t v = {acc0[i], acc0[i + 1]};
                *reinterpret_cast<pair_t *>(p0) = v;
            }
            if constexpr (LAY::CHUNK_VECS > 4) {
#pragma unroll
                for (int i = 0; i < EPV; i += 2) {
                    double *p1 = live && second && e1 + i + 1 < row_len ? orow + e1 + i : dump_lane;
                    pair_t v = {acc1[i], acc1[i + 1]};
                    *reinterpret_cast<pair_t *>(p1) = v;
                }
            }
            if constexpr (!EVEN) {
                const int64_t t = row_len - 1;       // even: the last element of the row is the first of a pair
                double tv = acc0[0];
                bool mine = t == e0;
#pragma unroll
                for (int i = 2; i < EPV; i += 2) {
                    tv = t == e0 + i ? acc0[i] : tv;
                    mine = mine || t == e0 + i;
                }
#pragma unroll
                for (int i = 0; i < EPV; i += 2) {
                    tv = second && t == e1 + i ? acc1[i] : tv;
                    mine = mine || (second && t == e1 + i);
                }
                double *pt = live && mine ? orow + t : dump_lane;
                *pt = tv;
            }
            S3_STAMP(ts5)
            __syncthreads();
#ifdef S3_PROBE_STAMPS
            {
                const long long ts6 = __builtin_amdgcn_s_memrealtime();
                pr_sum[0] += ts1 - ts0; pr_sum[1] += ts2 - ts1; pr_sum[2] += ts3 - ts2; pr_sum[3] += ts4 - ts3; pr_sum[4] += ts5 - ts4; pr_sum[5] += ts6 - ts5;
                pr_sum[6] += 1;
            }
#endif
        }
    }
#ifdef S3_PROBE_STAMPS
    if ((tid & 63) == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 8; ++i) s3_probe_stamps[(blockIdx.x * 4 + (tid >> 6)) * 8 + i] = i < 7 ? pr_sum[i] : 0;
#endif
#undef S3S_DECL
#undef S3S_RID
#undef S3S_LOAD
#undef S3S_ISSUE
#undef S3S_LOAD_IF
#undef S3S_STORE
}

#undef S3_REP16

// distinct rows a tile of `tc` cells may hold: what is left of the LDS budget (80 KiB -> two 256-thread workgroups per CU
// for tc = 64; 160 KiB -> one 512-thread workgroup per CU for tc = 128) after the tile's weights and positions
// rows of at most this many 16-byte vectors take the short-row kernel (S3_SHORT_ROW_VECS overrides, for A/B runs).
// Measured on MI355X, cylinder3D grid (461 130 cells, k = 26, fp32): 16 snapshots (4 vectors) 0.139 ms short / 0.162 ms
// chunk kernel / 0.170 ms direct gather; from 25 snapshots on the chunk kernel is faster (0.189 vs 0.282 ms)
static int short_row_vecs() {
    static const int v = [] {
        const char *e = getenv("S3_SHORT_ROW_VECS");
        return e ? atoi(e) : 4;
    }();
    return v;
}

// fewest workgroups a launch of the chunk kernel should have before the column chunks are split over blockIdx.y
// (S3_PLAN_MIN_BLOCKS overrides, for A/B runs)
// The S3_* switches of the planned launches (A/B runs, tests) are parsed ONCE, at the first launch: getenv on every launch raced with
// the interpreter's putenv from other threads.  A tool that flips a switch inside a process calls s3_debug_reload_env()
// afterwards (hipops.reload_env()) -- not while another thread launches.
struct LaunchSwitches {
    int64_t min_blocks = 2048, stream_min_tiles = 64;
    int stream_max_chunks = 24, inplace_shift = 1, shift_min_chunks = 6;
    int short_stream = -1;                      // -1 unset (decided by the table's size), 0 never, 1 always
    bool short_no_quad = false, short_lds_weights = false;
    int plan_split = 0, plan_brick = 0;         // 0 unset
    bool tail_given = false;
    int tail = 32, tail_split = 4;
    int out_hold = 0;
};
static LaunchSwitches parse_switches() {
    LaunchSwitches w;
    if (const char *e = getenv("S3_PLAN_MIN_BLOCKS")) w.min_blocks = atoll(e);
    if (const char *e = getenv("S3_STREAM_MIN_TILES")) w.stream_min_tiles = atoll(e);
    if (const char *e = getenv("S3_STREAM_MAX_CHUNKS")) w.stream_max_chunks = atoi(e);
    if (const char *e = getenv("S3_INPLACE_SHIFT")) w.inplace_shift = atoi(e);
    if (const char *e = getenv("S3_SHIFT_MIN_CHUNKS")) w.shift_min_chunks = atoi(e);
    if (const char *e = getenv("S3_SHORT_STREAM")) w.short_stream = e[0] == '1' ? 1 : 0;
    w.short_no_quad = getenv("S3_SHORT_NO_QUAD") != nullptr;
    w.short_lds_weights = getenv("S3_SHORT_LDS_WEIGHTS") != nullptr;
    if (const char *e = getenv("S3_PLAN_SPLIT")) w.plan_split = atoi(e) < 1 ? 1 : atoi(e);
    if (const char *e = getenv("S3_PLAN_BRICK")) w.plan_brick = atoi(e) < 1 ? 1 : atoi(e);
    if (const char *e = getenv("S3_PLAN_TAIL")) {
        w.tail_given = true;
        w.tail = atoi(e);
        const char *x = strchr(e, 'x');
        w.tail_split = x ? atoi(x + 1) : 4;
    }
    if (const char *e = getenv("S3_OUT_HOLD")) w.out_hold = atoi(e);
    return w;
}
static LaunchSwitches &switches() {
    static LaunchSwitches w = parse_switches();
    return w;
}
static int64_t min_blocks() { return switches().min_blocks; }

static int plan_ucap(int k, int tc) {
    const int budget = (tc == 128 ? 160 : 80) * 1024;
    int u = (budget - k * tc * (int)(sizeof(double) + sizeof(uint16_t))) / PL_SEG;
    u = (u / 16) * 16;
    const int cap = PL_NP * tc / 2;
    return u > cap ? cap : u;
}

}  // namespace s3

using namespace s3;

// the persistent kernel (interp_planned_stream_kernel) takes batches of up to this many 128-byte column chunks per row
// (S3_STREAM_MAX_CHUNKS overrides; 0 switches it off) on plans with at least S3_STREAM_MIN_TILES tiles.  24 (r4; 8 before the
// segment loads were spread over its accumulate phase): rows on the line grid, cylinder3D plan, one process -- 288 / 320 / 384 /
// 512 / 768 snapshots 1.049 / 1.046 / 1.259 / 1.767 / 2.478 ms against 1.117 / 1.090 / 1.307 / 1.841 / 2.608 with the chunk kernel;
// 1000 snapshots (32 chunks) 3.689 against 3.529: the long sweeps stay with the chunk kernel's two chunks of prefetch
static int stream_max_chunks() { return s3::switches().stream_max_chunks; }
static int64_t stream_min_tiles() { return s3::switches().stream_min_tiles; }
static int inplace_shift() { return s3::switches().inplace_shift; }   // 0: never (A/B runs), 1: rows off the 128-byte grid (default), 2: always (A/B runs)
// rows OFF the line grid of at least this many chunks take the shift kernel (whole aligned lines), shorter ones the persistent
// kernel with straddling segments.  6 (r4; 3 before): dense rows of 68 / 100 / 136 / 200 / 300 / 600 snapshots (3 / 4 / 5 / 7 / 10 /
// 19 chunks), one process: shift 0.409 / 0.535 / 0.618 / 0.842 / 1.214 / 2.176 ms, persistent 0.372 / 0.535 / 0.605 / 0.862 / 1.362 / 2.531
static int shift_min_chunks() { return s3::switches().shift_min_chunks; }
static int stream_workgroups() {
    static const int v = [] {
        const char *e = getenv("S3_STREAM_WORKGROUPS");
        if (e) return atoi(e);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 2 * cus;                                   // LDS: two workgroups per CU
    }();
    return v;
}

// (the neighbour counts of the reference's exports: 8 in 2-D, 26 in 3-D, export.py:84-85; any other k takes the chunk kernel)
static bool stream_can_take(const s3_interp_plan *p) { return p->tc == 64 && p->ucap <= 512 && (p->k == 8 || p->k == 26); }

// Tile lists of the persistent workgroups.  Workgroup b runs on XCD b % 8 (round-robin dispatch); XCD x owns the x-th eighth
// of the Hilbert-ordered tiles and hands them out IN ORDER to its workgroups; the lists are fixed here, so the kernel needs
// no atomics and can issue the loads of the next tiles' ids two tiles ahead.
static int plan_schedule(s3_interp_plan *p, hipStream_t st) {
    const size_t nt = (size_t)p->n_tiles;
    std::vector<int32_t> cb(nt + 1), rb(nt + 1);
    S3_HIP_CHECK(hipMemcpyAsync(cb.data(), p->tile_cell_begin, sizeof(int32_t) * (nt + 1), hipMemcpyDeviceToHost, st));
    S3_HIP_CHECK(hipMemcpyAsync(rb.data(), p->tile_row_begin, sizeof(int32_t) * (nt + 1), hipMemcpyDeviceToHost, st));
    S3_HIP_CHECK(hipStreamSynchronize(st));
    int wgs = stream_workgroups() / 8 * 8;
    if (wgs < 8) wgs = 8;
    const int slots = wgs / 8;
    const size_t per_xcd = (nt + 7) / 8;
    std::vector<std::vector<int32_t>> lists((size_t)wgs);
    std::vector<double> busy((size_t)slots);
    // Default: plain round robin -- the workgroups of an XCD then work on adjacent tiles at the same time and find most of a
    // tile's halo in the XCD's L2.  S3_STREAM_SCHEDULE=cost: each tile to the workgroup expected to be free first (list
    // scheduling with a byte-count cost model).  Measured on MI355X, cylinder3D grid, interleaved in one process
    // (tools/ab_plan.py): round robin 0.158 / 0.151 / 0.454 ms at 25 / 32 / 128 snapshots, cost model 0.162 / 0.153 / 0.476 --
    // the balance it buys is worth less than the locality it loses.
    const char *mode = getenv("S3_STREAM_SCHEDULE");
    const bool round_robin = !(mode && mode[0] == 'c');
    // (r5: balancing INSIDE a round -- the 64 tiles of a round, most expensive first, to the workgroups with the least work so far,
    // so that the rounds stay together in time -- measured too: 0.1344 / 0.1318 / 0.464 / 0.476 / 0.934 ms at 25 / 32 / 100 / 128 / 256
    // snapshots against 0.1317 / 0.1295 / 0.444 / 0.445 / 0.875 with plain round robin.  What round robin has is CORRELATION: workgroup
    // s always works on the tile next to workgroup s + 1's, neighbouring tiles cost about the same, so the two stay in phase and
    // share their halo through the L2; any reassignment breaks the pairs.)
    auto cost_of = [&](size_t t) { return 128.0 * (rb[t + 1] - rb[t]) + (10.0 * p->k + 256.0) * (cb[t + 1] - cb[t]) + 4096.0; };
    for (int x = 0; x < 8; ++x) {
        const size_t lo = std::min(nt, x * per_xcd), hi = std::min(nt, lo + per_xcd);
        std::fill(busy.begin(), busy.end(), 0.0);
        for (size_t t = lo; t < hi; ++t) {
            int best = 0;
            if (round_robin) {
                best = (int)((t - lo) % (size_t)slots);
            } else {
                for (int s = 1; s < slots; ++s)
                    if (busy[s] < busy[best]) best = s;                 // (first of equals: the first `slots` tiles go out in order)
            }
            busy[best] += cost_of(t);
            lists[(size_t)best * 8 + x].push_back((int32_t)t);
        }
    }
    std::vector<int32_t> begin((size_t)wgs + 1, 0), tiles;
    tiles.reserve(nt);
    for (int b = 0; b < wgs; ++b) {
        tiles.insert(tiles.end(), lists[b].begin(), lists[b].end());
        begin[b + 1] = (int32_t)tiles.size();
    }
    // what the persistent kernel needs to know of a tile, in list order: the descriptors of the tiles ahead sit at known
    // addresses and are fetched a whole step early (looked up through the tile id they were a chain of two dependent round trips
    // per tile, in front of the accumulate phase)
    std::vector<int4> desc(std::max<size_t>(nt, 1));
    for (size_t i = 0; i < tiles.size(); ++i) {
        const int32_t t = tiles[i];
        desc[i] = make_int4(rb[t], rb[t + 1] - rb[t], cb[t], cb[t + 1] - cb[t]);
    }
    // built into locals and handed to the plan only when every step has succeeded: a failure half way (a transient out-of-memory)
    // must not leave a plan whose next launch skips this function and runs with a grid of 0 workgroups and null tables
    int32_t *d_tiles = nullptr, *d_begin = nullptr;
    int4 *d_desc = nullptr;
    uint16_t *d_pl = nullptr;
    auto build = [&]() -> int {
        S3_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_tiles), sizeof(int32_t) * std::max<size_t>(nt, 1)));
        S3_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_begin), sizeof(int32_t) * ((size_t)wgs + 1)));
        S3_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_pl), sizeof(uint16_t) * (size_t)p->nc * 32));
        lane_positions_kernel<<<(unsigned)p->n_tiles, 256, 0, st>>>(p->tile_cell_begin, p->loc, p->k, d_pl);
        S3_LAUNCH_CHECK();
        S3_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_desc), sizeof(int4) * desc.size()));
        S3_HIP_CHECK(hipMemcpyAsync(d_desc, desc.data(), sizeof(int4) * desc.size(), hipMemcpyHostToDevice, st));
        S3_HIP_CHECK(hipMemcpyAsync(d_tiles, tiles.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice, st));
        S3_HIP_CHECK(hipMemcpyAsync(d_begin, begin.data(), sizeof(int32_t) * ((size_t)wgs + 1), hipMemcpyHostToDevice, st));
        S3_HIP_CHECK(hipStreamSynchronize(st));
        return S3_OK;
    };
    const int rc = build();
    if (rc != S3_OK) {
        if (d_tiles) (void)hipFree(d_tiles);
        if (d_begin) (void)hipFree(d_begin);
        if (d_pl) (void)hipFree(d_pl);
        if (d_desc) (void)hipFree(d_desc);
        return rc;
    }
    p->sched_tiles = d_tiles;
    p->sched_begin = d_begin;
    p->pl = d_pl;
    p->sched_desc = d_desc;
    p->sched_wgs = wgs;
    return S3_OK;
}

template <typename T, bool ALIGNED, bool EVEN, bool NARROW = false>
static int launch_stream_e(s3_interp_plan *p, const int32_t *rows, const void *data, int64_t row_len, int64_t in_stride,
                           double *out, hipStream_t st) {
    using LAY = std::conditional_t<NARROW, StreamLayoutNarrow, std::conditional_t<ALIGNED, StreamLayoutAligned, StreamLayoutUnaligned>>;
    constexpr int EPC = LAY::CHUNK_VECS * 16 / (int)sizeof(T);
    const int n_chunks = (int)((row_len + EPC - 1) / EPC);
    if (!p->sched_begin) {
        const int rc = plan_schedule(p, st);
        if (rc != S3_OK) return rc;
    }
    const size_t lds = (size_t)LAY::NPASS * (256 / LAY::LPR) * LAY::LP * 16 + 2 * 256 * sizeof(int32_t);
    const size_t dump_doubles = (size_t)p->sched_wgs * 256 * 2;        // one 16-byte slot per lane of the launch
    if (p->dump_doubles < dump_doubles) {
        if (p->dump) (void)hipFree(p->dump);
        p->dump = nullptr;
        p->dump_doubles = 0;
        S3_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p->dump), dump_doubles * sizeof(double)));
        p->dump_doubles = dump_doubles;
    }
#define S3_LAUNCH_STREAM(K)                                                                                                   \
    do {                                                                                                                      \
        auto kern = interp_planned_stream_kernel<T, K, ALIGNED, EVEN, NARROW>;                                                           \
        S3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                         (int)lds));                                                                          \
        kern<<<dim3((unsigned)p->sched_wgs), 256, lds, st>>>(p->perm, rows, p->pl,                                            \
                                                            p->wl, static_cast<const T *>(data), row_len, in_stride,           \
                                                            out, p->dump, p->sched_begin, p->sched_desc, n_chunks);           \
    } while (0)
    if (p->k == 8) S3_LAUNCH_STREAM(8);
    else S3_LAUNCH_STREAM(26);
#undef S3_LAUNCH_STREAM
    S3_LAUNCH_CHECK();
    return S3_OK;
}

template <typename T, bool ALIGNED, bool NARROW = false>
static int launch_stream(s3_interp_plan *p, const int32_t *rows, const void *data, int64_t row_len, int64_t in_stride,
                         double *out, hipStream_t st) {
    if (row_len & 1) return launch_stream_e<T, ALIGNED, false, NARROW>(p, rows, data, row_len, in_stride, out, st);
    return launch_stream_e<T, ALIGNED, true, NARROW>(p, rows, data, row_len, in_stride, out, st);
}

template <typename T>
static int launch_planned(s3_interp_plan *p, const int32_t *rows, int64_t n_rows, bool aligned, const void *data, int64_t row_len,
                          int64_t in_stride, double *out, hipStream_t st) {
    if (!aligned) {                              // element-aligned rows read where they lie: the persistent kernel only
        S3_REQUIRE(stream_can_take(p), "s3_interp_planned: rows that are not 16-byte aligned need k = 8 | 26 and 64-cell tiles");
        return launch_stream<T, false>(p, rows, data, row_len, in_stride, out, st);
    }
    constexpr int EPC = PL_SEG / (int)sizeof(T);
    constexpr int EPV = 16 / (int)sizeof(T);
    const int n_chunks = (int)((row_len + EPC - 1) / EPC);
    const int64_t tiles_per_xcd = (p->n_tiles + 7) / 8;
    const int64_t gx = tiles_per_xcd * 8;
    S3_REQUIRE(gx < ((int64_t)1 << 31), "s3_interp_planned: too many tiles");
    // rows off the 128-byte grid (a dense batch read where it lies) with three or more chunks are better off with whole aligned
    // lines (interp_planned_shift_kernel, below) than with the persistent kernel's straddling segments: 800-byte rows 0.850
    // against 0.906 ms, 400-byte rows 0.529 / 0.553, 272-byte rows 0.414 / 0.424 (cylinder3D grid, interleaved in one process,
    // tools/ab_inplace.py; S3_SHIFT_MIN_CHUNKS overrides)
    const bool off_line = reinterpret_cast<uintptr_t>(data) % PL_SEG != 0 || ((uint64_t)in_stride * sizeof(T)) % PL_SEG != 0;
    const bool shift_ok = p->tc == 64 && (off_line ? inplace_shift() >= 1 : inplace_shift() >= 2);
    if ((row_len + EPV - 1) / EPV > s3::short_row_vecs() && n_chunks <= stream_max_chunks() && stream_can_take(p) &&
        p->n_tiles >= stream_min_tiles() && !(off_line && shift_ok && n_chunks >= shift_min_chunks()))
        return launch_stream<T, true>(p, rows, data, row_len, in_stride, out, st);
    // (r4) rows of two to four vectors (5 .. 16 fp32 snapshots): the persistent kernel in its narrow layout -- four lanes per row,
    // eight gathers per lane and step -- unless the table is a large one read in place.  Measured, one box each: cylinder3D, 16 / 12
    // snapshots, pitched copy 0.089 / 0.088 ms against 0.105 / 0.117 with the short-row kernels, read in place (320-MB table) 0.100 /
    // 0.116 against 0.112 / 0.121; box5e7 (10 M cells, 16 snapshots) pitched copy 1.745 against 1.786 ms, but its 3.2-GB table read in
    // place 2.363 against 2.208: there five short-lived workgroups per CU hide the page-table walks of a scattered table better
    // than two persistent ones.  Rows of one vector stay with the short-row kernels (0.096 against 0.078 ms).  S3_SHORT_STREAM=0 / 1:
    // never / always, for A/B runs.
    {
        const int vpr_ = (int)((row_len + EPV - 1) / EPV);
        const int sw = s3::switches().short_stream;
        const bool in_place = rows != p->rows;
        const bool big_table = in_place && (uint64_t)n_rows * (uint64_t)in_stride * sizeof(T) > ((uint64_t)1 << 30);
        const bool want = sw >= 0 ? sw == 1 : !big_table;
        if (vpr_ >= 2 && vpr_ <= 4 && want && stream_can_take(p) && p->n_tiles >= stream_min_tiles())
            return launch_stream<T, true, true>(p, rows, data, row_len, in_stride, out, st);
    }
    if ((row_len + EPV - 1) / EPV <= s3::short_row_vecs() && p->tc == 64) {
        const int vpr = (int)((row_len + EPV - 1) / EPV);
        if (vpr == 4 && p->ucap * vpr <= 256 * 8 && p->k <= 32 && !s3::switches().short_no_quad && !s3::switches().short_lds_weights) {
            // four vectors per row: the lanes of a cell are a DPP quad and share the loads of its weights / positions
            const size_t lds = (size_t)p->ucap * vpr * 16;
#define S3_LAUNCH_SHORT_QUAD(KQ)                                                                                                 \
    do {                                                                                                                         \
        auto kern = interp_planned_short_quad_kernel<T, KQ>;                                                                     \
        kern<<<dim3((unsigned)gx), 256, lds, st>>>(p->perm, p->tile_cell_begin, p->tile_row_begin, rows, p->loc, p->wp, p->k, \
                                                   static_cast<const T *>(data), row_len, in_stride, out, p->n_tiles,            \
                                                   tiles_per_xcd);                                                               \
    } while (0)
            if (p->k <= 8) S3_LAUNCH_SHORT_QUAD(2);
            else if (p->k <= 28) S3_LAUNCH_SHORT_QUAD(7);
            else S3_LAUNCH_SHORT_QUAD(8);
#undef S3_LAUNCH_SHORT_QUAD
            S3_LAUNCH_CHECK();
            return S3_OK;
        }
        if (vpr <= 4 && p->ucap * vpr <= 256 * 8 && p->k <= 32 && !s3::switches().short_lds_weights) {
            // one lane per (cell, vector) pair, weights in registers
            const size_t lds = (size_t)p->ucap * vpr * 16;
#define S3_LAUNCH_SHORT_REG(KM)                                                                                                  \
    do {                                                                                                                         \
        auto kern = interp_planned_short_reg_kernel<T, KM>;                                                                      \
        kern<<<dim3((unsigned)gx), 256, lds, st>>>(p->perm, p->tile_cell_begin, p->tile_row_begin, rows, p->loc, p->wp, p->k, \
                                                   static_cast<const T *>(data), row_len, in_stride, out, p->n_tiles,            \
                                                   tiles_per_xcd, vpr);                                                          \
    } while (0)
            if (p->k <= 8) S3_LAUNCH_SHORT_REG(8);
            else if (p->k <= 26) S3_LAUNCH_SHORT_REG(26);
            else S3_LAUNCH_SHORT_REG(32);
#undef S3_LAUNCH_SHORT_REG
            S3_LAUNCH_CHECK();
            return S3_OK;
        }
        const int pitch = vpr < 8 ? vpr : 8;
        const size_t lds = (size_t)p->ucap * pitch * 16 + (size_t)p->k * p->tc * (sizeof(double) + sizeof(uint16_t)) +
                           (size_t)p->tc * sizeof(int32_t);
        auto kern = interp_planned_short_kernel<T, 64>;
        S3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<dim3((unsigned)gx), 256, lds, st>>>(p->perm, p->tile_cell_begin, p->tile_row_begin, rows, p->loc, p->wp, p->k,
                                                   p->ucap, static_cast<const T *>(data), row_len, in_stride, out, p->n_tiles,
                                                   tiles_per_xcd, vpr, pitch);
        S3_LAUNCH_CHECK();
        return S3_OK;
    }
    // The column chunks of a tile are split into runs over several workgroups when there are too few tiles to fill the chip
    // (S3_PLAN_MIN_BLOCKS), or on request (S3_PLAN_SPLIT = runs per tile, S3_PLAN_BRICK = tiles per XCD and brick; brick_map above).
    // One workgroup per tile over ALL chunks was fastest in round 1 (MI355X, cylinder3D workload: 3.7 ms vs 4.4 ms with 4 chunks per
    // workgroup, runs in 2-D grid order)
    int gy = 1;
    while (gx * gy < s3::min_blocks() && gy < n_chunks) gy *= 2;
    const LaunchSwitches &sw_ = s3::switches();
    if (sw_.plan_split > 0) gy = sw_.plan_split;
    if (gy > n_chunks) gy = n_chunks;
    if (gy < 1) gy = 1;
    const int chunks_per_block = (n_chunks + gy - 1) / gy;
    gy = (n_chunks + chunks_per_block - 1) / chunks_per_block;
    int brick = (int)std::min<int64_t>(tiles_per_xcd, 1 << 30);
    if (sw_.plan_brick > 0) brick = std::max(1, std::min(brick, sw_.plan_brick));
    // a finer grain for the last tiles of every XCD's share (tail_map): only where a tile is swept by ONE workgroup and a launch
    // has several rounds of tiles per slot to drain (S3_PLAN_TAIL="<tiles per XCD>x<runs>", 0 = off)
    int tail = 0, tail_split = 1;
    if (gy == 1 && (sw_.tail_given || (n_chunks >= 8 && tiles_per_xcd >= 4 * 64))) {
        tail = 32, tail_split = 4;          // (MI355X, cylinder3D, interleaved in one process: off 3.768 ms, 64x4 3.749, 32x4 3.736, 128x4 3.789, 64x8 3.781)
        if (sw_.tail_given) tail = sw_.tail, tail_split = sw_.tail_split;
        if (tail > tiles_per_xcd) tail = (int)tiles_per_xcd;
        if (tail_split > n_chunks) tail_split = n_chunks;
        if (tail < 1 || tail_split < 2) tail = 0, tail_split = 1;
    }
    const int64_t n_wg = tail > 0 ? 8 * (tiles_per_xcd + (int64_t)tail * (tail_split - 1)) : gx * gy;
    S3_REQUIRE(n_wg < ((int64_t)1 << 31), "s3_interp_planned: too many workgroups");
    const size_t lds = (size_t)p->ucap * PL_SEG + (size_t)p->k * p->tc * (sizeof(double) + sizeof(uint16_t));
    dim3 grid((unsigned)n_wg);
    // rows that do not start on 128-byte boundaries (a dense batch read where it lies): whole aligned lines per load, the
    // per-row phase undone on the way into LDS (S3_INPLACE_SHIFT=0: the kernel below with straddling segments, for A/B runs)
    if (shift_ok) {
        // S3_OUT_HOLD=1: whole-line output stores (HOLD = 1 above).  They bring WRITE_SIZE down to the output's size (3.955 -> 3.690 GB
        // per launch at 1000 snapshots, traffic 1.31 -> 1.29 x algorithmic) but cost the launch 0.2-0.9 % (3.564 against 3.531 ms,
        // 3.488 / 3.456, 3.575 / 3.569 on three boxes, interleaved in one process): the counter tallies two partial write-backs of a
        // line as more than the line, the DRAM bursts are the same, and the divergent store path is not free.  Off by default.
        // (A form with ONE store sequence for all lanes behind selects: 3.588 ms -- worse than the branch.)
        const int hold = sw_.out_hold;
#define S3_LAUNCH_SHIFT(H)                                                                                                       \
    do {                                                                                                                         \
        auto kern = interp_planned_shift_kernel<T, H>;                                                                           \
        S3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        kern<<<grid, 256, lds, st>>>(p->perm, p->tile_cell_begin, p->tile_row_begin, rows, p->loc, p->wp, p->k, p->ucap,        \
                                     static_cast<const T *>(data), row_len, in_stride, out, p->n_tiles, tiles_per_xcd,           \
                                     chunks_per_block, n_chunks, brick, gy, tail, tail_split);                                   \
    } while (0)
        if (hold == 0 || !std::is_same<T, float>::value) S3_LAUNCH_SHIFT(0);
        else S3_LAUNCH_SHIFT(1);
#undef S3_LAUNCH_SHIFT
        S3_LAUNCH_CHECK();
        return S3_OK;
    }
    if (p->tc == 128) {
        auto kern = interp_planned_kernel<T, 128>;
        S3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, 512, lds, st>>>(p->perm, p->tile_cell_begin, p->tile_row_begin, rows, p->loc, p->wp, p->k, p->ucap,
                                     static_cast<const T *>(data), row_len, in_stride, out, p->n_tiles, tiles_per_xcd,
                                     chunks_per_block, n_chunks, brick, gy, tail, tail_split);
    } else {
        auto kern = interp_planned_kernel<T, 64>;
        S3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, 256, lds, st>>>(p->perm, p->tile_cell_begin, p->tile_row_begin, rows, p->loc, p->wp, p->k, p->ucap,
                                     static_cast<const T *>(data), row_len, in_stride, out, p->n_tiles, tiles_per_xcd,
                                     chunks_per_block, n_chunks, brick, gy, tail, tail_split);
    }
    S3_LAUNCH_CHECK();
    return S3_OK;
}

// ---- yardstick (measurement aid of bench.py, not on any product path) ------------------------------------------------------
// The LOADS of a tile plan and nothing else: one workgroup per tile at the headline kernel's occupancy (256 threads, two per CU, the
// same dynamic LDS), eight lanes per 128-byte line, sixteen passes of 32 rows, whole aligned lines of the rows where they lie, two
// register sets of sixteen vectors rolling -- the shift kernel's load schedule with the LDS image, the conversions, the FMAs and the
// stores taken away.  What it measures is the time the memory system needs for the bytes this plan STAGES (halo included), i.e. the
// floor of the tiling on this table.  VARIANT 0: every visit of a row fetches ONE line (sets = {all rows} x {line c}, {all rows} x
// {line c + 1}: the schedule of interp_planned_shift_kernel).  VARIANT 1: every visit fetches TWO consecutive lines = 256 contiguous
// bytes, one address translation (sets = {rows of passes 0-7} x {lines c, c + 1}, {rows of passes 8-15} x {lines c, c + 1}): the
// same bytes, the same number of loads per issue and in flight -- the only difference is which lines share an issue.  That isolates
// what "256 bytes of a row per visit"  can be worth; TCP_UTCL1_TRANSLATION_MISS counts the translations.
template <int VARIANT>
__global__ void __launch_bounds__(256, 2)
plan_loads_kernel(const int32_t *__restrict__ tile_row_begin, const int32_t *__restrict__ rows, const char *__restrict__ data,
                  uint64_t stride_bytes, uint64_t row_bytes, int n_lines, int64_t n_tiles, int64_t tiles_per_xcd, float *__restrict__ sink) {
    extern __shared__ float4 lds_raw[];
    const int64_t tile = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);      // XCD-aware, tile order (brick_map, n_split = 1)
    if (tile >= n_tiles) return;
    const int r_begin = tile_row_begin[tile], n_r = tile_row_begin[tile + 1] - r_begin;
    const int srow = threadIdx.x >> 3, svec = threadIdx.x & 7;
    const uintptr_t base = reinterpret_cast<uintptr_t>(data);
    const char *const data128 = data - (base & 127);
    const char *ptr[16];
    int last[16];                                    // last line that holds bytes of the row (lines beyond it are clamped to it)
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const uintptr_t a = base + (uint64_t)(uint32_t)rows[r_begin + min(p * 32 + srow, n_r - 1)] * stride_bytes;
        ptr[p] = data128 + ((a & ~(uintptr_t)127) - (base & ~(uintptr_t)127)) + 16u * (unsigned)svec;
        last[p] = (int)(((a & 127) + row_bytes - 1) >> 7);
    }
    float acc = 0.f;
    float4 x[16], y[16];
    auto line = [&](int p, int j) { return *reinterpret_cast<const float4 *>(ptr[p] + (int64_t)min(j, last[p]) * 128); };
    if constexpr (VARIANT == 0) {
#pragma unroll
        for (int p = 0; p < 16; ++p) x[p] = line(p, 0);
#pragma unroll
        for (int p = 0; p < 16; ++p) y[p] = line(p, 1);
        for (int c = 0; c < n_lines; c += 2) {
#pragma unroll
            for (int p = 0; p < 16; ++p) acc += x[p].x + x[p].w;
            if (c + 2 < n_lines) {
#pragma unroll
                for (int p = 0; p < 16; ++p) x[p] = line(p, c + 2);
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) acc += y[p].x + y[p].w;
            if (c + 3 < n_lines) {
#pragma unroll
                for (int p = 0; p < 16; ++p) y[p] = line(p, c + 3);
            }
        }
    } else {
#pragma unroll
        for (int p = 0; p < 8; ++p) x[2 * p] = line(p, 0), x[2 * p + 1] = line(p, 1);
#pragma unroll
        for (int p = 0; p < 8; ++p) y[2 * p] = line(p + 8, 0), y[2 * p + 1] = line(p + 8, 1);
        for (int c = 0; c < n_lines; c += 2) {
#pragma unroll
            for (int p = 0; p < 16; ++p) acc += x[p].x + x[p].w;
            if (c + 2 < n_lines) {
#pragma unroll
                for (int p = 0; p < 8; ++p) x[2 * p] = line(p, c + 2), x[2 * p + 1] = line(p, c + 3);
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) acc += y[p].x + y[p].w;
            if (c + 2 < n_lines) {
#pragma unroll
                for (int p = 0; p < 8; ++p) y[2 * p] = line(p + 8, c + 2), y[2 * p + 1] = line(p + 8, c + 3);
            }
        }
    }
    if (acc == 123.456f) sink[0] = acc + lds_raw[threadIdx.x].x;          // never true: keeps the loads (and the LDS allocation)
}

extern "C" {

#ifdef S3_PROBE_STAMPS
int s3_probe_read(long long *h, int n) { return (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(s3_probe_stamps), sizeof(long long) * n); }
#endif

void s3_interp_plan_destroy(s3_interp_plan *p) {
    if (!p) return;
    if (p->perm) (void)hipFree(p->perm);
    if (p->tile_cell_begin) (void)hipFree(p->tile_cell_begin);
    if (p->tile_row_begin) (void)hipFree(p->tile_row_begin);
    if (p->rows) (void)hipFree(p->rows);
    if (p->loc) (void)hipFree(p->loc);
    if (p->wp) (void)hipFree(p->wp);
    if (p->dump) (void)hipFree(p->dump);
    if (p->rows_src) (void)hipFree(p->rows_src);
    if (p->sched_begin) (void)hipFree(p->sched_begin);
    if (p->sched_tiles) (void)hipFree(p->sched_tiles);
    if (p->sched_desc) (void)hipFree(p->sched_desc);
    if (p->wl) (void)hipFree(p->wl);
    if (p->pl) (void)hipFree(p->pl);
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

    s3_interp_plan *p = new s3_interp_plan();
    p->nc = nc; p->k = k; p->ucap = plan_ucap(k, PL_TC); p->tc = PL_TC; p->n_src = n_src;
    const int rc = build_plan_tables(d_idx, nc, k, n_src, d_centers, dim, PL_TC, p->ucap, st, p);
    if (rc != S3_OK) {
        s3_interp_plan_destroy(p);
        return rc;
    }
    if (stream_can_take(p)) {                // tile lists of the persistent kernel
        const int rs = plan_schedule(p, st);
        if (rs != S3_OK) {
            s3_interp_plan_destroy(p);
            return rs;
        }
    }
    *out = p;
    return S3_OK;
} catch (const std::exception &e) {      // host-side allocations of the builder
    s3::set_error("s3_interp_plan_create: %s", e.what());
    return S3_ENOMEM;
}

int s3_interp_plan_info(const s3_interp_plan *p, int64_t *n_tiles, int64_t *total_rows) {
    S3_REQUIRE(p != nullptr, "s3_interp_plan_info: null plan");
    if (n_tiles) *n_tiles = p->n_tiles;
    if (total_rows) *total_rows = p->total_rows;
    return S3_OK;
}

// Leaf-cell shards for `world` ranks: the plan's tiles -- cells in Hilbert order, so a run of tiles is a
// spatially compact blob -- are cut into `world` consecutive runs of (nearly) equal cost, the cost of a tile being what
// the interpolation kernel moves for it per snapshot: 4 bytes per staged source row (fp32; the halo of a tile counts,
// that is the point) + per cell 8 bytes of output and a measured equivalent of its accumulate work.  h_cuts[r] .. h_cuts[r+1] are positions in the
// plan's processing order (d_order, a copy of `perm`): rank r owns the cells d_order[h_cuts[r] .. h_cuts[r+1]).
// Equal cell counts in creation order (the survey's first suggestion) leave the slowest of 8 ranks with 1.8x the
// mean time on the cylinder3D grid: the early, coarse cells reference 26 distinct rows each, the fine ones share theirs.
// cost of tile t per snapshot: 4 bytes per staged row; per cell 8 bytes of output + the accumulate phase of its k neighbours,
// which is not hidden behind the row traffic in tiles full of cells: a fit of launch times of 15 shards of the cylinder3D
// grid (tools/shard_probe.py: t = a * rows + b * cells + c) gives b / a = 3.6 at k = 26
static int plan_tile_costs(const s3_interp_plan *p, hipStream_t st, int32_t *d_order, std::vector<int32_t> &cb, std::vector<double> &acc) {
    const size_t nt = (size_t)p->n_tiles;
    std::vector<int32_t> rb(nt + 1);
    cb.assign(nt + 1, 0);
    S3_HIP_CHECK(hipMemcpyAsync(cb.data(), p->tile_cell_begin, sizeof(int32_t) * (nt + 1), hipMemcpyDeviceToHost, st));
    S3_HIP_CHECK(hipMemcpyAsync(rb.data(), p->tile_row_begin, sizeof(int32_t) * (nt + 1), hipMemcpyDeviceToHost, st));
    if (d_order)
        S3_HIP_CHECK(hipMemcpyAsync(d_order, p->perm, sizeof(int32_t) * (size_t)p->nc, hipMemcpyDeviceToDevice, st));
    S3_HIP_CHECK(hipStreamSynchronize(st));
    const double ROW = 4.0, CELL = 8.0 + 6.0 * p->k / 26.0;
    acc.assign(nt + 1, 0.0);
    for (size_t t = 0; t < nt; ++t)
        acc[t + 1] = acc[t] + ROW * (rb[t + 1] - rb[t]) + CELL * (cb[t + 1] - cb[t]);
    return S3_OK;
}

int s3_interp_plan_partition(const s3_interp_plan *p, int world, int32_t *d_order, int64_t *h_cuts, s3_stream stream) try {
    S3_REQUIRE(p != nullptr && h_cuts != nullptr, "s3_interp_plan_partition: null argument");
    S3_REQUIRE(world >= 1, "s3_interp_plan_partition: world=%d", world);
    const size_t nt = (size_t)p->n_tiles;
    std::vector<int32_t> cb;
    std::vector<double> acc;
    const int rc = plan_tile_costs(p, as_stream(stream), d_order, cb, acc);
    if (rc != S3_OK) return rc;
    h_cuts[0] = 0;
    size_t t = 0;
    for (int r = 1; r < world; ++r) {
        const double goal = acc[nt] * r / world;
        while (t < nt && acc[t + 1] <= goal) ++t;                              // tiles [.., t) are below the goal
        if (t < nt && goal - acc[t] > acc[t + 1] - goal) ++t;                  // the nearer tile boundary
        h_cuts[r] = cb[t];
    }
    h_cuts[world] = p->nc;
    return S3_OK;
} catch (const std::exception &e) {
    s3::set_error("s3_interp_plan_partition: %s", e.what());
    return S3_ENOMEM;
}

int s3_interp_plan_cost_profile(const s3_interp_plan *p, int n_samples, double *h_out, s3_stream stream) try {
    S3_REQUIRE(p != nullptr && h_out != nullptr && n_samples >= 1, "s3_interp_plan_cost_profile: bad arguments");
    const size_t nt = (size_t)p->n_tiles;
    std::vector<int32_t> cb;
    std::vector<double> acc;
    const int rc = plan_tile_costs(p, as_stream(stream), nullptr, cb, acc);
    if (rc != S3_OK) return rc;
    size_t t = 0;
    for (int i = 0; i <= n_samples; ++i) {
        const double pos = (double)p->nc * i / n_samples;                      // cell position along the processing order
        while (t + 1 < nt && (double)cb[t + 1] <= pos) ++t;
        const double span = (double)(cb[t + 1] - cb[t]);
        const double f = span > 0 ? std::min(1.0, std::max(0.0, (pos - cb[t]) / span)) : 1.0;
        h_out[i] = acc[t] + f * (acc[t + 1] - acc[t]);
    }
    h_out[0] = 0.0;
    h_out[n_samples] = acc[nt];
    return S3_OK;
} catch (const std::exception &e) {
    s3::set_error("s3_interp_plan_cost_profile: %s", e.what());
    return S3_ENOMEM;
}

int s3_interp_plan_set_weights(s3_interp_plan *p, const double *d_w, s3_stream stream) {
    S3_REQUIRE(p != nullptr && d_w != nullptr, "s3_interp_plan_set_weights: null argument");
    if (!p->wp) {
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->wp), sizeof(double) * (size_t)p->nc * p->k);
        if (e != hipSuccess) {
            p->wp = nullptr;
            s3::set_error("s3_interp_plan_set_weights: hipMalloc failed: %s", hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP;
        }
    }
    permute_weights_kernel<<<(unsigned)p->n_tiles, 256, 0, as_stream(stream)>>>(p->perm, p->tile_cell_begin, d_w, p->k, p->wp);
    S3_LAUNCH_CHECK();
    if (stream_can_take(p)) {                // the persistent kernel's copy, in lane order
        const int nv = ((p->k + 3) / 4 + 1) / 2;
        if (!p->wl) {
            const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->wl), sizeof(double) * (size_t)p->nc * 4 * nv * 2);
            if (e != hipSuccess) {
                p->wl = nullptr;
                s3::set_error("s3_interp_plan_set_weights: hipMalloc failed: %s", hipGetErrorString(e));
                return e == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP;
            }
        }
        lane_weights_kernel<<<(unsigned)p->n_tiles, 256, 0, as_stream(stream)>>>(p->tile_cell_begin, p->wp, p->k, p->wl);
        S3_LAUNCH_CHECK();
    }
    p->has_weights = true;
    return S3_OK;
}

// common part of the two launches: `rows` = the plan's row list in the numbering of d_data's rows
static int planned_dispatch(s3_interp_plan *p, const int32_t *rows, int64_t n_rows, const char *who, const void *d_data, int dtype,
                            int64_t row_len, int64_t in_stride, double *d_out, s3_stream stream) {
    S3_REQUIRE(p->has_weights, "%s: no weights (pass d_w or call s3_interp_plan_set_weights first)", who);
    S3_REQUIRE(dtype == S3_DTYPE_F32 || dtype == S3_DTYPE_F64, "%s: unknown dtype %d", who, dtype);
    S3_REQUIRE(row_len >= 0, "%s: bad row_len", who);
    if (row_len == 0) return S3_OK;
    S3_REQUIRE(d_data && d_out, "%s: null array", who);
    const int epv = dtype == S3_DTYPE_F32 ? 4 : 2;
    const uintptr_t a_in = reinterpret_cast<uintptr_t>(d_data), a_out = reinterpret_cast<uintptr_t>(d_out);
    if (in_stride <= 0) in_stride = row_len;
    S3_REQUIRE(in_stride >= row_len && in_stride < ((int64_t)1 << 31), "%s: in_stride %lld < row_len %lld (or >= 2^31)", who,
               (long long)in_stride, (long long)row_len);
    // pitched rows: every source row starts on a 16-byte boundary and is readable up to the next multiple of 16 bytes (the
    // ragged tail of a row is loaded as a whole vector, the surplus lanes are never stored).  Anything else -- a dense
    // [N, n_comp * T] batch read where it lies -- is read with element alignment by the persistent kernel, which touches
    // nothing beyond the end of a row; it wants rows of at least one 16-byte vector.
    const bool aligned = in_stride % epv == 0 && in_stride >= (row_len + epv - 1) / epv * epv && a_in % 16 == 0;
    if (!aligned)
        S3_REQUIRE(a_in % (16 / epv) == 0 && row_len >= epv && stream_can_take(p),
                   "%s: source rows must be 16-byte aligned with a pitch >= the row length rounded up to %d elements, or "
                   "element-aligned rows of >= %d elements on a plan with k = 8 | 26 (row_len %lld, in_stride %lld)", who, epv, epv,
                   (long long)row_len, (long long)in_stride);
    // output rows start on 8-byte boundaries (pairs are written with element alignment where the row length is odd)
    S3_REQUIRE(a_out % ((row_len & 1) ? 8 : 16) == 0, "%s: output not aligned (row_len %lld)", who, (long long)row_len);
    if (dtype == S3_DTYPE_F32) return launch_planned<float>(p, rows, n_rows, aligned, d_data, row_len, in_stride, d_out, as_stream(stream));
    return launch_planned<double>(p, rows, n_rows, aligned, d_data, row_len, in_stride, d_out, as_stream(stream));
}

int s3_interp_planned(s3_interp_plan *p, const double *d_w, const void *d_data, int dtype, int64_t row_len,
                      int64_t in_stride, double *d_out, s3_stream stream) {
    S3_REQUIRE(p != nullptr, "s3_interp_planned: null plan");
    if (d_w != nullptr) {                            // weights handed over with the call: bring them into plan order first
        const int rc = s3_interp_plan_set_weights(p, d_w, stream);
        if (rc != S3_OK) return rc;
    }
    return planned_dispatch(p, p->rows, p->n_src, "s3_interp_planned", d_data, dtype, row_len, in_stride, d_out, stream);
}

__global__ void source_ids_kernel(const int32_t *__restrict__ rows, int64_t n, const int32_t *__restrict__ ids, int32_t n_table,
                                  int32_t *__restrict__ rows_src, int32_t *__restrict__ bad) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = ids[rows[i]];
    if (r < 0 || r >= n_table) atomicExch(bad, 1);
    rows_src[i] = r < 0 || r >= n_table ? 0 : r;
}

int s3_interp_plan_set_source_ids(s3_interp_plan *p, const int32_t *d_ids, int64_t n_table_rows, s3_stream stream) {
    S3_REQUIRE(p != nullptr && d_ids != nullptr, "s3_interp_plan_set_source_ids: null argument");
    S3_REQUIRE(n_table_rows >= 1 && n_table_rows < ((int64_t)1 << 31), "s3_interp_plan_set_source_ids: bad table size");
    hipStream_t st = as_stream(stream);
    if (!p->rows_src)
        S3_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p->rows_src), sizeof(int32_t) * (size_t)std::max<int64_t>(p->total_rows, 1)));
    p->n_table = 0;
    int32_t *d_bad = nullptr, bad = 0;
    S3_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_bad), sizeof(int32_t)));
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(int32_t), st);
    if (e == hipSuccess && p->total_rows > 0) {
        source_ids_kernel<<<s3::grid_for(p->total_rows, 256), 256, 0, st>>>(p->rows, p->total_rows, d_ids, (int32_t)n_table_rows,
                                                                            p->rows_src, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_bad);
    S3_HIP_CHECK(e);
    S3_REQUIRE(bad == 0, "s3_interp_plan_set_source_ids: id outside [0, %lld)", (long long)n_table_rows);
    p->n_table = n_table_rows;
    return S3_OK;
}

int s3_interp_planned_src(s3_interp_plan *p, const void *d_table, int dtype, int64_t n_table_rows, int64_t row_len,
                          int64_t in_stride, double *d_out, s3_stream stream) {
    S3_REQUIRE(p != nullptr, "s3_interp_planned_src: null plan");
    S3_REQUIRE(p->rows_src != nullptr && p->n_table > 0, "s3_interp_planned_src: call s3_interp_plan_set_source_ids first");
    S3_REQUIRE(n_table_rows == p->n_table, "s3_interp_planned_src: the table has %lld rows, the ids were given for %lld",
               (long long)n_table_rows, (long long)p->n_table);
    return planned_dispatch(p, p->rows_src, p->n_table, "s3_interp_planned_src", d_table, dtype, row_len, in_stride, d_out, stream);
}


int s3_debug_reload_env(void) {
    s3::switches() = s3::parse_switches();
    return S3_OK;
}

// yardstick: see plan_loads_kernel.  The table is the one s3_interp_planned_src reads (after s3_interp_plan_set_source_ids) or, with
// n_table_rows == 0, the compacted table of s3_interp_planned.  *h_staged_bytes = rows staged over all tiles x lines x 128.
int s3_yard_plan_loads(s3_interp_plan *p, const void *d_table, int64_t n_table_rows, int64_t row_bytes, int64_t stride_bytes, int variant,
                       s3_stream stream, int64_t *h_staged_bytes) {
    S3_REQUIRE(p != nullptr && d_table != nullptr, "s3_yard_plan_loads: null argument");
    S3_REQUIRE(p->tc == 64 && p->ucap <= 512, "s3_yard_plan_loads: 64-cell tiles of <= 512 rows only");
    S3_REQUIRE(row_bytes >= 128 && stride_bytes >= row_bytes && stride_bytes % 16 == 0 && reinterpret_cast<uintptr_t>(d_table) % 16 == 0,
               "s3_yard_plan_loads: rows of >= 128 bytes at a 16-byte aligned pitch");
    S3_REQUIRE(variant == 0 || variant == 1, "s3_yard_plan_loads: variant 0 | 1");
    const int32_t *rows = p->rows;
    if (n_table_rows > 0) {
        S3_REQUIRE(p->rows_src != nullptr && n_table_rows == p->n_table, "s3_yard_plan_loads: call s3_interp_plan_set_source_ids for this table first");
        rows = p->rows_src;
    }
    static float *d_sink = nullptr;
    if (!d_sink) S3_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_sink), 16));
    // lines of the row that starts furthest into a line: starts are (table start + i * pitch) mod 128
    int64_t g = 128, b = stride_bytes % 128;
    while (b) { const int64_t t = g % b; g = b; b = t; }
    const int64_t max_phase = 128 - g + (int64_t)(reinterpret_cast<uintptr_t>(d_table) & 127) % g;
    int n_lines = (int)((max_phase + row_bytes + 127) / 128);
    n_lines += n_lines & 1;
    const int64_t tiles_per_xcd = (p->n_tiles + 7) / 8;
    const size_t lds = (size_t)p->ucap * PL_SEG + (size_t)p->k * p->tc * (sizeof(double) + sizeof(uint16_t));
    auto k0 = plan_loads_kernel<0>;
    auto k1 = plan_loads_kernel<1>;
    auto kern = variant == 0 ? k0 : k1;
    S3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)(tiles_per_xcd * 8)), 256, lds, as_stream(stream)>>>(p->tile_row_begin, rows, static_cast<const char *>(d_table),
                                                                              (uint64_t)stride_bytes, (uint64_t)row_bytes, n_lines,
                                                                              p->n_tiles, tiles_per_xcd, d_sink);
    S3_LAUNCH_CHECK();
    if (h_staged_bytes) *h_staged_bytes = p->total_rows * (int64_t)n_lines * 128;
    return S3_OK;
}


}  // extern "C"
