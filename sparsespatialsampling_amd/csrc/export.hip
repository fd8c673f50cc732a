// KNN inverse-distance interpolation of snapshot fields onto the S^3 cells (the roofline kernel).  gfx950 only.
//
// Reference behaviour restated here: interpolate_data, export.py:446-468
//     out[c, j, t] = sum_m  w[c, m] * data[idx[c, m], j, t]          (f64 weights, f32/f64 data, f64 output)
// and the metric re-interpolation at export.py:215 (row_len = 1).
//
// HBM layout: data is the caller's [n_src][row_len] matrix with row_len = n_comp*T contiguous per source point, so a
// neighbour's contribution to one output row is one contiguous row read.  A workgroup (256 threads = 4 wavefronts)
// owns a tile of TC consecutive output cells: it stages the tile's k neighbour indices (i32) and weights (f64) in LDS
// once, then its threads sweep the tile's TC*row_len outputs in row-major order so that the 64 lanes of a wavefront read
// 64 consecutive VEC-wide pieces of the same source row (1 KiB per wave-instruction for f32 x4) and write 64
// consecutive pieces of the output row.  Accumulation is f64 FMA in neighbour order (the reference forms the rounded
// product and then sums; the difference is <= k ulp, the contract is 1e-5 relative).  No MFMA: this is a gather +
// weighted reduce with ~0.5 flop per byte.
//
// Workgroup -> tile mapping is XCD-aware: workgroups that share `blockIdx % 8` run on the same XCD (MI355X deals
// workgroups round-robin over its 8 XCDs), so XCD x sweeps the x-th contiguous eighth of the tiles and spatially
// adjacent cells (which share neighbours) hit the same 4 MiB L2.
#include "common.h"

namespace s3 {

constexpr int INTERP_BLOCK = 256;

template <typename T, int VEC>
struct VecT;
template <> struct VecT<float, 4> { using type = float4; };
template <> struct VecT<float, 2> { using type = float2; };
template <> struct VecT<float, 1> { using type = float; };
template <> struct VecT<double, 2> { using type = double2; };
template <> struct VecT<double, 1> { using type = double; };

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T *__restrict__ p, double (&v)[VEC]) {
    using V = typename VecT<T, VEC>::type;
    V raw = *reinterpret_cast<const V *>(p);
    const T *e = reinterpret_cast<const T *>(&raw);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = (double)e[i];
}

template <int VEC>
__device__ __forceinline__ void store_vec(double *__restrict__ p, const double (&a)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<double2 *>(p) = make_double2(a[0], a[1]);
        *reinterpret_cast<double2 *>(p + 2) = make_double2(a[2], a[3]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<double2 *>(p) = make_double2(a[0], a[1]);
    } else {
        p[0] = a[0];
    }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(INTERP_BLOCK)
interp_kernel(const double *__restrict__ w, const int32_t *__restrict__ idx, int64_t nc, int k,
              const T *__restrict__ data, int64_t row_len, double *__restrict__ out, int tc, int64_t n_tiles,
              int64_t tiles_per_xcd) {
    extern __shared__ double lds[];
    double *s_w = lds;                                                  // [tc*k]
    int32_t *s_idx = reinterpret_cast<int32_t *>(lds + (size_t)tc * k); // [tc*k]

    // XCD-aware tile assignment (speed only)
    const int64_t b = blockIdx.x;
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);
    if (tile >= n_tiles) return;
    const int64_t c0 = tile * tc;
    const int n_c = (int)min((int64_t)tc, nc - c0);

    const int n_stage = n_c * k;
    for (int i = threadIdx.x; i < n_stage; i += INTERP_BLOCK) {
        s_w[i] = w[c0 * k + i];
        s_idx[i] = idx[c0 * k + i];
    }
    __syncthreads();

    const int lv_count = (int)(row_len / VEC);            // VEC-wide pieces per row
    const int n_items = n_c * lv_count;
    for (int item = threadIdx.x; item < n_items; item += INTERP_BLOCK) {
        const int cl = item / lv_count;
        const int lv = item - cl * lv_count;
        const double *wp = s_w + cl * k;
        const int32_t *ip = s_idx + cl * k;
        const T *col = data + (int64_t)lv * VEC;
        double acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0;
        int m = 0;
        // eight independent row reads in flight per lane before the first FMA (the kernel is bound by gather latency /
        // Infinity-Cache bandwidth, not by arithmetic)
        for (; m + 8 <= k; m += 8) {
            double v[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; ++u) load_vec<T, VEC>(col + (int64_t)ip[m + u] * row_len, v[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double wu = wp[m + u];
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = fma(wu, v[u][i], acc[i]);
            }
        }
        for (; m + 4 <= k; m += 4) {
            double v0[VEC], v1[VEC], v2[VEC], v3[VEC];
            const int64_t r0 = ip[m], r1 = ip[m + 1], r2 = ip[m + 2], r3 = ip[m + 3];
            load_vec<T, VEC>(col + r0 * row_len, v0);
            load_vec<T, VEC>(col + r1 * row_len, v1);
            load_vec<T, VEC>(col + r2 * row_len, v2);
            load_vec<T, VEC>(col + r3 * row_len, v3);
            const double w0 = wp[m], w1 = wp[m + 1], w2 = wp[m + 2], w3 = wp[m + 3];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                acc[i] = fma(w0, v0[i], acc[i]);
                acc[i] = fma(w1, v1[i], acc[i]);
                acc[i] = fma(w2, v2[i], acc[i]);
                acc[i] = fma(w3, v3[i], acc[i]);
            }
        }
        for (; m < k; ++m) {
            double v0[VEC];
            load_vec<T, VEC>(col + (int64_t)ip[m] * row_len, v0);
            const double w0 = wp[m];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fma(w0, v0[i], acc[i]);
        }
        store_vec<VEC>(out + (c0 + cl) * row_len + (int64_t)lv * VEC, acc);
    }
}

// [nc][n_comp][T] -> [T][n_out][n_comp]: snapshot-major image of an interpolated batch for the HDF5 sink, which writes one
// dataset per snapshot (reference export.py:283-299 slices out[:, :, i] on the host).  32x32 tiles through LDS: reads run
// along t, writes along the cell axis.  `rows` (optional): input cell c is row rows[c] of the output (a rank's shard of the
// cells -- ascending ids, mostly runs of siblings -- written into the batch buffer all ranks share); NULL: row c, n_out = nc.
__global__ void __launch_bounds__(256)
snapshot_major_kernel(const double *__restrict__ in, int64_t nc, int n_comp, int64_t T, const int32_t *__restrict__ rows,
                      int64_t n_out, double *__restrict__ out) {
    __shared__ double tile[32][33];
    const int j = blockIdx.z;
    const int64_t c0 = (int64_t)blockIdx.x * 32, t0 = (int64_t)blockIdx.y * 32;   // cells on x: up to 2^31 tiles
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8 threads
    for (int r = ty; r < 32; r += 8) {
        const int64_t c = c0 + r, t = t0 + tx;
        if (c < nc && t < T) tile[r][tx] = in[(c * n_comp + j) * T + t];
    }
    __syncthreads();
    const int64_t c = c0 + tx;
    const int64_t row = c < nc ? (rows ? (int64_t)rows[c] : c) : 0;
    for (int r = ty; r < 32; r += 8) {
        const int64_t t = t0 + r;
        if (c < nc && t < T) out[(t * n_out + row) * n_comp + j] = tile[tx][r];
    }
}

template <typename T, int VEC>
static int launch_interp(const double *w, const int32_t *idx, int64_t nc, int k, const void *data, int64_t row_len,
                         double *out, hipStream_t st) {
    const int64_t lv_count = row_len / VEC;
    // tile height: enough outputs per workgroup to amortise the LDS staging, bounded by 48 KiB of LDS (12 B per
    // (cell, neighbour): a few workgroups stay resident per CU and the launch needs no opt-in to a larger LDS window)
    int64_t tc = (2048 + lv_count - 1) / lv_count;
    const int64_t tc_lds = (48 * 1024) / ((int64_t)k * (sizeof(double) + sizeof(int32_t)));
    if (tc > 256) tc = 256;
    if (tc > tc_lds) tc = tc_lds;
    if (tc < 1) tc = 1;
    if (tc > nc) tc = nc;
    const int64_t n_tiles = (nc + tc - 1) / tc;
    const int64_t tiles_per_xcd = (n_tiles + 7) / 8;
    const int64_t grid = tiles_per_xcd * 8;
    S3_REQUIRE(grid < ((int64_t)1 << 31), "s3_interp: too many tiles (%lld)", (long long)grid);
    S3_REQUIRE(tc * lv_count < ((int64_t)1 << 31), "s3_interp: row_len %lld too long", (long long)row_len);
    size_t lds = (size_t)tc * k * (sizeof(double) + sizeof(int32_t));
    interp_kernel<T, VEC><<<(unsigned)grid, INTERP_BLOCK, lds, st>>>(w, idx, nc, k, static_cast<const T *>(data),
                                                                    row_len, out, (int)tc, n_tiles, tiles_per_xcd);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

}  // namespace s3

using namespace s3;

extern "C" {

int s3_interp(const double *d_w, const int32_t *d_idx, int64_t nc, int k, const void *d_data, int dtype, int64_t n_src,
              int64_t row_len, double *d_out, s3_stream stream) {
    S3_REQUIRE(nc >= 0 && row_len >= 0 && n_src >= 1, "s3_interp: bad shape nc=%lld row_len=%lld n_src=%lld",
               (long long)nc, (long long)row_len, (long long)n_src);
    S3_REQUIRE(k >= 1 && k <= S3_MAX_K, "s3_interp: k=%d outside [1,%d]", k, S3_MAX_K);
    S3_REQUIRE(dtype == S3_DTYPE_F32 || dtype == S3_DTYPE_F64, "s3_interp: unknown dtype %d", dtype);
    if (nc == 0 || row_len == 0) return S3_OK;
    S3_REQUIRE(d_w && d_idx && d_data && d_out, "s3_interp: null array");
    S3_REQUIRE(n_src < ((int64_t)1 << 31), "s3_interp: n_src must fit int32");
    hipStream_t st = as_stream(stream);
    const uintptr_t a_in = reinterpret_cast<uintptr_t>(d_data), a_out = reinterpret_cast<uintptr_t>(d_out);
    if (dtype == S3_DTYPE_F32) {
        if (row_len % 4 == 0 && a_in % 16 == 0 && a_out % 16 == 0)
            return launch_interp<float, 4>(d_w, d_idx, nc, k, d_data, row_len, d_out, st);
        if (row_len % 2 == 0 && a_in % 8 == 0 && a_out % 16 == 0)
            return launch_interp<float, 2>(d_w, d_idx, nc, k, d_data, row_len, d_out, st);
        return launch_interp<float, 1>(d_w, d_idx, nc, k, d_data, row_len, d_out, st);
    }
    if (row_len % 2 == 0 && a_in % 16 == 0 && a_out % 16 == 0)
        return launch_interp<double, 2>(d_w, d_idx, nc, k, d_data, row_len, d_out, st);
    return launch_interp<double, 1>(d_w, d_idx, nc, k, d_data, row_len, d_out, st);
}

int s3_snapshot_major_rows(const double *d_in, int64_t nc, int n_comp, int64_t n_snapshots, const int32_t *d_rows,
                           int64_t n_out, double *d_out, s3_stream stream) {
    S3_REQUIRE(nc >= 0 && n_comp >= 1 && n_snapshots >= 0 && n_out >= nc, "s3_snapshot_major: bad shape");
    if (nc == 0 || n_snapshots == 0) return S3_OK;
    S3_REQUIRE(d_in && d_out && d_in != d_out, "s3_snapshot_major: null or aliased array");
    S3_REQUIRE(d_rows != nullptr || n_out == nc, "s3_snapshot_major: without a row list the output has the input's rows");
    const int64_t gx = (nc + 31) / 32, gy = (n_snapshots + 31) / 32;
    S3_REQUIRE(gx < ((int64_t)1 << 31) && gy <= 65535 && n_comp <= 65535, "s3_snapshot_major: shape too large for one launch");
    snapshot_major_kernel<<<dim3((unsigned)gx, (unsigned)gy, (unsigned)n_comp), 256, 0, as_stream(stream)>>>(
        d_in, nc, n_comp, n_snapshots, d_rows, n_out, d_out);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_snapshot_major(const double *d_in, int64_t nc, int n_comp, int64_t n_snapshots, double *d_out, s3_stream stream) {
    return s3_snapshot_major_rows(d_in, nc, n_comp, n_snapshots, nullptr, nc, d_out, stream);
}

}  // extern "C"
