// Metric computation upstream of S^3: per-row temporal mean and standard deviation of a snapshot
// matrix, the reduction the reference's example scripts do with torch on the CPU before the grid is generated
// (metric = pt.std(field, dim=1), examples/s3_for_OAT15_airfoil.py:91).  gfx950 only.
//
// HBM-bound streaming reduction: every element is read exactly once (N*T*s_in bytes), 16 bytes out per row.  A group of
// G lanes (G = 4..64, 8-16 vectors per lane) owns one row; each lane reads 16-byte vectors G vectors apart (a wavefront
// instruction covers up to 1 KiB of contiguous memory) and works in f64 -- two passes over the up to
// sixteen values it holds in registers -- and the lanes of the group combine their partial sums with xor-shuffles.
#include "common.h"

#include <cmath>

namespace s3 {

namespace {

struct Moments { double n, mean, m2; };

__device__ __forceinline__ Moments merge(const Moments &a, const Moments &b) {
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    const double n = a.n + b.n, delta = b.mean - a.mean;
    return Moments{n, a.mean + delta * (b.n / n), a.m2 + b.m2 + delta * delta * (a.n * b.n / n)};
}

template <typename T, int VEC>
struct VecLoad;
template <> struct VecLoad<float, 4> { using type = float4; };
template <> struct VecLoad<float, 2> { using type = float2; };
template <> struct VecLoad<float, 1> { using type = float; };
template <> struct VecLoad<double, 2> { using type = double2; };
template <> struct VecLoad<double, 1> { using type = double; };

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
    return v;                                   // xor butterfly: every lane of the group ends with the same bits
}

// G lanes per row, 256 threads per workgroup -> 256/G rows per workgroup.  The row is consumed in chunks of 4*G vectors
// (each lane four 16-byte loads in flight): chunk sum -> chunk mean (one division, the count is known), squared
// deviations from the values still in registers -> chunk M2; chunks are merged with Chan's update (group-uniform).
// A row of up to 4*G vectors (T <= 1024 for fp32 at G = 64) is one chunk: the classic two-pass formula, one read.
// ABS: moments of |x| (the time-mean of sum_c |U_c| of examples/s3_for_cylinder2D_Re100.py:55 is n_comp times the mean of |x|
// over the [n_comp * T] row of a cell)
template <typename T, int VEC, int G, bool ABS>
__global__ void __launch_bounds__(256)
row_moments_kernel(const T *__restrict__ data, int64_t n_rows, int64_t row_len, int64_t in_stride, int ddof,
                   double *__restrict__ mean_out, double *__restrict__ std_out) {
    using V = typename VecLoad<T, VEC>::type;
    constexpr int ROWS = 256 / G;
    const int64_t row = (int64_t)blockIdx.x * ROWS + threadIdx.x / G;
    const int lane = threadIdx.x % G;
    const bool live = row < n_rows;                          // dead rows of the last workgroup still take part in the shuffles
    const T *p = data + (live ? row : 0) * in_stride;
    const int64_t n_vec = live ? row_len / VEC : 0;
    const int64_t tail0 = n_vec * VEC, n_tail = live ? row_len - tail0 : 0;      // < VEC ragged elements, taken by lane 0
    Moments acc{0.0, 0.0, 0.0};
    for (int64_t base = 0; base < n_vec || (base == 0 && n_tail > 0); base += (int64_t)G * 4) {
        // statically indexed registers (a dynamically indexed local array would live in scratch memory): vectors beyond
        // the row are loaded from a clamped position and masked out
        double x[4][VEC];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t v = base + lane + (int64_t)u * G;
            ok[u] = v < n_vec;
            const V raw = *reinterpret_cast<const V *>(p + (ok[u] ? v : 0) * VEC);
            const T *e = reinterpret_cast<const T *>(&raw);
#pragma unroll
            for (int i = 0; i < VEC; ++i) x[u][i] = ABS ? fabs((double)e[i]) : (double)e[i];
        }
        const bool last_chunk = base + (int64_t)G * 4 >= n_vec;
        const bool with_tail = last_chunk && lane == 0 && n_tail > 0;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < VEC; ++i) s += ok[u] ? x[u][i] : 0.0;
        if (with_tail)
            for (int64_t i = 0; i < n_tail; ++i) s += ABS ? fabs((double)p[tail0 + i]) : (double)p[tail0 + i];
        const double n_chunk = (double)(max((int64_t)0, min((int64_t)G * 4, n_vec - base)) * VEC + (last_chunk ? n_tail : 0));
        const double m = group_sum<G>(s) / n_chunk;
        double q = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < VEC; ++i) q += ok[u] ? (x[u][i] - m) * (x[u][i] - m) : 0.0;
        if (with_tail)
            for (int64_t i = 0; i < n_tail; ++i) {
                const double xv = ABS ? fabs((double)p[tail0 + i]) : (double)p[tail0 + i];
                q += (xv - m) * (xv - m);
            }
        acc = merge(acc, Moments{n_chunk, m, group_sum<G>(q)});
    }
    if (live && lane == 0) {
        if (mean_out) mean_out[row] = acc.mean;
        if (std_out) std_out[row] = acc.n - (double)ddof > 0.0 ? sqrt(acc.m2 / (acc.n - (double)ddof)) : NAN;
    }
}

template <typename T, int VEC, bool ABS>
int launch_moments(const void *data, int64_t n_rows, int64_t row_len, int64_t in_stride, int ddof, double *mean, double *sd,
                   hipStream_t st) {
    const T *d = static_cast<const T *>(data);
    const int64_t n_vec = row_len / VEC;
#define S3_LAUNCH_G(G)                                                                                                  \
    do {                                                                                                                \
        const int64_t rows_per_block = 256 / G;                                                                         \
        const int64_t grid = (n_rows + rows_per_block - 1) / rows_per_block;                                            \
        S3_REQUIRE(grid < ((int64_t)1 << 31), "s3_row_moments: too many rows");                                         \
        row_moments_kernel<T, VEC, G, ABS><<<(unsigned)grid, 256, 0, st>>>(d, n_rows, row_len, in_stride, ddof, mean, sd); \
    } while (0)
    // lanes per row: few lanes with many vectors each beat many lanes with few (fewer shuffle reductions and divisions
    // per byte).  MI355X, 4 991 774 fp32 rows: T = 1000 (250 vectors) 64 / 32 / 16 lanes: 3.73 / 3.38 / 3.21 ms;
    // T = 256 (64 vectors) 16 / 8 / 4 lanes: 0.92 / 0.83 / 0.88 ms
    const int64_t per_lane = n_vec >= 128 ? 16 : 8;
    if (n_vec > 32 * per_lane) S3_LAUNCH_G(64);
    else if (n_vec > 16 * per_lane) S3_LAUNCH_G(32);
    else if (n_vec > 8 * per_lane) S3_LAUNCH_G(16);
    else if (n_vec > 4 * per_lane) S3_LAUNCH_G(8);
    else S3_LAUNCH_G(4);
#undef S3_LAUNCH_G
    S3_LAUNCH_CHECK();
    return S3_OK;
}

}  // namespace

}  // namespace s3

using namespace s3;

static int row_moments_impl(const void *d_data, int dtype, int64_t n_rows, int64_t row_len, int64_t in_stride, int ddof,
                            double *d_mean, double *d_std, bool absolute, s3_stream stream) {
    S3_REQUIRE(n_rows >= 0 && row_len >= 1, "s3_row_moments: bad shape n_rows=%lld row_len=%lld", (long long)n_rows,
               (long long)row_len);
    S3_REQUIRE(dtype == S3_DTYPE_F32 || dtype == S3_DTYPE_F64, "s3_row_moments: unknown dtype %d", dtype);
    S3_REQUIRE(ddof == 0 || ddof == 1, "s3_row_moments: ddof must be 0 or 1");
    if (n_rows == 0) return S3_OK;
    S3_REQUIRE(d_data && (d_mean || d_std), "s3_row_moments: null array");
    if (in_stride <= 0) in_stride = row_len;
    S3_REQUIRE(in_stride >= row_len, "s3_row_moments: in_stride %lld < row_len %lld", (long long)in_stride, (long long)row_len);
    hipStream_t st = as_stream(stream);
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_data);
#define S3_MOMENTS(T, VEC)                                                                                                        \
    return absolute ? launch_moments<T, VEC, true>(d_data, n_rows, row_len, in_stride, ddof, d_mean, d_std, st)                   \
                    : launch_moments<T, VEC, false>(d_data, n_rows, row_len, in_stride, ddof, d_mean, d_std, st)
    if (dtype == S3_DTYPE_F32) {
        if (in_stride % 4 == 0 && a % 16 == 0) S3_MOMENTS(float, 4);
        if (in_stride % 2 == 0 && a % 8 == 0) S3_MOMENTS(float, 2);
        S3_MOMENTS(float, 1);
    }
    if (in_stride % 2 == 0 && a % 16 == 0) S3_MOMENTS(double, 2);
    S3_MOMENTS(double, 1);
#undef S3_MOMENTS
}

extern "C" {

int s3_row_moments(const void *d_data, int dtype, int64_t n_rows, int64_t row_len, int64_t in_stride, int ddof,
                   double *d_mean, double *d_std, s3_stream stream) {
    return row_moments_impl(d_data, dtype, n_rows, row_len, in_stride, ddof, d_mean, d_std, false, stream);
}

int s3_row_abs_moments(const void *d_data, int dtype, int64_t n_rows, int64_t row_len, int64_t in_stride, int ddof,
                       double *d_mean, double *d_std, s3_stream stream) {
    return row_moments_impl(d_data, dtype, n_rows, row_len, in_stride, ddof, d_mean, d_std, true, stream);
}

}  // extern "C"
