// Weighted Gram matrix of the interpolated snapshot matrix on the f64 matrix cores -- the dense step of the weighted SVD
// downstream of S^3.  gfx950 only.
//
// Reference behaviour: utils.compute_svd (sparseSpatialSampling/utils.py:302-346): subtract the temporal mean of every row,
// scale every row by sqrt(cell area / volume) (data.py:240-247), SVD of the [N_cells * n_comp, T] matrix.  T (snapshots) is
// a few thousand at most while N is 10^5..10^7, so the SVD is taken through the T x T Gram matrix
//     G = sum_n a_n (x_n - mean_n)(x_n - mean_n)^T         (x_n = row n of the data matrix, a_n its cell area)
// (method of snapshots): G is accumulated here with v_mfma_f64_16x16x4_f64, the small eigenproblem is solved on the host,
// the modes follow from one plain library GEMM (svd.py).  Centring and weighting are fused into the operand staging: the
// matrix is read as it left the interpolation kernel, nothing is materialised.
//
// Kernel: one 256-thread workgroup per (128 x 128 block of the upper triangle of G, slice of the rows).  Per step 16 rows
// of the two column panels are loaded (coalesced 16-byte pieces), centred, weighted and stored to LDS; each wavefront
// owns a 64 x 64 quarter = 4 x 4 MFMA tiles (128 accumulator VGPRs) and issues 16 MFMAs per 4 rows.  The row slices'
// partial blocks are added in slice order by a second kernel (deterministic), which also mirrors the lower triangle.
#include "common.h"

#include <vector>

namespace s3 {

constexpr int GB = 128;            // block edge of G
constexpr int GK = 16;             // rows of the data matrix per step
constexpr int GLD = GB + 16;       // LDS row pitch in doubles: consecutive rows start 128 B apart modulo 256 B (no bank conflicts
                                   // between the four 16-lane groups of a ds_read_b64)

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
gram_block_kernel(const double *__restrict__ x, int64_t n_rows, int t, int64_t in_stride, const double *__restrict__ mean,
                  const double *__restrict__ weight, const int2 *__restrict__ pairs, int64_t rows_per_slice,
                  double *__restrict__ partial /*[slice][pair][GB][GB]*/) {
    __shared__ double sA[2][GK][GLD];
    __shared__ double sB[2][GK][GLD];
    const int pair = blockIdx.x, slice = blockIdx.y;
    const int bi = pairs[pair].x, bj = pairs[pair].y;
    const bool diagonal = bi == bj;
    const int64_t r0 = (int64_t)slice * rows_per_slice, r1 = min(n_rows, r0 + rows_per_slice);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};

    // staging role: 16 threads per row, each 4 pieces of 2 doubles per panel (columns c2, c2 + 32, c2 + 64, c2 + 96).  The
    // raw loads of step s + 1 are issued before the MFMAs of step s and land in registers; they are centred, weighted
    // and written to the other LDS buffer after the MFMAs (global latency hidden behind ~4000 cycles of matrix work)
    const int srow = threadIdx.x >> 4, c2 = (threadIdx.x & 15) * 2;
    double ra[4][2], rb[4][2], mu = 0.0, sw = 0.0;
    auto load = [&](int64_t row_base) {
        const int64_t row = row_base + srow;
        const bool ok = row < r1;
        mu = ok ? mean[row] : 0.0;
        sw = ok ? sqrt(weight[row]) : 0.0;                        // sw = 0 zeroes the padding rows
        const double *xr = x + (ok ? row : 0) * in_stride;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ca = bi * GB + c2 + 32 * p, cb = bj * GB + c2 + 32 * p;
            ra[p][0] = ca < t ? xr[ca] : mu;                      // columns past t contribute (mu - mu) * sw = 0
            ra[p][1] = ca + 1 < t ? xr[ca + 1] : mu;
            if (!diagonal) {
                rb[p][0] = cb < t ? xr[cb] : mu;
                rb[p][1] = cb + 1 < t ? xr[cb + 1] : mu;
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c = c2 + 32 * p;
            sA[buf][srow][c] = (ra[p][0] - mu) * sw;
            sA[buf][srow][c + 1] = (ra[p][1] - mu) * sw;
            if (!diagonal) {
                sB[buf][srow][c] = (rb[p][0] - mu) * sw;
                sB[buf][srow][c + 1] = (rb[p][1] - mu) * sw;
            }
        }
    };

    int buf = 0;
    if (r0 < r1) {
        load(r0);
        store(0);
    }
    __syncthreads();
    for (int64_t row = r0; row < r1; row += GK) {
        const bool more = row + GK < r1;
        if (more) load(row + GK);
        const double(*pa)[GLD] = sA[buf];
        const double(*pb)[GLD] = diagonal ? sA[buf] : sB[buf];
#pragma unroll
        for (int k4 = 0; k4 < GK / 4; ++k4) {
            const int kr = k4 * 4 + (lane >> 4), cl = lane & 15;
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = pa[kr][wi * 64 + i * 16 + cl];
                b[i] = pb[kr][wj * 64 + i * 16 + cl];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // C/D layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * register
    double *out = partial + ((int64_t)slice * gridDim.x + pair) * GB * GB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = wi * 64 + i * 16 + (lane >> 4) + 4 * r, gc = wj * 64 + j * 16 + (lane & 15);
                out[gr * GB + gc] = acc[i][j][r];
            }
}

// G block = sum over the row slices in slice order; upper block written as is, lower block mirrored
__global__ void __launch_bounds__(256)
gram_reduce_kernel(const double *__restrict__ partial, const int2 *__restrict__ pairs, int n_pairs, int n_slices, int t,
                   double *__restrict__ g) {
    const int pair = blockIdx.x;
    const int bi = pairs[pair].x, bj = pairs[pair].y;
    {
        const int e = blockIdx.y * 256 + threadIdx.x;            // one entry of the block per lane
        double s = 0.0;
        for (int sl = 0; sl < n_slices; ++sl) s += partial[((int64_t)sl * n_pairs + pair) * GB * GB + e];
        const int r = bi * GB + e / GB, c = bj * GB + e % GB;
        if (r < t && c < t) {
            if (bi != bj || c >= r) g[(int64_t)r * t + c] = s;
            if (bi != bj || c > r) g[(int64_t)c * t + r] = s;
        }
    }
}

// C = (L - lmean 1^T) B                       (E == nullptr)      the modes U = (X - mean) V S^-1 and the coefficients (X - mean) V
// C = (E - emean 1^T) - (L - lmean 1^T) B     (E != nullptr)      the residual (X - mean) - A V^T of a deflation level
// L [m][k] (row pitch l_stride), B [k][n] contiguous, E [m][n] (row pitch e_stride), C [m][n] contiguous, all f64
// (reference utils.py:302-346 gets U from the SVD itself; here the tall matrix never leaves HBM).  One 256-thread workgroup per
// 128 x 128 block of C; per step 16 columns of L and 16 rows of B go through LDS -- the L tile transposed on the way so that
// both MFMA operands are read like the Gram kernel reads its panels (sA[k][row], sB[k][column]) -- each wavefront owns a
// 64 x 64 quarter = 4 x 4 tiles of v_mfma_f64_16x16x4_f64; the raw loads of step s + 1 are issued before the MFMAs of step s.
// NJ = 16-column tiles per wavefront and row of tiles: 4 -> a 128 x 128 block of C, 2 -> 128 x 64 (right-hand sides of up to 64
// columns -- compute_svd(rank=50) -- would issue 128 columns' worth of MFMAs for 50 otherwise)
template <int NJ>
__global__ void __launch_bounds__(256, 2)      // 198 VGPRs at NJ = 4; without the second bound the compiler takes 316 = one wavefront per SIMD: 31 instead of 44 TFLOP/s
centered_gemm_kernel(const double *__restrict__ l, int64_t m, int k, int64_t l_stride, const double *__restrict__ lmean,
                     const double *__restrict__ b, int n, const double *__restrict__ e, int64_t e_stride,
                     const double *__restrict__ emean, double *__restrict__ c) {
    constexpr int BN = 32 * NJ;                      // columns of C per workgroup
    __shared__ double sA[2][GK][GLD];
    __shared__ double sB[2][GK][GLD];
    const int64_t m0 = (int64_t)blockIdx.x * GB;
    const int n0 = blockIdx.y * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;

    double4_t acc[4][NJ];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < NJ; ++q) acc[a][q] = double4_t{0.0, 0.0, 0.0, 0.0};

    // staging roles.  L: two threads per row of the block, eight consecutive columns each (64 contiguous bytes), written
    // transposed; B: sixteen threads per row of the step, four pieces of two columns (like the Gram kernel's panels)
    const int lrow = threadIdx.x >> 1, lk = (threadIdx.x & 1) * 8;
    const int64_t row_l = m0 + lrow;
    const bool row_ok = row_l < m;
    const double mu = row_ok && lmean ? lmean[row_l] : 0.0;
    const double *lr = l + (row_ok ? row_l : 0) * l_stride;
    const int brow = threadIdx.x >> 4, c2 = (threadIdx.x & 15) * 2;
    double ra[8], rb[NJ][2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kk = k0 + lk + i;
            ra[i] = row_ok && kk < k ? lr[kk] - mu : 0.0;          // rows / columns past the matrix contribute nothing
        }
        const int kb = k0 + brow;
        const double *br = b + (int64_t)(kb < k ? kb : 0) * n;
#pragma unroll
        for (int p = 0; p < NJ; ++p) {
            const int cb = n0 + c2 + 32 * p;
            rb[p][0] = kb < k && cb < n ? br[cb] : 0.0;
            rb[p][1] = kb < k && cb + 1 < n ? br[cb + 1] : 0.0;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sA[buf][lk + i][lrow] = ra[i];
#pragma unroll
        for (int p = 0; p < NJ; ++p) {
            sB[buf][brow][c2 + 32 * p] = rb[p][0];
            sB[buf][brow][c2 + 32 * p + 1] = rb[p][1];
        }
    };

    int buf = 0;
    load(0);
    store(0);
    __syncthreads();
    for (int k0 = 0; k0 < k; k0 += GK) {
        const bool more = k0 + GK < k;
        if (more) load(k0 + GK);
        const double(*pa)[GLD] = sA[buf];
        const double(*pb)[GLD] = sB[buf];
#pragma unroll
        for (int k4 = 0; k4 < GK / 4; ++k4) {
            const int kr = k4 * 4 + (lane >> 4), cl = lane & 15;
            double a[4], bb[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = pa[kr][wi * 64 + i * 16 + cl];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bb[j] = pb[kr][wj * (16 * NJ) + j * 16 + cl];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // C/D layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * register
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gr = m0 + wi * 64 + i * 16 + (lane >> 4) + 4 * r;
            if (gr >= m) continue;
            const double em = e && emean ? emean[gr] : 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int gc = n0 + wj * (16 * NJ) + j * 16 + (lane & 15);
                if (gc >= n) continue;
                const double v = acc[i][j][r];
                c[gr * n + gc] = e ? (e[gr * e_stride + gc] - em) - v : v;
            }
        }
}

}  // namespace s3

using namespace s3;

extern "C" {

int s3_centered_gemm(const double *d_l, int64_t m, int64_t k, int64_t l_stride, const double *d_lmean, const double *d_b,
                     int64_t n, const double *d_e, int64_t e_stride, const double *d_emean, double *d_c, s3_stream stream) {
    S3_REQUIRE(d_l && d_b && d_c, "s3_centered_gemm: null array");
    S3_REQUIRE(m >= 1 && k >= 1 && n >= 1 && k < (1 << 24) && n < (1 << 24) && l_stride >= k && (d_e == nullptr || e_stride >= n),
               "s3_centered_gemm: bad sizes (m %lld, k %lld, n %lld)", (long long)m, (long long)k, (long long)n);
    const bool narrow = n <= 64;                     // (one 64-column block: half the MFMAs of a 128-column one)
    const int64_t bn = narrow ? 64 : GB;
    const int64_t gx = (m + GB - 1) / GB, gy = (n + bn - 1) / bn;
    S3_REQUIRE(gx < ((int64_t)1 << 31) && gy <= 65535, "s3_centered_gemm: shape too large for one launch");
    if (narrow)
        centered_gemm_kernel<2><<<dim3((unsigned)gx, (unsigned)gy), 256, 0, as_stream(stream)>>>(d_l, m, (int)k, l_stride, d_lmean, d_b,
                                                                                                (int)n, d_e, e_stride, d_emean, d_c);
    else
        centered_gemm_kernel<4><<<dim3((unsigned)gx, (unsigned)gy), 256, 0, as_stream(stream)>>>(d_l, m, (int)k, l_stride, d_lmean, d_b,
                                                                                                (int)n, d_e, e_stride, d_emean, d_c);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

size_t s3_weighted_gram_scratch_bytes(int64_t n_rows, int64_t t) {
    if (n_rows < 1 || t < 1) return 0;
    const int64_t nb = (t + GB - 1) / GB, n_pairs = nb * (nb + 1) / 2;
    int64_t slices = (1024 + n_pairs - 1) / n_pairs;
    const int64_t max_slices = (n_rows + 16 * GK - 1) / (16 * GK);
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    return (size_t)(slices * n_pairs) * GB * GB * sizeof(double) + (size_t)n_pairs * sizeof(int2) + 64;
}

int s3_weighted_gram(const double *d_x, int64_t n_rows, int64_t t, int64_t in_stride, const double *d_mean,
                     const double *d_weight, double *d_gram, void *d_scratch, s3_stream stream) {
    S3_REQUIRE(d_x && d_mean && d_weight && d_gram && d_scratch, "s3_weighted_gram: null array");
    S3_REQUIRE(n_rows >= 1 && t >= 1 && t < (1 << 20) && in_stride >= t, "s3_weighted_gram: bad sizes (rows %lld, t %lld, stride %lld)",
               (long long)n_rows, (long long)t, (long long)in_stride);
    hipStream_t st = as_stream(stream);
    const int nb = (int)((t + GB - 1) / GB), n_pairs = nb * (nb + 1) / 2;
    int64_t slices = (1024 + n_pairs - 1) / n_pairs;
    const int64_t max_slices = (n_rows + 16 * GK - 1) / (16 * GK);
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    int64_t rows_per_slice = (n_rows + slices - 1) / slices;
    rows_per_slice = (rows_per_slice + GK - 1) / GK * GK;
    slices = (n_rows + rows_per_slice - 1) / rows_per_slice;
    S3_REQUIRE(slices < 65536, "s3_weighted_gram: too many row slices");
    double *d_partial = static_cast<double *>(d_scratch);
    int2 *d_pairs = reinterpret_cast<int2 *>(d_partial + (size_t)slices * n_pairs * GB * GB);
    std::vector<int2> pairs;
    for (int i = 0; i < nb; ++i)
        for (int j = i; j < nb; ++j) pairs.push_back(make_int2(i, j));
    S3_HIP_CHECK(hipMemcpyAsync(d_pairs, pairs.data(), sizeof(int2) * pairs.size(), hipMemcpyHostToDevice, st));
    S3_HIP_CHECK(hipStreamSynchronize(st));               // `pairs` is a local
    gram_block_kernel<<<dim3((unsigned)n_pairs, (unsigned)slices), 256, 0, st>>>(d_x, n_rows, (int)t, in_stride, d_mean, d_weight,
                                                                                d_pairs, rows_per_slice, d_partial);
    S3_LAUNCH_CHECK();
    gram_reduce_kernel<<<dim3((unsigned)n_pairs, GB * GB / 256), 256, 0, st>>>(d_partial, d_pairs, n_pairs, (int)slices, (int)t, d_gram);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

}  // extern "C"
