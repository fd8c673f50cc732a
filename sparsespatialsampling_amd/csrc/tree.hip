// Device-resident cell arrays of the S^3 sampling tree: child creation, geometry predicates, batch bookkeeping,
// captured-metric reduction and top-N gain selection.  gfx950 only.
//
// The tree itself (which cell gets which id, neighbour links, shared-node numbering) is decided on the host exactly as
// the reference does (ordering comes from CPython set iteration); the arrays below are the
// structure-of-arrays image of the reference's `Cell` objects (s_cube.py:32-83) that the kernels work on:
//   center[cap][dim] f64, level[cap] i32, metric[cap] f64, gain[cap] f64, leaf[cap] u8.
#include "common.h"

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <vector>

namespace s3 {

// ------------------------------------------------------------------------------------------------------------------
// a3: children.  s_cube.py:875 -> 399-445 with _factor = 0.25
// ------------------------------------------------------------------------------------------------------------------
template <int DIM>
__global__ void make_children_kernel(double *__restrict__ center, int32_t *__restrict__ level,
                                     const int32_t *__restrict__ parents, int64_t n_par, int64_t new_index,
                                     double quarter_width) {
    constexpr int NCH = 1 << DIM;
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n_par * NCH) return;
    int64_t i = t / NCH;
    int c = (int)(t - i * NCH);
    int64_t p = parents[i];
    int lv = level[p];
    double off = cell_offset(quarter_width, lv);
    int64_t child = new_index + t;
#pragma unroll
    for (int j = 0; j < DIM; ++j) center[child * DIM + j] = center[p * DIM + j] + dir_comp(DIM, c, j) * off;
    level[child] = lv + 1;
}

// ------------------------------------------------------------------------------------------------------------------
// a12: geometry predicates.  combine() = GeometryObject._apply_mask (geometry_base.py:40-76)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t combine(int n_in, int n_nodes, int refine_mode, int keep_inside) {
    bool all = n_in == n_nodes, any = n_in > 0;
    if (!refine_mode) return keep_inside ? !any : all;
    return keep_inside ? !all : any;
}

struct BoxParams { double lo[3], hi[3]; };
struct SphereParams { double pos[3], radius; };
struct CylParams { double p0[3], axis[3], norm, r0, r1; int is_cone; };
struct PolyParams { const double *poly; int nv; };
struct TriParams { double p[3][2]; };
struct PrismParams { double origin[3], axis[3], norm; int dims[2]; TriParams tri; };
struct TetParams { double pos[2][4][3], nrm[2][3][4]; int n_tets; };

template <int DIM>
__device__ __forceinline__ bool inside(const BoxParams &g, const double (&x)[DIM]) {
    bool in = true;
#pragma unroll
    for (int j = 0; j < DIM; ++j) in &= (x[j] >= g.lo[j]) & (x[j] <= g.hi[j]);
    return in;
}

template <int DIM>
__device__ __forceinline__ bool inside(const SphereParams &g, const double (&x)[DIM]) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        double t = x[j] - g.pos[j];
        s += t * t;
    }
    return sqrt(s) <= g.radius;
}

template <int DIM>
__device__ __forceinline__ bool inside(const CylParams &g, const double (&x)[DIM]) {
    static_assert(DIM == 3, "cylinder is 3-D only");
    double v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = x[j] - g.p0[j];
    double c0 = g.axis[1] * v[2] - g.axis[2] * v[1];
    double c1 = g.axis[2] * v[0] - g.axis[0] * v[2];
    double c2 = g.axis[0] * v[1] - g.axis[1] * v[0];
    double nd = sqrt(c0 * c0 + c1 * c1 + c2 * c2) / g.norm;
    double proj = ((v[0] * g.axis[0] + v[1] * g.axis[1]) + v[2] * g.axis[2]) / g.norm;
    double rad = g.is_cone ? g.r0 + proj / g.norm * (g.r1 - g.r0) : g.r0;
    return (0.0 <= proj) & (proj <= g.norm) & (nd <= rad);
}

template <int DIM>
__device__ __forceinline__ bool inside(const PolyParams &g, const double (&x)[DIM]) {
    static_assert(DIM == 2, "polygon is 2-D only");
    const double px = x[0], py = x[1];
    bool in = false;
    for (int i = 0; i < g.nv; ++i) {
        int j = i + 1 == g.nv ? 0 : i + 1;
        double xi = g.poly[2 * i], yi = g.poly[2 * i + 1], xj = g.poly[2 * j], yj = g.poly[2 * j + 1];
        double cross = (xj - xi) * (py - yi) - (yj - yi) * (px - xi);
        if (cross == 0.0 && fmin(xi, xj) <= px && px <= fmax(xi, xj) && fmin(yi, yj) <= py && py <= fmax(yi, yj))
            return false;   // on the boundary -> not strictly inside
        if ((yi > py) != (yj > py)) {
            double xint = xi + (py - yi) * (xj - xi) / (yj - yi);
            if (px < xint) in = !in;
        }
    }
    return in;
}

// triangle: sign test of the three edge cross products (triangle_geometry.py:80-103); edges 0->1, 1->2, 2->0, the third
// one measured from point 0 as the reference does; the outline counts as inside
__device__ __forceinline__ bool inside_triangle(const TriParams &g, double x, double y) {
    const double d1 = (g.p[1][0] - g.p[0][0]) * (y - g.p[0][1]) - (g.p[1][1] - g.p[0][1]) * (x - g.p[0][0]);
    const double d2 = (g.p[2][0] - g.p[1][0]) * (y - g.p[1][1]) - (g.p[2][1] - g.p[1][1]) * (x - g.p[1][0]);
    const double d3 = (g.p[0][0] - g.p[2][0]) * (y - g.p[0][1]) - (g.p[0][1] - g.p[2][1]) * (x - g.p[0][0]);
    const bool neg = (d1 < 0.0) | (d2 < 0.0) | (d3 < 0.0), pos = (d1 > 0.0) | (d2 > 0.0) | (d3 > 0.0);
    return !(neg & pos);
}

template <int DIM>
__device__ __forceinline__ bool inside(const TriParams &g, const double (&x)[DIM]) {
    static_assert(DIM == 2, "triangle is 2-D only");
    return inside_triangle(g, x[0], x[1]);
}

// prism: projection on the extrusion axis within [0, |axis|] and the in-plane coordinates inside the first triangle
// (prism_geometry.py:90-118)
template <int DIM>
__device__ __forceinline__ bool inside(const PrismParams &g, const double (&x)[DIM]) {
    static_assert(DIM == 3, "prism is 3-D only");
    double v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = x[j] - g.origin[j];
    const double proj = ((v[0] * g.axis[0] + v[1] * g.axis[1]) + v[2] * g.axis[2]) / g.norm;
    const double u0 = g.dims[0] == 0 ? x[0] : (g.dims[0] == 1 ? x[1] : x[2]);
    const double u1 = g.dims[1] == 0 ? x[0] : (g.dims[1] == 1 ? x[1] : x[2]);
    return (0.0 <= proj) & (proj <= g.norm) & inside_triangle(g.tri, u0, u1);
}

// tetrahedron / pyramid (= union of two tetrahedra): no face sees the point on its outer side
// (tetrahedron_geometry.py:121-140, pyramid_geometry.py:156-170).  The reference takes the products with torch.dot on a
// contiguous and a strided operand; its BLAS evaluates fma(a2, b2, a0*b0 + a1*b1) for three elements.
template <int DIM>
__device__ __forceinline__ bool inside(const TetParams &g, const double (&x)[DIM]) {
    static_assert(DIM == 3, "tetrahedron is 3-D only");
    bool in_any = false;
    for (int t = 0; t < g.n_tets; ++t) {
        bool outside = false;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const double a0 = x[0] - g.pos[t][p][0], a1 = x[1] - g.pos[t][p][1], a2 = x[2] - g.pos[t][p][2];
            const double d = __fma_rn(a2, g.nrm[t][2][p], a0 * g.nrm[t][0][p] + a1 * g.nrm[t][1][p]);
            outside |= d < 0.0;
        }
        in_any |= !outside;
    }
    return in_any;
}

template <int DIM, typename G>
__global__ void mask_kernel(const double *__restrict__ center, const int32_t *__restrict__ level,
                            const int32_t *__restrict__ cells, int64_t first, int64_t n, double half_width, G g,
                            int refine_mode, int keep_inside, uint8_t *__restrict__ invalid) {
    constexpr int NN = 1 << DIM;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t cell = cells ? (int64_t)cells[i] : first + i;
    double off = cell_offset(half_width, level[cell]);
    double cx[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) cx[j] = center[cell * DIM + j];
    int n_in = 0;
#pragma unroll
    for (int c = 0; c < NN; ++c) {
        double x[DIM];
#pragma unroll
        for (int j = 0; j < DIM; ++j) x[j] = cx[j] + dir_comp(DIM, c, j) * off;
        n_in += inside<DIM>(g, x) ? 1 : 0;
    }
    invalid[i] |= combine(n_in, NN, refine_mode, keep_inside);
}

static int check_mask_args(const void *center, const void *level, int64_t first, int64_t n, int dim, const void *inv,
                           const char *who) {
    S3_REQUIRE(dim == 2 || dim == 3, "%s: dim must be 2 or 3", who);
    S3_REQUIRE(n >= 0 && first >= 0, "%s: bad range", who);
    S3_REQUIRE(n == 0 || (center && level && inv), "%s: null array", who);
    return S3_OK;
}

// ------------------------------------------------------------------------------------------------------------------
__global__ void commit_batch_kernel(uint8_t *__restrict__ leaf, double *__restrict__ gain,
                                    const int32_t *__restrict__ parents, int64_t n_par, int64_t first, int64_t n_new,
                                    const uint8_t *__restrict__ invalid) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < n_par) leaf[parents[t]] = 0;
    if (t < n_new) {
        bool bad = invalid && invalid[t];
        leaf[first + t] = bad ? 0 : 1;
        if (bad) gain[first + t] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// a6: sum of metric^2 over leaves, fixed reduction tree (deterministic for a given range)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double *sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    return r;
}

__global__ void __launch_bounds__(256)
sumsq_partial_kernel(const double *__restrict__ metric, const uint8_t *__restrict__ leaf, int64_t begin, int64_t end,
                     double *__restrict__ partial) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t i = begin + blockIdx.x * (int64_t)256 + threadIdx.x; i < end; i += (int64_t)gridDim.x * 256)
        if (leaf[i]) {
            double m = metric[i];
            s += m * m;
        }
    double r = block_sum_256(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ void __launch_bounds__(256) sumsq_final_kernel(const double *__restrict__ partial, int nb, double *__restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    double r = block_sum_256(s, sh);
    if (threadIdx.x == 0) *out = r;
}

// captured metric in a form that does not depend on how the cells are split over ranks: one workgroup per 1024-cell block
// (4 cells per lane, fixed tree), then the block sums are added in index order by a single workgroup (fixed tree)
__global__ void __launch_bounds__(256)
sumsq_blocks_kernel(const double *__restrict__ metric, const uint8_t *__restrict__ leaf, int64_t n_cells, int64_t block_begin,
                    double *__restrict__ partial) {
    __shared__ double sh[4];
    const int64_t blk = block_begin + blockIdx.x;
    const int64_t base = blk * S3_SUMSQ_BLOCK;
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < S3_SUMSQ_BLOCK / 256; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < n_cells && leaf[i]) {
            const double m = metric[i];
            s += m * m;
        }
    }
    const double r = block_sum_256(s, sh);
    if (threadIdx.x == 0) partial[blk] = r;
}

__global__ void __launch_bounds__(256) sum_ordered_kernel(const double *__restrict__ v, int64_t n, double *__restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];        // lane l adds l, l + 256, ... in that order
    const double r = block_sum_256(s, sh);
    if (threadIdx.x == 0) *out = r;
}

// ------------------------------------------------------------------------------------------------------------------
// a8: top-N by (gain desc, id asc) via a radix select over the 96-bit composite key (orderable gain bits, ~id)
// ------------------------------------------------------------------------------------------------------------------
constexpr int SEL_BITS = 12, SEL_BINS = 1 << SEL_BITS, SEL_DIGITS = 8;

struct Key96 { uint64_t hi; uint32_t lo; };

__host__ __device__ __forceinline__ Key96 make_key(double gain, uint32_t id) {
    uint64_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (uint64_t)__double_as_longlong(gain);
#else
    memcpy(&b, &gain, 8);
#endif
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);   // total order of IEEE doubles as unsigned
    return Key96{b, ~id};
}

__host__ __device__ __forceinline__ uint32_t key_digit(const Key96 &k, int digit) {
    int shift = 84 - SEL_BITS * digit;                  // position of the digit's LSB in the 96-bit number
    if (shift >= 32) return (uint32_t)(k.hi >> (shift - 32)) & (SEL_BINS - 1);
    uint64_t low64 = (k.hi << 32) | k.lo;
    return (uint32_t)(low64 >> shift) & (SEL_BINS - 1);
}

// keep the `nd` most significant digits
__host__ __device__ __forceinline__ Key96 key_prefix(const Key96 &k, int nd) {
    int bits = SEL_BITS * nd;
    Key96 r;
    if (bits == 0) { r.hi = 0; r.lo = 0; }
    else if (bits <= 64) { r.hi = bits == 64 ? k.hi : (k.hi & (~0ull << (64 - bits))); r.lo = 0; }
    else { r.hi = k.hi; r.lo = bits >= 96 ? k.lo : (k.lo & (~0u << (96 - bits))); }
    return r;
}

// The digit passes are driven from the device: the state of the selection (prefix of the threshold so far, how many keys of
// the prefix class are still wanted, done) lives in the scratch buffer, a one-workgroup kernel reads each pass's histogram
// and advances it, and the next pass's kernels read it from there -- the host queues all passes at once and waits ONCE, for
// the selected ids (before: a histogram download and a stream synchronisation per digit, four or five per call, more than
// the kernels themselves at 5 * 10^5 cells).  Passes behind the one that finished return at once.
struct SelState {
    unsigned long long prefix_hi;
    uint32_t prefix_lo;
    int32_t done;
    long long need;
    unsigned long long count;                              // number of ids collected
};

__global__ void __launch_bounds__(256)
select_hist_dev_kernel(const double *__restrict__ gain, const uint8_t *__restrict__ leaf, int64_t n,
                       const SelState *__restrict__ st, int nd, uint32_t *__restrict__ hist) {
    if (st->done) return;
    const Key96 prefix{st->prefix_hi, st->prefix_lo};
    __shared__ uint32_t sh[SEL_BINS];
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) sh[i] = 0;
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (!leaf[i]) continue;
        Key96 k = make_key(gain[i], (uint32_t)i);
        Key96 p = key_prefix(k, nd);
        if (p.hi == prefix.hi && p.lo == prefix.lo) atomicAdd(&sh[key_digit(k, nd)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_BINS; i += 256)
        if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// one workgroup: the bin in which the count from the top reaches `need`; clears the histogram for the next pass
__global__ void __launch_bounds__(256)
select_pick_kernel(uint32_t *__restrict__ hist, SelState *__restrict__ st, int d) {
    constexpr int PER = SEL_BINS / 256;
    __shared__ long long part[256];
    __shared__ long long s_total, s_above, s_in_bin;
    __shared__ int s_bin;
    if (st->done) return;                                  // (uniform: every thread reads the same word)
    const int t = threadIdx.x;
    // thread t owns bins SEL_BINS - 1 - t * PER downwards (the scan runs from the top bin)
    uint32_t mine[PER];
    long long sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        mine[i] = hist[SEL_BINS - 1 - (t * PER + i)];
        sum += mine[i];
    }
    part[t] = sum;
    if (t == 0) s_bin = -1;
    __syncthreads();
    for (int i = t; i < SEL_BINS; i += 256) hist[i] = 0;
    if (t == 0) {                                          // exclusive prefix of the 256 sums (a few hundred cycles)
        long long run = 0;
        for (int i = 0; i < 256; ++i) {
            const long long v = part[i];
            part[i] = run;
            run += v;
        }
        s_total = run;
    }
    __syncthreads();
    const long long need = st->need;
    if (d == 0 && s_total <= need) {                       // fewer leaves than requested: take everything (threshold 0)
        if (t == 0) {
            st->prefix_hi = 0;
            st->prefix_lo = 0;
            st->done = 1;
        }
        return;
    }
    long long acc = part[t];
    if (acc < need && acc + sum >= need) {                 // exactly one thread: the count reaches `need` inside its bins
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (acc + mine[i] >= need) {
                s_bin = SEL_BINS - 1 - (t * PER + i);
                s_above = acc;
                s_in_bin = mine[i];
                break;
            }
            acc += mine[i];
        }
    }
    __syncthreads();
    if (t == 0 && s_bin >= 0) {                            // (the class of the prefix holds >= need keys: a bin is always found)
        const long long left = need - s_above;             // still wanted from bin s_bin
        const int shift = 84 - SEL_BITS * d;
        unsigned long long hi = st->prefix_hi;
        uint32_t lo = st->prefix_lo;
        if (shift >= 32) hi |= (unsigned long long)s_bin << (shift - 32);
        else {
            const unsigned long long low64 = (unsigned long long)s_bin << shift;
            hi |= low64 >> 32;
            lo |= (uint32_t)low64;
        }
        st->prefix_hi = hi;
        st->prefix_lo = lo;
        st->need = left;
        if (s_in_bin == left || d == SEL_DIGITS - 1) st->done = 1;     // the whole bin is taken: threshold = this prefix
    }
}

__global__ void __launch_bounds__(256)
select_collect_dev_kernel(const double *__restrict__ gain, const uint8_t *__restrict__ leaf, int64_t n, SelState *__restrict__ st,
                          int64_t cap, int32_t *__restrict__ out_id, double *__restrict__ out_gain) {
    const Key96 thr{st->prefix_hi, st->prefix_lo};
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (!leaf[i]) continue;
        double g = gain[i];
        Key96 k = make_key(g, (uint32_t)i);
        if (k.hi > thr.hi || (k.hi == thr.hi && k.lo >= thr.lo)) {
            unsigned long long pos = atomicAdd(&st->count, 1ull);
            if ((int64_t)pos < cap) {
                out_id[pos] = (int32_t)i;
                out_gain[pos] = g;
            }
        }
    }
}

// The selected ids in descending key order: a key's place is the number of selected keys above it (the keys are distinct:
// the id is part of them).  n^2 comparisons -- 2.5 * 10^7 at 5000 ids, 8 * 10^8 at 29 000: microseconds on the device, against
// 0.15 / 1 ms for the host's std::sort of the same records, which was the largest part of a call.
constexpr int64_t RANK_MAX = 262144;       // above this many ids the host sorts (n log n beats n^2 / 10^4 lanes)

constexpr int RANK_SPAN = 512;             // keys a workgroup compares its 256 keys with

// partial ranks: workgroup (x, y) counts, for each of the 256 keys of tile x, the keys above it among keys y * RANK_SPAN ..
__global__ void __launch_bounds__(256)
select_rank_kernel(const int32_t *__restrict__ ids, const double *__restrict__ gains, const SelState *__restrict__ st,
                   int64_t cap, int32_t *__restrict__ rank) {
    __shared__ unsigned long long t_hi[RANK_SPAN];
    __shared__ uint32_t t_lo[RANK_SPAN];
    const int64_t n = min((int64_t)st->count, cap);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, base = (int64_t)blockIdx.y * RANK_SPAN;
    if ((int64_t)blockIdx.x * 256 >= n || base >= n) return;              // (uniform)
    const int m = (int)min((int64_t)RANK_SPAN, n - base);
    for (int q = threadIdx.x; q < RANK_SPAN; q += 256) {
        const Key96 k = q < m ? make_key(gains[base + q], (uint32_t)ids[base + q]) : Key96{0, 0};   // (0, 0): above no key
        t_hi[q] = k.hi;
        t_lo[q] = k.lo;
    }
    __syncthreads();
    if (i >= n) return;
    const Key96 mine = make_key(gains[i], (uint32_t)ids[i]);
    int32_t above = 0;
#pragma unroll 8
    for (int q = 0; q < RANK_SPAN; ++q) above += (t_hi[q] > mine.hi || (t_hi[q] == mine.hi && t_lo[q] > mine.lo)) ? 1 : 0;
    atomicAdd(&rank[i], above);
}

__global__ void __launch_bounds__(256)
select_place_kernel(const int32_t *__restrict__ ids, const int32_t *__restrict__ rank, const SelState *__restrict__ st,
                    int64_t cap, int32_t *__restrict__ sorted) {
    const int64_t n = min((int64_t)st->count, cap);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) sorted[rank[i]] = ids[i];
}

}  // namespace s3

using namespace s3;

extern "C" {

int s3_make_children(double *d_center, int32_t *d_level, const int32_t *d_parents, int64_t n_par, int64_t new_index,
                     int dim, double width, s3_stream stream) {
    S3_REQUIRE(dim == 2 || dim == 3, "s3_make_children: dim must be 2 or 3");
    S3_REQUIRE(n_par >= 0 && new_index >= 1, "s3_make_children: bad range");
    S3_REQUIRE(n_par == 0 || (d_center && d_level && d_parents), "s3_make_children: null array");
    if (n_par == 0) return S3_OK;
    const double qw = 0.25 * width;
    if (dim == 2)
        make_children_kernel<2><<<grid_for(n_par * 4, 256), 256, 0, as_stream(stream)>>>(d_center, d_level, d_parents,
                                                                                        n_par, new_index, qw);
    else
        make_children_kernel<3><<<grid_for(n_par * 8, 256), 256, 0, as_stream(stream)>>>(d_center, d_level, d_parents,
                                                                                        n_par, new_index, qw);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_mask_box(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n, int dim,
                double width, const double *h_lo, const double *h_hi, int refine_mode, int keep_inside,
                uint8_t *d_invalid, s3_stream stream) {
    if (int rc = check_mask_args(d_center, d_level, first, n, dim, d_invalid, "s3_mask_box")) return rc;
    S3_REQUIRE(h_lo && h_hi, "s3_mask_box: null bounds");
    if (n == 0) return S3_OK;
    BoxParams g{};
    for (int j = 0; j < dim; ++j) { g.lo[j] = h_lo[j]; g.hi[j] = h_hi[j]; }
    const double hw = 0.5 * width;
    if (dim == 2)
        mask_kernel<2, BoxParams><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(d_center, d_level, d_cells, first, n,
                                                                                  hw, g, refine_mode, keep_inside, d_invalid);
    else
        mask_kernel<3, BoxParams><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(d_center, d_level, d_cells, first, n,
                                                                                  hw, g, refine_mode, keep_inside, d_invalid);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_mask_sphere(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                   int dim, double width, const double *h_pos, double radius, int refine_mode, int keep_inside,
                   uint8_t *d_invalid, s3_stream stream) {
    if (int rc = check_mask_args(d_center, d_level, first, n, dim, d_invalid, "s3_mask_sphere")) return rc;
    S3_REQUIRE(h_pos, "s3_mask_sphere: null position");
    if (n == 0) return S3_OK;
    SphereParams g{};
    for (int j = 0; j < dim; ++j) g.pos[j] = h_pos[j];
    g.radius = radius;
    const double hw = 0.5 * width;
    if (dim == 2)
        mask_kernel<2, SphereParams><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(
            d_center, d_level, d_cells, first, n, hw, g, refine_mode, keep_inside, d_invalid);
    else
        mask_kernel<3, SphereParams><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(
            d_center, d_level, d_cells, first, n, hw, g, refine_mode, keep_inside, d_invalid);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_mask_cylinder(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                     double width, const double *h_p0, const double *h_axis, double norm, double r0, double r1,
                     int is_cone, int refine_mode, int keep_inside, uint8_t *d_invalid, s3_stream stream) {
    if (int rc = check_mask_args(d_center, d_level, first, n, 3, d_invalid, "s3_mask_cylinder")) return rc;
    S3_REQUIRE(h_p0 && h_axis && norm > 0, "s3_mask_cylinder: bad axis");
    if (n == 0) return S3_OK;
    CylParams g{};
    for (int j = 0; j < 3; ++j) { g.p0[j] = h_p0[j]; g.axis[j] = h_axis[j]; }
    g.norm = norm; g.r0 = r0; g.r1 = r1; g.is_cone = is_cone;
    mask_kernel<3, CylParams><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(d_center, d_level, d_cells, first, n,
                                                                              0.5 * width, g, refine_mode, keep_inside,
                                                                              d_invalid);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_mask_polygon(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                    double width, const double *d_poly, int nv, int refine_mode, int keep_inside, uint8_t *d_invalid,
                    s3_stream stream) {
    if (int rc = check_mask_args(d_center, d_level, first, n, 2, d_invalid, "s3_mask_polygon")) return rc;
    S3_REQUIRE(d_poly && nv >= 3, "s3_mask_polygon: need >= 3 vertices");
    if (n == 0) return S3_OK;
    PolyParams g{d_poly, nv};
    mask_kernel<2, PolyParams><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(d_center, d_level, d_cells, first, n,
                                                                               0.5 * width, g, refine_mode, keep_inside,
                                                                               d_invalid);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_mask_triangle(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                     double width, const double *h_points, int refine_mode, int keep_inside, uint8_t *d_invalid,
                     s3_stream stream) {
    if (int rc = check_mask_args(d_center, d_level, first, n, 2, d_invalid, "s3_mask_triangle")) return rc;
    S3_REQUIRE(h_points != nullptr, "s3_mask_triangle: null points");
    if (n == 0) return S3_OK;
    TriParams g{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) g.p[i][j] = h_points[2 * i + j];
    mask_kernel<2, TriParams><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(d_center, d_level, d_cells, first, n,
                                                                              0.5 * width, g, refine_mode, keep_inside,
                                                                              d_invalid);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_mask_prism(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                  double width, const double *h_origin, const double *h_axis, double norm, const int32_t *h_dims,
                  const double *h_triangle, int refine_mode, int keep_inside, uint8_t *d_invalid, s3_stream stream) {
    if (int rc = check_mask_args(d_center, d_level, first, n, 3, d_invalid, "s3_mask_prism")) return rc;
    S3_REQUIRE(h_origin && h_axis && h_dims && h_triangle && norm > 0, "s3_mask_prism: bad axis");
    S3_REQUIRE(h_dims[0] >= 0 && h_dims[0] < 3 && h_dims[1] >= 0 && h_dims[1] < 3 && h_dims[0] != h_dims[1],
               "s3_mask_prism: the triangle plane must be two different coordinate directions");
    if (n == 0) return S3_OK;
    PrismParams g{};
    for (int j = 0; j < 3; ++j) { g.origin[j] = h_origin[j]; g.axis[j] = h_axis[j]; }
    g.norm = norm; g.dims[0] = h_dims[0]; g.dims[1] = h_dims[1];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) g.tri.p[i][j] = h_triangle[2 * i + j];
    mask_kernel<3, PrismParams><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(d_center, d_level, d_cells, first, n,
                                                                                0.5 * width, g, refine_mode,
                                                                                keep_inside, d_invalid);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_mask_tetrahedra(const double *d_center, const int32_t *d_level, const int32_t *d_cells, int64_t first, int64_t n,
                       double width, const double *h_positions, const double *h_normals, int n_tets, int refine_mode,
                       int keep_inside, uint8_t *d_invalid, s3_stream stream) {
    if (int rc = check_mask_args(d_center, d_level, first, n, 3, d_invalid, "s3_mask_tetrahedra")) return rc;
    S3_REQUIRE(h_positions && h_normals && (n_tets == 1 || n_tets == 2), "s3_mask_tetrahedra: one or two tetrahedra");
    if (n == 0) return S3_OK;
    TetParams g{};
    g.n_tets = n_tets;
    for (int t = 0; t < n_tets; ++t) {
        for (int p = 0; p < 4; ++p)
            for (int j = 0; j < 3; ++j) g.pos[t][p][j] = h_positions[(t * 4 + p) * 3 + j];
        for (int j = 0; j < 3; ++j)
            for (int p = 0; p < 4; ++p) g.nrm[t][j][p] = h_normals[(t * 3 + j) * 4 + p];
    }
    mask_kernel<3, TetParams><<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(d_center, d_level, d_cells, first, n,
                                                                              0.5 * width, g, refine_mode, keep_inside,
                                                                              d_invalid);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_commit_batch(uint8_t *d_leaf, double *d_gain, const int32_t *d_parents, int64_t n_par, int64_t first,
                    int64_t n_new, const uint8_t *d_invalid, s3_stream stream) {
    S3_REQUIRE(n_par >= 0 && n_new >= 0 && first >= 0, "s3_commit_batch: bad range");
    S3_REQUIRE(d_leaf && d_gain && (n_par == 0 || d_parents), "s3_commit_batch: null array");
    int64_t n = std::max(n_par, n_new);
    if (n == 0) return S3_OK;
    commit_batch_kernel<<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(d_leaf, d_gain, d_parents, n_par, first, n_new,
                                                                        d_invalid);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_sumsq_leaf(const double *d_metric, const uint8_t *d_leaf, int64_t begin, int64_t end, double *d_out,
                  double *d_scratch, s3_stream stream) {
    S3_REQUIRE(d_metric && d_leaf && d_out && d_scratch, "s3_sumsq_leaf: null array");
    S3_REQUIRE(begin >= 0 && end >= begin, "s3_sumsq_leaf: bad range");
    int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (end - begin + 1023) / 1024));
    sumsq_partial_kernel<<<nb, 256, 0, as_stream(stream)>>>(d_metric, d_leaf, begin, end, d_scratch);
    sumsq_final_kernel<<<1, 256, 0, as_stream(stream)>>>(d_scratch, nb, d_out);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_sumsq_blocks(const double *d_metric, const uint8_t *d_leaf, int64_t n_cells, int64_t block_begin, int64_t block_end,
                    double *d_partial, s3_stream stream) {
    S3_REQUIRE(d_metric && d_leaf && d_partial, "s3_sumsq_blocks: null array");
    const int64_t n_blocks = (n_cells + S3_SUMSQ_BLOCK - 1) / S3_SUMSQ_BLOCK;
    S3_REQUIRE(n_cells >= 0 && block_begin >= 0 && block_end >= block_begin && block_end <= n_blocks,
               "s3_sumsq_blocks: bad block range [%lld, %lld) of %lld", (long long)block_begin, (long long)block_end, (long long)n_blocks);
    if (block_end == block_begin) return S3_OK;
    S3_REQUIRE(block_end - block_begin < ((int64_t)1 << 31), "s3_sumsq_blocks: too many blocks");
    sumsq_blocks_kernel<<<(unsigned)(block_end - block_begin), 256, 0, as_stream(stream)>>>(d_metric, d_leaf, n_cells, block_begin,
                                                                                          d_partial);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_sum_ordered(const double *d_values, int64_t n, double *d_out, s3_stream stream) {
    S3_REQUIRE(d_values && d_out && n >= 0, "s3_sum_ordered: bad arguments");
    sum_ordered_kernel<<<1, 256, 0, as_stream(stream)>>>(d_values, n, d_out);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

size_t s3_topn_scratch_bytes(int64_t n_cells, int64_t n_top) {
    (void)n_cells;
    if (n_top < 1) n_top = 1;
    return sizeof(uint32_t) * SEL_BINS + 64 + (size_t)n_top * (sizeof(int32_t) + sizeof(double) + 2 * sizeof(int32_t)) + 192;
}

int s3_topn_leaf(const double *d_gain, const uint8_t *d_leaf, int64_t n_cells, int64_t n_top, int32_t *h_out,
                 int64_t *h_count, void *d_scratch, s3_stream stream) {
    S3_REQUIRE(d_gain && d_leaf && h_out && h_count && d_scratch, "s3_topn_leaf: null argument");
    S3_REQUIRE(n_cells >= 0 && n_cells < ((int64_t)1 << 31) && n_top >= 0, "s3_topn_leaf: bad sizes");
    *h_count = 0;
    if (n_cells == 0 || n_top == 0) return S3_OK;
    hipStream_t st = as_stream(stream);
    char *base = static_cast<char *>(d_scratch);
    uint32_t *d_hist = reinterpret_cast<uint32_t *>(base);
    SelState *d_state = reinterpret_cast<SelState *>(base + sizeof(uint32_t) * SEL_BINS);
    static_assert(sizeof(SelState) <= 64, "the state fits the gap behind the histogram");
    double *d_og = reinterpret_cast<double *>(base + sizeof(uint32_t) * SEL_BINS + 64);
    int32_t *d_oi = reinterpret_cast<int32_t *>(base + sizeof(uint32_t) * SEL_BINS + 64 + sizeof(double) * n_top);
    unsigned grid = grid_for(n_cells, 256, 2048);

    // all digit passes and the collection are queued at once (the state of the selection lives on the device); one wait
    SelState init{0ull, 0u, 0, (long long)n_top, 0ull};
    S3_HIP_CHECK(hipMemsetAsync(d_hist, 0, sizeof(uint32_t) * SEL_BINS, st));
    S3_HIP_CHECK(hipMemcpyAsync(d_state, &init, sizeof(init), hipMemcpyHostToDevice, st));
    for (int d = 0; d < SEL_DIGITS; ++d) {
        select_hist_dev_kernel<<<grid, 256, 0, st>>>(d_gain, d_leaf, n_cells, d_state, d, d_hist);
        select_pick_kernel<<<1, 256, 0, st>>>(d_hist, d_state, d);
    }
    select_collect_dev_kernel<<<grid, 256, 0, st>>>(d_gain, d_leaf, n_cells, d_state, n_top, d_oi, d_og);
    const size_t cap = (size_t)std::min<int64_t>(n_top, n_cells);       // at most this many ids can have been selected
    // (d_oi ends on a multiple of 4 bytes behind 8-byte entries: the sorted ids follow 64-byte aligned)
    int32_t *d_sorted = reinterpret_cast<int32_t *>(base + ((sizeof(uint32_t) * SEL_BINS + 64 + (sizeof(double) + sizeof(int32_t)) * (size_t)n_top + 63) / 64) * 64);
    const bool on_device = (int64_t)cap <= RANK_MAX;
    int32_t *d_rank = d_sorted + ((cap + 15) / 16) * 16;
    if (on_device) {
        S3_HIP_CHECK(hipMemsetAsync(d_rank, 0, sizeof(int32_t) * cap, st));
        const dim3 rgrid(grid_for((int64_t)cap, 256), (unsigned)((cap + RANK_SPAN - 1) / RANK_SPAN));
        select_rank_kernel<<<rgrid, 256, 0, st>>>(d_oi, d_og, d_state, (int64_t)cap, d_rank);
        select_place_kernel<<<grid_for((int64_t)cap, 256), 256, 0, st>>>(d_oi, d_rank, d_state, (int64_t)cap, d_sorted);
    }
    S3_LAUNCH_CHECK();
    SelState fin{};
    std::vector<int32_t> ids(cap);
    std::vector<double> gs(on_device ? 0 : cap);
    S3_HIP_CHECK(hipMemcpyAsync(&fin, d_state, sizeof(fin), hipMemcpyDeviceToHost, st));
    S3_HIP_CHECK(hipMemcpyAsync(ids.data(), on_device ? d_sorted : d_oi, sizeof(int32_t) * cap, hipMemcpyDeviceToHost, st));
    if (!on_device) S3_HIP_CHECK(hipMemcpyAsync(gs.data(), d_og, sizeof(double) * cap, hipMemcpyDeviceToHost, st));
    S3_HIP_CHECK(hipStreamSynchronize(st));
    const unsigned long long cnt = fin.count;
    if ((int64_t)cnt > n_top || !fin.done) {
        s3::set_error("s3_topn_leaf: internal selection error (%llu selected, %lld wanted, done %d)", cnt, (long long)n_top, fin.done);
        return S3_EHIP;
    }
    if (on_device) {
        for (size_t i = 0; i < cnt; ++i) h_out[i] = ids[i];
    } else {
        // descending (gain, -id): the keys are formed once, then (key, id) records are sorted as a whole
        struct Rec { uint64_t hi; uint32_t lo; int32_t id; };
        std::vector<Rec> rec(cnt);
        for (size_t i = 0; i < cnt; ++i) {
            const Key96 kk = make_key(gs[i], (uint32_t)ids[i]);
            rec[i] = Rec{kk.hi, kk.lo, ids[i]};
        }
        std::sort(rec.begin(), rec.end(), [](const Rec &a, const Rec &b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); });
        for (size_t i = 0; i < cnt; ++i) h_out[i] = rec[i].id;
    }
    *h_count = (int64_t)cnt;
    return S3_OK;
}

}  // extern "C"
