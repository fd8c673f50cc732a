// Exact k-nearest-neighbour search on a uniform bucket grid + the KNN consumers of the S^3 refine path
// (inverse-distance regression, child-centre gain).  gfx950 only.
//
// Reference behaviour restated here (file:line relative to the reference checkout):
//   KNeighborsRegressor(k, weights="distance").fit/predict   s_cube.py:161-163,224,328,372
//   NearestNeighbors(k).fit/kneighbors                       export.py:120,423-425,438
//   SamplingTree._update_gain + numba _update_gain           s_cube.py:207-241,1840-1859
//
// Layout in HBM: the point cloud is copied once into bucket order (counting sort by grid cell) as an array of
// structures [n][dim] f64 so that one query thread streams whole points from consecutive addresses; cell_start[]
// (int32, ncell+1) gives each bucket's range; orig[] maps bucket order back to the caller's point ids; y[] holds the
// regression target in bucket order.  One thread owns one query; its running k-best list (squared distance f64 +
// bucket position i32) lives in LDS, laid out [slot][thread] so the 64 lanes of a wavefront hit 64 distinct banks.
// The search visits Chebyshev rings of buckets around the query's bucket and stops when the k-th best squared
// distance is smaller than the squared distance to the nearest unvisited bucket face.
//
// Arithmetic is kept bit-identical to the reference stack (no FMA contraction: this file is compiled with
// -ffp-contract=off): rdist = sum_j (q_j-p_j)^2 in dimension order, dist = sqrt(rdist), w = 1/dist, numpy pairwise
// summation of y*w and w.  Ties in rdist are ordered by the original point id.
#include "common.h"

#include <cfloat>
#include <cstdlib>
#include <cmath>
#include <utility>
#include <vector>

struct s3_knn {
    int dim = 0;
    int device = 0;
    int64_t n = 0;
    double lo[3] = {0, 0, 0}, h[3] = {1, 1, 1}, inv_h[3] = {1, 1, 1};
    int res[3] = {1, 1, 1};
    int64_t ncell = 1;
    double *pts = nullptr;         // [n][dim] bucket order
    int32_t *orig = nullptr;       // [n]
    int32_t *cell_start = nullptr; // [ncell+1]
    double *y = nullptr;           // [n] bucket order (optional)
    // second level: buckets holding more than `split` points carry their own r x r (x r) sub-lattice
    uint8_t *sub_res = nullptr;    // [ncell] 0 = plain bucket, else r
    int32_t *sub_off = nullptr;    // [ncell] offset of the bucket's table in sub_start
    int32_t *sub_start = nullptr;  // pooled tables, r^dim + 1 absolute positions each
    int64_t n_refined = 0;
};

namespace s3 {

constexpr int KNN_BLOCK = 128;

template <int DIM>
struct Grid {
    double lo[DIM], h[DIM], inv_h[DIM];
    int res[DIM];
    const uint8_t *sub_res;     // nullptr when no bucket is refined
    const int32_t *sub_off;
    const int32_t *sub_start;
};

constexpr int SUB_RES_MAX = 32;

template <int DIM>
static Grid<DIM> make_grid(const s3_knn *k) {
    Grid<DIM> g;
    for (int j = 0; j < DIM; ++j) {
        g.lo[j] = k->lo[j];
        g.h[j] = k->h[j];
        g.inv_h[j] = k->inv_h[j];
        g.res[j] = k->res[j];
    }
    g.sub_res = k->n_refined > 0 ? k->sub_res : nullptr;
    g.sub_off = k->sub_off;
    g.sub_start = k->sub_start;
    return g;
}

template <int DIM>
__device__ __forceinline__ int cell_coord(const Grid<DIM> &g, double x, int j) {
    double t = (x - g.lo[j]) * g.inv_h[j];
    t = fmin(fmax(t, 0.0), (double)(g.res[j] - 1));   // also maps NaN to 0
    return (int)t;
}

// ------------------------------------------------------------------------------------------------------------------
// build kernels
// ------------------------------------------------------------------------------------------------------------------
__global__ void bbox_kernel(const double *__restrict__ pts, int64_t n, int dim, double *__restrict__ partial) {
    __shared__ double smin[3][256], smax[3][256];
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int j = 0; j < dim; ++j) {
            double v = pts[i * dim + j];
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

template <int DIM>
__global__ void cell_count_kernel(Grid<DIM> g, const double *__restrict__ pts, int64_t n, int32_t *__restrict__ cid,
                                  int32_t *__restrict__ count) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t c = 0;
    for (int j = DIM - 1; j >= 0; --j) c = c * g.res[j] + cell_coord<DIM>(g, pts[i * DIM + j], j);
    cid[i] = (int32_t)c;
    atomicAdd(&count[c], 1);
}

// exclusive scan, 1024 items per block (4 per thread), three passes
__global__ void scan_block_kernel(int32_t *__restrict__ data, int64_t n, int32_t *__restrict__ block_sums) {
    __shared__ int32_t s[256];
    int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int32_t v[4], t = 0;
    for (int j = 0; j < 4; ++j) {
        v[j] = (base + j < n) ? data[base + j] : 0;
        t += v[j];
    }
    s[threadIdx.x] = t;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int32_t add = ((int)threadIdx.x >= off) ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += add;
        __syncthreads();
    }
    int32_t excl = s[threadIdx.x] - t;
    for (int j = 0; j < 4; ++j) {
        if (base + j < n) data[base + j] = excl;
        excl += v[j];
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = s[255];
}

__global__ void scan_sums_kernel(int32_t *__restrict__ block_sums, int64_t nb) {
    // single block, serial over chunks of 256
    __shared__ int32_t s[256];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 256) {
        int64_t i = base + threadIdx.x;
        int32_t t = i < nb ? block_sums[i] : 0;
        s[threadIdx.x] = t;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int32_t add = ((int)threadIdx.x >= off) ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nb) block_sums[i] = carry + s[threadIdx.x] - t;
        __syncthreads();
        if (threadIdx.x == 255) carry += s[255];
        __syncthreads();
    }
}

__global__ void scan_add_kernel(int32_t *__restrict__ data, int64_t n, const int32_t *__restrict__ block_sums) {
    int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int32_t add = block_sums[blockIdx.x];
    for (int j = 0; j < 4; ++j)
        if (base + j < n) data[base + j] += add;
}

template <int DIM>
__global__ void scatter_kernel(const double *__restrict__ pts, int64_t n, const int32_t *__restrict__ cid,
                               const int32_t *__restrict__ cell_start, int32_t *__restrict__ cursor,
                               double *__restrict__ out_pts, int32_t *__restrict__ orig) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t c = cid[i];
    int32_t pos = cell_start[c] + atomicAdd(&cursor[c], 1);
    for (int j = 0; j < DIM; ++j) out_pts[(int64_t)pos * DIM + j] = pts[i * DIM + j];
    orig[pos] = (int32_t)i;
}

// ---- second level ---------------------------------------------------------------------------------------------
template <int DIM>
__device__ __forceinline__ int64_t top_cell_of(const Grid<DIM> &g, const double *__restrict__ x, int (&cc)[3]) {
    int64_t c = 0;
    cc[0] = cc[1] = cc[2] = 0;
    for (int j = DIM - 1; j >= 0; --j) {
        cc[j] = cell_coord<DIM>(g, x[j], j);
        c = c * g.res[j] + cc[j];
    }
    return c;
}

template <int DIM>
__device__ __forceinline__ int sub_cell_of(const Grid<DIM> &g, const int (&cc)[3], int r, const double *__restrict__ x) {
    int s = 0;
    for (int j = DIM - 1; j >= 0; --j) {
        double t = (x[j] - (g.lo[j] + (double)cc[j] * g.h[j])) * ((double)r * g.inv_h[j]);
        t = fmin(fmax(t, 0.0), (double)(r - 1));
        s = s * r + (int)t;
    }
    return s;
}

// which buckets get a sub-lattice, and how fine: r = ceil((count / occupancy)^(1/dim)), capped
__global__ void sub_plan_kernel(const int32_t *__restrict__ cell_start, int64_t ncell, int split, double occ, int dim,
                                uint8_t *__restrict__ sub_res, int32_t *__restrict__ sub_size,
                                unsigned long long *__restrict__ n_refined) {
    int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c > ncell) return;
    if (c == ncell) { sub_size[c] = 0; return; }
    const int cnt = cell_start[c + 1] - cell_start[c];
    int r = 0;
    if (cnt > split) {
        r = (int)ceil(pow((double)cnt / occ, 1.0 / dim));
        r = min(max(r, 2), SUB_RES_MAX);
        atomicAdd(n_refined, 1ull);
    }
    sub_res[c] = (uint8_t)r;
    int sz = 0;
    if (r > 0) sz = (dim == 2 ? r * r : r * r * r) + 1;
    sub_size[c] = sz;
}

template <int DIM>
__global__ void sub_count_kernel(Grid<DIM> g, const double *__restrict__ pts, int64_t n, const uint8_t *__restrict__ sub_res,
                                 const int32_t *__restrict__ sub_off, int32_t *__restrict__ sub_start,
                                 int32_t *__restrict__ sid) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    int cc[3];
    const int64_t c = top_cell_of<DIM>(g, pts + p * DIM, cc);
    const int r = sub_res[c];
    if (r == 0) { sid[p] = -1; return; }
    const int sc = sub_cell_of<DIM>(g, cc, r, pts + p * DIM);
    sid[p] = sc;
    atomicAdd(&sub_start[sub_off[c] + sc], 1);
}

// per refined bucket: counts -> absolute start positions (exclusive scan + bucket base), closing entry = bucket end
__global__ void __launch_bounds__(64)
sub_scan_kernel(const int32_t *__restrict__ cell_start, const uint8_t *__restrict__ sub_res,
                const int32_t *__restrict__ sub_off, int32_t *__restrict__ sub_start, int dim) {
    const int64_t c = blockIdx.x;
    const int r = sub_res[c];
    if (r == 0) return;
    const int n = dim == 2 ? r * r : r * r * r;
    int32_t *t = sub_start + sub_off[c];
    int32_t carry = cell_start[c];
    for (int base = 0; base < n; base += 64) {
        const int i = base + threadIdx.x;
        const int32_t v = i < n ? t[i] : 0;
        int32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t u = __shfl_up(incl, off, 64);
            if ((int)threadIdx.x >= off) incl += u;
        }
        if (i < n) t[i] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (threadIdx.x == 0) t[n] = cell_start[c + 1];
}

template <int DIM>
__global__ void sub_scatter_kernel(Grid<DIM> g, const double *__restrict__ pts, const int32_t *__restrict__ orig, int64_t n,
                                   const int32_t *__restrict__ sid, const int32_t *__restrict__ sub_off,
                                   const int32_t *__restrict__ sub_start,
                                   int32_t *__restrict__ cursor, double *__restrict__ out_pts, int32_t *__restrict__ out_orig) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    int64_t dst = p;
    const int sc = sid[p];
    if (sc >= 0) {
        int cc[3];
        const int64_t c = top_cell_of<DIM>(g, pts + p * DIM, cc);
        dst = sub_start[sub_off[c] + sc] + atomicAdd(&cursor[sub_off[c] + sc], 1);
    }
    for (int j = 0; j < DIM; ++j) out_pts[dst * DIM + j] = pts[p * DIM + j];
    out_orig[dst] = orig[p];
}

__global__ void permute_values_kernel(const double *__restrict__ y, const int32_t *__restrict__ orig, int64_t n,
                                      double *__restrict__ out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = y[orig[i]];
}

// ------------------------------------------------------------------------------------------------------------------
// the search
// ------------------------------------------------------------------------------------------------------------------
struct KBest {
    double *sd;    // LDS, this thread's column: slot m at sd[m * stride]
    int32_t *sp;
    int k;
    int cnt;
    int stride;    // threads that share the LDS block (KNN_BLOCK; a few lanes when a wavefront searches for one cell)
    double worst;  // sd[k-1] once the list is full, +inf before
};

__device__ __forceinline__ bool kb_less(double d, int32_t p, double d2, int32_t p2, const int32_t *__restrict__ orig) {
    return d < d2 || (d == d2 && orig[p] < orig[p2]);
}

__device__ __forceinline__ void kb_offer(KBest &b, double d, int32_t p, const int32_t *__restrict__ orig) {
    if (b.cnt == b.k) {
        if (d > b.worst) return;
        if (d == b.worst && !(orig[p] < orig[b.sp[(b.k - 1) * b.stride]])) return;
    }
    int j = b.cnt < b.k ? b.cnt : b.k - 1;
    while (j > 0 && kb_less(d, p, b.sd[(j - 1) * b.stride], b.sp[(j - 1) * b.stride], orig)) {
        b.sd[j * b.stride] = b.sd[(j - 1) * b.stride];
        b.sp[j * b.stride] = b.sp[(j - 1) * b.stride];
        --j;
    }
    b.sd[j * b.stride] = d;
    b.sp[j * b.stride] = p;
    if (b.cnt < b.k) ++b.cnt;
    if (b.cnt == b.k) b.worst = b.sd[(b.k - 1) * b.stride];
}

template <int DIM>
__device__ __forceinline__ void scan_range(KBest &b, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                                           const double (&q)[DIM], int32_t p0, int32_t p1) {
    if (p0 >= p1) return;
    // the coordinates of the next two points are requested before the current one is offered to the list (whose
    // insertion is a data-dependent LDS loop the compiler does not move loads across): the search is bound by the
    // latency of these loads
    double c0[DIM], c1[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        c0[j] = pts[(int64_t)p0 * DIM + j];
        c1[j] = pts[(int64_t)min(p0 + 1, p1 - 1) * DIM + j];
    }
    for (int32_t p = p0; p < p1; ++p) {
        double c2[DIM];
#pragma unroll
        for (int j = 0; j < DIM; ++j) c2[j] = pts[(int64_t)min(p + 2, p1 - 1) * DIM + j];
        double d = 0.0;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            double t = q[j] - c0[j];
            d += t * t;
        }
        kb_offer(b, d, p, orig);
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            c0[j] = c1[j];
            c1[j] = c2[j];
        }
    }
}

// a uniform lattice of buckets: the top-level grid, or the sub-lattice of one refined bucket
template <int DIM>
struct Lattice {
    double lo[DIM], h[DIM], inv_h[DIM];
    int res[DIM];
};

// Visit the buckets of a lattice in Chebyshev rings around the bucket of q until the k-best list is full and its worst
// squared distance is below the squared distance to everything not yet visited (or the lattice is exhausted).
// run(row, y, z, xa, xb) is called for runs of buckets [xa, xb] of one lattice row (linear index row + x).
template <int DIM, typename RunFn>
__device__ __forceinline__ void ring_search(const Lattice<DIM> &L, const double (&q)[DIM], KBest &b, RunFn &&run) {
    int c[DIM];
    int rmax = 0;
    double hmin = DBL_MAX;
    // how far q lies outside the lattice along each axis (0 inside), shrunk a little to stay conservative
    double out[DIM], out2 = 0.0;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        double t = (q[j] - L.lo[j]) * L.inv_h[j];
        t = fmin(fmax(t, 0.0), (double)(L.res[j] - 1));   // also maps NaN to 0
        c[j] = (int)t;
        rmax = max(rmax, max(c[j], L.res[j] - 1 - c[j]));
        hmin = fmin(hmin, L.h[j]);
        const double hi = L.lo[j] + (double)L.res[j] * L.h[j];
        out[j] = fmax(0.0, fmax(L.lo[j] - q[j], q[j] - hi)) * (1.0 - 1e-9);
        out2 += out[j] * out[j];
    }
    const int c2 = DIM == 3 ? c[DIM - 1] : 0;
    const int res2 = DIM == 3 ? L.res[DIM - 1] : 1;

    // squared distance from q to the slab of bucket index i along axis j, shrunk a little to stay conservative (0 inside)
    auto gap2 = [&](int j, int i) {
        const double lo_i = L.lo[j] + (double)i * L.h[j];
        const double g = fmax(lo_i - q[j], q[j] - (lo_i + L.h[j])) - 1e-9 * hmin;
        return g > 0.0 ? g * g : 0.0;
    };
    for (int r = 0; r <= rmax; ++r) {
        const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, L.res[0] - 1);
        const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, L.res[1] - 1);
        const int z0 = DIM == 3 ? max(c2 - r, 0) : 0, z1 = DIM == 3 ? min(c2 + r, res2 - 1) : 0;
        for (int z = z0; z <= z1; ++z) {
            const int dz = DIM == 3 ? abs(z - c2) : 0;
            const double gz2 = DIM == 3 ? gap2(DIM - 1, z) : 0.0;
            for (int y = y0; y <= y1; ++y) {
                const int dy = abs(y - c[1]);
                const int64_t row = ((int64_t)z * L.res[1] + y) * L.res[0];
                // once the list is full, a bucket whose box lies farther away than the current worst entry cannot
                // contribute (the cube of a ring has eight times the volume of the sphere that matters): rows are
                // skipped and runs clipped by that test -- a superset of the needed buckets is still visited, the result
                // does not change
                const bool full = b.cnt == b.k;
                const double rest2 = gz2 + gap2(1, y);
                if (full && rest2 > b.worst) continue;
                if (max(dz, dy) == r) {
                    int xa = x0, xb = x1;             // the whole x-run of this row belongs to the ring
                    if (full) {
                        while (xa <= xb && rest2 + gap2(0, xa) > b.worst) ++xa;
                        while (xb >= xa && rest2 + gap2(0, xb) > b.worst) --xb;
                    }
                    if (xa <= xb) run(row, y, z, xa, xb);
                } else {
                    if (c[0] - r >= 0 && !(full && rest2 + gap2(0, c[0] - r) > b.worst)) run(row, y, z, c[0] - r, c[0] - r);
                    if (c[0] + r < L.res[0] && !(full && rest2 + gap2(0, c[0] + r) > b.worst)) run(row, y, z, c[0] + r, c[0] + r);
                }
            }
        }
        if (b.cnt == b.k) {
            // lower bound on the squared distance from q to any point in a bucket that has not been visited: such a
            // point lies beyond at least one face j of the visited box (distance >= face_j along j) and, like every
            // point of this lattice, inside it (distance >= out_i along every other axis i on which q lies outside)
            double bound2 = DBL_MAX;
#pragma unroll
            for (int j = 0; j < DIM; ++j) {
                const double rest = fmax(0.0, out2 - out[j] * out[j]);
                if (c[j] - r > 0) {
                    double f = q[j] - (L.lo[j] + (double)(c[j] - r) * L.h[j]) - 1e-9 * hmin;
                    if (f > 0.0) bound2 = fmin(bound2, f * f + rest); else bound2 = fmin(bound2, rest);
                }
                if (c[j] + r < L.res[j] - 1) {
                    double f = (L.lo[j] + (double)(c[j] + r + 1) * L.h[j]) - q[j] - 1e-9 * hmin;
                    if (f > 0.0) bound2 = fmin(bound2, f * f + rest); else bound2 = fmin(bound2, rest);
                }
            }
            if (bound2 == DBL_MAX) break;         // the box covers the whole lattice
            if (b.worst < bound2) break;
        }
    }
}

// sub-lattice of the refined top-level bucket with integer coordinates (x, y, z): r x r (x r) buckets over its box
template <int DIM>
__device__ __forceinline__ Lattice<DIM> sub_lattice(const Grid<DIM> &g, int x, int y, int z, int r) {
    Lattice<DIM> L;
    const int cc[3] = {x, y, z};
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        L.lo[j] = g.lo[j] + (double)cc[j] * g.h[j];
        L.h[j] = g.h[j] / (double)r;
        L.inv_h[j] = (double)r * g.inv_h[j];
        L.res[j] = r;
    }
    return L;
}

template <int DIM>
__device__ void knn_search(const Grid<DIM> &g, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                           const int32_t *__restrict__ cs, const double (&q)[DIM], KBest &b) {
    Lattice<DIM> top;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        top.lo[j] = g.lo[j];
        top.h[j] = g.h[j];
        top.inv_h[j] = g.inv_h[j];
        top.res[j] = g.res[j];
    }
    if (g.sub_res == nullptr) {
        // no refined bucket: a run of buckets is one contiguous range of points
        ring_search<DIM>(top, q, b, [&](int64_t row, int, int, int xa, int xb) {
            scan_range<DIM>(b, pts, orig, q, cs[row + xa], cs[row + xb + 1]);
        });
        return;
    }
    ring_search<DIM>(top, q, b, [&](int64_t row, int y, int z, int xa, int xb) {
        int64_t pending = -1;                       // first bucket of a run of plain buckets not scanned yet
        for (int x = xa; x <= xb; ++x) {
            const int64_t cell = row + x;
            const int r = g.sub_res[cell];
            if (r == 0) {
                if (pending < 0) pending = cell;
                continue;
            }
            if (pending >= 0) {
                scan_range<DIM>(b, pts, orig, q, cs[pending], cs[cell]);
                pending = -1;
            }
            // refined bucket: the same ring search on its sub-lattice; it stops as soon as nothing left in this bucket
            // can improve the list, the outer search then carries on with the next bucket
            const int32_t *st = g.sub_start + g.sub_off[cell];
            const Lattice<DIM> sub = sub_lattice<DIM>(g, x, y, z, r);
            ring_search<DIM>(sub, q, b, [&](int64_t srow, int, int, int sa, int sb) {
                scan_range<DIM>(b, pts, orig, q, st[srow + sa], st[srow + sb + 1]);
            });
        }
        if (pending >= 0) scan_range<DIM>(b, pts, orig, q, cs[pending], cs[row + xb + 1]);
    });
}

// numpy's pairwise summation of f(m), m = 0..k-1, for k <= 128 (numpy/_core/src/umath/loops_utils.h.src): eight
// interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), tail added sequentially
template <typename F>
__device__ __forceinline__ double numpy_pairwise(int k, F f) {
    if (k < 8) {
        double r = 0.0;
        for (int i = 0; i < k; ++i) r += f(i);
        return r;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f(j);
    int i = 8;
    for (; i < k - (k % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += f(i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < k; ++i) res += f(i);
    return res;
}

__device__ __forceinline__ double idw_from_list(const KBest &b, const double *__restrict__ y) {
    bool has_zero = false;
    for (int m = 0; m < b.k; ++m) has_zero |= (b.sd[m * b.stride] == 0.0);
    auto wgt = [&](int m) {
        double rd = b.sd[m * b.stride];
        return has_zero ? (rd == 0.0 ? 1.0 : 0.0) : 1.0 / sqrt(rd);
    };
    double num = numpy_pairwise(b.k, [&](int m) { return y[b.sp[m * b.stride]] * wgt(m); });
    double den = numpy_pairwise(b.k, wgt);
    return num / den;
}

__device__ __forceinline__ KBest kb_init(double *lds, int k) {
    KBest b;
    b.sd = lds + threadIdx.x;
    b.sp = reinterpret_cast<int32_t *>(lds + (size_t)k * KNN_BLOCK) + threadIdx.x;
    b.k = k;
    b.cnt = 0;
    b.stride = KNN_BLOCK;
    b.worst = DBL_MAX;
    return b;
}

template <int DIM>
__global__ void __launch_bounds__(KNN_BLOCK)
knn_query_kernel(Grid<DIM> g, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                 const int32_t *__restrict__ cs, const double *__restrict__ q_in, int64_t nq, int k,
                 int32_t *__restrict__ idx_out, double *__restrict__ dist_out) {
    extern __shared__ double lds[];
    int64_t i = blockIdx.x * (int64_t)KNN_BLOCK + threadIdx.x;
    if (i >= nq) return;
    double q[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) q[j] = q_in[i * DIM + j];
    KBest b = kb_init(lds, k);
    knn_search<DIM>(g, pts, orig, cs, q, b);
    for (int m = 0; m < k; ++m) {
        idx_out[i * k + m] = orig[b.sp[m * KNN_BLOCK]];
        dist_out[i * k + m] = sqrt(b.sd[m * KNN_BLOCK]);
    }
}

template <int DIM>
__global__ void __launch_bounds__(KNN_BLOCK)
idw_predict_kernel(Grid<DIM> g, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                   const int32_t *__restrict__ cs, const double *__restrict__ y, const double *__restrict__ q_in,
                   int64_t nq, int k, double *__restrict__ yhat) {
    extern __shared__ double lds[];
    int64_t i = blockIdx.x * (int64_t)KNN_BLOCK + threadIdx.x;
    if (i >= nq) return;
    double q[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) q[j] = q_in[i * DIM + j];
    KBest b = kb_init(lds, k);
    knn_search<DIM>(g, pts, orig, cs, q, b);
    yhat[i] = idw_from_list(b, y);
}

// a3+a4 first half: query t = (cell i, point j) with j = 0 the cell centre and j = 1..2^DIM its candidate children.
// REUSE: the centre of new cell first + i IS candidate child (i mod 2^DIM) of its parent, a point whose metric was
// predicted (by this kernel, from the same coordinates: make_children_kernel and the expression below are the same) when
// the parent was created -- the reference predicts it again (s_cube.py:221-233) and gets the same number, so only the
// 2^DIM child points are searched here and the centre's value is taken from child_metric[parent].  Either way the child
// values of every new cell are kept in child_metric[cell] for the day the cell is refined.
template <int DIM, bool REUSE>
__global__ void __launch_bounds__(KNN_BLOCK)
child_metric_kernel(Grid<DIM> g, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                    const int32_t *__restrict__ cs, const double *__restrict__ y, const double *__restrict__ center,
                    const int32_t *__restrict__ level, int64_t first, int64_t n, double quarter_width, int k,
                    double *__restrict__ metric_all, const int32_t *__restrict__ parents, int64_t parents_offset,
                    double *__restrict__ child_metric) {
    constexpr int NCH = 1 << DIM, NQ = NCH + 1, PER = REUSE ? NCH : NQ;
    extern __shared__ double lds[];
    int64_t t = blockIdx.x * (int64_t)KNN_BLOCK + threadIdx.x;
    if (t >= n * PER) return;
    int64_t i = t / PER;
    int jq = (int)(t - i * PER) + (REUSE ? 1 : 0);
    int64_t cell = first + i;
    double q[DIM];
    double off = cell_offset(quarter_width, level[cell]);
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        double cj = center[cell * DIM + j];
        q[j] = jq == 0 ? cj : cj + dir_comp(DIM, jq - 1, j) * off;
    }
    KBest b = kb_init(lds, k);
    knn_search<DIM>(g, pts, orig, cs, q, b);
    const double m = idw_from_list(b, y);
    metric_all[i * NQ + jq] = m;
    if (child_metric != nullptr && jq > 0) child_metric[cell * NCH + jq - 1] = m;
    if (REUSE && jq == 1) {
        const int64_t ii = parents_offset + i;               // position of the cell among the batch's children
        metric_all[i * NQ] = child_metric[(int64_t)parents[ii / NCH] * NCH + (ii % NCH)];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same 2^DIM child predictions of a new cell by ONE WAVEFRONT (s3_child_gain_reuse): selection instead of insertion.
//
// The per-lane search above is bound by instruction issue at 24 % lane utilisation: every lane walks its own rings, rows
// and runs, and inserts into its own sorted list at its own moments.  The 2^DIM child points of one cell lie within a
// quarter of the cell's width of each other, so they share their candidates:
//   1. box: the buckets within the child points' reach of the cell centre's bucket, per axis and side (2 or 3 of them at two
//      points per bucket; rows of buckets = contiguous runs of points: one lane per row fetches their bounds, a wave scan
//      gives every candidate its slot); at most COOP_M points, no refined bucket, or the cell is searched the per-lane way;
//   2. the points' coordinates go to LDS once (coalesced), 64 / 2^DIM lanes per child point then share the candidates;
//   3. pass A: squared distances (same expression as the per-lane search: no contraction, dimension order) below the
//      SAFE radius of the point -- its distance to the nearest face of the box that has unvisited buckets behind it -- are
//      counted in 64 bins; the bin where the count reaches k gives a threshold with k .. k + a few candidates below it;
//      fewer than k candidates inside the safe radius, or more than COOP_CAP below the threshold (ties in bulk): per-lane kernel;
//   4. pass B: the candidates below the threshold are compacted into a short list, every entry is ranked by (distance,
//      original point id) against the others, the k best land in order -- the k nearest neighbours exactly as the per-lane
//      search returns them, since every point nearer than the safe radius is in the box;
//   5. inverse-distance prediction with numpy's pairwise order (eight partial sums on eight lanes).
// Cells the wavefront cannot take are searched the per-lane way by 2^DIM of its lanes on the spot: the results are the same
// bits either way (tools/knn_coop_probe.py, every refine golden).
// ------------------------------------------------------------------------------------------------------------------
constexpr int COOP_M = 384;        // candidates per cell held in LDS
constexpr int COOP_SLOTS = COOP_M / 64;
constexpr int COOP_NB = 64;        // histogram bins per child point
constexpr int COOP_CAP = 48;       // entries of the short list per child point (k plus the rest of the k-th bin)
constexpr int COOP_RMAX = 3;       // at most this many buckets on either side of the centre's bucket
constexpr int COOP_ROWS = 64;      // >= (2 RMAX + 1)^(DIM - 1) rows of buckets, one lane each
constexpr int COOP_WAVES = 2;      // wavefronts (cells) per workgroup

// 17 KiB per wavefront: nine wavefronts per CU.  A wavefront's life is a chain of dependent memory round trips (cell -> bucket
// bounds -> points -> values), so what counts is how many of them a CU holds and how few trips each needs: the loops below
// are unrolled so that the loads of several slots are in flight together.
template <int DIM>
struct CoopLds {
    union {
        double xyz[COOP_M][DIM];                   // candidates' coordinates (until the short lists are built)
        struct {
            double list_y[1 << DIM][COOP_CAP];     // afterwards: value and weight of the k best
            double list_w[1 << DIM][COOP_CAP];
        } best;
    };
    int32_t pos[COOP_M];
    uint32_t hist[1 << DIM][COOP_NB + 1];          // (+1: the child points' rows start in different banks)
    double list_d[1 << DIM][COOP_CAP];
    int32_t list_p[1 << DIM][COOP_CAP];
    int32_t row_start[COOP_ROWS], row_prefix[COOP_ROWS + 1];
    uint32_t count[1 << DIM];
};
static_assert(sizeof(double) * COOP_M * 2 >= sizeof(double) * 4 * COOP_CAP * 2, "the lists of the best fit the coordinates' space");

__device__ __forceinline__ void wave_sync_lds() {
    // one wavefront: LDS operations execute in program order; this only keeps the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One cooperative search: the 64 / LPQ query points c + dir(jq) * off (LPQ lanes each; LPQ = 64: the single point c) share
// the box around the bucket of c.  reach = radius (in bucket sides, per axis) the box must cover around the query points:
// the expected distance of the k-th neighbour at the index's average occupancy times a margin.  Returns false (for the
// whole wavefront) when the scheme cannot answer -- reach beyond COOP_RMAX buckets, a refined bucket in the box, more than
// COOP_M candidates, fewer than k inside a point's safe radius, more than COOP_CAP below the threshold, a tie among the
// entries that matter; otherwise lane 0 of every group returns its point's prediction in `result`.
// (returns 0 = answered, 1 = the points are too far apart to share a box -- a coarse cell --, 2 = any other reason)
template <int DIM, int LPQ>
__device__ __forceinline__ int coop_solve(const Grid<DIM> &g, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                                           const int32_t *__restrict__ cs, const double *__restrict__ y, CoopLds<DIM> &L,
                                           const double (&c)[DIM], double off, int k, double reach, int lane, double &result) {
    constexpr int BPL = COOP_NB / LPQ < 1 ? 1 : COOP_NB / LPQ;
    constexpr int CHUNK = LPQ >= 64 ? 2 : 4;                  // candidates whose coordinates a lane reads together
    constexpr int ROUNDS = COOP_CAP / LPQ < 1 ? 1 : COOP_CAP / LPQ;
    constexpr int NQB = 64 / LPQ;
    static_assert(LPQ >= 8 && (COOP_NB % LPQ == 0 || LPQ > COOP_NB) && COOP_CAP <= 64, "lane layout");

    // ---- 1. the box: as many buckets on either side of the centre's bucket as the query points' reach needs ---------------
    int lo_i[3] = {0, 0, 0}, hi_i[3] = {0, 0, 0};
    double hmin = DBL_MAX;
    bool too_far = false;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const int b = cell_coord<DIM>(g, c[j], j);
        const double frac = fmin(fmax((c[j] - g.lo[j]) * g.inv_h[j] - (double)b, 0.0), 1.0);   // position inside the bucket
        const double want = off * g.inv_h[j] + reach;                                          // in bucket sides
        const int r_lo = (int)ceil(want - frac), r_hi = (int)ceil(want - (1.0 - frac));
        too_far = too_far || r_lo > COOP_RMAX || r_hi > COOP_RMAX;
        lo_i[j] = max(b - max(r_lo, 0), 0);
        hi_i[j] = min(b + max(r_hi, 0), g.res[j] - 1);
        hmin = fmin(hmin, g.h[j]);
    }
    if (too_far) return 1;                                    // the points do not share candidates (coarse cells)
    const int ny = hi_i[1] - lo_i[1] + 1, nz = DIM == 3 ? hi_i[2] - lo_i[2] + 1 : 1, n_rows = ny * nz;
    int my_start = 0, my_cnt = 0;
    bool refined = false;
    if (lane < n_rows) {
        const int zz = lane / ny, yy = lane - zz * ny;
        const int64_t row = ((int64_t)(DIM == 3 ? lo_i[2] + zz : 0) * g.res[1] + (lo_i[1] + yy)) * g.res[0];
        my_start = cs[row + lo_i[0]];
        my_cnt = cs[row + hi_i[0] + 1] - my_start;
        if (g.sub_res != nullptr)
            for (int x = lo_i[0]; x <= hi_i[0]; ++x) refined |= g.sub_res[row + x] != 0;
    }
    int incl = my_cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(incl, d, 64);
        if (lane >= d) incl += u;
    }
    const int M = __shfl(incl, 63, 64);
    if (__ballot(refined) != 0ull || M > COOP_M || M < k) return 2;
    wave_sync_lds();                                          // (an earlier attempt's reads of these arrays are done)
    if (lane < n_rows) {
        L.row_start[lane] = my_start;
        L.row_prefix[lane] = incl - my_cnt;
    }
    if (lane == 0) L.row_prefix[COOP_ROWS] = M;
    if (lane < NQB) L.count[lane] = 0;
    for (int t = lane; t < NQB * (COOP_NB + 1); t += 64) (&L.hist[0][0])[t] = 0;
    wave_sync_lds();

    // ---- 2. candidates -> LDS: four slots of a lane at a time searched, then loaded together, then stored -------------------
#pragma unroll 1
    for (int base = 0; base < M; base += 64 * 4) {
        int slot_p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = min(base + lane + 64 * u, M - 1);
            int r = 0;                                        // the row whose slots hold t: last r with row_prefix[r] <= t
#pragma unroll
            for (int step = COOP_ROWS / 2; step > 0; step >>= 1)
                if (r + step < n_rows && L.row_prefix[r + step] <= t) r += step;
            slot_p[u] = L.row_start[r] + (t - L.row_prefix[r]);
        }
        double slot_x[4][DIM];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < DIM; ++j) slot_x[u][j] = pts[(int64_t)slot_p[u] * DIM + j];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = base + lane + 64 * u;
            if (t < M) {
                L.pos[t] = slot_p[u];
#pragma unroll
                for (int j = 0; j < DIM; ++j) L.xyz[t][j] = slot_x[u][j];
            }
        }
    }
    wave_sync_lds();

    // ---- 3. this lane's query point, its safe radius, ONE pass over the distances (kept in registers) ----------------------
    const int jq = lane / LPQ, gl = lane - jq * LPQ;
    double q[DIM], safe2 = DBL_MAX, far2 = 0.0;
    bool inside = true;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        q[j] = NQB == 1 ? c[j] : c[j] + dir_comp(DIM, jq, j) * off;
        const double face_lo = g.lo[j] + (double)lo_i[j] * g.h[j], face_hi = g.lo[j] + (double)(hi_i[j] + 1) * g.h[j];
        if (lo_i[j] > 0) {                                    // buckets behind this face exist and were not visited
            const double f = q[j] - face_lo - 1e-9 * hmin;
            inside = inside && f > 0.0;
            safe2 = fmin(safe2, f * f);
        }
        if (hi_i[j] < g.res[j] - 1) {
            const double f = face_hi - q[j] - 1e-9 * hmin;
            inside = inside && f > 0.0;
            safe2 = fmin(safe2, f * f);
        }
        const double span = fmax(fabs(q[j] - face_lo), fabs(face_hi - q[j]));
        far2 += span * span;
    }
    if (__ballot(!inside) != 0ull) return 1;                  // a query point at or beyond a face of the box
    const double lim2 = fmin(safe2, far2 * 1.0000001 + 1e-300);   // no limiting face: every point of the grid is in the box
    const double to_bin = (double)COOP_NB / lim2;
    auto dist2 = [&](int t) {
        double d = 0.0;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            const double w_ = q[j] - L.xyz[t][j];
            d += w_ * w_;
        }
        return d;
    };
    // (groups of CHUNK candidates: their coordinates are read together, one LDS round trip per group instead of per candidate)
    for (int t0 = gl; t0 < M; t0 += CHUNK * LPQ) {
        double d[CHUNK];
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) d[u] = dist2(min(t0 + u * LPQ, M - 1));
#pragma unroll
        for (int u = 0; u < CHUNK; ++u)
            if (t0 + u * LPQ < M && d[u] < lim2) atomicAdd(&L.hist[jq][min(COOP_NB - 1, (int)(d[u] * to_bin))], 1u);
    }
    wave_sync_lds();
    // the bin in which the count reaches k: lane gl of the group owns bins gl * BPL .. (lanes beyond the bins: none)
    uint32_t bins[BPL], mine = 0;
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
        bins[b] = gl * BPL + b < COOP_NB ? L.hist[jq][gl * BPL + b] : 0u;
        mine += bins[b];
    }
    uint32_t upto = mine;
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) {
        const uint32_t u = __shfl_up(upto, d, LPQ);
        if (gl >= d) upto += u;
    }
    const uint32_t total = __shfl(upto, LPQ - 1, LPQ);
    int b_star = -1;
    uint32_t below = 0;
    {
        uint32_t run = upto - mine;
        const bool owner = run < (uint32_t)k && upto >= (uint32_t)k;       // exactly one lane of the group
#pragma unroll
        for (int b = 0; b < BPL; ++b) {
            run += bins[b];
            if (owner && b_star < 0 && run >= (uint32_t)k) { b_star = gl * BPL + b; below = run; }
        }
    }
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) {                       // the owner's values to the whole group
        b_star = max(b_star, __shfl_xor(b_star, d, LPQ));
        below = max(below, __shfl_xor(below, d, LPQ));
    }
    if (__ballot(total < (uint32_t)k || below > (uint32_t)COOP_CAP) != 0ull) return 2;

    // ---- 4. short list, ranks, the k best in order -----------------------------------------------------------------------
    for (int t0 = gl; t0 < M; t0 += CHUNK * LPQ) {           // (the distances again: cheaper than 2 x 48 registers per lane)
        double d[CHUNK];
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) d[u] = dist2(min(t0 + u * LPQ, M - 1));
#pragma unroll
        for (int u = 0; u < CHUNK; ++u)
            if (t0 + u * LPQ < M && d[u] < lim2 && min(COOP_NB - 1, (int)(d[u] * to_bin)) <= b_star) {
                const uint32_t at = atomicAdd(&L.count[jq], 1u);
                L.list_d[jq][at] = d[u];
                L.list_p[jq][at] = L.pos[t0 + u * LPQ];
            }
    }
    wave_sync_lds();
    const int n_list = (int)below;
    double e_d[ROUNDS];
    int32_t e_p[ROUNDS], e_less[ROUNDS], e_same[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int e = min(gl + r * LPQ, n_list - 1);
        e_d[r] = L.list_d[jq][e];
        e_p[r] = L.list_p[jq][e];
        e_less[r] = e_same[r] = 0;
    }
    const int n_rounds = (n_list + LPQ - 1) / LPQ;            // (uniform within the group, nearly always within the wavefront)
#pragma unroll 4
    for (int m = 0; m < n_list; ++m) {                        // every entry of the list against this lane's entries
        const double dm = L.list_d[jq][m];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if (r < n_rounds) {
                e_less[r] += dm < e_d[r] ? 1 : 0;
                e_same[r] += dm == e_d[r] ? 1 : 0;
            }
    }
    // two entries at exactly the same distance among the ones that matter (lattices of points): the order among them is
    // the original point id -- left to the per-lane search, which breaks the tie that way
    bool tie = false;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) tie |= gl + r * LPQ < n_list && e_same[r] > 1 && e_less[r] < k;
    if (__ballot(tie) != 0ull) return 2;
    wave_sync_lds();
    // value and inverse-distance weight of the k best, by the lanes that hold them: one round trip for all values
    double e_y[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (!(gl + r * LPQ < n_list)) e_less[r] = COOP_CAP;
        e_y[r] = e_less[r] < k ? y[e_p[r]] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        if (e_less[r] < k) {
            L.list_d[jq][e_less[r]] = e_d[r];
            L.best.list_y[jq][e_less[r]] = e_y[r];
            L.best.list_w[jq][e_less[r]] = 1.0 / sqrt(e_d[r]);
        }
    wave_sync_lds();

    // ---- 5. inverse-distance prediction, numpy's pairwise order (idw_from_list / numpy_pairwise above) --------------------
    bool zero = false;
    for (int m = gl; m < k; m += LPQ) zero |= L.list_d[jq][m] == 0.0;
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) zero |= __shfl_xor((int)zero, d, LPQ) != 0;
    auto wgt = [&](int m) { return zero ? (L.list_d[jq][m] == 0.0 ? 1.0 : 0.0) : L.best.list_w[jq][m]; };
    auto term = [&](int m) { return L.best.list_y[jq][m] * wgt(m); };
    double num = 0.0, den = 0.0;
    if (k < 8) {
        if (gl == 0)
            for (int m = 0; m < k; ++m) {
                num += term(m);
                den += wgt(m);
            }
    } else {
        double rn = 0.0, rw = 0.0;
        if (gl < 8) {
            rn = term(gl);
            rw = wgt(gl);
            for (int m = 8; m < k - (k % 8); m += 8) {
                rn += term(m + gl);
                rw += wgt(m + gl);
            }
        }
        // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) on lane 0 of the group
        rn += __shfl_xor(rn, 1, LPQ); rw += __shfl_xor(rw, 1, LPQ);        // lanes 0,2,4,6: r0+r1, r2+r3, ...
        rn += __shfl_xor(rn, 2, LPQ); rw += __shfl_xor(rw, 2, LPQ);        // lanes 0,4: (r0+r1)+(r2+r3), (r4+r5)+(r6+r7)
        rn += __shfl_xor(rn, 4, LPQ); rw += __shfl_xor(rw, 4, LPQ);
        num = rn;
        den = rw;
        if (gl == 0)
            for (int m = k - (k % 8); m < k; ++m) {
                num += term(m);
                den += wgt(m);
            }
    }
    result = num / den;
    wave_sync_lds();
    return 0;
}

template <int DIM>
__global__ void __launch_bounds__(64 * COOP_WAVES)
child_metric_coop_kernel(Grid<DIM> g, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                         const int32_t *__restrict__ cs, const double *__restrict__ y, const double *__restrict__ center,
                         const int32_t *__restrict__ level, int64_t first, int64_t n, double quarter_width, int k, double reach,
                         double *__restrict__ metric_all, const int32_t *__restrict__ parents, int64_t parents_offset,
                         double *__restrict__ child_metric, int32_t *__restrict__ rest /*[0]: count, [2 ..]: queries left over*/) {
    constexpr int NCH = 1 << DIM, NQ = NCH + 1, LPQ = 64 / NCH;
    __shared__ CoopLds<DIM> lds_all[COOP_WAVES];
    const int lane = threadIdx.x & 63;
    const int64_t i = blockIdx.x * (int64_t)COOP_WAVES + (threadIdx.x >> 6);
    if (i >= n) return;                                       // (uniform per wavefront)
    CoopLds<DIM> &L = lds_all[threadIdx.x >> 6];
    const int64_t cell = first + i;
    // the centre's value from the parent's entry
    if (lane == 0) {
        const int64_t ii = parents_offset + i;
        metric_all[i * NQ] = child_metric[(int64_t)parents[ii / NCH] * NCH + (ii % NCH)];
    }
    double c[DIM];
    const double off = cell_offset(quarter_width, level[cell]);
#pragma unroll
    for (int j = 0; j < DIM; ++j) c[j] = center[cell * DIM + j];
    double m = 0.0;
    // (a second attempt with a wider box for the cells that find too few candidates inside their safe radius was measured:
    //  it halves the cells left over but the kernel carrying both attempts is slower by more than that saves -- the cells
    //  near a body or the edge of the cloud need searches much wider than any box that fits here)
    const int code = k <= COOP_CAP ? coop_solve<DIM, LPQ>(g, pts, orig, cs, y, L, c, off, k, reach, lane, m) : 2;
    if (code == 0) {
        if (lane % LPQ == 0) {
            metric_all[i * NQ + 1 + lane / LPQ] = m;
            child_metric[cell * NCH + lane / LPQ] = m;
        }
        return;
    }
    // Coarse cells (child points several buckets apart: no shared candidates), more candidates than the box holds, refined
    // buckets in the box, a thin stretch of the cloud, ties in distance, k > COOP_CAP: the cell's child points are listed for
    // child_metric_near_kernel
    if (lane < NCH) rest[2 + atomicAdd(&rest[0], 1)] = (int32_t)(i * NCH + lane);
}

// ------------------------------------------------------------------------------------------------------------------
// The queries the wavefront-per-cell kernel leaves over -- child points of cells too large for one shared box, points next to
// a body, at the edge of the cloud or outside it (the tree's root cell is a cube) -- need searches of their own, and the
// per-lane search takes 0.2 - 1.5 ms of dependent loads for them however few there are.  child_metric_far_kernel streams the
// candidates instead of caching them, in two stages: near_solve (2^DIM queries per wavefront, each group of lanes its own small
// box) and, for what that leaves, far_solve: ONE WAVEFRONT PER QUERY, boxes that hold balls of growing radius (FAR_BALLS x reach
// beyond the grid's nearest point; rows of buckets = contiguous runs of points, their bounds and a prefix sum in LDS), two
// passes over the box's points read straight from the index (histogram of the squared distances inside the safe radius ->
// threshold; compaction of the candidates below it), then ranks by counting and the pairwise-ordered prediction exactly as
// in coop_solve.  What is still left (refined buckets, ties in distance, k > COOP_CAP, a box of more than FAR_MAX_POINTS
// points) goes to the per-lane kernel.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FAR_ROWS = 1024;            // >= (2 * max radius + 1)^(DIM - 1)
constexpr int FAR_NB = 256;               // histogram bins
constexpr int FAR_MAX_POINTS = 1 << 16;
constexpr int FAR_UNROLL = 4;            // slots of the box a lane has in flight per pass step (one round trip each step)
constexpr int FAR_RMAX = 15;             // buckets on either side of the query's bucket, at most
__device__ constexpr double FAR_BALLS[4] = {1.8, 3.5, 7.0, 13.0};   // radius of the ball the box must hold, in units of `reach`

struct FarLds {
    int32_t row_start[FAR_ROWS];
    int32_t row_prefix[FAR_ROWS + 1];
    uint32_t hist[FAR_NB];
    double list_d[COOP_CAP], list_y[COOP_CAP], list_w[COOP_CAP];
    int32_t list_p[COOP_CAP];
    uint32_t count;
};

template <int DIM>
__device__ __forceinline__ bool far_solve(const Grid<DIM> &g, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                                          const int32_t *__restrict__ cs, const double *__restrict__ y, FarLds &L,
                                          const double (&q)[DIM], int k, double ball, int lane, bool &hopeless,
                                          double &result) {
    // ---- the box: what the ball of `ball` bucket sides beyond the grid's nearest point needs, per axis and side ----------------
    // (a query outside the grid at distance D_i along axis i -- the tree's root cell is a cube around an oblong cloud -- : every
    //  point is at least D_i away along that axis, so the box need not be a cube around the query and the points beyond a face
    //  of axis j are at least sqrt(f_j^2 + sum_{i != j} D_i^2) away)
    int lo_i[3] = {0, 0, 0}, hi_i[3] = {0, 0, 0};
    double hmin = DBL_MAX, hmax = 0.0, safe2 = DBL_MAX, far2 = 0.0, out[3] = {0.0, 0.0, 0.0}, out2 = 0.0;
    bool whole_grid = true;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        hmin = fmin(hmin, g.h[j]);
        hmax = fmax(hmax, g.h[j]);
    }
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const double top = g.lo[j] + (double)g.res[j] * g.h[j];
        out[j] = fmax(fmax(g.lo[j] - q[j], q[j] - top) - 1e-9 * hmin, 0.0);
        out2 += out[j] * out[j];
    }
    // inside the grid: the ball itself.  At distance D outside it: the cap of the ball around q that reaches into the grid is to
    // hold as much as that ball would, height^2 (3 D + 2 height) = 4 ball^3 -- both bounds below overestimate the height
    const double dist_out = sqrt(out2), ball_phys = ball * hmax;
    const double cap = dist_out > 0.0 ? fmin(1.26 * ball_phys, sqrt(4.0 * ball_phys * ball_phys * ball_phys / (3.0 * dist_out)))
                                      : ball_phys;
    const double ball_len = dist_out + cap;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const int b = cell_coord<DIM>(g, q[j], j);
        const double pos = (q[j] - g.lo[j]) * g.inv_h[j] - (double)b;             // inside the grid: 0 .. 1
        const double want = sqrt(fmax(ball_len * ball_len - (out2 - out[j] * out[j]), 0.0)) * g.inv_h[j];
        const int r_lo = min(max((int)ceil(want - pos), 0), FAR_RMAX), r_hi = min(max((int)ceil(want - (1.0 - pos)), 0), FAR_RMAX);
        lo_i[j] = max(b - r_lo, 0);
        hi_i[j] = min(b + r_hi, g.res[j] - 1);
    }
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const double face_lo = g.lo[j] + (double)lo_i[j] * g.h[j], face_hi = g.lo[j] + (double)(hi_i[j] + 1) * g.h[j];
        const double others2 = fmax(out2 - out[j] * out[j], 0.0);
        if (lo_i[j] > 0) {
            const double f = q[j] - face_lo - 1e-9 * hmin;
            safe2 = fmin(safe2, f > 0.0 ? f * f + others2 : 0.0);
            whole_grid = false;
        }
        if (hi_i[j] < g.res[j] - 1) {
            const double f = face_hi - q[j] - 1e-9 * hmin;
            safe2 = fmin(safe2, f > 0.0 ? f * f + others2 : 0.0);
            whole_grid = false;
        }
        const double span = fmax(fabs(q[j] - face_lo), fabs(face_hi - q[j]));
        far2 += span * span;
    }
    const int ny = hi_i[1] - lo_i[1] + 1, nz = DIM == 3 ? hi_i[2] - lo_i[2] + 1 : 1, n_rows = ny * nz;
    if (n_rows > FAR_ROWS) { hopeless = true; return false; }
    wave_sync_lds();
    // every lane a contiguous share of the rows: bounds, local prefix, wave scan
    const int per = (n_rows + 63) / 64, r0 = lane * per, r1 = min(r0 + per, n_rows);
    int mine = 0;
    bool refined = false;
    for (int r = r0; r < r1; ++r) {
        const int zz = r / ny, yy = r - zz * ny;
        const int64_t row = ((int64_t)(DIM == 3 ? lo_i[2] + zz : 0) * g.res[1] + (lo_i[1] + yy)) * g.res[0];
        const int start = cs[row + lo_i[0]], cnt = cs[row + hi_i[0] + 1] - start;
        L.row_start[r] = start;
        L.row_prefix[r] = mine;                            // (local; the lane's offset is added below)
        mine += cnt;
        if (g.sub_res != nullptr)
            for (int x = lo_i[0]; x <= hi_i[0]; ++x) refined |= g.sub_res[row + x] != 0;
    }
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(incl, d, 64);
        if (lane >= d) incl += u;
    }
    const int M = __shfl(incl, 63, 64);
    for (int r = r0; r < r1; ++r) L.row_prefix[r] += incl - mine;
    if (lane == 0) L.count = 0;
    for (int t = lane; t < FAR_NB; t += 64) L.hist[t] = 0;
    if (__ballot(refined) != 0ull || M > FAR_MAX_POINTS) { hopeless = true; return false; }
    if (M < k) { hopeless = whole_grid; return false; }      // a wider box will do (unless this one is the whole grid)
    wave_sync_lds();
    const double lim2 = fmin(safe2, far2 * 1.0000001 + 1e-300);
    if (!(lim2 > out2)) return false;
    // (no point is nearer than the grid: the bins span out2 .. lim2, or the candidates of a query far outside would share a few)
    const double to_bin = (double)FAR_NB / (lim2 - out2);
    auto bin_of = [&](double d) { return min(FAR_NB - 1, max((int)((d - out2) * to_bin), 0)); };
    auto point_of = [&](int t) {                             // slot t of the box -> position in the index
        int r = 0;
#pragma unroll
        for (int step = FAR_ROWS / 2; step > 0; step >>= 1)
            if (r + step < n_rows && L.row_prefix[r + step] <= t) r += step;
        return L.row_start[r] + (t - L.row_prefix[r]);
    };
    auto dist2 = [&](int p) {
        double d = 0.0;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            const double u = q[j] - pts[(int64_t)p * DIM + j];
            d += u * u;
        }
        return d;
    };
    // ---- pass A: histogram ---------------------------------------------------------------------------------------------------
    for (int t0 = lane; t0 < M; t0 += FAR_UNROLL * 64) {
        int p[FAR_UNROLL];
        double d[FAR_UNROLL];
#pragma unroll
        for (int u = 0; u < FAR_UNROLL; ++u) p[u] = point_of(min(t0 + 64 * u, M - 1));
#pragma unroll
        for (int u = 0; u < FAR_UNROLL; ++u) d[u] = dist2(p[u]);
#pragma unroll
        for (int u = 0; u < FAR_UNROLL; ++u)
            if (t0 + 64 * u < M && d[u] < lim2) atomicAdd(&L.hist[bin_of(d[u])], 1u);
    }
    wave_sync_lds();
    constexpr int BPL = FAR_NB / 64;
    uint32_t bins[BPL], own = 0;
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
        bins[b] = L.hist[lane * BPL + b];
        own += bins[b];
    }
    uint32_t upto = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = __shfl_up(upto, d, 64);
        if (lane >= d) upto += u;
    }
    const uint32_t total = __shfl(upto, 63, 64);
    int b_star = -1;
    uint32_t below = 0;
    {
        uint32_t run = upto - own;
        const bool owner = run < (uint32_t)k && upto >= (uint32_t)k;
#pragma unroll
        for (int b = 0; b < BPL; ++b) {
            run += bins[b];
            if (owner && b_star < 0 && run >= (uint32_t)k) { b_star = lane * BPL + b; below = run; }
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        b_star = max(b_star, __shfl_xor(b_star, d, 64));
        below = max(below, __shfl_xor(below, d, 64));
    }
    if (total < (uint32_t)k) { hopeless = whole_grid; return false; }
    if (below > (uint32_t)COOP_CAP) { hopeless = true; return false; }     // ties in bulk
    // ---- pass B: the candidates below the threshold ---------------------------------------------------------------------------
    for (int t0 = lane; t0 < M; t0 += FAR_UNROLL * 64) {
        int p[FAR_UNROLL];
        double d[FAR_UNROLL];
#pragma unroll
        for (int u = 0; u < FAR_UNROLL; ++u) p[u] = point_of(min(t0 + 64 * u, M - 1));
#pragma unroll
        for (int u = 0; u < FAR_UNROLL; ++u) d[u] = dist2(p[u]);
#pragma unroll
        for (int u = 0; u < FAR_UNROLL; ++u)
            if (t0 + 64 * u < M && d[u] < lim2 && bin_of(d[u]) <= b_star) {
                const uint32_t at = min(atomicAdd(&L.count, 1u), (uint32_t)(COOP_CAP - 1));
                L.list_d[at] = d[u];
                L.list_p[at] = p[u];
            }
    }
    wave_sync_lds();
    const int n_list = (int)below;
    const int e = min(lane, n_list - 1);
    const double e_d = L.list_d[e];
    const int e_p = L.list_p[e];
    int less = 0, same = 0;
    for (int m = 0; m < n_list; ++m) {
        const double dm = L.list_d[m];
        less += dm < e_d ? 1 : 0;
        same += dm == e_d ? 1 : 0;
    }
    if (__ballot(lane < n_list && same > 1 && less < k) != 0ull) { hopeless = true; return false; }   // a tie that matters
    wave_sync_lds();
    const bool best = lane < n_list && less < k;
    const double e_y = best ? y[e_p] : 0.0;
    if (best) {
        L.list_d[less] = e_d;
        L.list_y[less] = e_y;
        L.list_w[less] = 1.0 / sqrt(e_d);
    }
    wave_sync_lds();
    bool zero = false;
    for (int m = lane; m < k; m += 64) zero |= L.list_d[m] == 0.0;
    zero = __ballot(zero) != 0ull;
    auto wgt = [&](int m) { return zero ? (L.list_d[m] == 0.0 ? 1.0 : 0.0) : L.list_w[m]; };
    auto term = [&](int m) { return L.list_y[m] * wgt(m); };
    double num = 0.0, den = 0.0;
    if (k < 8) {
        if (lane == 0)
            for (int m = 0; m < k; ++m) {
                num += term(m);
                den += wgt(m);
            }
    } else {
        double rn = 0.0, rw = 0.0;
        if (lane < 8) {
            rn = term(lane);
            rw = wgt(lane);
            for (int m = 8; m < k - (k % 8); m += 8) {
                rn += term(m + lane);
                rw += wgt(m + lane);
            }
        }
        rn += __shfl_xor(rn, 1, 64); rw += __shfl_xor(rw, 1, 64);
        rn += __shfl_xor(rn, 2, 64); rw += __shfl_xor(rw, 2, 64);
        rn += __shfl_xor(rn, 4, 64); rw += __shfl_xor(rw, 4, 64);
        num = rn;
        den = rw;
        if (lane == 0)
            for (int m = k - (k % 8); m < k; ++m) {
                num += term(m);
                den += wgt(m);
            }
    }
    result = num / den;
    return true;
}

// ---- stage one of the streaming kernel: 2^DIM left-over queries per wavefront, 64 / 2^DIM lanes each ---------------------------
// Most of what the wavefront-per-cell kernel leaves over in a batch of mid-sized cells are ordinary queries in ordinary
// surroundings whose only fault is a cell too large for one shared box: each needs the ~30 - 80 buckets around ITSELF, a
// hundred or two candidates.  A whole wavefront per such query leaves most lanes idle and the kernel bound by instruction
// issue (measured: ~1500 wave instructions per query).  Here every group of lanes owns one query and its own box -- the
// smallest box that holds the ball of `reach` bucket sides around the query, at most NEAR_SIDE buckets per axis --, streams its
// candidates from the index (two passes: histogram, compaction), ranks and predicts exactly like coop_solve.  A group that
// cannot answer (box too large, refined bucket, too few candidates inside the safe radius, ties) reports so; the wavefront then
// searches for its query with all lanes (far_solve).
constexpr int NEAR_SIDE = 5;
constexpr int NEAR_MAX_POINTS = 1024;

template <int DIM>
struct NearLds {
    static constexpr int NQB = 1 << DIM;
    static constexpr int ROWS = DIM == 3 ? NEAR_SIDE * NEAR_SIDE : NEAR_SIDE;
    int32_t row_start[NQB][ROWS + 1], row_prefix[NQB][ROWS + 1];
    uint32_t hist[NQB][COOP_NB + 1];
    double list_d[NQB][COOP_CAP], list_y[NQB][COOP_CAP], list_w[NQB][COOP_CAP];
    int32_t list_p[NQB][COOP_CAP];
    uint32_t count[NQB];
};

template <int DIM>
__device__ __forceinline__ bool near_solve(const Grid<DIM> &g, const double *__restrict__ pts, const int32_t *__restrict__ cs,
                                           const double *__restrict__ y, NearLds<DIM> &L, const double (&q)[DIM], bool active,
                                           int k, double reach, int lane, double &result) {
    constexpr int NQB = 1 << DIM, LPQ = 64 / NQB, ROWS = NearLds<DIM>::ROWS;
    constexpr int MAXC = (ROWS + LPQ - 1) / LPQ;              // rows per lane
    constexpr int BPL = COOP_NB / LPQ, ROUNDS = COOP_CAP / LPQ;
    static_assert(COOP_NB % LPQ == 0 && COOP_CAP % LPQ == 0, "lane layout");
    const int jq = lane / LPQ, gl = lane - jq * LPQ;
    bool fail = !active || k > COOP_CAP;

    // ---- the box and its rows ----------------------------------------------------------------------------------------------
    int lo_i[3] = {0, 0, 0}, hi_i[3] = {0, 0, 0};
    double hmin = DBL_MAX, safe2 = DBL_MAX, far2 = 0.0;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const int b = cell_coord<DIM>(g, q[j], j);
        const double frac = fmin(fmax((q[j] - g.lo[j]) * g.inv_h[j] - (double)b, 0.0), 1.0);
        int r_lo = max((int)ceil(reach - frac), 0), r_hi = max((int)ceil(reach - (1.0 - frac)), 0);
        if (r_lo + r_hi + 1 > NEAR_SIDE) { fail = true; r_lo = r_hi = 0; }
        lo_i[j] = max(b - r_lo, 0);
        hi_i[j] = min(b + r_hi, g.res[j] - 1);
        hmin = fmin(hmin, g.h[j]);
    }
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const double face_lo = g.lo[j] + (double)lo_i[j] * g.h[j], face_hi = g.lo[j] + (double)(hi_i[j] + 1) * g.h[j];
        if (lo_i[j] > 0) {
            const double f = q[j] - face_lo - 1e-9 * hmin;
            safe2 = fmin(safe2, f > 0.0 ? f * f : 0.0);
        }
        if (hi_i[j] < g.res[j] - 1) {
            const double f = face_hi - q[j] - 1e-9 * hmin;
            safe2 = fmin(safe2, f > 0.0 ? f * f : 0.0);
        }
        const double span = fmax(fabs(q[j] - face_lo), fabs(face_hi - q[j]));
        far2 += span * span;
    }
    const int ny = hi_i[1] - lo_i[1] + 1, nz = DIM == 3 ? hi_i[2] - lo_i[2] + 1 : 1, n_rows = fail ? 0 : ny * nz;
    int r_start[MAXC], r_cnt[MAXC];
    bool refined = false;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int r = c * LPQ + gl;
        r_start[c] = r_cnt[c] = 0;
        if (r < n_rows) {
            const int zz = r / ny, yy = r - zz * ny;
            const int64_t row = ((int64_t)(DIM == 3 ? lo_i[2] + zz : 0) * g.res[1] + (lo_i[1] + yy)) * g.res[0];
            r_start[c] = cs[row + lo_i[0]];
            r_cnt[c] = cs[row + hi_i[0] + 1] - r_start[c];
            if (g.sub_res != nullptr)
                for (int x = lo_i[0]; x <= hi_i[0]; ++x) refined |= g.sub_res[row + x] != 0;
        }
    }
    wave_sync_lds();                                          // (the previous query's reads of these arrays are done)
    int M = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        int incl = r_cnt[c];
#pragma unroll
        for (int d = 1; d < LPQ; d <<= 1) {
            const int u = __shfl_up(incl, d, LPQ);
            if (gl >= d) incl += u;
        }
        const int r = c * LPQ + gl;
        if (r < n_rows) {
            L.row_start[jq][r] = r_start[c];
            L.row_prefix[jq][r] = M + incl - r_cnt[c];
        }
        M += __shfl(incl, LPQ - 1, LPQ);
    }
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) refined |= __shfl_xor((int)refined, d, LPQ) != 0;
    const double lim2 = fmin(safe2, far2 * 1.0000001 + 1e-300);
    fail = fail || refined || M < k || M > NEAR_MAX_POINTS || !(lim2 > 0.0);
    if (fail) M = 0;
    if (lane < NQB) L.count[lane] = 0;
    for (int t = lane; t < NQB * (COOP_NB + 1); t += 64) (&L.hist[0][0])[t] = 0;
    wave_sync_lds();
    const double to_bin = (double)COOP_NB / (lim2 > 0.0 ? lim2 : 1.0);
    auto point_of = [&](int t) {                             // slot t of the group's box -> position in the index
        int r = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1)
            if (step < ROWS && r + step < n_rows && L.row_prefix[jq][r + step] <= t) r += step;
        return L.row_start[jq][r] + (t - L.row_prefix[jq][r]);
    };
    auto dist2 = [&](int p) {
        double d = 0.0;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            const double u = q[j] - pts[(int64_t)p * DIM + j];
            d += u * u;
        }
        return d;
    };
    // ---- pass A: histogram of the squared distances inside the safe radius ------------------------------------------------------
    for (int t0 = gl; t0 < M; t0 += 4 * LPQ) {
        int p[4];
        double d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = point_of(min(t0 + LPQ * u, M - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = dist2(p[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + LPQ * u < M && d[u] < lim2) atomicAdd(&L.hist[jq][min(COOP_NB - 1, (int)(d[u] * to_bin))], 1u);
    }
    wave_sync_lds();
    uint32_t bins[BPL], mine = 0;
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
        bins[b] = L.hist[jq][gl * BPL + b];
        mine += bins[b];
    }
    uint32_t upto = mine;
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) {
        const uint32_t u = __shfl_up(upto, d, LPQ);
        if (gl >= d) upto += u;
    }
    const uint32_t total = __shfl(upto, LPQ - 1, LPQ);
    int b_star = -1;
    uint32_t below = 0;
    {
        uint32_t run = upto - mine;
        const bool owner = run < (uint32_t)k && upto >= (uint32_t)k;
#pragma unroll
        for (int b = 0; b < BPL; ++b) {
            run += bins[b];
            if (owner && b_star < 0 && run >= (uint32_t)k) { b_star = gl * BPL + b; below = run; }
        }
    }
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) {
        b_star = max(b_star, __shfl_xor(b_star, d, LPQ));
        below = max(below, __shfl_xor(below, d, LPQ));
    }
    fail = fail || total < (uint32_t)k || below > (uint32_t)COOP_CAP;
    if (fail) { M = 0; below = 0; }
    // ---- pass B: the candidates below the threshold -> the group's short list ---------------------------------------------------
    for (int t0 = gl; t0 < M; t0 += 4 * LPQ) {
        int p[4];
        double d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = point_of(min(t0 + LPQ * u, M - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = dist2(p[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + LPQ * u < M && d[u] < lim2 && min(COOP_NB - 1, (int)(d[u] * to_bin)) <= b_star) {
                const uint32_t at = min(atomicAdd(&L.count[jq], 1u), (uint32_t)(COOP_CAP - 1));
                L.list_d[jq][at] = d[u];
                L.list_p[jq][at] = p[u];
            }
    }
    wave_sync_lds();
    const int n_list = (int)below;
    double e_d[ROUNDS];
    int32_t e_p[ROUNDS], e_less[ROUNDS], e_same[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int e = max(min(gl + r * LPQ, n_list - 1), 0);
        e_d[r] = L.list_d[jq][e];
        e_p[r] = L.list_p[jq][e];
        e_less[r] = e_same[r] = 0;
    }
    for (int m = 0; m < n_list; ++m) {
        const double dm = L.list_d[jq][m];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            e_less[r] += dm < e_d[r] ? 1 : 0;
            e_same[r] += dm == e_d[r] ? 1 : 0;
        }
    }
    bool tie = false;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) tie |= gl + r * LPQ < n_list && e_same[r] > 1 && e_less[r] < k;
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) tie |= __shfl_xor((int)tie, d, LPQ) != 0;
    fail = fail || tie;
    wave_sync_lds();
    const int kk = fail ? 0 : k;                              // (a failed group loads and sums nothing: its lists hold no data)
    double e_y[ROUNDS];
    bool e_best[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        e_best[r] = gl + r * LPQ < n_list && e_less[r] < kk;
        e_y[r] = e_best[r] ? y[e_p[r]] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        if (e_best[r]) {
            L.list_d[jq][e_less[r]] = e_d[r];
            L.list_y[jq][e_less[r]] = e_y[r];
            L.list_w[jq][e_less[r]] = 1.0 / sqrt(e_d[r]);
        }
    wave_sync_lds();
    bool zero = false;
    for (int m = gl; m < kk; m += LPQ) zero |= L.list_d[jq][m] == 0.0;
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) zero |= __shfl_xor((int)zero, d, LPQ) != 0;
    auto wgt = [&](int m) { return zero ? (L.list_d[jq][m] == 0.0 ? 1.0 : 0.0) : L.list_w[jq][m]; };
    auto term = [&](int m) { return L.list_y[jq][m] * wgt(m); };
    double num = 0.0, den = 0.0;
    if (kk < 8) {
        if (gl == 0)
            for (int m = 0; m < kk; ++m) {
                num += term(m);
                den += wgt(m);
            }
    } else {
        double rn = 0.0, rw = 0.0;
        if (gl < 8) {
            rn = term(gl);
            rw = wgt(gl);
            for (int m = 8; m < kk - (kk % 8); m += 8) {
                rn += term(m + gl);
                rw += wgt(m + gl);
            }
        }
        rn += __shfl_xor(rn, 1, LPQ); rw += __shfl_xor(rw, 1, LPQ);
        rn += __shfl_xor(rn, 2, LPQ); rw += __shfl_xor(rw, 2, LPQ);
        rn += __shfl_xor(rn, 4, LPQ); rw += __shfl_xor(rw, 4, LPQ);
        num = rn;
        den = rw;
        if (gl == 0)
            for (int m = kk - (kk % 8); m < kk; ++m) {
                num += term(m);
                den += wgt(m);
            }
    }
    result = num / den;
    return !fail;
}

// entry of the left-over lists -> query point / where its prediction goes
template <int DIM>
__device__ __forceinline__ void leftover_query(int32_t entry, const double *__restrict__ center, const int32_t *__restrict__ level,
                                               int64_t first, double quarter_width, double (&q)[DIM]) {
    constexpr int NCH = 1 << DIM;
    const int64_t cell = first + entry / NCH;
    const double off = cell_offset(quarter_width, level[cell]);
#pragma unroll
    for (int j = 0; j < DIM; ++j) q[j] = center[cell * DIM + j] + dir_comp(DIM, entry % NCH, j) * off;
}

template <int DIM>
__device__ __forceinline__ void leftover_store(int32_t entry, double m, int64_t first, double *__restrict__ metric_all,
                                               double *__restrict__ child_metric) {
    constexpr int NCH = 1 << DIM, NQ = NCH + 1;
    metric_all[(int64_t)(entry / NCH) * NQ + 1 + entry % NCH] = m;
    child_metric[(first + entry / NCH) * NCH + entry % NCH] = m;
}

// stage one: 2^DIM entries of `rest` per wavefront and round; what it cannot answer goes to `next`
template <int DIM>
__global__ void __launch_bounds__(64 * COOP_WAVES)
child_metric_near_kernel(Grid<DIM> g, const double *__restrict__ pts, const int32_t *__restrict__ cs,
                         const double *__restrict__ y, const double *__restrict__ center, const int32_t *__restrict__ level,
                         int64_t first, double quarter_width, int k, double reach, double *__restrict__ metric_all,
                         double *__restrict__ child_metric, const int32_t *__restrict__ rest, int32_t *__restrict__ next) {
    constexpr int NCH = 1 << DIM, LPQ = 64 / NCH;
    __shared__ NearLds<DIM> lds_all[COOP_WAVES];
    const int lane = threadIdx.x & 63, jg = lane / LPQ;
    NearLds<DIM> &L = lds_all[threadIdx.x >> 6];
    const int count = rest[0];
    for (int e0 = (blockIdx.x * COOP_WAVES + (threadIdx.x >> 6)) * NCH; e0 < count; e0 += gridDim.x * COOP_WAVES * NCH) {
        const bool active = e0 + jg < count;
        const int32_t entry = active ? rest[2 + e0 + jg] : 0;
        double q[DIM], m = 0.0;
        leftover_query<DIM>(entry, center, level, first, quarter_width, q);
        const bool ok = near_solve<DIM>(g, pts, cs, y, L, q, active, k, reach, lane, m);
        if (lane % LPQ == 0 && active) {
            if (ok) leftover_store<DIM>(entry, m, first, metric_all, child_metric);
            else next[2 + atomicAdd(&next[0], 1)] = entry;
        }
    }
}

// stage two: one wavefront per entry of `rest`; what it cannot answer goes to `next` (the per-lane kernel's list)
template <int DIM>
__global__ void __launch_bounds__(64 * COOP_WAVES)
child_metric_far_kernel(Grid<DIM> g, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                        const int32_t *__restrict__ cs, const double *__restrict__ y, const double *__restrict__ center,
                        const int32_t *__restrict__ level, int64_t first, double quarter_width, int k, double reach,
                        double *__restrict__ metric_all, double *__restrict__ child_metric, const int32_t *__restrict__ rest,
                        int32_t *__restrict__ next) {
    __shared__ FarLds lds_all[COOP_WAVES];
    const int lane = threadIdx.x & 63;
    FarLds &L = lds_all[threadIdx.x >> 6];
    const int count = rest[0];
    for (int e = blockIdx.x * COOP_WAVES + (threadIdx.x >> 6); e < count; e += gridDim.x * COOP_WAVES) {
        const int32_t entry = rest[2 + e];
        double q[DIM], m = 0.0;
        leftover_query<DIM>(entry, center, level, first, quarter_width, q);
        bool ok = false, hopeless = k > COOP_CAP;
#pragma unroll 1
        for (int attempt = 0; attempt < 4 && !ok && !hopeless; ++attempt)
            ok = far_solve<DIM>(g, pts, orig, cs, y, L, q, k, FAR_BALLS[attempt] * reach, lane, hopeless, m);
        if (lane == 0) {
            if (ok) leftover_store<DIM>(entry, m, first, metric_all, child_metric);
            else next[2 + atomicAdd(&next[0], 1)] = entry;
        }
    }
}

// the queries the wavefronts above could not answer (rest[0] of them): the per-lane search, lanes densely packed
template <int DIM>
__global__ void __launch_bounds__(KNN_BLOCK)
child_metric_rest_kernel(Grid<DIM> g, const double *__restrict__ pts, const int32_t *__restrict__ orig,
                         const int32_t *__restrict__ cs, const double *__restrict__ y, const double *__restrict__ center,
                         const int32_t *__restrict__ level, int64_t first, double quarter_width, int k,
                         double *__restrict__ metric_all, double *__restrict__ child_metric, const int32_t *__restrict__ rest) {
    constexpr int NCH = 1 << DIM, NQ = NCH + 1;
    extern __shared__ double lds[];
    const int count = rest[0];
    for (int e = blockIdx.x * KNN_BLOCK + threadIdx.x; e < count; e += gridDim.x * KNN_BLOCK) {
        const int64_t i = rest[2 + e] / NCH;
        const int jq = rest[2 + e] % NCH;
        const int64_t cell = first + i;
        const double off = cell_offset(quarter_width, level[cell]);
        double q[DIM];
#pragma unroll
        for (int j = 0; j < DIM; ++j) q[j] = center[cell * DIM + j] + dir_comp(DIM, jq, j) * off;
        KBest b = kb_init(lds, k);
        knn_search<DIM>(g, pts, orig, cs, q, b);
        const double m = idw_from_list(b, y);
        metric_all[i * NQ + 1 + jq] = m;
        child_metric[cell * NCH + jq] = m;
    }
}

// summation order of torch's CPU sum over a contiguous inner dimension of n <= 64 doubles (ATen SumKernel.cpp,
// vectorized_inner_sum with 4-wide vectors and 4 interleaved vector accumulators)
template <typename F>
__device__ __forceinline__ double torch_inner_sum(int n, F f) {
    const int nv = n / 4;
    double part[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int l = 0; l < 4; ++l) part[a][l] = 0.0;
    const int size_ilp = nv / 4;
    for (int i = 0; i < size_ilp; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int l = 0; l < 4; ++l) part[a][l] += f((i * 4 + a) * 4 + l);
    for (int i = size_ilp * 4; i < nv; ++i)
#pragma unroll
        for (int l = 0; l < 4; ++l) part[0][l] += f(i * 4 + l);
#pragma unroll
    for (int a = 1; a < 4; ++a)
#pragma unroll
        for (int l = 0; l < 4; ++l) part[0][l] += part[a][l];
    double acc = 0.0;
    for (int i = nv * 4; i < n; ++i) acc += f(i);
#pragma unroll
    for (int l = 0; l < 4; ++l) acc += part[0][l];
    return acc;
}

template <int DIM>
__global__ void child_gain_kernel(const double *__restrict__ metric_all, const int32_t *__restrict__ level,
                                  int64_t first, int64_t n, const double *__restrict__ level_factor, double gain0,
                                  double *__restrict__ metric, double *__restrict__ gain) {
    constexpr int NCH = 1 << DIM, NQ = NCH + 1;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *m = metric_all + i * NQ;
    double m0 = m[0];
    double sd = torch_inner_sum(NCH, [&](int c) { return fabs(m0 - m[1 + c]); });
    int lv = level[first + i];
    metric[first + i] = m0;
    gain[first + i] = level_factor[lv] * sd / gain0;
}

__global__ void idw_weights_kernel(const double *__restrict__ dist, int64_t nc, int k, double *__restrict__ w) {
    int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= nc) return;
    const double *d = dist + c * k;
    auto inv = [&](int m) {
        double v = d[m];
        return 1.0 / (v < 1e-12 ? 1e-12 : v);
    };
    double s = torch_inner_sum(k, inv);
    for (int m = 0; m < k; ++m) w[c * k + m] = inv(m) / s;
}

static size_t knn_lds_bytes(int k) { return (size_t)k * KNN_BLOCK * (sizeof(double) + sizeof(int32_t)); }

static int check_query_args(const s3_knn *knn, const void *q, int64_t nq, int k, const char *who) {
    S3_REQUIRE(knn != nullptr && knn->pts != nullptr, "%s: null index", who);
    S3_REQUIRE(nq >= 0 && (nq == 0 || q != nullptr), "%s: bad query array", who);
    S3_REQUIRE(k >= 1 && k <= S3_MAX_K, "%s: k=%d outside [1,%d]", who, k, S3_MAX_K);
    S3_REQUIRE((int64_t)k <= knn->n, "%s: k=%d exceeds the number of points %lld", who, k, (long long)knn->n);
    return S3_OK;
}

}  // namespace s3

using namespace s3;

extern "C" {

int s3_knn_create(const double *d_pts, int64_t n, int dim, double target_occupancy, s3_stream stream, s3_knn **out) {
    S3_REQUIRE(out != nullptr, "s3_knn_create: null output");
    *out = nullptr;
    S3_REQUIRE(dim == 2 || dim == 3, "s3_knn_create: dim must be 2 or 3, got %d", dim);
    S3_REQUIRE(n >= 1 && n < ((int64_t)1 << 31), "s3_knn_create: n=%lld outside [1, 2^31)", (long long)n);
    S3_REQUIRE(d_pts != nullptr, "s3_knn_create: null points");
    hipStream_t st = as_stream(stream);
    int dev = 0;
    S3_HIP_CHECK(hipGetDevice(&dev));

    // bounding box
    const int nb = 256;
    double *d_partial = nullptr;
    S3_HIP_CHECK(hipMalloc(&d_partial, sizeof(double) * nb * 6));
    bbox_kernel<<<nb, 256, 0, st>>>(d_pts, n, dim, d_partial);
    std::vector<double> part(nb * 6);
    hipError_t e_bbox = hipGetLastError();
    if (e_bbox == hipSuccess) e_bbox = hipMemcpyAsync(part.data(), d_partial, sizeof(double) * nb * 6, hipMemcpyDeviceToHost, st);
    if (e_bbox == hipSuccess) e_bbox = hipStreamSynchronize(st);
    (void)hipFree(d_partial);
    S3_HIP_CHECK(e_bbox);
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int b = 0; b < nb; ++b)
        for (int j = 0; j < dim; ++j) {
            lo[j] = std::fmin(lo[j], part[b * 6 + j]);
            hi[j] = std::fmax(hi[j], part[b * 6 + 3 + j]);
        }
    for (int j = 0; j < dim; ++j)
        S3_REQUIRE(std::isfinite(lo[j]) && std::isfinite(hi[j]), "s3_knn_create: non-finite coordinates");

    s3_knn *k = new s3_knn();
    k->dim = dim;
    k->n = n;
    k->device = dev;
    double occ = target_occupancy > 0 ? target_occupancy : (dim == 2 ? 3.0 : 8.0);
    double ext[3], vol = 1.0;
    int nz = 0;
    for (int j = 0; j < dim; ++j) {
        ext[j] = hi[j] - lo[j];
        if (ext[j] > 0) {
            vol *= ext[j];
            ++nz;
        }
    }
    double target_cells = std::fmax(1.0, (double)n / occ);
    double s = nz > 0 ? std::pow(vol / target_cells, 1.0 / nz) : 1.0;
    const int rmax = dim == 2 ? 8192 : 512;
    int64_t ncell = 1;
    for (int j = 0; j < dim; ++j) {
        int r = 1;
        if (ext[j] > 0 && s > 0) r = (int)std::fmin((double)rmax, std::fmax(1.0, std::ceil(ext[j] / s)));
        k->res[j] = r;
        k->lo[j] = lo[j];
        double e = ext[j] > 0 ? ext[j] * (1.0 + 1e-12) : 1.0;
        k->h[j] = e / r;
        k->inv_h[j] = r / e;
        ncell *= r;
    }
    k->ncell = ncell;

    int32_t *cid = nullptr, *cursor = nullptr, *block_sums = nullptr;
    int64_t nscan = ncell + 1;
    int64_t nblk = (nscan + 1023) / 1024;
    auto fail = [&](int rc) {
        if (cid) (void)hipFree(cid);
        if (cursor) (void)hipFree(cursor);
        if (block_sums) (void)hipFree(block_sums);
        s3_knn_destroy(k);
        return rc;
    };
#define S3_TRY(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            s3::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
            return fail(_e == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP);                              \
        }                                                                                              \
    } while (0)
    S3_TRY(hipMalloc(&k->pts, sizeof(double) * n * dim));
    S3_TRY(hipMalloc(&k->orig, sizeof(int32_t) * n));
    S3_TRY(hipMalloc(&k->cell_start, sizeof(int32_t) * nscan));
    S3_TRY(hipMalloc(&cid, sizeof(int32_t) * n));
    S3_TRY(hipMalloc(&cursor, sizeof(int32_t) * ncell));
    S3_TRY(hipMalloc(&block_sums, sizeof(int32_t) * nblk));
    S3_TRY(hipMemsetAsync(k->cell_start, 0, sizeof(int32_t) * nscan, st));
    S3_TRY(hipMemsetAsync(cursor, 0, sizeof(int32_t) * ncell, st));
    if (dim == 2)
        cell_count_kernel<2><<<grid_for(n, 256), 256, 0, st>>>(make_grid<2>(k), d_pts, n, cid, k->cell_start);
    else
        cell_count_kernel<3><<<grid_for(n, 256), 256, 0, st>>>(make_grid<3>(k), d_pts, n, cid, k->cell_start);
    S3_TRY(hipGetLastError());
    scan_block_kernel<<<(unsigned)nblk, 256, 0, st>>>(k->cell_start, nscan, block_sums);
    scan_sums_kernel<<<1, 256, 0, st>>>(block_sums, nblk);
    scan_add_kernel<<<(unsigned)nblk, 256, 0, st>>>(k->cell_start, nscan, block_sums);
    S3_TRY(hipGetLastError());
    if (dim == 2)
        scatter_kernel<2><<<grid_for(n, 256), 256, 0, st>>>(d_pts, n, cid, k->cell_start, cursor, k->pts, k->orig);
    else
        scatter_kernel<3><<<grid_for(n, 256), 256, 0, st>>>(d_pts, n, cid, k->cell_start, cursor, k->pts, k->orig);
    S3_TRY(hipGetLastError());
    S3_TRY(hipStreamSynchronize(st));
    (void)hipFree(cid);
    (void)hipFree(cursor);
    (void)hipFree(block_sums);
#undef S3_TRY
    // ---- second level: buckets with more than 8x the target occupancy get their own sub-lattice, so that strongly graded
    //      point clouds (boundary-layer meshes) do not degenerate into scanning thousands of points per bucket --------
    {
        const int split = (int)std::ceil(8.0 * occ);
        int32_t *sub_size = nullptr, *bsum = nullptr, *sid = nullptr, *cur2 = nullptr, *orig2 = nullptr;
        double *pts2 = nullptr;
        unsigned long long *d_nref = nullptr;
        auto fail2 = [&](int rc) {
            for (void *q : {(void *)sub_size, (void *)bsum, (void *)sid, (void *)cur2, (void *)orig2, (void *)pts2, (void *)d_nref})
                if (q) (void)hipFree(q);
            s3_knn_destroy(k);
            return rc;
        };
#define S3_TRY2(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            s3::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
            return fail2(_e == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP);                             \
        }                                                                                              \
    } while (0)
        const int64_t nscan2 = ncell + 1, nblk2 = (nscan2 + 1023) / 1024;
        S3_TRY2(hipMalloc(&k->sub_res, ncell));
        S3_TRY2(hipMalloc(&sub_size, sizeof(int32_t) * nscan2));
        S3_TRY2(hipMalloc(&bsum, sizeof(int32_t) * nblk2));
        S3_TRY2(hipMalloc(&d_nref, sizeof(unsigned long long)));
        S3_TRY2(hipMemsetAsync(d_nref, 0, sizeof(unsigned long long), st));
        sub_plan_kernel<<<grid_for(nscan2, 256), 256, 0, st>>>(k->cell_start, ncell, split, occ, dim, k->sub_res, sub_size, d_nref);
        scan_block_kernel<<<(unsigned)nblk2, 256, 0, st>>>(sub_size, nscan2, bsum);
        scan_sums_kernel<<<1, 256, 0, st>>>(bsum, nblk2);
        scan_add_kernel<<<(unsigned)nblk2, 256, 0, st>>>(sub_size, nscan2, bsum);
        S3_TRY2(hipGetLastError());
        unsigned long long nref = 0;
        int32_t pool = 0;
        S3_TRY2(hipMemcpyAsync(&nref, d_nref, sizeof(nref), hipMemcpyDeviceToHost, st));
        S3_TRY2(hipMemcpyAsync(&pool, sub_size + ncell, sizeof(pool), hipMemcpyDeviceToHost, st));
        S3_TRY2(hipStreamSynchronize(st));
        k->n_refined = (int64_t)nref;
        k->sub_off = sub_size;           // the scanned sizes are the table offsets
        sub_size = nullptr;
        if (nref > 0) {
            S3_TRY2(hipMalloc(&k->sub_start, sizeof(int32_t) * (size_t)pool));
            S3_TRY2(hipMalloc(&cur2, sizeof(int32_t) * (size_t)pool));
            S3_TRY2(hipMalloc(&sid, sizeof(int32_t) * n));
            S3_TRY2(hipMalloc(&pts2, sizeof(double) * n * dim));
            S3_TRY2(hipMalloc(&orig2, sizeof(int32_t) * n));
            S3_TRY2(hipMemsetAsync(k->sub_start, 0, sizeof(int32_t) * (size_t)pool, st));
            S3_TRY2(hipMemsetAsync(cur2, 0, sizeof(int32_t) * (size_t)pool, st));
            if (dim == 2) {
                sub_count_kernel<2><<<grid_for(n, 256), 256, 0, st>>>(make_grid<2>(k), k->pts, n, k->sub_res, k->sub_off, k->sub_start, sid);
                sub_scan_kernel<<<(unsigned)ncell, 64, 0, st>>>(k->cell_start, k->sub_res, k->sub_off, k->sub_start, dim);
                sub_scatter_kernel<2><<<grid_for(n, 256), 256, 0, st>>>(make_grid<2>(k), k->pts, k->orig, n, sid, k->sub_off,
                                                                       k->sub_start, cur2, pts2, orig2);
            } else {
                sub_count_kernel<3><<<grid_for(n, 256), 256, 0, st>>>(make_grid<3>(k), k->pts, n, k->sub_res, k->sub_off, k->sub_start, sid);
                sub_scan_kernel<<<(unsigned)ncell, 64, 0, st>>>(k->cell_start, k->sub_res, k->sub_off, k->sub_start, dim);
                sub_scatter_kernel<3><<<grid_for(n, 256), 256, 0, st>>>(make_grid<3>(k), k->pts, k->orig, n, sid, k->sub_off,
                                                                       k->sub_start, cur2, pts2, orig2);
            }
            S3_TRY2(hipGetLastError());
            S3_TRY2(hipStreamSynchronize(st));
            std::swap(k->pts, pts2);
            std::swap(k->orig, orig2);
        }
        for (void *q : {(void *)bsum, (void *)sid, (void *)cur2, (void *)orig2, (void *)pts2, (void *)d_nref})
            if (q) (void)hipFree(q);
#undef S3_TRY2
    }
    *out = k;
    return S3_OK;
}

void s3_knn_destroy(s3_knn *knn) {
    if (!knn) return;
    if (knn->pts) (void)hipFree(knn->pts);
    if (knn->orig) (void)hipFree(knn->orig);
    if (knn->cell_start) (void)hipFree(knn->cell_start);
    if (knn->y) (void)hipFree(knn->y);
    if (knn->sub_res) (void)hipFree(knn->sub_res);
    if (knn->sub_off) (void)hipFree(knn->sub_off);
    if (knn->sub_start) (void)hipFree(knn->sub_start);
    delete knn;
}

int s3_knn_info(const s3_knn *knn, int64_t *h_n_buckets, int64_t *h_n_refined) {
    S3_REQUIRE(knn != nullptr, "s3_knn_info: null index");
    if (h_n_buckets) *h_n_buckets = knn->ncell;
    if (h_n_refined) *h_n_refined = knn->n_refined;
    return S3_OK;
}

int s3_knn_set_values(s3_knn *knn, const double *d_y, s3_stream stream) {
    S3_REQUIRE(knn != nullptr && d_y != nullptr, "s3_knn_set_values: null argument");
    if (!knn->y) S3_HIP_CHECK(hipMalloc(&knn->y, sizeof(double) * knn->n));
    permute_values_kernel<<<grid_for(knn->n, 256), 256, 0, as_stream(stream)>>>(d_y, knn->orig, knn->n, knn->y);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_knn_query(const s3_knn *knn, const double *d_q, int64_t nq, int k, int32_t *d_idx, double *d_dist,
                 s3_stream stream) {
    if (int rc = check_query_args(knn, d_q, nq, k, "s3_knn_query")) return rc;
    S3_REQUIRE(nq == 0 || (d_idx && d_dist), "s3_knn_query: null output");
    if (nq == 0) return S3_OK;
    unsigned grid = grid_for(nq, KNN_BLOCK);
    size_t lds = knn_lds_bytes(k);
    if (knn->dim == 2)
        knn_query_kernel<2><<<grid, KNN_BLOCK, lds, as_stream(stream)>>>(make_grid<2>(knn), knn->pts, knn->orig,
                                                                         knn->cell_start, d_q, nq, k, d_idx, d_dist);
    else
        knn_query_kernel<3><<<grid, KNN_BLOCK, lds, as_stream(stream)>>>(make_grid<3>(knn), knn->pts, knn->orig,
                                                                         knn->cell_start, d_q, nq, k, d_idx, d_dist);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_idw_predict(const s3_knn *knn, const double *d_q, int64_t nq, int k, double *d_yhat, s3_stream stream) {
    if (int rc = check_query_args(knn, d_q, nq, k, "s3_idw_predict")) return rc;
    S3_REQUIRE(knn->y != nullptr, "s3_idw_predict: call s3_knn_set_values first");
    S3_REQUIRE(nq == 0 || d_yhat, "s3_idw_predict: null output");
    if (nq == 0) return S3_OK;
    unsigned grid = grid_for(nq, KNN_BLOCK);
    size_t lds = knn_lds_bytes(k);
    if (knn->dim == 2)
        idw_predict_kernel<2><<<grid, KNN_BLOCK, lds, as_stream(stream)>>>(
            make_grid<2>(knn), knn->pts, knn->orig, knn->cell_start, knn->y, d_q, nq, k, d_yhat);
    else
        idw_predict_kernel<3><<<grid, KNN_BLOCK, lds, as_stream(stream)>>>(
            make_grid<3>(knn), knn->pts, knn->orig, knn->cell_start, knn->y, d_q, nq, k, d_yhat);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

static bool knn_coop_enabled() {
    const char *e = getenv("S3_KNN_COOP");
    return !(e && e[0] == '0');
}

static int64_t knn_coop_min_cells() {
    const char *e = getenv("S3_KNN_COOP_MIN");
    return e ? atoll(e) : 1ll;
}
static bool knn_coop_forced() {
    const char *e = getenv("S3_KNN_COOP");
    return e && e[0] == '1';
}

static double knn_coop_margin() {
    const char *e = getenv("S3_KNN_COOP_MARGIN");
    return e ? atof(e) : 1.15;
}

static int child_gain_impl(const s3_knn *knn, int k, const double *d_center, const int32_t *d_level, int64_t first, int64_t n,
                           int dim, double width, const double *d_level_factor, double gain0, double *d_metric, double *d_gain,
                           double *d_scratch, const int32_t *d_parents, int64_t parents_offset, double *d_child_metric,
                           s3_stream stream, const char *who) {
    if (int rc = check_query_args(knn, d_center, n, k, who)) return rc;
    S3_REQUIRE(knn->y != nullptr, "%s: call s3_knn_set_values first", who);
    S3_REQUIRE(dim == knn->dim, "%s: dim %d does not match the index (%d)", who, dim, knn->dim);
    S3_REQUIRE(first >= 0 && n >= 0 && parents_offset >= 0, "%s: bad range", who);
    S3_REQUIRE(n == 0 || (d_level && d_level_factor && d_metric && d_gain && d_scratch), "%s: null array", who);
    S3_REQUIRE(gain0 != 0.0, "%s: gain0 must be non-zero", who);
    S3_REQUIRE(d_parents == nullptr || d_child_metric != nullptr, "%s: parents given without the child-metric table", who);
    if (n == 0) return S3_OK;
    hipStream_t st = as_stream(stream);
    size_t lds = knn_lds_bytes(k);
    const double qw = 0.25 * width;
#define S3_CHILD_METRIC(DIM, REUSE, PER)                                                                                      \
    child_metric_kernel<DIM, REUSE><<<grid_for(n * (PER), KNN_BLOCK), KNN_BLOCK, lds, st>>>(                                   \
        make_grid<DIM>(knn), knn->pts, knn->orig, knn->cell_start, knn->y, d_center, d_level, first, n, qw, k, d_scratch,     \
        d_parents, parents_offset, d_child_metric)
#define S3_CHILD_COOP(DIM)                                                                                                    \
    do {                                                                                                                      \
    child_metric_coop_kernel<DIM><<<grid_for(n, COOP_WAVES), 64 * COOP_WAVES, 0, st>>>(                                       \
        make_grid<DIM>(knn), knn->pts, knn->orig, knn->cell_start, knn->y, d_center, d_level, first, n, qw, k, reach,        \
        d_scratch, d_parents, parents_offset, d_child_metric, d_rest);                                                        \
    child_metric_near_kernel<DIM><<<grid_for(n, COOP_WAVES, 8192), 64 * COOP_WAVES, 0, st>>>(                                  \
        make_grid<DIM>(knn), knn->pts, knn->cell_start, knn->y, d_center, d_level, first, qw, k, reach, d_scratch,            \
        d_child_metric, d_rest, d_rest2);                                                                                     \
    S3_HIP_CHECK(hipMemsetAsync(d_rest, 0, 2 * sizeof(int32_t), st));                                                         \
    child_metric_far_kernel<DIM><<<grid_for(n * (1 << DIM), COOP_WAVES, 8192), 64 * COOP_WAVES, 0, st>>>(                      \
        make_grid<DIM>(knn), knn->pts, knn->orig, knn->cell_start, knn->y, d_center, d_level, first, qw, k, reach, d_scratch, \
        d_child_metric, d_rest2, d_rest);                                                                                     \
    child_metric_rest_kernel<DIM><<<grid_for(n * (1 << DIM), KNN_BLOCK, 256), KNN_BLOCK, lds, st>>>(                           \
        make_grid<DIM>(knn), knn->pts, knn->orig, knn->cell_start, knn->y, d_center, d_level, first, qw, k, d_scratch,        \
        d_child_metric, d_rest);                                                                                              \
    } while (0)
    // The wavefront kernels (S3_KNN_COOP=0: the per-lane search for every child point, as for cells without a known parent):
    //   coop -- one wavefront per cell, one shared box;  near -- what that leaves (cells too large for one box: the uniform
    //   levels, the first adaptive batches), 2^dim queries per wavefront, each its own small box;  far -- what that leaves (next
    //   to a body, at the edge of the cloud, outside it), one wavefront per query, boxes of growing size;  rest -- per lane.
    // Measured on MI355X (rocprofv3 per launch): cylinder3D, adaptive batches of ~40 000 new cells: 0.37 + 0.08 + 0.21 ms against
    // 1.16 ms for the per-lane kernel alone, the first adaptive batch 0.47 + 0.53 + 1.1 ms, the batches of the uniform levels
    // 0.1 - 0.3 ms instead of 0.4 - 1.4 ms (a per-lane wavefront lives that long however few queries it has); refine
    // 0.058 -> 0.050 s on one box.  box5e7, batches of ~230 000 fine cells: 3.8 - 4.2 + 0.5 - 1.1 + 0.05 ms against 7.0 ms,
    // refine 0.70 -> 0.57 s.
    const bool coop = d_parents != nullptr && knn_coop_enabled() && (n >= knn_coop_min_cells() || knn_coop_forced());
    // how far (in bucket sides) the box reaches beyond the child points: the radius of the ball that holds k points at the
    // index's average occupancy, plus a margin (S3_KNN_COOP_MARGIN, default 1.15); where the cloud is thinner than that the
    // wavefront notices (fewer than k candidates inside its safe radius) and leaves the cell to the per-lane kernel
    const double occupancy = (double)knn->n / (double)knn->ncell;
    const double ball = dim == 3 ? std::cbrt(3.0 * k / (4.0 * 3.14159265358979323846 * occupancy))
                                 : std::sqrt(k / (3.14159265358979323846 * occupancy));
    const double reach = ball * knn_coop_margin();
    // behind the n * (2^dim + 1) doubles of d_scratch: two lists (counter, pad, entries) that the kernels hand on in turns:
    // coop -> rest -> near -> rest2 -> far -> rest (emptied in between) -> per-lane kernel
    int32_t *d_rest = reinterpret_cast<int32_t *>(d_scratch + n * ((1 << dim) + 1));
    int32_t *d_rest2 = d_rest + 2 + n * (1 << dim);
    if (coop) {
        S3_HIP_CHECK(hipMemsetAsync(d_rest, 0, 2 * sizeof(int32_t), st));
        S3_HIP_CHECK(hipMemsetAsync(d_rest2, 0, 2 * sizeof(int32_t), st));
    }
    if (dim == 2) {
        if (coop) S3_CHILD_COOP(2); else if (d_parents) S3_CHILD_METRIC(2, true, 4); else S3_CHILD_METRIC(2, false, 5);
        child_gain_kernel<2><<<grid_for(n, 256), 256, 0, st>>>(d_scratch, d_level, first, n, d_level_factor, gain0,
                                                              d_metric, d_gain);
    } else {
        if (coop) S3_CHILD_COOP(3); else if (d_parents) S3_CHILD_METRIC(3, true, 8); else S3_CHILD_METRIC(3, false, 9);
        child_gain_kernel<3><<<grid_for(n, 256), 256, 0, st>>>(d_scratch, d_level, first, n, d_level_factor, gain0,
                                                              d_metric, d_gain);
    }
#undef S3_CHILD_METRIC
#undef S3_CHILD_COOP
    S3_LAUNCH_CHECK();
    return S3_OK;
}

int s3_child_gain(const s3_knn *knn, int k, const double *d_center, const int32_t *d_level, int64_t first, int64_t n,
                  int dim, double width, const double *d_level_factor, double gain0, double *d_metric, double *d_gain,
                  double *d_scratch, s3_stream stream) {
    return child_gain_impl(knn, k, d_center, d_level, first, n, dim, width, d_level_factor, gain0, d_metric, d_gain, d_scratch,
                           nullptr, 0, nullptr, stream, "s3_child_gain");
}

int s3_child_gain_reuse(const s3_knn *knn, int k, const double *d_center, const int32_t *d_level, int64_t first, int64_t n,
                        int dim, double width, const double *d_level_factor, double gain0, double *d_metric, double *d_gain,
                        double *d_scratch, const int32_t *d_parents, int64_t parents_offset, double *d_child_metric,
                        s3_stream stream) {
    S3_REQUIRE(d_child_metric != nullptr, "s3_child_gain_reuse: null child-metric table");
    return child_gain_impl(knn, k, d_center, d_level, first, n, dim, width, d_level_factor, gain0, d_metric, d_gain, d_scratch,
                           d_parents, parents_offset, d_child_metric, stream, "s3_child_gain_reuse");
}

int s3_idw_weights(const double *d_dist, int64_t nc, int k, double *d_w, s3_stream stream) {
    S3_REQUIRE(nc >= 0 && k >= 1 && k <= S3_MAX_K, "s3_idw_weights: bad shape nc=%lld k=%d", (long long)nc, k);
    S3_REQUIRE(nc == 0 || (d_dist && d_w), "s3_idw_weights: null array");
    if (nc == 0) return S3_OK;
    idw_weights_kernel<<<grid_for(nc, 256), 256, 0, as_stream(stream)>>>(d_dist, nc, k, d_w);
    S3_LAUNCH_CHECK();
    return S3_OK;
}

}  // extern "C"
