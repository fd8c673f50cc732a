// Symmetric eigenproblem of the T x T Gram matrix behind the weighted SVD (reference utils.py:302-346 takes the
// decomposition from flowtorch.analysis.SVD -> torch.linalg.svd).  The ONE library call of that path: rocSOLVER's dense solver
// (dsyevd: tridiagonalisation + divide and conquer), looked up with dlopen like RCCL in comm.hip -- libs3hip.so does not link it, a
// host without rocSOLVER loses this entry point only.  Around the call, hand-written: the scaling to a unit diagonal maximum (the
// solver works with an absolute tolerance; on the Gram matrix of a deflated residual, entries of 1e-12 and below, the unscaled call
// returned eigenvalues with a relative error of 4e-6 where the scaled one gives 1e-14, tools/svd_accuracy_probe.py) and the scaling
// back.  A plain-C host reaches the whole Gram -> eigenvectors -> modes chain through the C ABI with this (tests/native/c_host.c).
#include <dlfcn.h>

#include <mutex>

#include "common.h"

namespace {

typedef struct _rocblas_handle *rocblas_handle;
typedef int rocblas_int;
typedef int rocblas_status;                 // 0 = success
constexpr int ROCBLAS_EVECT_ORIGINAL = 211; // rocsolver-extra-types.h
constexpr int ROCBLAS_FILL_UPPER = 121;     // rocblas-types.h

struct Solver {
    void *lib = nullptr;
    rocblas_status (*create_handle)(rocblas_handle *) = nullptr;
    rocblas_status (*destroy_handle)(rocblas_handle) = nullptr;
    rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
    rocblas_status (*dsyevd)(rocblas_handle, int, int, rocblas_int, double *, rocblas_int, double *, double *, rocblas_int *) = nullptr;
    rocblas_handle handle = nullptr;
};

std::mutex g_solver_mutex;

Solver *solver() {
    static Solver s;
    static bool tried = false;
    if (tried) return s.lib ? &s : nullptr;
    tried = true;
    // (by soname first: in a process that holds torch the loader hands back the copy torch brought, next to torch's HIP runtime)
    for (const char *name : {"librocsolver.so.0", "/opt/rocm/lib/librocsolver.so.0", "librocsolver.so"}) {
        s.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (s.lib) break;
    }
    if (!s.lib) return nullptr;
    bool ok = true;
    auto sym = [&](const char *n) {
        void *p = dlsym(s.lib, n);                    // (rocblas_* resolve through rocSOLVER's dependency on rocBLAS)
        if (!p) p = dlsym(RTLD_DEFAULT, n);
        ok = ok && p != nullptr;
        return p;
    };
    s.create_handle = reinterpret_cast<decltype(s.create_handle)>(sym("rocblas_create_handle"));
    s.destroy_handle = reinterpret_cast<decltype(s.destroy_handle)>(sym("rocblas_destroy_handle"));
    s.set_stream = reinterpret_cast<decltype(s.set_stream)>(sym("rocblas_set_stream"));
    s.dsyevd = reinterpret_cast<decltype(s.dsyevd)>(sym("rocsolver_dsyevd"));
    if (!ok) {
        s.lib = nullptr;
        return nullptr;
    }
    return &s;
}

// scratch layout: [0] scale, [1] info (as int), [2 .. 2 + t) the solver's off-diagonal workspace E
__global__ void __launch_bounds__(256) diag_max_kernel(const double *__restrict__ g, int64_t t, double *__restrict__ scratch) {
    __shared__ double part[256];
    double m = 0.0;
    for (int64_t i = threadIdx.x; i < t; i += 256) m = fmax(m, g[i * t + i]);
    part[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] = fmax(part[threadIdx.x], part[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scratch[0] = part[0] > 0.0 ? part[0] : 1.0;             // a zero / negative diagonal: solved as it stands
        *reinterpret_cast<int *>(scratch + 1) = 0;
    }
}

__global__ void __launch_bounds__(256) scaled_copy_kernel(const double *__restrict__ g, int64_t n, const double *__restrict__ scratch,
                                                         double *__restrict__ out) {
    const double scale = scratch[0];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = g[i] / scale;
}

__global__ void __launch_bounds__(256) scale_back_kernel(double *__restrict__ lam, int64_t t, const double *__restrict__ scratch) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < t) lam[i] *= scratch[0];
}

}  // namespace

extern "C" {

int s3_sym_eig_available(void) {
    std::lock_guard<std::mutex> guard(g_solver_mutex);
    return solver() != nullptr ? 1 : 0;
}

size_t s3_sym_eig_scratch_bytes(int64_t t) { return sizeof(double) * (size_t)(t + 2); }

int s3_sym_eig(const double *d_g, int64_t t, double *d_lam, double *d_vec, void *d_scratch, s3_stream stream) {
    S3_REQUIRE(t >= 1 && t < ((int64_t)1 << 30), "s3_sym_eig: bad size %lld", (long long)t);
    S3_REQUIRE(d_g && d_lam && d_vec && d_scratch && d_g != d_vec, "s3_sym_eig: null / aliased array");
    std::lock_guard<std::mutex> guard(g_solver_mutex);
    Solver *S = solver();
    S3_REQUIRE(S != nullptr, "s3_sym_eig: rocSOLVER (librocsolver.so.0) cannot be loaded in this process");
    if (!S->handle) S3_REQUIRE(S->create_handle(&S->handle) == 0, "s3_sym_eig: rocblas_create_handle failed");
    hipStream_t st = s3::as_stream(stream);
    S3_REQUIRE(S->set_stream(S->handle, st) == 0, "s3_sym_eig: rocblas_set_stream failed");
    double *scratch = static_cast<double *>(d_scratch);
    diag_max_kernel<<<1, 256, 0, st>>>(d_g, t, scratch);
    scaled_copy_kernel<<<s3::grid_for(t * t, 256, 4096), 256, 0, st>>>(d_g, t * t, scratch, d_vec);
    S3_LAUNCH_CHECK();
    // symmetric: the row-major matrix IS its column-major self; the solver leaves eigenvector j in COLUMN j of the column-major
    // array, i.e. in ROW j of d_vec read row-major
    const rocblas_status rs = S->dsyevd(S->handle, ROCBLAS_EVECT_ORIGINAL, ROCBLAS_FILL_UPPER, (rocblas_int)t, d_vec, (rocblas_int)t, d_lam,
                                        scratch + 2, reinterpret_cast<rocblas_int *>(scratch + 1));
    S3_REQUIRE(rs == 0, "s3_sym_eig: rocsolver_dsyevd returned status %d", rs);
    scale_back_kernel<<<s3::grid_for(t, 256), 256, 0, st>>>(d_lam, t, scratch);
    S3_LAUNCH_CHECK();
    int info = 0;
    S3_HIP_CHECK(hipMemcpyAsync(&info, scratch + 1, sizeof(int), hipMemcpyDeviceToHost, st));
    S3_HIP_CHECK(hipStreamSynchronize(st));
    S3_REQUIRE(info == 0, "s3_sym_eig: the solver did not converge (info = %d)", info);
    return S3_OK;
}

}  // extern "C"
