// dev probe (standalone): issue cost of the accumulate phase's vector instructions on gfx950 -- v_fma_f64, v_cvt_f64_f32 and the
// mix of the interpolation kernels (8 cvt + 8 fma per neighbour), one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_f64_rate tools/valu_f64_rate.hip && tools/bin/valu_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) probe(const float *in, double *out, int iters, long long *cycles) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x * 8 + i];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double w = in[0] + 1.0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {                       // fma only
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fma(w, acc[(i + 1) & 7] , acc[i]);
            } else if (MODE == 1) {                // cvt only (kept alive through a cheap xor of the bits)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    double d = (double)x[i];
                    x[i] = __int_as_float(__float_as_int(x[i]) ^ (int)__double2loint(d));
                }
            } else {                               // the kernel's mix
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fma(w, (double)x[i], acc[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __int_as_float(__float_as_int(x[i]) + 1);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i] + x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main() {
    float *in; double *out; long long *cyc;
    hipMalloc(&in, 256 * 8 * 4); hipMalloc(&out, 4096 * 256 * 8); hipMalloc(&cyc, 4096 * 8);
    hipMemset(in, 0, 256 * 8 * 4);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int wgs : {256, 512, 1024}) {       // 256-thread workgroups: 1, 2, 4 waves per SIMD
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) probe<0><<<wgs, 256>>>(in, out, iters, cyc);
                if (mode == 1) probe<1><<<wgs, 256>>>(in, out, iters, cyc);
                if (mode == 2) probe<2><<<wgs, 256>>>(in, out, iters, cyc);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
            const double insts = (double)iters * 8 * (mode == 2 ? 24 : (mode == 1 ? 16 : 8));
            printf("mode %d (%s) wgs %4d: %.3f ms, memtime ticks %lld, per wave: %.2f ticks per instruction-slot (%g slots)\n", mode,
                   mode == 0 ? "fma_f64" : mode == 1 ? "cvt_f64_f32 + xor" : "8 cvt + 8 fma + 8 add", wgs, ms, c0, c0 / insts, insts);
        }
    return 0;
}
