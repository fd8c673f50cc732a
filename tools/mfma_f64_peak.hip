// dev helper (standalone): sustained rate of v_mfma_f64_16x16x4_f64 with operands in registers -- the practical ceiling next to
// the 78.6 TFLOP/s datasheet figure that HISTORY 5.6 prices the Gram kernel against.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_f64_peak tools/mfma_f64_peak.hip && tools/bin/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(double *out, int iters, double seed) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 2e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        a += 1e-9;
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int wgs_per_cu, double *d_out) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * wgs_per_cu, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, 256>>>(d_out, 100, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, 256>>>(d_out, iters, 1.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * NACC * 2.0 * 16 * 16 * 4;
    printf("accumulators per wave %2d, %d workgroups of 256 per CU (%d CUs): %.2f ms -> %.1f TFLOP/s\n", NACC, wgs_per_cu,
           prop.multiProcessorCount, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    double *d_out;
    (void)hipMalloc(&d_out, sizeof(double) * 256 * 4096);
    run<4>(1, d_out); run<8>(1, d_out); run<16>(1, d_out); run<16>(2, d_out); run<8>(2, d_out); run<4>(4, d_out);
    return 0;
}
