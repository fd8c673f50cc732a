// dev probe (standalone): rate of the planned kernel's access pattern for 128-byte vs 64-byte row segments.
// Every workgroup owns ~440 random rows of a [rows][pitch] table and sweeps the columns chunk by chunk (like a tile);
// SEG = 128: 8 lanes x 16 B per row and step, SEG = 64: 4 lanes x 16 B (the second half of a line is asked for one step later).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gather_probe tools/gather_probe.hip && tools/bin/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

template <int SEG>
__global__ void __launch_bounds__(256) sweep(const float4 *__restrict__ data, const int *__restrict__ rows, int rows_per_wg,
                                            long pitch16, int row_bytes, float *__restrict__ sink) {
    constexpr int LPS = SEG / 16;                   // lanes per segment
    constexpr int RPP = 256 / LPS;                  // rows per pass
    const int *my = rows + (long)blockIdx.x * rows_per_wg;
    const int srow = threadIdx.x / LPS, sv = threadIdx.x % LPS;
    float acc = 0.f;
    const int n_chunks = row_bytes / SEG;
    for (int c = 0; c < n_chunks; ++c) {
        for (int pb = 0; pb * 16 * RPP < rows_per_wg; ++pb) {          // (segments longer than 128 B: several blocks of 16 passes)
            float4 v[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int r = (pb * 16 + p) * RPP + srow;
                v[p] = r < rows_per_wg ? data[(long)my[r] * pitch16 + (long)c * LPS + sv] : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) acc += v[p].x + v[p].w;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const long n_rows = 2430607, pitch = 4480, row_bytes = 4096;
    const int rows_per_wg = 440, n_wg = 9500;
    float4 *d; int *r; float *sink;
    hipMalloc(&d, n_rows * pitch); hipMalloc(&r, sizeof(int) * (long)n_wg * rows_per_wg); hipMalloc(&sink, 4);
    hipMemset(d, 0, n_rows * pitch);
    std::vector<int> h((size_t)n_wg * rows_per_wg);
    std::mt19937 g(1);
    // spatially coherent rows: a workgroup's rows come from a window of the table, neighbouring workgroups overlap (like tiles)
    for (int w = 0; w < n_wg; ++w) {
        const long base = (long)((double)w / n_wg * (n_rows - 3000));
        for (int i = 0; i < rows_per_wg; ++i) h[(size_t)w * rows_per_wg + i] = (int)(base + g() % 3000);
    }
    hipMemcpy(r, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int seg : {128, 64, 256, 512}) {
            auto run = [&]() {
                if (seg == 128) sweep<128><<<n_wg, 256>>>(d, r, rows_per_wg, pitch / 16, (int)row_bytes, sink);
                else if (seg == 256) sweep<256><<<n_wg, 256>>>(d, r, rows_per_wg, pitch / 16, (int)row_bytes, sink);
                else if (seg == 512) sweep<512><<<n_wg, 256>>>(d, r, rows_per_wg, pitch / 16, (int)row_bytes, sink);
                else sweep<64><<<n_wg, 256>>>(d, r, rows_per_wg, pitch / 16, (int)row_bytes, sink);
            };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) run();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double bytes = (double)n_wg * rows_per_wg * row_bytes;
            printf("segment %3d B: %.3f ms, %.2f TB/s of staged bytes (%.1f GB per sweep)\n", seg, ms, bytes / ms / 1e9, bytes / 1e9);
        }
    return 0;
}
