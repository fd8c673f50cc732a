// dev probe (standalone): what a hand-written streaming kernel reaches on this chip -- copy, read-only, write-only and a 7:2 read / write
// mix -- for several shapes of the loop (vectors per lane and iteration, workgroups, nontemporal or not).  Decides the form of
// csrc/yardstick.hip.     hipcc --offload-arch=gfx950 -O3 -o tools/bin/copy_probe tools/copy_probe.hip && tools/bin/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));

// every lane: U x R loads first, then U x W stores; consecutive lanes consecutive vectors; one iteration of a workgroup = a contiguous
// block of 256 * U * R vectors (TILE = true) or vectors G apart (TILE = false)
template <int R, int W, int U, bool NT, bool TILE>
__global__ void __launch_bounds__(256) k(const v4f *__restrict__ src, v4f *__restrict__ dst, int64_t n_iter) {
    const int64_t G = (int64_t)gridDim.x * 256, gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t it = 0; it < n_iter; ++it) {
        v4f v[U * R > 0 ? U * R : 1];
#pragma unroll
        for (int i = 0; i < U * R; ++i) {
            const int64_t a = TILE ? ((it * gridDim.x + blockIdx.x) * (U * R) + i) * 256 + threadIdx.x : (it * (U * R) + i) * G + gid;
            v[i] = NT ? __builtin_nontemporal_load(src + a) : src[a];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v4f s = R > 0 ? v[u * R] : (v4f){1.f, 2.f, 3.f, 4.f};
#pragma unroll
            for (int r = 1; r < R; ++r) s += v[u * R + r];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const int i = u * W + w;
                const int64_t a = TILE ? ((it * gridDim.x + blockIdx.x) * (U * W) + i) * 256 + threadIdx.x : (it * (U * W) + i) * G + gid;
                if (NT) __builtin_nontemporal_store(s, dst + a); else dst[a] = s;
            }
            if (W == 0 && s.x == 123.456f && s.w == 6.5f) dst[0] = s;
        }
    }
}

template <int R, int W, int U, bool NT, bool TILE>
void run(const char *name, const v4f *src, v4f *dst, int64_t n_vec, int blocks) {
    const int64_t per_iter = (int64_t)blocks * 256 * U;
    int64_t n_iter = n_vec / (per_iter * (R > W ? R : W));
    if (n_iter < 1) { printf("%-40s skipped\n", name); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<R, W, U, NT, TILE><<<blocks, 256>>>(src, dst, n_iter); hipDeviceSynchronize();
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k<R, W, U, NT, TILE><<<blocks, 256>>>(src, dst, n_iter);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); sum += ms; if (ms < best) best = ms;
    }
    const double bytes = (double)n_iter * per_iter * (R + W) * 16;
    printf("%-44s R%d W%d U%d %s %s %6d wg x %5lld it: mean %.3f ms  %.3f TB/s (best %.3f)\n", name, R, W, U, NT ? "nt" : "  ", TILE ? "tile" : "grid",
           blocks, (long long)n_iter, sum / 5, bytes / (sum / 5) / 1e9, bytes / best / 1e9);
}

int main() {
    const int64_t bytes = (int64_t)4 << 30, n_vec = bytes / 16;
    v4f *src, *dst; hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
    for (int blocks : {2048, 4096, 16384, 65536}) {
        run<1, 1, 1, false, false>("copy", src, dst, n_vec, blocks);
        run<1, 1, 4, false, false>("copy", src, dst, n_vec, blocks);
        run<1, 1, 4, true, false>("copy", src, dst, n_vec, blocks);
        run<1, 1, 4, false, true>("copy", src, dst, n_vec, blocks);
        run<1, 1, 4, true, true>("copy", src, dst, n_vec, blocks);
        run<1, 1, 8, true, true>("copy", src, dst, n_vec, blocks);
        run<1, 1, 16, true, true>("copy", src, dst, n_vec, blocks);
        run<7, 2, 1, false, false>("mix", src, dst, n_vec, blocks);
        run<7, 2, 1, true, true>("mix", src, dst, n_vec, blocks);
        run<7, 2, 2, true, true>("mix", src, dst, n_vec, blocks);
        run<7, 2, 4, true, true>("mix", src, dst, n_vec, blocks);
        run<4, 0, 1, true, false>("read", src, dst, n_vec, blocks);
        run<4, 0, 4, true, true>("read", src, dst, n_vec, blocks);
        run<0, 1, 4, true, true>("write", src, dst, n_vec, blocks);
        run<0, 1, 4, false, false>("write", src, dst, n_vec, blocks);
    }
    // one vector per thread, no loop
    run<1, 1, 1, false, false>("copy, one vector per thread", src, dst, n_vec, (int)(n_vec / 256));
    run<1, 1, 1, true, false>("copy, one vector per thread", src, dst, n_vec, (int)(n_vec / 256));
    run<1, 1, 4, true, true>("copy, four vectors per thread", src, dst, n_vec, (int)(n_vec / 1024));
    return 0;
}
