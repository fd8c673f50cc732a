// Yardsticks for the roofline line of bench.py (measurement aids, not on any product path): what this chip's memory system
// gives a hand-written streaming kernel -- a float4 copy, reads only, and a read / write mix in the headline launch's own ratio
// (7 vectors read per 2 written = 78 % / 22 %) -- so that the launch's bytes per second are compared with kernels of this
// library's own making, not with the runtime's blit (dst.copy_(src) = __amd_rocclr_copyBuffer, 4.6-4.9 TB/s).
// The form of the loop was chosen by measurement (tools/copy_probe.hip, MI355X, 4-GiB buffers, read + written bytes per second):
//   * a grid-stride loop of long-lived workgroups is the SLOWEST form: copy 4.5-4.8 TB/s, mix 4.6-4.8 TB/s;
//   * contiguous blocks per workgroup and iteration, loads of an iteration issued before its stores, nontemporal: copy 5.2-5.6,
//     mix 5.4-5.9 TB/s -- the fewer iterations per workgroup the better;
//   * SHORT-LIVED workgroups that touch their block once and end: copy 6.2-6.45 TB/s (the guide's 6.29, MI355X_MICROARCH.md:36),
//     mix 5.95-6.0 TB/s, reads only 7.0 TB/s, writes only 5.6-5.9 TB/s.
// So every workgroup here handles ONE contiguous block of 256 x U x R vectors and ends.
#include <algorithm>

#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));        // (the nontemporal builtins take native vectors, not HIP's float4 struct)

// every lane: U x R loads (consecutive lanes, consecutive vectors; the loads of a lane 256 vectors apart), then U x W stores
template <int R, int W, int U, bool NT>
__global__ void __launch_bounds__(256) stream_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst) {
    v4f v[U * R];
    const v4f *s0 = src + (int64_t)blockIdx.x * (U * R) * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < U * R; ++i) v[i] = NT ? __builtin_nontemporal_load(s0 + i * 256) : s0[i * 256];
    v4f *d0 = dst + (int64_t)blockIdx.x * (U * W) * 256 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        v4f s = v[u * R];
#pragma unroll
        for (int r = 1; r < R; ++r) s += v[u * R + r];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            v4f o = s;
            o.x += (float)w;
            if constexpr (NT) __builtin_nontemporal_store(o, d0 + (u * W + w) * 256);
            else d0[(u * W + w) * 256] = o;
        }
        if constexpr (W == 0) {
            if (s.x == 123.456f && s.w == 654.321f) dst[0] = s;      // never true: keeps the loads
        }
    }
}

template <int R, int W, int U>
int launch_stream(const void *d_src, void *d_dst, int64_t src_bytes, int64_t dst_bytes, bool nt, hipStream_t st, int64_t *h_read,
                  int64_t *h_written) {
    int64_t blocks = src_bytes / 16 / (256 * U * R);
    if constexpr (W > 0) blocks = std::min<int64_t>(blocks, dst_bytes / 16 / (256 * U * W));
    S3_REQUIRE(blocks >= 1 && blocks < ((int64_t)1 << 31), "s3_yard_stream: buffers too small or too large");
    if (nt) stream_kernel<R, W, U, true><<<dim3((unsigned)blocks), 256, 0, st>>>(static_cast<const v4f *>(d_src), static_cast<v4f *>(d_dst));
    else stream_kernel<R, W, U, false><<<dim3((unsigned)blocks), 256, 0, st>>>(static_cast<const v4f *>(d_src), static_cast<v4f *>(d_dst));
    S3_LAUNCH_CHECK();
    if (h_read) *h_read = blocks * 256 * U * R * 16;
    if (h_written) *h_written = blocks * 256 * U * W * 16;
    return S3_OK;
}

}  // namespace

extern "C" int s3_yard_stream(const void *d_src, void *d_dst, int64_t src_bytes, int64_t dst_bytes, int reads, int writes,
                              int nontemporal, s3_stream stream, int64_t *h_bytes_read, int64_t *h_bytes_written) {
    S3_REQUIRE(d_src && d_dst && src_bytes > 0 && dst_bytes >= 16, "s3_yard_stream: null / empty buffer");
    S3_REQUIRE(reinterpret_cast<uintptr_t>(d_src) % 16 == 0 && reinterpret_cast<uintptr_t>(d_dst) % 16 == 0, "s3_yard_stream: 16-byte alignment");
    hipStream_t st = s3::as_stream(stream);
    const bool nt = nontemporal != 0;
    if (reads == 1 && writes == 1) return launch_stream<1, 1, 1>(d_src, d_dst, src_bytes, dst_bytes, nt, st, h_bytes_read, h_bytes_written);
    if (reads == 7 && writes == 2) return launch_stream<7, 2, 2>(d_src, d_dst, src_bytes, dst_bytes, nt, st, h_bytes_read, h_bytes_written);
    if (reads == 4 && writes == 0) return launch_stream<4, 0, 4>(d_src, d_dst, src_bytes, dst_bytes, nt, st, h_bytes_read, h_bytes_written);
    S3_REQUIRE(false, "s3_yard_stream: (reads, writes) must be (1, 1), (7, 2) or (4, 0)");
    return S3_EINVAL;
}
