// Shared host/device helpers of libs3hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "s3hip.h"

namespace s3 {

void set_error(const char *fmt, ...);

#define S3_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            s3::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorNoDevice || _e == hipErrorInvalidDevice ? S3_ENODEV : S3_EHIP;  \
        }                                                                                        \
    } while (0)

#define S3_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            s3::set_error(__VA_ARGS__);    \
            return S3_EINVAL;              \
        }                                  \
    } while (0)

#define S3_LAUNCH_CHECK() S3_HIP_CHECK(hipGetLastError())

inline hipStream_t as_stream(s3_stream s) { return reinterpret_cast<hipStream_t>(s); }

inline unsigned grid_for(int64_t n, int block, int64_t cap = (int64_t)1 << 30) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// child / node direction table of the reference (s_cube.py:188-194); component j of direction c
__device__ __forceinline__ double dir_comp(int /*dim*/, int c, int j) {
    // 2-D: (-1,-1) (-1,1) (1,1) (1,-1); 3-D: the same four with z=+1, then with z=-1
    if (j == 0) return (c & 3) >= 2 ? 1.0 : -1.0;
    if (j == 1) return ((c & 3) == 1 || (c & 3) == 2) ? 1.0 : -1.0;
    return c < 4 ? 1.0 : -1.0;
}

// (factor*width)/2^level exactly as torch evaluates it at s_cube.py:441 (all operations are exact scalings)
__device__ __forceinline__ double cell_offset(double factor_width, int level) {
    return factor_width / (double)(1ull << level);
}

}  // namespace s3
