// Runtime plumbing of libs3hip.so: error string, device selection, raw memory helpers for hosts without torch.
#include "common.h"

#include <cstring>

namespace s3 {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace s3

extern "C" {

const char *s3_last_error(void) { return s3::g_err; }

int s3_abi_version(void) { return 1; }

int s3_device_count(int *h_count) {
    S3_REQUIRE(h_count != nullptr, "s3_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *h_count = 0;
        s3::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return S3_ENODEV;
    }
    *h_count = n;
    return S3_OK;
}

int s3_set_device(int device) {
    S3_HIP_CHECK(hipSetDevice(device));
    return S3_OK;
}

int s3_malloc(void **d_ptr, size_t bytes) {
    S3_REQUIRE(d_ptr != nullptr, "s3_malloc: null output");
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        s3::set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? S3_ENOMEM : S3_EHIP;
    }
    return S3_OK;
}

int s3_free(void *d_ptr) {
    if (d_ptr) S3_HIP_CHECK(hipFree(d_ptr));
    return S3_OK;
}

int s3_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, s3_stream stream) {
    S3_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s3::as_stream(stream)));
    return S3_OK;
}

int s3_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, s3_stream stream) {
    S3_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s3::as_stream(stream)));
    return S3_OK;
}

int s3_stream_synchronize(s3_stream stream) {
    S3_HIP_CHECK(hipStreamSynchronize(s3::as_stream(stream)));
    return S3_OK;
}

}  // extern "C"
