// Host-only entry points of libiak.so: version, error strings, and the memory / stream
// helpers that stand in for cudaMalloc / cudaMemcpy / cudaFree in a host program.
#include "iak_common.hpp"

#include <atomic>

using namespace iak;

namespace iak {
std::atomic<int> g_contraction{0};
}

extern "C" {

int iak_version(void) { return 100; }

int iak_set_contraction(int enabled) {
    iak::g_contraction.store(enabled ? 1 : 0);
    return 0;
}
int iak_get_contraction(void) { return iak::g_contraction.load(); }

const char *iak_error_string(int code) {
    switch (code) {
    case 0: return "success";
    case IAK_E_ARG: return "iak: invalid argument";
    case IAK_E_UNSUPPORTED: return "iak: unsupported configuration";
    case IAK_E_TOO_LARGE: return "iak: size exceeds a fixed capacity";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "iak: unknown error";
}

int iak_malloc(void **dev_ptr, size_t bytes) {
    if (!dev_ptr) return IAK_E_ARG;
    IAK_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return 0;
}

int iak_free(void *dev_ptr) {
    IAK_HIP(hipFree(dev_ptr));
    return 0;
}

int iak_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream) {
    if (!dst_dev || !src_host) return IAK_E_ARG;
    IAK_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    IAK_HIP(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

int iak_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream) {
    if (!dst_host || !src_dev) return IAK_E_ARG;
    IAK_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    IAK_HIP(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

int iak_memset(void *dst_dev, int value, size_t bytes, void *stream) {
    if (!dst_dev) return IAK_E_ARG;
    IAK_HIP(hipMemsetAsync(dst_dev, value, bytes, as_stream(stream)));
    return 0;
}

int iak_stream_sync(void *stream) {
    IAK_HIP(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

int iak_device_count(int *count) {
    if (!count) return IAK_E_ARG;
    IAK_HIP(hipGetDeviceCount(count));
    return 0;
}

int iak_set_device(int device) {
    IAK_HIP(hipSetDevice(device));
    return 0;
}

}  // extern "C"
