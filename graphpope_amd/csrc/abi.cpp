// Error string and version behind the C ABI (include/graphpope_hip.h).
#include <cstring>

#include "common.h"

namespace pope {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

void clear_error() { g_error[0] = '\0'; }

}  // namespace pope

extern "C" const char *pope_last_error(void) { return pope::g_error; }

extern "C" const char *pope_version(void) { return "graphpope_hip 0.2 gfx950"; }

extern "C" int pope_require_device(int32_t *cu_count_host) {
    pope::clear_error();
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        pope::set_error("no gfx950 device visible (%s)", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return POPE_ERR_NO_DEVICE;
    }
    int dev = 0;
    POPE_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    POPE_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pope::set_error("device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
        return POPE_ERR_NO_DEVICE;
    }
    if (cu_count_host) *cu_count_host = prop.multiProcessorCount;
    return POPE_OK;
}

extern "C" int pope_copy_2d_to_host(const void *src, int64_t src_pitch_bytes, void *dst_host, int64_t dst_pitch_bytes,
                                    int64_t row_bytes, int64_t rows, void *stream) {
    pope::clear_error();
    POPE_REQUIRE(src && dst_host, "pope_copy_2d_to_host: null pointer");
    POPE_REQUIRE(row_bytes > 0 && rows > 0 && src_pitch_bytes >= row_bytes && dst_pitch_bytes >= row_bytes, "pope_copy_2d_to_host: bad size");
    POPE_HIP(hipMemcpy2DAsync(dst_host, (size_t)dst_pitch_bytes, src, (size_t)src_pitch_bytes, (size_t)row_bytes, (size_t)rows,
                              hipMemcpyDeviceToHost, (hipStream_t)stream));
    return POPE_OK;
}
