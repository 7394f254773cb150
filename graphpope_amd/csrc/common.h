// Shared host-side helpers for libgraphpope_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "graphpope_hip.h"

namespace pope {

typedef unsigned long long u64;

// Per-thread error string behind pope_last_error().
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();

inline int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return POPE_ERR_HIP;
}

#define POPE_HIP(call)                                                         \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) return ::pope::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

#define POPE_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::pope::set_error(__VA_ARGS__);     \
            return POPE_ERR_INVALID;            \
        }                                       \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// "The LDS opt-in of these kernels has been applied on the current device": one bit per device, set AFTER the attribute
// calls have returned, so neither a second device nor a second host thread can launch a >64 KB-LDS kernel without it
// (two threads may both apply it: hipFuncSetAttribute is idempotent).  Usage:
//     static LdsOptIn once;  if (!once.done()) { POPE_HIP(hipFuncSetAttribute(...)); ...; once.mark(); }
struct LdsOptIn {
    std::atomic<unsigned long long> mask[2] = {{0}, {0}};          // devices 0 .. 127
    static int device() {
        int dev = 0;
        return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 128 ? dev : -1;
    }
    bool done() const {
        const int d = device();
        return d >= 0 && ((mask[d >> 6].load(std::memory_order_acquire) >> (d & 63)) & 1ull);
    }
    void mark() {
        const int d = device();
        if (d >= 0) mask[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release);
    }
};

// Words of 64 anchors per node, padded so a node's words form whole tiles of 1, 2 or 4 words.
inline int words_for(int K) {
    int w = (K + 63) / 64;
    if (w <= 2) return w < 1 ? 1 : w;
    return (w + 3) / 4 * 4;
}

// Device extents (include/graphpope_hip.h, "dims"): a size that is only known on the device.  `cap` is the host-side capacity
// the launch was sized for; a null pointer means the capacity IS the size.
__device__ __forceinline__ int dyn_extent(const int *dev, int cap) {
    if (!dev) return cap;
    const int v = *dev;
    return v < cap ? (v < 0 ? 0 : v) : cap;
}

// Memory-bound grids: enough blocks to fill 256 CUs several times over, grid-stride the rest.
inline unsigned capped_grid(size_t work_items, unsigned block, unsigned cap = 256u * 16u) {
    size_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace pope
