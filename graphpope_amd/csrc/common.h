// Shared host-side helpers for libgraphpope_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

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

// Words of 64 anchors per node, padded so a node's words form whole tiles of 1, 2 or 4 words.
inline int words_for(int K) {
    int w = (K + 63) / 64;
    if (w <= 2) return w < 1 ? 1 : w;
    return (w + 3) / 4 * 4;
}

// Memory-bound grids: enough blocks to fill 256 CUs several times over, grid-stride the rest.
inline unsigned capped_grid(size_t work_items, unsigned block, unsigned cap = 256u * 16u) {
    size_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace pope
