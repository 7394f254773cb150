// Host-side half of the drop-in boundary (no device code): assembling the reference's [N, F+K] result tensor on the
// host cores while the GPU computes / ships only the K embedding columns.
//
// Replaces /root/reference/utils.py:129-135 concat_into_features -> torch.cat((data.x, embedding), 1) for the
// host -> host call: the features never cross PCIe (they do not change); they are copied once, host to host, by a
// few threads with streaming stores, underneath the GPU work.
#include <emmintrin.h>

#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "graphpope_hip.h"

namespace {

// One row segment of `bytes` bytes; dst 16-byte aligned segments go out as non-temporal stores (the destination is
// written once and not read back by these threads: no read-for-ownership traffic, the caches keep the sources).
inline void copy_segment(const char *src, char *dst, size_t bytes) {
    size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
    if (head > bytes) head = bytes;
    if (head) {
        memcpy(dst, src, head);
        src += head; dst += head; bytes -= head;
    }
    const size_t vec = bytes / 64 * 64;
    for (size_t o = 0; o < vec; o += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + o));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + o + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + o + 32));
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + o + 48));
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + o), a);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + o + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + o + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + o + 48), d);
    }
    if (bytes > vec) memcpy(dst + vec, src + vec, bytes - vec);
}

void copy_rows(const char *src, size_t src_pitch, char *dst, size_t dst_pitch, size_t row_bytes, int64_t r0, int64_t r1) {
    if (src_pitch == row_bytes && dst_pitch == row_bytes) {             // contiguous on both sides: one long segment
        copy_segment(src + (size_t)r0 * row_bytes, dst + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes);
    } else {
        for (int64_t r = r0; r < r1; ++r) copy_segment(src + (size_t)r * src_pitch, dst + (size_t)r * dst_pitch, row_bytes);
    }
    _mm_sfence();
}

}  // namespace

extern "C" int pope_host_copy_2d(const void *src_host, int64_t src_pitch_bytes, void *dst_host, int64_t dst_pitch_bytes,
                                 int64_t row_bytes, int64_t rows, int32_t threads) {
    if (!src_host || !dst_host || row_bytes < 0 || rows < 0 || src_pitch_bytes < row_bytes || dst_pitch_bytes < row_bytes)
        return POPE_ERR_INVALID;
    if (rows == 0 || row_bytes == 0) return POPE_OK;
    const size_t total = (size_t)rows * (size_t)row_bytes;
    int t = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (t < 1) t = 1;
    if (t > 64) t = 64;
    if (total < ((size_t)1 << 20)) t = 1;                               // not worth a thread below 1 MiB
    if ((int64_t)t > rows) t = (int)rows;
    const char *s = static_cast<const char *>(src_host);
    char *d = static_cast<char *>(dst_host);
    if (t == 1) {
        copy_rows(s, (size_t)src_pitch_bytes, d, (size_t)dst_pitch_bytes, (size_t)row_bytes, 0, rows);
        return POPE_OK;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)t - 1);
    const int64_t per = (rows + t - 1) / t;
    for (int i = 1; i < t; ++i) {
        const int64_t r0 = i * per, r1 = r0 + per < rows ? r0 + per : rows;
        if (r0 >= rows) break;
        pool.emplace_back(copy_rows, s, (size_t)src_pitch_bytes, d, (size_t)dst_pitch_bytes, (size_t)row_bytes, r0, r1);
    }
    copy_rows(s, (size_t)src_pitch_bytes, d, (size_t)dst_pitch_bytes, (size_t)row_bytes, 0, per < rows ? per : rows);
    for (auto &th : pool) th.join();
    return POPE_OK;
}
