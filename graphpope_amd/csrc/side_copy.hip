// The feature copy beside the embedding kernels (side_copy.h).
#include "side_copy.h"

namespace pope {

typedef float f32x4c __attribute__((ext_vector_type(4)));

int g_copy_batches_per_wave = 0;

// 16-byte pieces, PER per lane and row (F <= 256 PER), 16 / PER rows per wave and batch, all 16 loads issued before the
// first store: 357 MB in 60 us alone (5.9 TB/s).  At most ~80 registers a wave, so that two of its waves fit on a SIMD
// next to two of the MFMA tile kernel's (~160 registers each; that kernel owns the CU's whole LDS, this one uses none).
template <int PER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(80))) void k_copy_features(const float *__restrict__ x, unsigned F4,
                                                                                            float *__restrict__ out, unsigned out_cols, int N) {
    constexpr int R = 16 / PER;
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, W = (gridDim.x * blockDim.x) >> 6;
    const unsigned xpitch = F4 * 16u, opitch = out_cols * 4u;
    for (int b = gw; b * R < N; b += W) {
        f32x4c v[R][PER];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int row = b * R + rr;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const unsigned q = lane + 64 * j;
                if (row < N && q < F4)
                    v[rr][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4c *>(reinterpret_cast<const char *>(x) + ((unsigned)row * xpitch + q * 16u)));
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int row = b * R + rr;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const unsigned q = lane + 64 * j;
                if (row < N && q < F4) *reinterpret_cast<f32x4c *>(reinterpret_cast<char *>(out) + ((unsigned)row * opitch + q * 16u)) = v[rr][j];
            }
        }
    }
}

struct SideStream {
    std::mutex mu;
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
constexpr int SIDE_MAX_DEVICES = 64;
static SideStream g_side[SIDE_MAX_DEVICES];
static std::mutex g_side_create;

bool SideCopy::eligible(const float *x, int32_t F, const float *out, int64_t out_cols, int64_t N) {
    int dev = 0;                                               // a device without a side-stream slot takes the serial pope_concat path
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SIDE_MAX_DEVICES) return false;
    return x && out && F > 0 && (F & 3) == 0 && F <= 4096 && (out_cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0 &&
           (uint64_t)N * (uint64_t)out_cols * 4u < (1ull << 32);
}

int SideCopy::fork(hipStream_t main) {
    int dev = 0;
    POPE_HIP(hipGetDevice(&dev));
    POPE_REQUIRE(dev >= 0 && dev < SIDE_MAX_DEVICES, "feature copy: device %d", dev);
    SideStream &s = g_side[dev];
    {
        std::lock_guard<std::mutex> once(g_side_create);
        if (!s.stream) {
            POPE_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
            POPE_HIP(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
            POPE_HIP(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
        }
    }
    hold_ = std::unique_lock<std::mutex>(s.mu);
    side_ = &s;
    POPE_HIP(hipEventRecord(s.fork, main));
    return POPE_OK;
}

// out[r, :F] = x[r, :] for r < N on `stream` (the kernel above; eligible() shapes only).
int enqueue_copy_features(const float *x, int32_t F, float *out, int64_t out_cols, int64_t N, hipStream_t stream) {
    const unsigned F4 = (unsigned)F / 4;
    const int per = (int)((F4 + 63) / 64);
    const int PER = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : per <= 8 ? 8 : 16;
    // Short-lived blocks (one batch of 16 / PER rows per wave by default): a grid of persistent blocks that is running when a
    // kernel that needs a whole CU's LDS is enqueued holds the wave slots and registers that kernel needs to start.
    const int64_t batches = (N + 16 / PER - 1) / (16 / PER), per_wave = g_copy_batches_per_wave > 0 ? g_copy_batches_per_wave : 1;
    // (A few resident grid-stride blocks per CU instead: 2 per CU copy at half the rate, 8 per CU hold the slots as well.)
    const dim3 grid((unsigned)((batches + 4 * per_wave - 1) / (4 * per_wave))), block(256);
    if (PER == 1) hipLaunchKernelGGL(k_copy_features<1>, grid, block, 0, stream, x, F4, out, (unsigned)out_cols, (int)N);
    else if (PER == 2) hipLaunchKernelGGL(k_copy_features<2>, grid, block, 0, stream, x, F4, out, (unsigned)out_cols, (int)N);
    else if (PER == 4) hipLaunchKernelGGL(k_copy_features<4>, grid, block, 0, stream, x, F4, out, (unsigned)out_cols, (int)N);
    else if (PER == 8) hipLaunchKernelGGL(k_copy_features<8>, grid, block, 0, stream, x, F4, out, (unsigned)out_cols, (int)N);
    else hipLaunchKernelGGL(k_copy_features<16>, grid, block, 0, stream, x, F4, out, (unsigned)out_cols, (int)N);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

int SideCopy::launch(const float *x, int32_t F, float *out, int64_t out_cols, int64_t N) {
    POPE_REQUIRE(side_, "feature copy: launch before fork");
    POPE_HIP(hipStreamWaitEvent(side_->stream, side_->fork, 0));
    hipStream_t stream = side_->stream;
    int rc = enqueue_copy_features(x, F, out, out_cols, N, stream);
    if (rc) return rc;
    POPE_HIP(hipEventRecord(side_->join, stream));
    return POPE_OK;
}

int SideCopy::join(hipStream_t main) {
    POPE_REQUIRE(side_, "feature copy: join before fork");
    POPE_HIP(hipStreamWaitEvent(main, side_->join, 0));
    return POPE_OK;
}

}  // namespace pope
