// node2vec-space GraphPOPE embedding on MI355X (gfx950): N x K pairwise distance to the anchor rows as an
// exact-f32 MFMA tile, then per-column min-max scaling.
//
// Replaces /root/reference/utils.py:158-176 (sklearn cosine_similarity / cosine_distances / euclidean_distances
// followed by MinMaxScaler().fit/transform).  Two pipelines:
//  (A) depth <= 128 in 16-byte pieces, out below 4 GB -- the node2vec table is [N, 128]:
//   k_pairwise_persistent  (pairwise_persistent.h) one block per CU, the anchor rows resident in LDS, the table streamed
//                     through by LDS-DMA; row norms, metric epilogue, raw values into out[:, F:], per-block column min / max
//   k_copy_features   (side_copy.hip) out[:, :F] = x on a side stream, beside that kernel
//   k_minmax_finish   column min / max over the partial rows -> scale_ = 1/range (range < 10 eps -> 1), min_ = 0 - min*scale_
//   k_minmax_apply    y = e * scale_ + min_ in place (two roundings, like NumPy's X *= scale_; X += min_)
//  (B) everything else -- round 1's pipeline:
//   k_sqnorm          ||x||^2 of every node2vec row and anchor row, accumulated in f64, as {(float)|r|^2, 1/|r|}
//   k_pairwise        dot(X, A^T) with v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate; 64 x 128 tiles on the
//                     machinery of gemm_tile.h: register double buffering, batched LDS fragment reads), metric
//                     epilogue, raw values written straight into the [N, F+K] output, per-block column min/max.
//                     The blocks also carry the feature copy out[:, :F] = x (utils.py:177 concat_into_features): each
//                     streams its own rows' share between its tile product and its epilogue.
//   k_minmax_fold / k_minmax_reduce   column min/max over blocks (two stages), then k_minmax_apply
// Measured and rejected (round 2): computing the tile TWICE (statistics pass, then a pass that scales in its epilogue and
// stores once) instead of raw store + apply pass: with kernel (B) two tile passes took 244 us against 161 us for tile +
// apply; with kernel (A) the second pass would cost more than the 35 us scaling pass it removes.
// Euclidean: sklearn upcasts f32 inputs to f64 (pairwise.py:582-653).  Here the row norms are accumulated in f64,
// d2 = xx + aa - 2 dot is one f32 fma on the f32-accumulated MFMA dot (whose own error, ~1e-6 |x||a|, dominates);
// where d2 is small against the norms (cancellation) the entry is recomputed as a direct sum of squared
// differences, so coincident rows give exactly 0 instead of ~1e-2.  Gate: 1e-5 absolute on the scaled values.
#include "gemm_tile.h"
#include "side_copy.h"

#include <type_traits>

namespace pope {

typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef PW_TILE
#define PW_TILE 0
#endif
#if PW_TILE == 1
constexpr int PM = 128, PN = 128, PWM = 4, PWN = 1;   // A/B: 4 waves stacked, each 32 x 128 (four MFMA tiles)
#else
constexpr int PM = 64, PN = 128, PWM = 2, PWN = 2;    // rows of X x anchor columns per block: 4 waves as 2 x 2, each 32 x 64 (two MFMA tiles)
#endif

extern int g_pairwise_kernel;          // pope_debug_set(POPE_KNOB_PAIRWISE_KERNEL, ...) in geodesic.hip

// Sum of squares of every row in f64 (sklearn row_norms on the upcast chunk), handed to the tile kernels' epilogues the
// way they use it: {(float)|row|^2, 1 / |row|} with 1 for a zero row (sklearn normalize(): zero rows stay zero).
// VEC: rows of 16-byte pieces (D % 4 == 0, aligned base): 16 lanes per row, four rows per wave and step -- the [N, 128]
// table at 4-byte loads, one wave per row, read at 2 TB/s.
// One launch for both matrices: rows [0, rows) of m, then rows [0, rows2) of m2 (same depth).
template <bool VEC>
__global__ __launch_bounds__(256) void k_sqnorm(const float *__restrict__ m, long long rows, float2 *__restrict__ out,
                                                const float *__restrict__ m2, long long rows2, float2 *__restrict__ out2, int D) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long total = rows + rows2;
    if (VEC) {
        const int sub = lane & 15, D4 = D >> 2;
        for (long long r = wave * 4 + (lane >> 4); r < (total + 3) / 4 * 4; r += nwaves * 4) {
            const float *row = r < rows ? m + r * D : m2 + (r - rows) * D;
            double acc = 0.0;
            if (r < total)
                for (int k = sub; k < D4; k += 16) {
                    const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v *>(row) + k);
                    acc += (double)v.x * (double)v.x;
                    acc += (double)v.y * (double)v.y;
                    acc += (double)v.z * (double)v.z;
                    acc += (double)v.w * (double)v.w;
                }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (sub == 0 && r < total) (r < rows ? out[r] : out2[r - rows]) = make_float2((float)acc, acc == 0.0 ? 1.0f : 1.0f / sqrtf((float)acc));
        }
        return;
    }
    for (long long r = wave; r < total; r += nwaves) {
        const float *row = r < rows ? m + r * D : m2 + (r - rows) * D;
        double acc = 0.0;
        for (int k = lane; k < D; k += 64) {
            const double v = (double)row[k];
            acc += v * v;
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) (r < rows ? out[r] : out2[r - rows]) = make_float2((float)acc, acc == 0.0 ? 1.0f : 1.0f / sqrtf((float)acc));
    }
}

static void launch_sqnorm(const float *X, int64_t N, float2 *xn, const float *A, int64_t K, float2 *an, int32_t D, hipStream_t stream) {
    if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(A)) & 15u) == 0)
        hipLaunchKernelGGL(k_sqnorm<true>, dim3(capped_grid((size_t)(N + K) * 16, 256)), dim3(256), 0, stream, X, (long long)N, xn, A, (long long)K, an, D);
    else
        hipLaunchKernelGGL(k_sqnorm<false>, dim3(capped_grid((size_t)(N + K) * 64, 256)), dim3(256), 0, stream, X, (long long)N, xn, A, (long long)K, an, D);
}

// sum_k (x[k] - a[k])^2 by a whole wave: lane l takes k = l, l + 64, ...; fixed shuffle tree, so the value does not
// depend on scheduling.  Every lane returns the total.
__device__ __forceinline__ float wave_sqdist(const float *__restrict__ x, const float *__restrict__ a, int D, int lane) {
    float acc = 0.0f;
    for (int k = lane; k < D; k += 64) {
        const float d = x[k] - a[k];
        acc = fmaf(d, d, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    return acc;
}

}  // namespace pope
#include "pairwise_persistent.h"
namespace pope {

// grid = (ceil(N / PM), ceil(K / PN)).  dot(X, A^T) on the shared MFMA tile machinery (gemm_tile.h), then the metric
// epilogue, the raw values into out[:, c0:], and this block's column min / max.
// The feature copy a pass carries: float4 columns [c_lo, c_hi) of x [N, F4 * 4] -> out[:, 0 : F4 * 4]; x == nullptr: none.
struct PwCopy {
    const float *x;
    int F4, c_lo, c_hi;
};

template <int LAYOUT>
__global__ __launch_bounds__(256) void k_pairwise(const float *__restrict__ X, int N, int D, const float *__restrict__ A,
                                                  int K, int metric, const float2 *__restrict__ xx, const float2 *__restrict__ aa,
                                                  float *__restrict__ out, long long out_cols, int c0,
                                                  float *__restrict__ part_min, float *__restrict__ part_max, int Kpad, PwCopy cp) {
    constexpr int NT = PN / PWN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);
    float *Bs = As + Tile<PM>::FLOATS;
    __shared__ float red_min[PWM][PN], red_max[PWM][PN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % PWM, wn = wave / PWM;
    const int row0 = blockIdx.x * PM, col0 = blockIdx.y * PN;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const Operand Xo{X, D, 1}, Ao{A, D, 1}, none{nullptr, 0, 0};
    // This block's share of the feature copy: its PM rows, the float4 columns split evenly over the tile columns.  The loads
    // are issued BEFORE the tile product (up to CPN 16-byte pieces per thread wait in registers while the MFMAs run) and
    // stored after it: as a load-store loop behind the product the copy added 95 us to the kernel, more than a pass of its own.
    constexpr int CPN = 16;
    f32x4v cpv[CPN];
    int cp_q0 = 0, cp_width = 0, cp_items = 0;
    if (cp.x) {
        const int per = (cp.c_hi - cp.c_lo + (int)gridDim.y - 1) / (int)gridDim.y;
        cp_q0 = cp.c_lo + (int)blockIdx.y * per;
        cp_width = max(0, min(cp.c_hi, cp_q0 + per) - cp_q0);
        cp_items = min(PM, N - row0) * cp_width;
        const f32x4v *src = reinterpret_cast<const f32x4v *>(cp.x) + (size_t)row0 * cp.F4;
#pragma unroll
        for (int j = 0; j < CPN; ++j) {
            const int i = tid + 256 * j;
            if (i < cp_items) {
                const int r = i / cp_width;
                cpv[j] = __builtin_nontemporal_load(src + (size_t)r * cp.F4 + cp_q0 + (i - r * cp_width));
            }
        }
    }
    mfma_accumulate<PM, PN, PWM, PWN, LAYOUT, LAYOUT>(acc, Xo, Ao, 0, D, none, none, 0, 0, row0, col0, N, K, As, Bs);
    if (cp.x) {
#pragma unroll
        for (int j = 0; j < CPN; ++j) {
            const int i = tid + 256 * j;
            if (i < cp_items) {
                const int r = i / cp_width;
                *reinterpret_cast<f32x4v *>(out + (size_t)(row0 + r) * out_cols + 4 * (cp_q0 + (i - r * cp_width))) = cpv[j];
            }
        }
        const f32x4v *src = reinterpret_cast<const f32x4v *>(cp.x) + (size_t)row0 * cp.F4;
        for (int i = tid + 256 * CPN; i < cp_items; i += 256) {       // wider shares than CPN pieces per thread (F > 1000 or K <= 128)
            const int r = i / cp_width, q = cp_q0 + (i - r * cp_width);
            *reinterpret_cast<f32x4v *>(out + (size_t)(row0 + r) * out_cols + 4 * q) = __builtin_nontemporal_load(src + (size_t)r * cp.F4 + q);
        }
    }

    // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // It is instruction-bound (32 outputs per thread), so everything that depends on the row only is hoisted out of the
    // tile loop and the per-element arithmetic is f32: the f32-accumulated MFMA dot carries ~1e-6 of |x||a| already, a
    // f64 sum of the norms on top of it buys nothing (the reference's f64 accuracy is restored where it matters, by the
    // cancellation fix-up below).  80 -> ~25 instructions per output: 48 us -> 15 us of the kernel.
    const float inf = __builtin_huge_valf();
    float x2f[16], rnx[16];
    long long obase[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float2 xn = row < N ? xx[row] : make_float2(1.0f, 1.0f);
        x2f[r] = xn.x;
        rnx[r] = xn.y;
        obase[r] = (long long)row * out_cols + c0;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int cl = wn * (PN / PWN) + t * 32 + (lane & 31);            // column inside the block
        const int col = col0 + cl;
        const bool col_ok = col < K;
        const float2 an = col_ok ? aa[col] : make_float2(0.0f, 1.0f);
        const float a2f = an.x, rna = an.y;
        float cmin = inf, cmax = -inf;
        float e[16];
        if (metric == POPE_METRIC_EUCLIDEAN) {
            unsigned flagged = 0;                                      // bit r: output r of this lane lost its digits
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float norms = x2f[r] + a2f;
                e[r] = fmaf(-2.0f, acc[t][r], norms);                  // squared distance
                if (row < N && col_ok && e[r] < 1e-2f * norms) flagged |= 1u << r;
            }
            // cancellation (d2 small against the norms, e.g. an anchor against its own row): recompute exactly as a sum of
            // squared differences.  Rare, and kept out of the straight-line code above: ONE wave-wide test per tile, then
            // the whole wave does each flagged output together -- 64 lanes over the depth, shuffle reduction -- instead
            // of one lane walking D dependent loads while the other 63 wait (that was a 90 us tail on a 140 us kernel).
            if (__any(flagged != 0)) {
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    unsigned long long fix = __ballot((flagged >> r) & 1u);
                    while (fix) {
                        const int src = __ffsll((long long)fix) - 1;
                        fix &= fix - 1;
                        const float v = wave_sqdist(X + (size_t)__shfl(row, src) * D, A + (size_t)__shfl(col, src) * D, D, lane);
                        if (lane == src) {
#pragma unroll
                            for (int q = 0; q < 16; ++q)
                                if (q == r) e[q] = v;
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_sqrtf(fmaxf(e[r], 0.0f));   // v_sqrt_f32 (1 ulp): the IEEE expansion is ~15 instructions per output
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sim = acc[t][r] * rnx[r] * rna;
                e[r] = metric == POPE_METRIC_COSINE_SIMILARITY ? sim : fminf(fmaxf(1.0f - sim, 0.0f), 2.0f);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < N && col_ok) {
                out[obase[r] + col] = e[r];
                cmin = fminf(cmin, e[r]);
                cmax = fmaxf(cmax, e[r]);
            }
        }
        cmin = fminf(cmin, __shfl_xor(cmin, 32));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        if (lane < 32) {
            red_min[wm][cl] = cmin;
            red_max[wm][cl] = cmax;
        }
    }
    __syncthreads();
    if (tid < PN && col0 + tid < K) {
        float mn = red_min[0][tid], mx = red_max[0][tid];
#pragma unroll
        for (int w = 1; w < PWM; ++w) {
            mn = fminf(mn, red_min[w][tid]);
            mx = fmaxf(mx, red_max[w][tid]);
        }
        part_min[(size_t)blockIdx.x * Kpad + col0 + tid] = mn;
        part_max[(size_t)blockIdx.x * Kpad + col0 + tid] = mx;
    }
}

// Column min / max over the per-block partials, two stages (a single pass over ~1 400 partial rows by one thread per
// column is a 400 us chain of dependent loads): stage 1 folds the partial rows into RSPLIT rows, stage 2 finishes and
// applies sklearn MinMaxScaler.fit (_data.py:456-567) with feature_range (0, 1), all in float32.
constexpr int RSPLIT = 64;

__global__ __launch_bounds__(256) void k_minmax_fold(const float *__restrict__ part_min, const float *__restrict__ part_max,
                                                     int nblocks, int K, int Kpad, float *__restrict__ fold_min,
                                                     float *__restrict__ fold_max) {
    __shared__ float smin[4][64], smax[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int per = (nblocks + RSPLIT - 1) / RSPLIT;
    const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
    float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
    if (j < K)
        for (int b = b0 + wave; b < b1; b += 4) {
            mn = fminf(mn, part_min[(size_t)b * Kpad + j]);
            mx = fmaxf(mx, part_max[(size_t)b * Kpad + j]);
        }
    smin[wave][lane] = mn;
    smax[wave][lane] = mx;
    __syncthreads();
    if (wave == 0 && j < K) {
        fold_min[(size_t)blockIdx.y * Kpad + j] = fminf(fminf(smin[0][lane], smin[1][lane]), fminf(smin[2][lane], smin[3][lane]));
        fold_max[(size_t)blockIdx.y * Kpad + j] = fmaxf(fmaxf(smax[0][lane], smax[1][lane]), fmaxf(smax[2][lane], smax[3][lane]));
    }
}

__global__ __launch_bounds__(256) void k_minmax_reduce(const float *__restrict__ fold_min, const float *__restrict__ fold_max,
                                                       int K, int Kpad, float *__restrict__ scale,
                                                       float *__restrict__ shift) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= K) return;
    float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
#pragma unroll 8
    for (int b = 0; b < RSPLIT; ++b) {
        mn = fminf(mn, fold_min[(size_t)b * Kpad + j]);
        mx = fmaxf(mx, fold_max[(size_t)b * Kpad + j]);
    }
    float range = mx - mn;
    if (range < 10.0f * 1.1920929e-07f) range = 1.0f;             // _handle_zeros_in_scale: < 10 * eps -> 1
    const float sc = 1.0f / range;
    scale[j] = sc;
    shift[j] = 0.0f - __fmul_rn(mn, sc);
}

// Both stages in one launch when there are few partial rows (the persistent kernel leaves one per CU): block = 64 columns,
// its sixteen waves fold rows wave, wave + 16, ..., then wave 0 finishes as k_minmax_reduce does.
__global__ __launch_bounds__(1024) void k_minmax_finish(const float *__restrict__ part_min, const float *__restrict__ part_max, int nrows,
                                                        int K, int Kpad, float *__restrict__ scale, float *__restrict__ shift) {
    __shared__ float smin[16][64], smax[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
    if (j < K) {
#pragma unroll 4
        for (int b = wave; b < nrows; b += 16) {
            mn = fminf(mn, part_min[(size_t)b * Kpad + j]);
            mx = fmaxf(mx, part_max[(size_t)b * Kpad + j]);
        }
    }
    smin[wave][lane] = mn;
    smax[wave][lane] = mx;
    __syncthreads();
    if (wave == 0 && j < K) {
#pragma unroll
        for (int w = 1; w < 16; ++w) {
            mn = fminf(mn, smin[w][lane]);
            mx = fmaxf(mx, smax[w][lane]);
        }
        float range = mx - mn;
        if (range < 10.0f * 1.1920929e-07f) range = 1.0f;
        const float sc = 1.0f / range;
        scale[j] = sc;
        shift[j] = 0.0f - __fmul_rn(mn, sc);
    }
}

// MinMaxScaler.transform: X *= scale_; X += min_  (two separately rounded operations: plain operators under contract(off);
// HIP's __fmul_rn / __fadd_rn wrappers are plain operators too and would let hipcc's default -ffp-contract=fast fuse them).
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_minmax_apply(float *__restrict__ out, int N, int K, long long out_cols, int c0,
                                                      const float *__restrict__ scale, const float *__restrict__ shift) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int v = wave; v < N; v += nwaves) {
        float *row = out + (size_t)v * out_cols + c0;
        for (int j = lane; j < K; j += 64) {
            const float scaled = row[j] * scale[j];
            row[j] = scaled + shift[j];
        }
    }
}

// The same on 16-byte pieces (K, c0, out_cols multiples of 4, aligned bases): a lane owns four columns of a row, a wave one row per
// pass -- the whole 1 KB of a 256-anchor row in ONE load and ONE store instruction, where the scalar form above issues four
// load - wait - store rounds of 256 bytes (round 4: the finalise kernel's lesson).  The arithmetic per element is
// identical: two separately rounded operations.
__global__ __launch_bounds__(256) void k_minmax_apply4(float *__restrict__ out, int N, int K4, long long out_cols, int c0,
                                                       const float *__restrict__ scale, const float *__restrict__ shift) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int q0 = 0; q0 < K4; q0 += 64) {                       // one trip for K <= 256
        const int q = q0 + lane;
        if (q >= K4) continue;
        const float4 sc = reinterpret_cast<const float4 *>(scale)[q], sh = reinterpret_cast<const float4 *>(shift)[q];
        for (int v = wave; v < N; v += nwaves) {
            float4 *p = reinterpret_cast<float4 *>(out + (size_t)v * out_cols + c0) + q;
            float4 e = *p;
            e.x = e.x * sc.x; e.x = e.x + sh.x;
            e.y = e.y * sc.y; e.y = e.y + sh.y;
            e.z = e.z * sc.z; e.z = e.z + sh.z;
            e.w = e.w * sc.w; e.w = e.w + sh.w;
            *p = e;
        }
    }
}

#pragma clang fp contract(fast)

static void launch_minmax_apply(float *out, int64_t N, int32_t K, int64_t out_cols, int32_t c0, const float *scale, const float *shift, hipStream_t stream) {
    const bool vec = (K & 3) == 0 && (c0 & 3) == 0 && (out_cols & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15u) == 0;
    if (vec) {                                                  // one row per wave: short-lived waves in row order (the finalise kernel's grid rule)
        const unsigned grid = (unsigned)std::min<int64_t>(std::max<int64_t>((N + 3) / 4, 256), 32768);
        hipLaunchKernelGGL(k_minmax_apply4, dim3(grid), dim3(256), 0, stream, out, (int)N, K / 4, (long long)out_cols, c0, scale, shift);
    } else {
        hipLaunchKernelGGL(k_minmax_apply, dim3(capped_grid((size_t)N * 64, 256)), dim3(256), 0, stream, out, (int)N, K, (long long)out_cols, c0, scale, shift);
    }
}

static size_t pw_lds_bytes() { return tile_lds_bytes<PM, PN>(); }

struct PwLayout {
    size_t xx, aa, pmin, pmax, fmin, fmax, scale, shift, total;
    int nblocks, part_rows, Kpad;
};

static PwLayout pw_layout(int64_t N, int32_t K) {
    PwLayout L;
    L.nblocks = (int)((N + PM - 1) / PM);
    L.part_rows = std::max<int64_t>(L.nblocks, 2 * std::min<int64_t>((N + PP_ROWS - 1) / PP_ROWS, PP_MAX_GRID));   // either tile kernel's partial rows fit
    L.Kpad = (K + PN - 1) / PN * PN;
    size_t o = 0;
    L.xx = o;    o += align_up((size_t)N * sizeof(float2), 256);
    L.aa = o;    o += align_up((size_t)K * sizeof(float2), 256);
    L.pmin = o;  o += align_up((size_t)L.part_rows * L.Kpad * sizeof(float), 256);
    L.pmax = o;  o += align_up((size_t)L.part_rows * L.Kpad * sizeof(float), 256);
    L.fmin = o;  o += align_up((size_t)RSPLIT * L.Kpad * sizeof(float), 256);
    L.fmax = o;  o += align_up((size_t)RSPLIT * L.Kpad * sizeof(float), 256);
    L.scale = o; o += align_up((size_t)K * sizeof(float), 256);
    L.shift = o; o += align_up((size_t)K * sizeof(float), 256);
    L.total = o;
    return L;
}

}  // namespace pope

using namespace pope;

extern "C" size_t pope_pairwise_scratch_bytes(int64_t N, int32_t K, int32_t D) {
    (void)D;
    if (N <= 0 || K <= 0) return 0;
    return pw_layout(N, K).total;
}

template <int LAYOUT>
static int pairwise_passes(const float *X, int64_t N, int32_t D, const float *A, int32_t K, int32_t metric, const float *x, int32_t F,
                           float *out, int64_t out_cols, int32_t c0, const PwLayout &L, char *base, hipStream_t stream) {
    const float2 *xx = (const float2 *)(base + L.xx), *aa = (const float2 *)(base + L.aa);
    float *pmin = (float *)(base + L.pmin), *pmax = (float *)(base + L.pmax);
    float *fmin = (float *)(base + L.fmin), *fmax = (float *)(base + L.fmax);
    float *scale = (float *)(base + L.scale), *shift = (float *)(base + L.shift);
    static LdsOptIn lds_opt_in;
    if (!lds_opt_in.done()) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_pairwise<LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pw_lds_bytes()));
        lds_opt_in.mark();
    }
    const PwCopy copy{x, F / 4, 0, F / 4};
    hipLaunchKernelGGL(k_pairwise<LAYOUT>, dim3(L.nblocks, L.Kpad / PN), dim3(256), pw_lds_bytes(), stream, X, (int)N, D, A, K, metric, xx, aa,
                       out, (long long)out_cols, c0, pmin, pmax, L.Kpad, copy);
    hipLaunchKernelGGL(k_minmax_fold, dim3((K + 63) / 64, RSPLIT), dim3(256), 0, stream, pmin, pmax, L.nblocks, K, L.Kpad, fmin, fmax);
    hipLaunchKernelGGL(k_minmax_reduce, dim3((K + 255) / 256), dim3(256), 0, stream, fmin, fmax, K, L.Kpad, scale, shift);
    launch_minmax_apply(out, N, K, out_cols, c0, scale, shift, stream);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

// The persistent kernel (pairwise_persistent.h): depths up to 128 in 16-byte pieces.
static int pairwise_persistent(const float *X, int64_t N, int32_t D, const float *A, int32_t K, int32_t metric, const float *x, int32_t F,
                               float *out, int64_t out_cols, int32_t c0, const PwLayout &L, char *base, hipStream_t stream,
                               const long long *anchor_rows = nullptr) {
    static LdsOptIn lds_opt_in;
    if (!lds_opt_in.done()) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_pairwise_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
        lds_opt_in.mark();
    }
    int dev = 0, cus = 0;
    POPE_HIP(hipGetDevice(&dev));
    POPE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const float *zero = nullptr;
    POPE_HIP(hipGetSymbolAddress((void **)&zero, HIP_SYMBOL(g_sk_zero)));
    const int n_tiles = (int)((N + PP_ROWS - 1) / PP_ROWS);
    const int grid = std::min(std::min(cus, PP_MAX_GRID), n_tiles);
    float *pmin = (float *)(base + L.pmin), *pmax = (float *)(base + L.pmax);
    float *scale = (float *)(base + L.scale), *shift = (float *)(base + L.shift);
    // The feature copy: forked onto the side stream here, joined at the end.  Its kernel is enqueued AFTER the tile kernel, so
    // that one's blocks (one per CU, the whole LDS) are resident first and the copy fills the wave slots they leave; started
    // first, the copy's blocks kept a quarter of the CUs busy until it ended and the tile kernel ran 139 us instead of 80.
    SideCopy copy;
    if (x) {
        int rc = copy.fork(stream);
        if (rc) return rc;
    }
    // Two consumer sets keep the matrix cores busy through the epilogues; with the feature copy beside the kernel the call is
    // HBM-bound and the second set's registers are worth more to the copy's waves (configs[2]: 0.193 ms against 0.205).
    const int sets = g_pairwise_kernel == 2 ? 1 : g_pairwise_kernel == 3 ? 2 : (x ? 1 : 2);
    PpArgs a{X, anchor_rows ? X : A, anchor_rows, (int)N, D, K, metric, (float2 *)(base + L.xx), out, (unsigned)out_cols, c0,
             pmin, pmax, L.Kpad, zero, sets};
    hipLaunchKernelGGL(k_pairwise_persistent, dim3(grid, (K + PP_COLS - 1) / PP_COLS), dim3(PP_THREADS), PP_LDS_BYTES, stream, a);
    if (x) {
        int rc = copy.launch(x, F, out, out_cols, N);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_minmax_finish, dim3((K + 63) / 64), dim3(1024), 0, stream, pmin, pmax, sets * grid, K, L.Kpad, scale, shift);
    // (Round 4, VERDICT item 6: the two-pass form -- a statistics-only tile pass that stores nothing, then the tile computed
    //  AGAIN and stored scaled, no scaling pass -- was built and measured: 0.219 ms against 0.171 for the whole call, 0.177
    //  against 0.119 for the embedding alone (profiles/r04_pairwise_two_pass.txt).  A second exact-f32 MFMA pass costs 58 us;
    //  the 182 MB scaling pass it replaces costs 35.  Removed again)
    launch_minmax_apply(out, N, K, out_cols, c0, scale, shift, stream);
    if (x) {
        int rc = copy.join(stream);
        if (rc) return rc;
    }
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

// rows[j, :] = X[ids[j], :] (one wave per anchor): the anchor matrix for the kernel path that cannot follow the ids
__global__ __launch_bounds__(64) void k_gather_anchor_rows(const float *__restrict__ X, const long long *__restrict__ ids, int N, int D, float *__restrict__ rows) {
    const float *src = X + anchor_row(ids, blockIdx.x, N) * D;       // out-of-range ids are clamped (pairwise_persistent.h)
    for (int c = threadIdx.x; c < D; c += 64) rows[(size_t)blockIdx.x * D + c] = src[c];
}

static bool aligned16p(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// anchor_ids == nullptr: A holds the K anchor rows.  Otherwise anchor j is row anchor_ids[j] of X (device int64) and A is unused:
// the persistent kernel reads the rows through the ids, the other path gathers them into the scratch tail first.
static int pairwise_features_impl(const float *x, int32_t F, const float *X, int64_t N, int32_t D, const float *A, const long long *anchor_ids, int32_t K,
                                      int32_t metric, float *out, int64_t out_cols, int32_t c0, void *scratch, size_t scratch_bytes,
                                      void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(X && (A || anchor_ids) && out && scratch, "pope_pairwise_features: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && K > 0 && D > 0 && F >= 0 && c0 >= 0 && out_cols >= (int64_t)c0 + K, "pope_pairwise_features: bad size");
    POPE_REQUIRE(!x || c0 >= F, "pope_pairwise_features: the embedding columns (c0 = %d) overlap the %d feature columns", c0, F);
    POPE_REQUIRE(metric >= 0 && metric <= 2, "pope_pairwise_features: unknown metric %d", metric);
    const PwLayout L = pw_layout(N, K);
    const size_t need = L.total + (anchor_ids ? align_up((size_t)K * D * sizeof(float), 256) : 0);
    if (scratch_bytes < need) {
        set_error("pope_pairwise_features: scratch %zu < %zu bytes", scratch_bytes, need);
        return POPE_ERR_WORKSPACE;
    }
    char *base = (char *)scratch;
    if (x && F > 0 && !((F & 3) == 0 && (out_cols & 3) == 0 && aligned16p(x) && aligned16p(out))) {
        // odd widths / unaligned bases: the copy runs as its own pass
        int rc = pope_concat(x, N, F, out, out_cols, stream_);
        if (rc) return rc;
        x = nullptr;
    }
    if (F == 0) x = nullptr;
    if (g_pairwise_kernel != 1 && D <= PP_DMAX && (D & 3) == 0 && aligned16p(X) && (anchor_ids || aligned16p(A)) && (uint64_t)N * (uint64_t)out_cols * 4u < (1ull << 32)) {
        if (x && !SideCopy::eligible(x, F, out, out_cols, N)) {          // rows wider than the side copy's 16 pieces per lane: the copy as its own pass
            int rc = pope_concat(x, N, F, out, out_cols, stream_);
            if (rc) return rc;
            x = nullptr;
        }
        return pairwise_persistent(X, N, D, A, K, metric, x, F, out, out_cols, c0, L, base, stream, anchor_ids);
    }
    if (anchor_ids) {                                               // the round-1 pipeline wants the anchor rows as a matrix
        float *rows = (float *)(base + L.total);
        hipLaunchKernelGGL(k_gather_anchor_rows, dim3(K), dim3(64), 0, stream, X, anchor_ids, (int)N, D, rows);
        A = rows;
    }
    launch_sqnorm(X, N, (float2 *)(base + L.xx), A, K, (float2 *)(base + L.aa), D, stream);
    const Operand Xo{X, D, 1}, Ao{A, D, 1};
    const bool vec = pick_layout(Xo, (int)N, D) == LAYOUT_KC_VEC && pick_layout(Ao, K, D) == LAYOUT_KC_VEC;
    if (vec) return pairwise_passes<LAYOUT_KC_VEC>(X, N, D, A, K, metric, x, F, out, out_cols, c0, L, base, stream);
    return pairwise_passes<LAYOUT_GENERIC>(X, N, D, A, K, metric, x, F, out, out_cols, c0, L, base, stream);
}

extern "C" int pope_pairwise_features(const float *x, int32_t F, const float *X, int64_t N, int32_t D, const float *A, int32_t K,
                                      int32_t metric, float *out, int64_t out_cols, int32_t c0, void *scratch, size_t scratch_bytes,
                                      void *stream_) {
    return pairwise_features_impl(x, F, X, N, D, A, nullptr, K, metric, out, out_cols, c0, scratch, scratch_bytes, stream_);
}

// utils.py:165-167 as it is written there -- embedding[anchor_nodes] and the distances in one call: anchor j is row
// anchor_ids[j] (device int64, each in [0, N)) of X; no gathered copy of the anchor rows is made on the common path.
extern "C" size_t pope_pairwise_by_id_scratch_bytes(int64_t N, int32_t K, int32_t D) {
    if (N <= 0 || K <= 0 || D <= 0) return 0;
    return pw_layout(N, K).total + align_up((size_t)K * D * sizeof(float), 256);
}

extern "C" int pope_pairwise_features_by_id(const float *x, int32_t F, const float *X, int64_t N, int32_t D, const int64_t *anchor_ids, int32_t K,
                                            int32_t metric, float *out, int64_t out_cols, int32_t c0, void *scratch, size_t scratch_bytes,
                                            void *stream_) {
    clear_error();
    POPE_REQUIRE(anchor_ids, "pope_pairwise_features_by_id: null pointer");
    return pairwise_features_impl(x, F, X, N, D, nullptr, (const long long *)anchor_ids, K, metric, out, out_cols, c0, scratch, scratch_bytes, stream_);
}

extern "C" int pope_pairwise_minmax(const float *X, int64_t N, int32_t D, const float *A, int32_t K, int32_t metric,
                                    float *out, int64_t out_cols, int32_t c0, void *scratch, size_t scratch_bytes,
                                    void *stream_) {
    return pope_pairwise_features(nullptr, 0, X, N, D, A, K, metric, out, out_cols, c0, scratch, scratch_bytes, stream_);
}

#ifdef POPE_STAMP
extern "C" int pope_debug_read_pairwise_stamps(unsigned long long *host, int count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_stamps), (size_t)count * sizeof(unsigned long long));
}
#endif
