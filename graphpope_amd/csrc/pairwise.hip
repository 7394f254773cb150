// node2vec-space GraphPOPE embedding on MI355X (gfx950): N x K pairwise distance to the anchor rows as an
// exact-f32 MFMA tile, then per-column min-max scaling.
//
// Replaces /root/reference/utils.py:158-176 (sklearn cosine_similarity / cosine_distances / euclidean_distances
// followed by MinMaxScaler().fit/transform).  Pipeline:
//   k_sqnorm          ||x||^2 of every node2vec row and anchor row, accumulated in f64
//   k_pairwise        dot(X, A^T) with v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate; 64 x 128 tiles on the
//                     machinery of gemm_tile.h: register double buffering, batched LDS fragment reads), metric
//                     epilogue, raw values written straight into the [N, F+K] output, per-block column min/max
//   k_minmax_fold / k_minmax_reduce   column min/max over blocks (two stages) -> scale_ = 1/range (range < 10 eps -> 1), min_ = 0 - min*scale_
//   k_minmax_apply    y = e * scale_ + min_ in place (two roundings, like NumPy's X *= scale_; X += min_)
// Euclidean: sklearn upcasts f32 inputs to f64 (pairwise.py:582-653).  Here d2 = xx + aa - 2 dot is formed in f64
// from the f32 MFMA dot; where d2 is small against the norms (cancellation) the entry is recomputed as a direct
// sum of squared differences, so coincident rows give exactly 0 instead of ~1e-2.
#include "gemm_tile.h"

namespace pope {

constexpr int PM = 64, PN = 128;     // rows of X x anchor columns per block: 4 waves as 2 x 2, each 32 x 64 (two MFMA tiles)

// One wave per row: sum of squares in f64 (sklearn row_norms on the upcast chunk).
__global__ __launch_bounds__(256) void k_sqnorm(const float *__restrict__ m, long long rows, int D, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave; r < rows; r += nwaves) {
        double acc = 0.0;
        for (int k = lane; k < D; k += 64) {
            const double v = (double)m[r * D + k];
            acc += v * v;
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) out[r] = acc;
    }
}

__device__ __forceinline__ float direct_sqdist(const float *__restrict__ x, const float *__restrict__ a, int D) {
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) {
        const float d = x[k] - a[k];
        acc = fmaf(d, d, acc);
    }
    return acc;
}

// grid = (ceil(N / PM), ceil(K / PN)).  dot(X, A^T) on the shared MFMA tile machinery (gemm_tile.h), then the metric
// epilogue, the raw values into out[:, c0:], and this block's column min / max.
template <int LAYOUT>
__global__ __launch_bounds__(256) void k_pairwise(const float *__restrict__ X, int N, int D, const float *__restrict__ A,
                                                  int K, int metric, const double *__restrict__ xx, const double *__restrict__ aa,
                                                  float *__restrict__ out, long long out_cols, int c0,
                                                  float *__restrict__ part_min, float *__restrict__ part_max, int Kpad) {
    constexpr int NT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);
    float *Bs = As + Tile<PM>::FLOATS;
    __shared__ float red_min[2][PN], red_max[2][PN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % 2, wn = wave / 2;
    const int row0 = blockIdx.x * PM, col0 = blockIdx.y * PN;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const Operand Xo{X, D, 1}, Ao{A, D, 1}, none{nullptr, 0, 0};
    mfma_accumulate<PM, PN, 2, 2, LAYOUT, LAYOUT>(acc, Xo, Ao, 0, D, none, none, 0, 0, row0, col0, N, K, As, Bs);

    // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const float inf = __builtin_huge_valf();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int cl = wn * (PN / 2) + t * 32 + (lane & 31);              // column inside the block
        const int col = col0 + cl;
        const bool col_ok = col < K;
        const double a2 = col_ok ? aa[col] : 0.0;
        const float na = col_ok ? (a2 == 0.0 ? 1.0f : (float)sqrt(a2)) : 1.0f;
        float cmin = inf, cmax = -inf;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < N && col_ok) {
                const float dot = acc[t][r];
                const double x2 = xx[row];
                float e;
                if (metric == POPE_METRIC_EUCLIDEAN) {
                    const double d2 = x2 + a2 - 2.0 * (double)dot;
                    float d2f = (float)d2;
                    if (d2 < 1e-2 * (x2 + a2))                                 // cancellation: recompute exactly
                        d2f = direct_sqdist(X + (size_t)row * D, A + (size_t)col * D, D);
                    e = sqrtf(fmaxf(d2f, 0.0f));
                } else {
                    const float nx = x2 == 0.0 ? 1.0f : (float)sqrt(x2);
                    const float s = dot / (nx * na);
                    e = metric == POPE_METRIC_COSINE_SIMILARITY ? s : fminf(fmaxf(1.0f - s, 0.0f), 2.0f);
                }
                out[(size_t)row * out_cols + c0 + col] = e;
                cmin = fminf(cmin, e);
                cmax = fmaxf(cmax, e);
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
        part_min[(size_t)blockIdx.x * Kpad + col0 + tid] = fminf(red_min[0][tid], red_min[1][tid]);
        part_max[(size_t)blockIdx.x * Kpad + col0 + tid] = fmaxf(red_max[0][tid], red_max[1][tid]);
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

// MinMaxScaler.transform: X *= scale_; X += min_  (two separately rounded operations).
__global__ __launch_bounds__(256) void k_minmax_apply(float *__restrict__ out, int N, int K, long long out_cols, int c0,
                                                      const float *__restrict__ scale, const float *__restrict__ shift) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int v = wave; v < N; v += nwaves) {
        float *row = out + (size_t)v * out_cols + c0;
        for (int j = lane; j < K; j += 64) row[j] = __fadd_rn(__fmul_rn(row[j], scale[j]), shift[j]);
    }
}

static size_t pw_lds_bytes() { return tile_lds_bytes<PM, PN>(); }

struct PwLayout {
    size_t xx, aa, pmin, pmax, fmin, fmax, scale, shift, total;
    int nblocks, Kpad;
};

static PwLayout pw_layout(int64_t N, int32_t K) {
    PwLayout L;
    L.nblocks = (int)((N + PM - 1) / PM);
    L.Kpad = (K + PN - 1) / PN * PN;
    size_t o = 0;
    L.xx = o;    o += align_up((size_t)N * sizeof(double), 256);
    L.aa = o;    o += align_up((size_t)K * sizeof(double), 256);
    L.pmin = o;  o += align_up((size_t)L.nblocks * L.Kpad * sizeof(float), 256);
    L.pmax = o;  o += align_up((size_t)L.nblocks * L.Kpad * sizeof(float), 256);
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

extern "C" int pope_pairwise_minmax(const float *X, int64_t N, int32_t D, const float *A, int32_t K, int32_t metric,
                                    float *out, int64_t out_cols, int32_t c0, void *scratch, size_t scratch_bytes,
                                    void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(X && A && out && scratch, "pope_pairwise_minmax: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && K > 0 && D > 0 && c0 >= 0 && out_cols >= (int64_t)c0 + K, "pope_pairwise_minmax: bad size");
    POPE_REQUIRE(metric >= 0 && metric <= 2, "pope_pairwise_minmax: unknown metric %d", metric);
    const PwLayout L = pw_layout(N, K);
    if (scratch_bytes < L.total) {
        set_error("pope_pairwise_minmax: scratch %zu < %zu bytes", scratch_bytes, L.total);
        return POPE_ERR_WORKSPACE;
    }
    char *base = (char *)scratch;
    double *xx = (double *)(base + L.xx), *aa = (double *)(base + L.aa);
    float *pmin = (float *)(base + L.pmin), *pmax = (float *)(base + L.pmax);
    float *fmin = (float *)(base + L.fmin), *fmax = (float *)(base + L.fmax);
    float *scale = (float *)(base + L.scale), *shift = (float *)(base + L.shift);

    hipLaunchKernelGGL(k_sqnorm, dim3(capped_grid((size_t)N * 64, 256)), dim3(256), 0, stream, X, (long long)N, D, xx);
    hipLaunchKernelGGL(k_sqnorm, dim3(capped_grid((size_t)K * 64, 256)), dim3(256), 0, stream, A, (long long)K, D, aa);
    const Operand Xo{X, D, 1}, Ao{A, D, 1};
    const bool vec = pick_layout(Xo, (int)N, D) == LAYOUT_KC_VEC && pick_layout(Ao, K, D) == LAYOUT_KC_VEC;
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_pairwise<LAYOUT_KC_VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pw_lds_bytes()));
        POPE_HIP(hipFuncSetAttribute((const void *)k_pairwise<LAYOUT_GENERIC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pw_lds_bytes()));
        lds_opt_in = true;
    }
    if (vec)
        hipLaunchKernelGGL(k_pairwise<LAYOUT_KC_VEC>, dim3(L.nblocks, L.Kpad / PN), dim3(256), pw_lds_bytes(), stream, X, (int)N, D, A, K, metric,
                           xx, aa, out, (long long)out_cols, c0, pmin, pmax, L.Kpad);
    else
        hipLaunchKernelGGL(k_pairwise<LAYOUT_GENERIC>, dim3(L.nblocks, L.Kpad / PN), dim3(256), pw_lds_bytes(), stream, X, (int)N, D, A, K, metric,
                           xx, aa, out, (long long)out_cols, c0, pmin, pmax, L.Kpad);
    hipLaunchKernelGGL(k_minmax_fold, dim3((K + 63) / 64, RSPLIT), dim3(256), 0, stream, pmin, pmax, L.nblocks, K, L.Kpad, fmin, fmax);
    hipLaunchKernelGGL(k_minmax_reduce, dim3((K + 255) / 256), dim3(256), 0, stream, fmin, fmax, K, L.Kpad, scale, shift);
    hipLaunchKernelGGL(k_minmax_apply, dim3(capped_grid((size_t)N * 64, 256)), dim3(256), 0, stream, out, (int)N, K,
                       (long long)out_cols, c0, scale, shift);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}
