// node2vec-space GraphPOPE embedding on MI355X (gfx950): N x K pairwise distance to the anchor rows as an
// exact-f32 MFMA tile, then per-column min-max scaling.
//
// Replaces /root/reference/utils.py:158-176 (sklearn cosine_similarity / cosine_distances / euclidean_distances
// followed by MinMaxScaler().fit/transform).  Pipeline:
//   k_sqnorm          ||x||^2 of every node2vec row and anchor row, accumulated in f64
//   k_pairwise        dot(X, A^T) with v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate), metric
//                     epilogue, raw values written straight into the [N, F+K] output, per-block column min/max
//   k_minmax_reduce   column min/max over blocks -> scale_ = 1/range (range < 10 eps -> 1), min_ = 0 - min*scale_
//   k_minmax_apply    y = e * scale_ + min_ in place (two roundings, like NumPy's X *= scale_; X += min_)
// Euclidean: sklearn upcasts f32 inputs to f64 (pairwise.py:582-653).  Here d2 = xx + aa - 2 dot is formed in f64
// from the f32 MFMA dot; where d2 is small against the norms (cancellation) the entry is recomputed as a direct
// sum of squared differences, so coincident rows give exactly 0 instead of ~1e-2.
#include "common.h"

namespace pope {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;          // rows of X per block (4 waves x 32)
constexpr int BN = 256;          // anchor columns per block (8 MFMA tiles of 32 per wave)
constexpr int BK = 64;           // depth staged in LDS at a time
constexpr int LDP = BK + 1;      // padded leading dimension: fragment reads and staging writes conflict-free
constexpr int NT = BN / 32;

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

// grid = (ceil(N / BM), ceil(K / BN)).  LDS: X tile [BM][LDP] + anchor tile [BN][LDP] + norms + reduction scratch.
__global__ __launch_bounds__(256) void k_pairwise(const float *__restrict__ X, int N, int D, const float *__restrict__ A,
                                                  int K, int metric, const double *__restrict__ xx, const double *__restrict__ aa,
                                                  float *__restrict__ out, long long out_cols, int c0,
                                                  float *__restrict__ part_min, float *__restrict__ part_max, int Kpad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Xs = reinterpret_cast<float *>(smem);                   // [BM][LDP]
    float *As = Xs + BM * LDP;                                     // [BN][LDP]
    float *red_min = As + BN * LDP;                                // [4][BN]
    float *red_max = red_min + 4 * BN;                             // [4][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    for (int k0 = 0; k0 < D; k0 += BK) {
        // stage: 16 consecutive threads read one row's 64 floats (256 contiguous bytes); zero beyond N / K / D
        for (int idx = tid; idx < BM * (BK / 4); idx += 256) {
            const int r = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (row0 + r < N) {
                const float *p = X + (size_t)(row0 + r) * D + k0 + kq;
                if (k0 + kq + 3 < D && (D & 3) == 0) {
                    const float4 q = *reinterpret_cast<const float4 *>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
                    for (int i = 0; i < 4; ++i) if (k0 + kq + i < D) v[i] = p[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) Xs[r * LDP + kq + i] = v[i];
        }
        for (int idx = tid; idx < BN * (BK / 4); idx += 256) {
            const int r = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (col0 + r < K) {
                const float *p = A + (size_t)(col0 + r) * D + k0 + kq;
                if (k0 + kq + 3 < D && (D & 3) == 0) {
                    const float4 q = *reinterpret_cast<const float4 *>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
                    for (int i = 0; i < 4; ++i) if (k0 + kq + i < D) v[i] = p[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) As[r * LDP + kq + i] = v[i];
        }
        __syncthreads();
        // v_mfma_f32_32x32x2_f32: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
        const float *xa = Xs + (wave * 32 + (lane & 31)) * LDP + (lane >> 5);
        const float *ab = As + (lane & 31) * LDP + (lane >> 5);
#pragma unroll 4
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = xa[kk];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ab[t * 32 * LDP + kk], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const float inf = __builtin_huge_valf();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = col0 + t * 32 + (lane & 31);
        const bool col_ok = col < K;
        const double a2 = col_ok ? aa[col] : 0.0;
        const float na = col_ok ? (a2 == 0.0 ? 1.0f : (float)sqrt(a2)) : 1.0f;
        float cmin = inf, cmax = -inf;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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
            red_min[wave * BN + t * 32 + lane] = cmin;
            red_max[wave * BN + t * 32 + lane] = cmax;
        }
    }
    __syncthreads();
    if (tid < BN && col0 + tid < K) {
        const float mn = fminf(fminf(red_min[tid], red_min[BN + tid]), fminf(red_min[2 * BN + tid], red_min[3 * BN + tid]));
        const float mx = fmaxf(fmaxf(red_max[tid], red_max[BN + tid]), fmaxf(red_max[2 * BN + tid], red_max[3 * BN + tid]));
        part_min[(size_t)blockIdx.x * Kpad + col0 + tid] = mn;
        part_max[(size_t)blockIdx.x * Kpad + col0 + tid] = mx;
    }
}

// sklearn MinMaxScaler.fit (_data.py:456-567) with feature_range (0, 1), all in float32.
__global__ __launch_bounds__(256) void k_minmax_reduce(const float *__restrict__ part_min, const float *__restrict__ part_max,
                                                       int nblocks, int K, int Kpad, float *__restrict__ scale,
                                                       float *__restrict__ shift) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= K) return;
    float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
    for (int b = 0; b < nblocks; ++b) {
        mn = fminf(mn, part_min[(size_t)b * Kpad + j]);
        mx = fmaxf(mx, part_max[(size_t)b * Kpad + j]);
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

static size_t pw_lds_bytes() { return (size_t)(BM * LDP + BN * LDP + 8 * BN) * sizeof(float); }

struct PwLayout {
    size_t xx, aa, pmin, pmax, scale, shift, total;
    int nblocks, Kpad;
};

static PwLayout pw_layout(int64_t N, int32_t K) {
    PwLayout L;
    L.nblocks = (int)((N + BM - 1) / BM);
    L.Kpad = (K + BN - 1) / BN * BN;
    size_t o = 0;
    L.xx = o;    o += align_up((size_t)N * sizeof(double), 256);
    L.aa = o;    o += align_up((size_t)K * sizeof(double), 256);
    L.pmin = o;  o += align_up((size_t)L.nblocks * L.Kpad * sizeof(float), 256);
    L.pmax = o;  o += align_up((size_t)L.nblocks * L.Kpad * sizeof(float), 256);
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
    float *scale = (float *)(base + L.scale), *shift = (float *)(base + L.shift);

    hipLaunchKernelGGL(k_sqnorm, dim3(capped_grid((size_t)N * 64, 256)), dim3(256), 0, stream, X, (long long)N, D, xx);
    hipLaunchKernelGGL(k_sqnorm, dim3(capped_grid((size_t)K * 64, 256)), dim3(256), 0, stream, A, (long long)K, D, aa);
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_pairwise, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pw_lds_bytes()));
        lds_opt_in = true;
    }
    hipLaunchKernelGGL(k_pairwise, dim3(L.nblocks, L.Kpad / BN), dim3(256), pw_lds_bytes(), stream, X, (int)N, D, A, K, metric,
                       xx, aa, out, (long long)out_cols, c0, pmin, pmax, L.Kpad);
    hipLaunchKernelGGL(k_minmax_reduce, dim3((K + 255) / 256), dim3(256), 0, stream, pmin, pmax, L.nblocks, K, L.Kpad, scale, shift);
    hipLaunchKernelGGL(k_minmax_apply, dim3(capped_grid((size_t)N * 64, 256)), dim3(256), 0, stream, out, (int)N, K,
                       (long long)out_cols, c0, scale, shift);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}
