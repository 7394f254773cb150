// K-means anchors of the node2vec branch on MI355X (gfx950).
//
// Replaces /root/reference/utils.py:168-170  KMeans(n_clusters=K).fit(node2vec_embeddings).cluster_centers_  (scikit-learn
// defaults: k-means++ seeding with 2 + int(log K) local trials, one run, Lloyd iterations on the mean-centred data until
// the labels stop changing or the squared centre shift drops below 1e-4 x the mean column variance, at most 300).
// Same algorithm, same consumption of the global NumPy random stream (the host binding draws the numbers and hands them
// over), distances and reductions on the device:
//   seeding    k_pp_candidates  squared distance of every point to the candidate rows (f64 accumulation),
//              min with the running closest distance and the potential of every candidate in one pass;
//              rocPRIM inclusive scan (float64) + k_pp_pick = searchsorted(cumsum(closest), u * potential);
//              k_pp_select = argmin of the candidate potentials.  255 steps of K = 256 without one host synchronisation.
//   assignment k_assign: the N x K dot products on the f32 MFMA tile of gemm_tile.h (the pairwise kernel's machinery);
//              the epilogue forms ||c||^2 - 2 x.c, reduces each row over the tile's columns with shuffles and merges the
//              column tiles with a 64-bit atomicMin on (ordered distance bits, centre index): first minimum wins, as argmin.
//   update     rocPRIM radix sort of (label, point) pairs, then one block per centre sums its points in sorted order in
//              float64: deterministic, no float atomics.  An empty cluster keeps its centre (scikit-learn relocates it to
//              the point farthest from its centre; k-means++ seeding makes that a corner case).
// Parity: not bit-level (scikit-learn's own result depends on its BLAS chunking); on well-separated data the same points
// are seeded in the same order and the centres agree to float32 rounding (tests/golden/node2vec_kmeans512.npz).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "gemm_tile.h"

namespace pope {

constexpr int KM_PARTS = 256;             // row slabs of the two-stage reductions
constexpr int KM_MAX_TRIALS = 16;         // 2 + int(log K) <= 16 for K < 1.2e6
constexpr int KM_TM = 64, KM_TN = 128;    // assignment tile: rows of X x centres per block (4 waves as 2 x 2)

// ---- column moments: sum and sum of squares of every column, float64, two deterministic stages ----
__global__ __launch_bounds__(256) void k_moments_partial(const float *__restrict__ X, long long N, int D, double *__restrict__ part) {
    const long long per = (N + gridDim.x - 1) / gridDim.x;
    const long long r0 = blockIdx.x * per, r1 = min(N, r0 + per);
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        double s = 0.0, ss = 0.0;
        for (long long r = r0; r < r1; ++r) {
            const double v = (double)X[r * D + c];
            s += v;
            ss += v * v;
        }
        part[((size_t)blockIdx.x * 2 + 0) * D + c] = s;
        part[((size_t)blockIdx.x * 2 + 1) * D + c] = ss;
    }
}

__global__ __launch_bounds__(256) void k_moments_final(const double *__restrict__ part, int parts, int D, double *__restrict__ sum,
                                                       double *__restrict__ sumsq) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    double s = 0.0, ss = 0.0;
    for (int p = 0; p < parts; ++p) {
        s += part[((size_t)p * 2 + 0) * D + c];
        ss += part[((size_t)p * 2 + 1) * D + c];
    }
    sum[c] = s;
    sumsq[c] = ss;
}

// out[r, c] = X[r, c] - shift[c]  (KMeans.fit subtracts the column means first; they are added back to the centres)
__global__ __launch_bounds__(256) void k_shift_columns(const float *__restrict__ X, const float *__restrict__ shift, long long N, int D,
                                                       float sign, float *__restrict__ out) {
    const long long total = N * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        out[i] = X[i] + sign * shift[i % D];
}

// ---- k-means++ seeding ----
struct PpState {                // device control block of the seeding
    double pot;                 // current potential = sum of the closest squared distances
    int cand[KM_MAX_TRIALS];    // candidate rows of the current step
};

// One wave per point: lane l holds the point's dimensions l, l + 64, ... (coalesced row read), the T candidate rows come from
// L1; squared distances accumulate in float64 and are rounded to float32 like scikit-learn's result.  Lane 0 of every wave
// sums its points' contributions to the T potentials; the block's four waves are folded through LDS.
//   first >= 0 (the seeding's first centre): T = 1, the candidate is row `first`, no running minimum yet.
__global__ __launch_bounds__(256) void k_pp_candidates(const float *__restrict__ X, long long N, int D, const PpState *__restrict__ st, int T,
                                                       long long first, const float *__restrict__ closest, float *__restrict__ newdist,
                                                       double *__restrict__ part) {
    __shared__ double red[4][KM_MAX_TRIALS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    double pot[KM_MAX_TRIALS];
#pragma unroll
    for (int t = 0; t < KM_MAX_TRIALS; ++t) pot[t] = 0.0;
    for (long long i = gw; i < N; i += nw) {
        const float *x = X + i * D;
        const float prev = first >= 0 ? __builtin_huge_valf() : closest[i];
#pragma unroll
        for (int t = 0; t < KM_MAX_TRIALS; ++t) {
            if (t >= T) continue;                                  // (no break: the loop must unroll so that pot[] stays in registers)
            const float *c = X + (first >= 0 ? first : (long long)st->cand[t]) * D;
            double acc = 0.0;
            for (int k = lane; k < D; k += 64) {
                const double d = (double)x[k] - (double)c[k];
                acc += d * d;
            }
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            const float d = fminf(prev, (float)acc);
            if (lane == 0) newdist[(size_t)t * N + i] = d;
            pot[t] += (double)d;
        }
    }
    if (lane == 0)
        for (int t = 0; t < T; ++t) red[wave][t] = pot[t];
    __syncthreads();
    if ((int)threadIdx.x < T)
        part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// One block.  step < 0: the potential after the first centre.  Otherwise: argmin of the T candidate potentials (first minimum),
// the winner's distances become `closest`, its row is centre `step`.
__global__ __launch_bounds__(256) void k_pp_select(const double *__restrict__ part, int blocks, int T, int step, PpState *st,
                                                   const float *__restrict__ newdist, float *__restrict__ closest, long long N,
                                                   long long *__restrict__ chosen) {
    __shared__ double pots[KM_MAX_TRIALS];
    __shared__ int best_s;
    if ((int)threadIdx.x < max(T, 1)) {
        double s = 0.0;
        for (int b = 0; b < blocks; ++b) s += part[(size_t)threadIdx.x * blocks + b];
        pots[threadIdx.x] = s;
    }
    __syncthreads();
    if (step < 0) {
        if (threadIdx.x == 0) st->pot = pots[0];
        return;
    }
    if (threadIdx.x == 0) {
        int best = 0;
        for (int t = 1; t < T; ++t)
            if (pots[t] < pots[best]) best = t;
        best_s = best;
        st->pot = pots[best];
        chosen[step] = st->cand[best];
    }
    __syncthreads();
    const float *src = newdist + (size_t)best_s * N;
    for (long long i = threadIdx.x; i < N; i += blockDim.x) closest[i] = src[i];
}

// cand[t] = searchsorted(cum, u[t] * pot) (side = left), clipped to N - 1
__global__ void k_pp_pick(const double *__restrict__ cum, long long N, const double *__restrict__ u, int T, PpState *st) {
    const int t = threadIdx.x;
    if (t >= T) return;
    const double v = u[t] * st->pot;
    long long lo = 0, hi = N;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (cum[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    st->cand[t] = (int)min(lo, N - 1);
}

struct F32ToF64 {
    __host__ __device__ double operator()(float v) const { return (double)v; }
};

// ---- Lloyd iteration ----
__global__ __launch_bounds__(256) void k_row_sqnorm32(const float *__restrict__ C, int K, int D, float *__restrict__ c2) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < K; r += nwaves) {
        double acc = 0.0;
        for (int k = lane; k < D; k += 64) acc += (double)C[(size_t)r * D + k] * (double)C[(size_t)r * D + k];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) c2[r] = (float)acc;
    }
}

__device__ __forceinline__ unsigned ordered_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // unsigned order = float order
}

// keys[i] = min over this block's centres of (ordered(||c||^2 - 2 x_i . c), centre index), merged across column tiles by atomicMin.
template <int LAYOUT>
__global__ __launch_bounds__(256) void k_assign(const float *__restrict__ X, int N, int D, const float *__restrict__ C, int K,
                                                const float *__restrict__ c2, unsigned long long *__restrict__ keys) {
    constexpr int WM = 2, WN = 2, NT = KM_TN / WN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);
    float *Bs = As + Tile<KM_TM>::FLOATS;
    __shared__ unsigned long long red[WN][KM_TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int row0 = blockIdx.x * KM_TM, col0 = blockIdx.y * KM_TN;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const Operand Xo{X, D, 1}, Co{C, D, 1}, none{nullptr, 0, 0};
    mfma_accumulate<KM_TM, KM_TN, WM, WN, LAYOUT, LAYOUT>(acc, Xo, Co, 0, D, none, none, 0, 0, row0, col0, N, K, As, Bs);
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned long long best = ~0ull;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = col0 + wn * (KM_TN / WN) + t * 32 + (lane & 31);
            if (col < K) {
                const float score = fmaf(-2.0f, acc[t][r], c2[col]);
                const unsigned long long key = ((unsigned long long)ordered_bits(score) << 32) | (unsigned)col;
                best = key < best ? key : best;
            }
        }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {                        // over the 32 lanes that hold this row's other columns
            const unsigned long long other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
        if ((lane & 31) == 0) red[wn][wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = best;
    }
    __syncthreads();
    if (tid < KM_TM && row0 + tid < N) {
        unsigned long long best = red[0][tid];
#pragma unroll
        for (int w = 1; w < WN; ++w) best = red[w][tid] < best ? red[w][tid] : best;
        atomicMin(&keys[row0 + tid], best);
    }
}

// labels from the keys; stats[0] |= (some label changed)
__global__ __launch_bounds__(256) void k_labels(const unsigned long long *__restrict__ keys, long long N, int *__restrict__ labels,
                                                const int *__restrict__ labels_prev, int *__restrict__ sort_idx, int *stats) {
    bool changed = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(keys[i] & 0xFFFFFFFFull);
        labels[i] = l;
        sort_idx[i] = (int)i;
        changed |= l != labels_prev[i];
    }
    if (changed) atomicOr(&stats[0], 1);
}

// One block per centre: mean of its points (sorted_idx[first .. last) found by binary search in the sorted labels), float64
// sums in sorted order; shift2[k] = |new - old|^2.  An empty cluster keeps its centre.
__global__ __launch_bounds__(128) void k_update(const float *__restrict__ X, int D, const int *__restrict__ sorted_labels,
                                                const int *__restrict__ sorted_idx, long long N, const float *__restrict__ C_old,
                                                float *__restrict__ C_new, double *__restrict__ shift2) {
    __shared__ double red[128];
    const int k = blockIdx.x;
    long long lo = 0, hi = N;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (sorted_labels[mid] < k) lo = mid + 1; else hi = mid; }
    const long long first = lo;
    hi = N;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (sorted_labels[mid] <= k) lo = mid + 1; else hi = mid; }
    const long long last = lo;
    double sh = 0.0;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float v = C_old[(size_t)k * D + c];
        if (last > first) {
            double s = 0.0;
            for (long long q = first; q < last; ++q) s += (double)X[(size_t)sorted_idx[q] * D + c];
            v = (float)(s / (double)(last - first));
        }
        const double d = (double)v - (double)C_old[(size_t)k * D + c];
        sh += d * d;
        C_new[(size_t)k * D + c] = v;
    }
    red[threadIdx.x] = sh;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) shift2[k] = red[0];
}

__global__ void k_shift_total(const double *__restrict__ shift2, int K, double *out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += shift2[k];
        *out = s;
    }
}

struct KmLayout {
    size_t part, pp_state, cum, scan_tmp, newdist, u, keys, c2, sort_labels, sort_idx, idx, sort_tmp, shift2, total;
};

static size_t km_scan_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, rocprim::make_transform_iterator((const float *)nullptr, F32ToF64()), (double *)nullptr, n,
                                  rocprim::plus<double>());
    return bytes;
}

static size_t km_sort_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int *)nullptr, (int *)nullptr, (const int *)nullptr, (int *)nullptr, n);
    return bytes;
}

static KmLayout km_layout(int64_t N, int32_t D, int32_t K) {
    KmLayout L;
    size_t o = 0;
    const size_t part_d = (size_t)KM_PARTS * 2 * (size_t)(D > KM_MAX_TRIALS ? D : KM_MAX_TRIALS) * sizeof(double);
    L.part = o;        o += align_up(part_d, 256);
    L.pp_state = o;    o += 256;
    L.cum = o;         o += align_up((size_t)N * sizeof(double), 256);
    L.scan_tmp = o;    o += align_up(km_scan_bytes((size_t)N), 256);
    L.newdist = o;     o += align_up((size_t)KM_MAX_TRIALS * N * sizeof(float), 256);
    L.u = o;           o += align_up((size_t)K * KM_MAX_TRIALS * sizeof(double), 256);
    L.keys = o;        o += align_up((size_t)N * sizeof(unsigned long long), 256);
    L.c2 = o;          o += align_up((size_t)K * sizeof(float), 256);
    L.sort_labels = o; o += align_up((size_t)N * sizeof(int), 256);
    L.sort_idx = o;    o += align_up((size_t)N * sizeof(int), 256);
    L.idx = o;         o += align_up((size_t)N * sizeof(int), 256);
    L.sort_tmp = o;    o += align_up(km_sort_bytes((size_t)N), 256);
    L.shift2 = o;      o += align_up((size_t)K * sizeof(double), 256);
    L.total = o;
    return L;
}

}  // namespace pope

using namespace pope;

extern "C" size_t pope_kmeans_scratch_bytes(int64_t N, int32_t D, int32_t K) {
    if (N <= 0 || D <= 0 || K <= 0) return 0;
    return km_layout(N, D, K).total;
}

extern "C" int pope_column_moments(const float *X, int64_t N, int32_t D, double *sum, double *sumsq, void *scratch, size_t scratch_bytes,
                                   void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(X && sum && sumsq && scratch && N > 0 && D > 0, "pope_column_moments: bad argument");
    POPE_REQUIRE(scratch_bytes >= (size_t)KM_PARTS * 2 * D * sizeof(double), "pope_column_moments: scratch too small");
    hipLaunchKernelGGL(k_moments_partial, dim3(KM_PARTS), dim3(256), 0, stream, X, (long long)N, D, (double *)scratch);
    hipLaunchKernelGGL(k_moments_final, dim3((D + 255) / 256), dim3(256), 0, stream, (const double *)scratch, KM_PARTS, D, sum, sumsq);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int pope_shift_columns(const float *X, const float *shift, int64_t N, int32_t D, float sign, float *out, void *stream_) {
    clear_error();
    POPE_REQUIRE(X && shift && out && N > 0 && D > 0, "pope_shift_columns: bad argument");
    hipLaunchKernelGGL(k_shift_columns, dim3(capped_grid((size_t)N * D, 256)), dim3(256), 0, (hipStream_t)stream_, X, shift, (long long)N, D,
                       sign, out);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int pope_kmeans_plusplus(const float *X, int64_t N, int32_t D, int32_t K, int64_t first_id, const double *uniforms_host,
                                    int32_t n_trials, int64_t *chosen, void *scratch, size_t scratch_bytes, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(X && chosen && scratch && (uniforms_host || K == 1), "pope_kmeans_plusplus: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && D > 0 && K > 0 && K <= N && first_id >= 0 && first_id < N && n_trials >= 1 &&
                 n_trials <= KM_MAX_TRIALS, "pope_kmeans_plusplus: bad size");
    const KmLayout L = km_layout(N, D, K);
    if (scratch_bytes < L.total) {
        set_error("pope_kmeans_plusplus: scratch %zu < %zu bytes", scratch_bytes, L.total);
        return POPE_ERR_WORKSPACE;
    }
    char *base = (char *)scratch;
    double *part = (double *)(base + L.part), *cum = (double *)(base + L.cum), *u = (double *)(base + L.u);
    PpState *st = (PpState *)(base + L.pp_state);
    float *closest = (float *)(base + L.keys);                 // the keys region is free during seeding
    float *newdist = (float *)(base + L.newdist);
    void *scan_tmp = base + L.scan_tmp;
    size_t scan_bytes = km_scan_bytes((size_t)N);
    const int T = n_trials;
    if (K > 1) POPE_HIP(hipMemcpyAsync(u, uniforms_host, (size_t)(K - 1) * T * sizeof(double), hipMemcpyHostToDevice, stream));
    POPE_HIP(hipMemcpyAsync(chosen, &first_id, sizeof(int64_t), hipMemcpyHostToDevice, stream));
    POPE_HIP(hipStreamSynchronize(stream));                     // first_id / uniforms may live on the caller's stack
    hipLaunchKernelGGL(k_pp_candidates, dim3(KM_PARTS), dim3(256), 0, stream, X, (long long)N, D, (const PpState *)st, 1, (long long)first_id,
                       (const float *)nullptr, closest, part);                      // distances to the first centre ARE the closest
    hipLaunchKernelGGL(k_pp_select, dim3(1), dim3(256), 0, stream, (const double *)part, KM_PARTS, 1, -1, st, (const float *)nullptr,
                       (float *)nullptr, (long long)N, (long long *)chosen);
    for (int c = 1; c < K; ++c) {
        POPE_HIP(rocprim::inclusive_scan(scan_tmp, scan_bytes, rocprim::make_transform_iterator((const float *)closest, F32ToF64()), cum,
                                         (size_t)N, rocprim::plus<double>(), stream));
        hipLaunchKernelGGL(k_pp_pick, dim3(1), dim3(64), 0, stream, (const double *)cum, (long long)N, (const double *)(u + (size_t)(c - 1) * T), T, st);
        hipLaunchKernelGGL(k_pp_candidates, dim3(KM_PARTS), dim3(256), 0, stream, X, (long long)N, D, (const PpState *)st, T, (long long)-1,
                           (const float *)closest, newdist, part);
        hipLaunchKernelGGL(k_pp_select, dim3(1), dim3(256), 0, stream, (const double *)part, KM_PARTS, T, c, st, (const float *)newdist, closest,
                           (long long)N, (long long *)chosen);
    }
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int pope_kmeans_lloyd_step(const float *X, int64_t N, int32_t D, const float *centers, int32_t K, float *centers_new,
                                      int32_t *labels, const int32_t *labels_prev, int32_t *changed, double *shift_total,
                                      void *scratch, size_t scratch_bytes, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(X && centers && centers_new && labels && labels_prev && changed && shift_total && scratch, "pope_kmeans_lloyd_step: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && D > 0 && K > 0 && centers != centers_new && labels != labels_prev, "pope_kmeans_lloyd_step: bad argument");
    const KmLayout L = km_layout(N, D, K);
    if (scratch_bytes < L.total) {
        set_error("pope_kmeans_lloyd_step: scratch %zu < %zu bytes", scratch_bytes, L.total);
        return POPE_ERR_WORKSPACE;
    }
    char *base = (char *)scratch;
    unsigned long long *keys = (unsigned long long *)(base + L.keys);
    float *c2 = (float *)(base + L.c2);
    int *sl = (int *)(base + L.sort_labels), *si = (int *)(base + L.sort_idx), *idx = (int *)(base + L.idx);
    double *shift2 = (double *)(base + L.shift2);
    size_t sort_bytes = km_sort_bytes((size_t)N);
    POPE_HIP(hipMemsetAsync(keys, 0xFF, (size_t)N * sizeof(unsigned long long), stream));
    POPE_HIP(hipMemsetAsync(changed, 0, sizeof(int), stream));
    hipLaunchKernelGGL(k_row_sqnorm32, dim3(capped_grid((size_t)K * 64, 256)), dim3(256), 0, stream, centers, K, D, c2);
    const Operand Xo{X, D, 1}, Co{centers, D, 1};
    const bool vec = pick_layout(Xo, (int)N, D) == LAYOUT_KC_VEC && pick_layout(Co, K, D) == LAYOUT_KC_VEC;
    const size_t lds = tile_lds_bytes<KM_TM, KM_TN>();
    static LdsOptIn opt_in;
    if (!opt_in.done()) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_assign<LAYOUT_KC_VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        POPE_HIP(hipFuncSetAttribute((const void *)k_assign<LAYOUT_GENERIC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        opt_in.mark();
    }
    const dim3 grid((unsigned)((N + KM_TM - 1) / KM_TM), (unsigned)((K + KM_TN - 1) / KM_TN));
    if (vec) hipLaunchKernelGGL(k_assign<LAYOUT_KC_VEC>, grid, dim3(256), lds, stream, X, (int)N, D, centers, K, (const float *)c2, keys);
    else hipLaunchKernelGGL(k_assign<LAYOUT_GENERIC>, grid, dim3(256), lds, stream, X, (int)N, D, centers, K, (const float *)c2, keys);
    hipLaunchKernelGGL(k_labels, dim3(capped_grid((size_t)N, 256)), dim3(256), 0, stream, (const unsigned long long *)keys, (long long)N, labels,
                       labels_prev, idx, changed);
    POPE_HIP(rocprim::radix_sort_pairs(base + L.sort_tmp, sort_bytes, (const int *)labels, sl, (const int *)idx, si, (size_t)N, 0, 32, stream));
    hipLaunchKernelGGL(k_update, dim3(K), dim3(128), 0, stream, X, D, (const int *)sl, (const int *)si, (long long)N, centers, centers_new, shift2);
    hipLaunchKernelGGL(k_shift_total, dim3(1), dim3(1), 0, stream, (const double *)shift2, K, shift_total);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}
