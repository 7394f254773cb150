// Biased anchor selection on the device: PageRank scores for sampling_method='pagerank'.
//
// Replaces /root/reference/utils.py:26-30  nx.pagerank_scipy(to_networkx(data))  -- in NetworkX 3 the same SciPy power
// iteration lives in nx.pagerank (_pagerank_scipy): x <- alpha * (x @ A + sum(x[dangling]) * p) + (1 - alpha) * p with
// A = D^-1 * adjacency of the DiGraph (one edge per distinct (u, v) pair), p = 1 / N, until sum |x - xlast| < N * tol.
// The anchors are the last K nodes of an ascending stable sort by score, so the scores are reproduced BIT FOR BIT:
//   * x @ A is SciPy's csc_matvec on A^T: y[i] accumulates A[j, i] * x[j] over the in-neighbours j in ASCENDING order,
//     one rounded multiply and one rounded add per edge (no fused multiply-add) -- k_pagerank_pull walks the rows of the
//     by-target CSR, whose sources are sorted (pope_csr_build_canonical), one thread per row, in that order;
//   * the elementwise update is evaluated with the same association, every operation rounded separately;
//   * sum(x[dangling]) (a Python sum in index order) and the convergence norm (NumPy's pairwise sum) are evaluated by the
//     host binding on the copied-back vector, so the iteration count is the reference's too.
#include "common.h"

// Every multiply and add in this file is rounded on its own, as SciPy's and NumPy's compiled loops do: hipcc's default
// -ffp-contract=fast would fuse  acc + w * x  into one fma (HIP's __dmul_rn / __dadd_rn are plain operators and do not
// prevent it) and the scores would differ from NetworkX's in the last bits.
#pragma clang fp contract(off)

namespace pope {

// w[j] = 1 / (number of distinct targets of j), 0 for a dangling node.  Rows of the canonical CSR are sorted by target:
// repeated edges are adjacent and count once (the DiGraph keeps one edge per pair).
__global__ __launch_bounds__(256) void k_outdeg_weights(const int *__restrict__ rowptr, const int *__restrict__ col, int N,
                                                        double *__restrict__ w) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        const int beg = rowptr[j], end = rowptr[j + 1];
        int distinct = 0;
        for (int p = beg; p < end; ++p) distinct += (p == beg || col[p] != col[p - 1]);
        w[j] = distinct ? 1.0 / (double)distinct : 0.0;
    }
}

// x_out[i] = alpha * (sum_{j -> i, j ascending} w[j] * x[j]  +  dsum * p) + (1 - alpha) * p,   p = 1 / N.
__global__ __launch_bounds__(256) void k_pagerank_pull(const int *__restrict__ rowptr_t, const int *__restrict__ src_t, int N,
                                                       const double *__restrict__ x, const double *__restrict__ w, double dsum,
                                                       double alpha, double *__restrict__ x_out) {
    const double p = 1.0 / (double)N;
    const double dangling = dsum * p;                             // sum(x[is_dangling]) * dangling_weights
    const double teleport = (1.0 - alpha) * p;                    // (1 - alpha) * p
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int beg = rowptr_t[i], end = rowptr_t[i + 1];
        double acc = 0.0;
        for (int q = beg; q < end; ++q) {
            const int j = src_t[q];
            if (q > beg && j == src_t[q - 1]) continue;           // a repeated edge: one entry in the DiGraph
            const double prod = w[j] * x[j];                       // plain operators: under the pragma above they carry no
            acc = acc + prod;                                      // 'contract' flag (the __dmul_rn / __dadd_rn wrappers do)
        }
        const double inner = acc + dangling;
        const double scaled = alpha * inner;
        x_out[i] = scaled + teleport;
    }
}

}  // namespace pope

using namespace pope;

extern "C" int pope_pagerank_weights(const int32_t *rowptr, const int32_t *col, int64_t N, double *w, void *stream_) {
    clear_error();
    POPE_REQUIRE(rowptr && w && N > 0 && N < INT32_MAX, "pope_pagerank_weights: bad argument");
    hipLaunchKernelGGL(k_outdeg_weights, dim3(capped_grid((size_t)N, 256)), dim3(256), 0, (hipStream_t)stream_, rowptr, col, (int)N, w);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int pope_pagerank_step(const int32_t *rowptr_by_target, const int32_t *sources, int64_t N, const double *x, const double *w,
                                  double dangling_sum, double alpha, double *x_out, void *stream_) {
    clear_error();
    POPE_REQUIRE(rowptr_by_target && x && w && x_out && x != x_out && N > 0 && N < INT32_MAX, "pope_pagerank_step: bad argument");
    hipLaunchKernelGGL(k_pagerank_pull, dim3(capped_grid((size_t)N, 256)), dim3(256), 0, (hipStream_t)stream_, rowptr_by_target, sources,
                       (int)N, x, w, dangling_sum, alpha, x_out);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}
