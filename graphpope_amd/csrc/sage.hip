// SAGEConv forward / backward on MI355X (gfx950) for the sampled bipartite blocks GraphPOPE trains on.
//
// Replaces the PyG SAGEConv call at /root/reference/main.py:206  x = convs[i]((x, x[:n_dst]), adj_t):
//     out = lin_l(mean_{j in N(i)} x_src[j]) + lin_r(x_src[i])          lin_l: weight + bias, lin_r: weight only
// (torch_sparse matmul(adj_t, x, reduce='mean') + two torch Linear layers in the reference stack).
//
//   k_gather_mean    neighbour gather + mean over the CSR block: one wave per destination row, 16-byte lane
//                    accesses along the feature dimension, several neighbour rows in flight.  HBM / L2 bound.
//   k_gemm           exact-f32 MFMA (v_mfma_f32_32x32x2_f32) tile, 128 x 256 or 64 x 128 per block, operands staged through
//                    LDS; C = sum_p A_p * B_p (+ bias) with p <= 2, so lin_l(agg) + lin_r(x_dst) is ONE pass
//                    that never materialises either product.  Arbitrary strides cover NT / NN / TN shapes;
//                    split-K (grid.z) with partial slabs + k_slab_reduce for the weight gradients
//                    (reduction over ~40 000 rows), deterministic: no float atomics there.
//   k_gemm_streamk_ld / k_gemm_streamk_tn   (gemm_streamk.h, gemm_streamk_tn.h) the two big products of a layer -- the forward
//                    projection and the weight-gradient twin -- as stream-K over one persistent block per CU: LDS-DMA staging by
//                    loader waves, MFMA-only consumer waves, partial tiles to slabs + a fix-up launch that sums them in block
//                    order; shapes they do not take (depth % 4, alignment, offsets beyond 32 bits, too few units) stay on k_gemm.
//   k_scatter_mean   grad_x[col[p]] += grad_agg[i] / deg(i) (float atomics, whole 16-byte-aligned row segments)
//   k_colsum_*       grad_bias, two deterministic stages
#include <mutex>

#include "gemm_streamk_tn.h"
#include "gemm_tile16.h"
#include "side_copy.h"

namespace pope {

// ------------------------------------------------------------------------------------------------
// neighbour gather + mean
// ------------------------------------------------------------------------------------------------
// n_id != nullptr (indexed mode): x is the WHOLE feature matrix and block-local source j lives in row n_id[j]
// (convert_batch's x = data.x[n_id], main.py:118-123, never materialised); destination i's own row goes to x_dst[i].
__device__ __forceinline__ long long src_row(int j, const long long *__restrict__ n_id) { return n_id ? n_id[j] : (long long)j; }

template <bool VEC>
__global__ __launch_bounds__(256) void k_gather_mean(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                     int n_dst, const float *__restrict__ x, int C,
                                                     float *__restrict__ agg, const long long *__restrict__ n_id,
                                                     float *__restrict__ x_dst, const int *__restrict__ n_dst_dev) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    n_dst = dyn_extent(n_dst_dev, n_dst);
    for (int i = wave; i < n_dst; i += nwaves) {
        const int beg = rowptr[i], end = rowptr[i + 1];
        const float inv = end > beg ? 1.0f / (float)(end - beg) : 0.0f;
        if (VEC) {
            const int C4 = C >> 2;
            for (int q = lane; q < C4; q += 64) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                int p = beg;
                for (; p + 3 < end; p += 4) {                              // four neighbour rows in flight
                    const long long j0 = src_row(col[p], n_id), j1 = src_row(col[p + 1], n_id), j2 = src_row(col[p + 2], n_id), j3 = src_row(col[p + 3], n_id);
                    const float4 a = reinterpret_cast<const float4 *>(x + (size_t)j0 * C)[q];
                    const float4 b = reinterpret_cast<const float4 *>(x + (size_t)j1 * C)[q];
                    const float4 c = reinterpret_cast<const float4 *>(x + (size_t)j2 * C)[q];
                    const float4 d = reinterpret_cast<const float4 *>(x + (size_t)j3 * C)[q];
                    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                    s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
                    s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
                    s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
                }
                for (; p < end; ++p) {
                    const float4 a = reinterpret_cast<const float4 *>(x + (size_t)src_row(col[p], n_id) * C)[q];
                    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                }
                s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
                reinterpret_cast<float4 *>(agg + (size_t)i * C)[q] = s;
                if (x_dst) reinterpret_cast<float4 *>(x_dst + (size_t)i * C)[q] = reinterpret_cast<const float4 *>(x + (size_t)n_id[i] * C)[q];
            }
        } else {
            for (int c = lane; c < C; c += 64) {
                float s = 0.f;
                for (int p = beg; p < end; ++p) s += x[(size_t)src_row(col[p], n_id) * C + c];
                agg[(size_t)i * C + c] = s * inv;
                if (x_dst) x_dst[(size_t)i * C + c] = x[(size_t)n_id[i] * C + c];
            }
        }
    }
}

// grad_x[col[p], :] += grad_agg[i, :] / deg(i).  One wave per (destination row, 256-column slab): the row's neighbour
// ids are loaded once, 64 at a time, and broadcast with shuffles, and the slab's four gradient values per lane are in
// registers before the first atomic -- the loop issues nothing but no-return atomics (256 contiguous bytes per wave
// instruction), instead of a dependent col[p] load in front of every one (38 us -> see DESIGN.md §7).
__device__ __forceinline__ void scatter_mean_block(const int *__restrict__ rowptr, const int *__restrict__ col, int n_dst,
                                                   const float *__restrict__ gagg, int C, float *__restrict__ gx,
                                                   const int *__restrict__ n_dst_dev, const int block, const int nblocks) {
    const int lane = threadIdx.x & 63;
    const int wave = (block * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (nblocks * blockDim.x) >> 6;
    const int slabs = (C + 255) / 256;
    n_dst = dyn_extent(n_dst_dev, n_dst);
    for (int w = wave; w < n_dst * slabs; w += nwaves) {
        const int i = w / slabs, c0 = (w - i * slabs) * 256;
        const int beg = rowptr[i], end = rowptr[i + 1];
        if (end == beg) continue;
        const float inv = 1.0f / (float)(end - beg);
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k * 64 + lane;
            g[k] = c < C ? gagg[(size_t)i * C + c] * inv : 0.f;
        }
        for (int p0 = beg; p0 < end; p0 += 64) {
            const int mine = p0 + lane < end ? col[p0 + lane] : 0;
            const int cnt = min(64, end - p0);
            for (int p = 0; p < cnt; ++p) {
                float *row = gx + (size_t)__shfl(mine, p) * C;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + k * 64 + lane;
                    if (c < C) atomicAdd(&row[c], g[k]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_scatter_mean(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                      int n_dst, const float *__restrict__ gagg, int C,
                                                      float *__restrict__ gx, const int *__restrict__ n_dst_dev) {
    scatter_mean_block(rowptr, col, n_dst, gagg, C, gx, n_dst_dev, blockIdx.x, gridDim.x);
}

// x[r, :] = 0 for r in [r0, r1): the rows of grad_x that only the scatter adds to.  Both bounds may live on the device.
__global__ __launch_bounds__(256) void k_zero_rows(float *__restrict__ x, int C, int r0, int r1, const int *__restrict__ r0_dev,
                                                   const int *__restrict__ r1_dev) {
    r0 = dyn_extent(r0_dev, r0);
    r1 = dyn_extent(r1_dev, r1);
    if (r1 <= r0) return;
    const size_t lo = (size_t)r0 * C, n = (size_t)(r1 - r0) * C;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float *base = x + lo;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(base)) & 15) == 0) {
        const size_t n4 = n >> 2;
        for (size_t q = i; q < n4; q += stride) reinterpret_cast<float4 *>(base)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t q = (n4 << 2) + i; q < n; q += stride) base[q] = 0.f;
    } else {
        for (size_t q = i; q < n; q += stride) base[q] = 0.f;
    }
}

// Column sums in two deterministic stages: part[s][c] = sum over row slice s, then out[c] = sum_s part[s][c].
constexpr int COLSUM_SPLITS = 256;

// VEC: lane l owns 4 adjacent columns (16-byte loads, a wave covers 256 columns of a row); else one column per lane.
// (bx, by) of (gx, gy): the block's place in the launch -- a kernel of its own (k_colsum_partial) or a role of k_gemm_dual.
template <bool VEC>
__device__ __forceinline__ void colsum_partial_block(const float *__restrict__ g, int rows, int C, float *__restrict__ part, int bx, int by, int gy) {
    constexpr int W = VEC ? 4 : 1;
    __shared__ float red[4][64 * W];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (bx * 64 + lane) * W;
    const int per = (rows + gy - 1) / gy;
    const int r0 = by * per, r1 = min(rows, r0 + per);
    float s[W];
#pragma unroll
    for (int k = 0; k < W; ++k) s[k] = 0.f;
    if (c < C) {
        // four rows requested before the first is added (round 4: one load per trip compiled to load - wait - add, a wave's ten
        // rows ten serial round trips); the additions keep their order, so the sums keep their bits
        int i = r0 + wave;
        for (; i + 12 < r1; i += 16) {
            if constexpr (VEC) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(g + (size_t)(i + 4 * u) * C + c);
#pragma unroll
                for (int u = 0; u < 4; ++u) { s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w; }
            } else {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = g[(size_t)(i + 4 * u) * C + c];
#pragma unroll
                for (int u = 0; u < 4; ++u) s[0] += v[u];
            }
        }
        for (; i < r1; i += 4) {
            if constexpr (VEC) {
                const float4 v = *reinterpret_cast<const float4 *>(g + (size_t)i * C + c);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            } else {
                s[0] += g[(size_t)i * C + c];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) red[wave][lane * W + k] = s[k];
    __syncthreads();
    if (wave == 0 && c < C)
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const int j = lane * W + k;
            part[(size_t)by * C + c + k] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
        }
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ g, int rows, int C, float *__restrict__ part,
                                                        const int *__restrict__ rows_dev) {
    colsum_partial_block<VEC>(g, dyn_extent(rows_dev, rows), C, part, blockIdx.x, blockIdx.y, gridDim.y);
}

// 16 columns x 16 split-groups per block: a one-thread-per-column loop over the 64 partials is 64 dependent-latency
// loads (15 us measured); here every thread adds 4 and LDS folds the 16 groups in a fixed order.
__global__ __launch_bounds__(256) void k_colsum_final(const float *__restrict__ part, int splits, int C, float *__restrict__ out) {
    colsum_final_block(part, splits, C, out, blockIdx.x);            // (gemm_streamk_tn.h: also a role of the weight gradients' fix-up launch)
}

// ------------------------------------------------------------------------------------------------
// f32 MFMA GEMM (tile machinery in gemm_tile.h)
// ------------------------------------------------------------------------------------------------
// C[M, N] = sum_{p < 2} A_p[M, K_p] * B_p[N, K_p]^T (+ bias[n]).  grid = (ceil(M/TM), ceil(N/TN), splits).
// splits > 1: every z handles a slice of the depth of every product and writes its partial tile to
// slab[z][M][N]; k_slab_reduce adds them (fixed order: deterministic).
// Twin mode (twin.tiles_n > 0): TWO results that share the A operand, C = A0 * B0^T and twin.C = A0 * twin.B^T, in one
// launch -- grid.y covers the tile columns of both (the weight gradients of lin_l and lin_r both multiply grad_out^T;
// grad_x and grad_agg both multiply grad_out): twice the blocks per launch, half the split-K slabs, one launch less.
struct Twin {
    Operand B;
    float *C;
    int tiles_n;               // tile columns of the first result; 0 = plain GEMM
};

// Device extents of a k_gemm launch (null: the host-side size is the size): the true M, the true depth of product 0.
struct GemmDyn {
    const int *m = nullptr, *k0 = nullptr;
};

struct GemmArgs {
    Operand A0, B0;
    int K0;
    Operand A1, B1;
    int K1, M, N;
    const float *bias;
    float *C;
    long long ldc;
    float *slab;
    Twin twin;
    const int *m_dev, *k0_dev;     // device extents (GemmDyn)
    int gx, gy, gz;                // the launch shape this problem was planned for: row tiles, column tiles (x2 in twin mode), splits
};

// One block of a planned GEMM: tile (bx, by), split bz.
template <int TM, int TN, int WM, int WN, int LA, int LB>
__device__ __forceinline__ void gemm_block(GemmArgs p, int bx, int by, int z, float *As, float *Bs) {
    constexpr int NT = TN / WN / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int splits = p.gz;
    // device extents: the true row count / the true depth of product 0 (the launch covers the capacities).  The slab
    // stride stays the capacity M: k_slab_reduce is given the same.
    const int M_cap = p.M;
    const int M = dyn_extent(p.m_dev, p.M), N = p.N, K0 = dyn_extent(p.k0_dev, p.K0), K1 = p.K1;
    if (bx * TM >= M) return;
    Operand B0 = p.B0;
    float *C = p.C, *slab = p.slab;
    if (p.twin.tiles_n > 0 && by >= p.twin.tiles_n) {       // second result: its own B, C and slab region
        by -= p.twin.tiles_n;
        B0 = p.twin.B;
        C = p.twin.C;
        slab += (size_t)splits * M_cap * N;
    }
    const int m0 = bx * TM, n0 = by * TN;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // depth range of this split inside each product, in whole LDS stages
    const int per0 = K0 > 0 ? ((K0 + splits - 1) / splits + GK - 1) / GK * GK : 0;
    const int per1 = K1 > 0 ? ((K1 + splits - 1) / splits + GK - 1) / GK * GK : 0;
    const int kb0 = z * per0, ke0 = min(K0, kb0 + per0), kb1 = z * per1, ke1 = min(K1, kb1 + per1);
    mfma_accumulate<TM, TN, WM, WN, LA, LB>(acc, p.A0, B0, kb0, ke0, p.A1, p.B1, kb1, ke1, m0, n0, M, N, As, Bs);
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *dst = splits > 1 ? slab + (size_t)z * M_cap * N : C;
    const long long ld = splits > 1 ? (long long)N : p.ldc;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + wn * (TN / WN) + t * 32 + (lane & 31);
        if (n >= N) continue;
        const float b = (p.bias && splits == 1) ? p.bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) dst[(size_t)m * ld + n] = acc[t][r] + b;
        }
    }
}

template <int TM, int TN, int WM, int WN, int LA, int LB>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);
    gemm_block<TM, TN, WM, WN, LA, LB>(p, blockIdx.x, blockIdx.y, blockIdx.z, As, As + Tile<TM>::FLOATS);
}

// Four independent pieces of a small layer's backward pass that only share grad_out, as ONE launch (sage_conv_backward):
//   blocks [0, n0)            grad_x[:n_dst] = grad_out * W_r  and  grad_agg = grad_out * W_l   (twin GEMM p0, operands KC x OC)
//   blocks [n0, n0 + n1)      the split-K slabs of grad_w_l / grad_w_r                          (twin GEMM p1, operands OC x OC)
//   the next zero_blocks      grad_x[n_dst : n_src] = 0 (the rows only the scatter adds to)
//   the last cs_gx * cs_gy    partial column sums of grad_out (the bias gradient)
// Each of them alone is a 5-13 us launch of ~200 blocks that leaves half the chip idle: 36 us in sequence, one launch together.
struct DualAux {
    float *zero_x;                 // grad_x, or null
    int zero_C, zero_r0, zero_r1;
    const int *zero_r0_dev, *zero_r1_dev;
    int zero_blocks;
    const float *cs_g;             // grad_out for the column sums, or null
    int cs_rows, cs_C;
    float *cs_part;
    const int *cs_rows_dev;
    int cs_gx, cs_gy;
};

__device__ __forceinline__ void zero_rows_block(float *__restrict__ x, int C, int r0, int r1, int b, int nb) {
    if (r1 <= r0) return;
    const size_t n = (size_t)(r1 - r0) * C;
    const size_t stride = (size_t)nb * blockDim.x;
    float *base = x + (size_t)r0 * C;
    const size_t i = (size_t)b * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
        const size_t n4 = n >> 2;
        for (size_t q = i; q < n4; q += stride) reinterpret_cast<float4 *>(base)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t q = (n4 << 2) + i; q < n; q += stride) base[q] = 0.f;
    } else {
        for (size_t q = i; q < n; q += stride) base[q] = 0.f;
    }
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void k_gemm_dual(GemmArgs p0, GemmArgs p1, DualAux aux) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem), *Bs = As + Tile<TM>::FLOATS;
    int b = blockIdx.x;
    const int n0 = p0.gx * p0.gy * p0.gz, n1 = p1.gx * p1.gy * p1.gz;
    if (b < n0) {
        gemm_block<TM, TN, WM, WN, LAYOUT_KC_VEC, LAYOUT_OC_VEC>(p0, b % p0.gx, (b / p0.gx) % p0.gy, b / (p0.gx * p0.gy), As, Bs);
        return;
    }
    b -= n0;
    if (b < n1) {
        gemm_block<TM, TN, WM, WN, LAYOUT_OC_VEC, LAYOUT_OC_VEC>(p1, b % p1.gx, (b / p1.gx) % p1.gy, b / (p1.gx * p1.gy), As, Bs);
        return;
    }
    b -= n1;
    if (b < aux.zero_blocks) {
        zero_rows_block(aux.zero_x, aux.zero_C, dyn_extent(aux.zero_r0_dev, aux.zero_r0), dyn_extent(aux.zero_r1_dev, aux.zero_r1), b, aux.zero_blocks);
        return;
    }
    b -= aux.zero_blocks;
    if (aux.cs_g) colsum_partial_block<true>(aux.cs_g, dyn_extent(aux.cs_rows_dev, aux.cs_rows), aux.cs_C, aux.cs_part, b % aux.cs_gx, b / aux.cs_gx, aux.cs_gy);
}

// results = 2 in twin mode: the second result's slabs follow the first's.
__global__ __launch_bounds__(256) void k_slab_reduce(const float *__restrict__ slab, int splits, size_t elems, int N,
                                                     float *__restrict__ C, float *__restrict__ C2, int results, long long ldc) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < elems * results; t += (size_t)gridDim.x * blockDim.x) {
        const int which = t >= elems;
        const size_t i = t - (which ? elems : 0);
        const float *src = slab + (size_t)which * splits * elems + i;
        float s = 0.f;
#pragma unroll 4
        for (int z = 0; z < splits; ++z) s += src[(size_t)z * elems];
        (which ? C2 : C)[(i / N) * ldc + (i % N)] = s;
    }
}

static GemmArgs gemm_args(const Operand &A0, const Operand &B0, int K0, const Operand &A1, const Operand &B1, int K1, int M, int N,
                          const float *bias, float *C, long long ldc, int splits, float *slab, const Twin &twin, const GemmDyn &dyn,
                          int TM, int TN) {
    GemmArgs p;
    p.A0 = A0; p.B0 = B0; p.K0 = K0; p.A1 = A1; p.B1 = B1; p.K1 = K1; p.M = M; p.N = N; p.bias = bias; p.C = C; p.ldc = ldc;
    p.slab = slab; p.twin = twin; p.m_dev = dyn.m; p.k0_dev = dyn.k0;
    const int tiles_n = (N + TN - 1) / TN;
    if (p.twin.C) p.twin.tiles_n = tiles_n;
    p.gx = (M + TM - 1) / TM; p.gy = p.twin.C ? 2 * tiles_n : tiles_n; p.gz = splits;
    return p;
}

// The two reductions behind k_gemm_dual in one launch: blocks [0, reduce_blocks) add the split-K slabs of both weight
// gradients, the rest fold the partial column sums of the bias gradient (k_colsum_final's block shape).
// (colsum_final_block: gemm_streamk_tn.h)
struct BwdFinals {
    const float *slab;
    int splits;
    size_t elems;
    int N;
    float *C, *C2;
    long long ldc;
    int reduce_blocks;
    const float *cs_part;
    int cs_splits, cs_C;
    float *cs_out;
};

__device__ __forceinline__ void bwd_finals_block(const BwdFinals &f, const int block) {
    if (block < f.reduce_blocks) {
        for (size_t t = (size_t)block * blockDim.x + threadIdx.x; t < f.elems * 2; t += (size_t)f.reduce_blocks * blockDim.x) {
            const int which = t >= f.elems;
            const size_t i = t - (which ? f.elems : 0);
            const float *src = f.slab + (size_t)which * f.splits * f.elems + i;
            float s = 0.f;
#pragma unroll 4
            for (int z = 0; z < f.splits; ++z) s += src[(size_t)z * f.elems];
            (which ? f.C2 : f.C)[(i / f.N) * f.ldc + (i % f.N)] = s;
        }
        return;
    }
    colsum_final_block(f.cs_part, f.cs_splits, f.cs_C, f.cs_out, block - f.reduce_blocks);
}

__global__ __launch_bounds__(256) void k_bwd_finals(BwdFinals f) { bwd_finals_block(f, (int)blockIdx.x); }

// The scatter of grad_agg and the two reductions behind k_gemm_dual read nothing of one another: one launch, the (few, short)
// reduction blocks first.
__global__ __launch_bounds__(256) void k_scatter_and_finals(const int *__restrict__ rowptr, const int *__restrict__ col, int n_dst,
                                                            const float *__restrict__ gagg, int C, float *__restrict__ gx,
                                                            const int *__restrict__ n_dst_dev, BwdFinals f, int finals_blocks) {
    if ((int)blockIdx.x < finals_blocks) bwd_finals_block(f, (int)blockIdx.x);
    else scatter_mean_block(rowptr, col, n_dst, gagg, C, gx, n_dst_dev, (int)blockIdx.x - finals_blocks, (int)gridDim.x - finals_blocks);
}

template <int TM, int TN, int WM, int WN, int LA, int LB>
static int launch_gemm_layout(const Operand &A0, const Operand &B0, int K0, const Operand &A1, const Operand &B1, int K1,
                              int M, int N, const float *bias, float *C, long long ldc, int splits, float *slab,
                              const Twin &twin, hipStream_t stream, const GemmDyn &dyn) {
    const size_t lds = tile_lds_bytes<TM, TN>();
    static LdsOptIn opt_in;
    if (!opt_in.done()) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm<TM, TN, WM, WN, LA, LB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        opt_in.mark();
    }
    const GemmArgs p = gemm_args(A0, B0, K0, A1, B1, K1, M, N, bias, C, ldc, splits, slab, twin, dyn, TM, TN);
    hipLaunchKernelGGL((k_gemm<TM, TN, WM, WN, LA, LB>), dim3(p.gx, p.gy, p.gz), dim3(256), lds, stream, p);
    return POPE_OK;
}

// Operand layouts are compile-time (gemm_tile.h): the three combinations SAGEConv produces get straight-line vector
// prefetch code, anything else the generic kernel.
template <int TM, int TN, int WM, int WN>
static int launch_gemm(const Operand &A0, const Operand &B0, int K0, const Operand &A1, const Operand &B1, int K1, int M,
                       int N, const float *bias, float *C, long long ldc, int splits, float *slab, const Twin &twin,
                       hipStream_t stream, const GemmDyn &dyn) {
    Layout la = pick_layout(A0, M, K0), lb = pick_layout(B0, N, K0);
    if (K1 > 0 && (pick_layout(A1, M, K1) != la || pick_layout(B1, N, K1) != lb)) la = lb = LAYOUT_GENERIC;
    if (twin.C && pick_layout(twin.B, N, K0) != lb) la = lb = LAYOUT_GENERIC;
    if (la == LAYOUT_KC_VEC && lb == LAYOUT_KC_VEC)
        return launch_gemm_layout<TM, TN, WM, WN, LAYOUT_KC_VEC, LAYOUT_KC_VEC>(A0, B0, K0, A1, B1, K1, M, N, bias, C, ldc, splits, slab, twin, stream, dyn);
    if (la == LAYOUT_OC_VEC && lb == LAYOUT_OC_VEC)
        return launch_gemm_layout<TM, TN, WM, WN, LAYOUT_OC_VEC, LAYOUT_OC_VEC>(A0, B0, K0, A1, B1, K1, M, N, bias, C, ldc, splits, slab, twin, stream, dyn);
    if (la == LAYOUT_KC_VEC && lb == LAYOUT_OC_VEC)
        return launch_gemm_layout<TM, TN, WM, WN, LAYOUT_KC_VEC, LAYOUT_OC_VEC>(A0, B0, K0, A1, B1, K1, M, N, bias, C, ldc, splits, slab, twin, stream, dyn);
    return launch_gemm_layout<TM, TN, WM, WN, LAYOUT_GENERIC, LAYOUT_GENERIC>(A0, B0, K0, A1, B1, K1, M, N, bias, C, ldc, splits, slab, twin, stream, dyn);
}

extern int g_gemm_force_tile;          // pope_debug_set(POPE_KNOB_GEMM_TILE, ...) in geodesic.hip: 0 = automatic choice
extern int g_gemm_small_tile16;        // POPE_KNOB_GEMM_SMALL_TILE16: 1 (default) = forward products too small for stream-K take 16 / 32-row whole tiles (layer 1: 14.4 us against 21)
extern int g_sage_forward_overlap;     // pope_debug_set(POPE_KNOB_SAGE_FORWARD_OVERLAP, ...): 1 (default) gather beside half of the projection, 0 one after the other
extern int g_streamk_xcd;            // POPE_KNOB_STREAMK_XCD
extern int g_gemm_tile16_buffers;     // pope_debug_set(POPE_KNOB_GEMM_TILE16_BUFFERS, ...)

static long long tiles(int M, int N, int tm, int tn) { return (long long)((M + tm - 1) / tm) * ((N + tn - 1) / tn); }

// Tile choice: the largest tile that still gives every CU about two blocks (2 x 256 = 512): co-resident blocks overlap
// one block's staging waits with another's MFMAs, and many small blocks balance better over 256 CUs than 1.2 per CU.
static int gemm(const Operand &A0, const Operand &B0, int K0, const Operand &A1, const Operand &B1, int K1, int M, int N,
                const float *bias, float *C, long long ldc, int splits, float *slab, hipStream_t stream,
                const Twin &twin = Twin{Operand{nullptr, 0, 0}, nullptr, 0}, const GemmDyn &dyn = GemmDyn{}) {
    int rc;
    const int results = twin.C ? 2 : 1;
    if (g_gemm_force_tile == 3 || (g_gemm_force_tile == 0 && tiles(M, N, 128, 256) * splits * results >= 512))
        rc = launch_gemm<128, 256, 4, 1>(A0, B0, K0, A1, B1, K1, M, N, bias, C, ldc, splits, slab, twin, stream, dyn);
    else if (g_gemm_force_tile == 2 || (g_gemm_force_tile == 0 && tiles(M, N, 64, 128) * splits * results >= 384))
        rc = launch_gemm<64, 128, 2, 2>(A0, B0, K0, A1, B1, K1, M, N, bias, C, ldc, splits, slab, twin, stream, dyn);
    else
        rc = launch_gemm<64, 64, 2, 2>(A0, B0, K0, A1, B1, K1, M, N, bias, C, ldc, splits, slab, twin, stream, dyn);
    if (rc) return rc;
    if (splits > 1)
        hipLaunchKernelGGL(k_slab_reduce, dim3(capped_grid((size_t)M * N * results, 256)), dim3(256), 0, stream, slab, splits,
                           (size_t)M * N, N, C, twin.C, results, ldc);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

// Weight gradients reduce over the n_dst rows: split that depth so the 64 x 128 tiles give ~400 blocks
// (measured: 85 us per gradient at 756 x 256 x 10 000 against 148 us with 64 x 64 tiles and fewer, longer splits).
static int weight_grad_splits(int64_t n_dst, int c_in, int c_out) {
    const long long t = 2 * tiles(c_out, c_in, 64, 128);            // both weight gradients in one twin launch
    int s = (int)((400 + t - 1) / t);
    const int max_s = (int)((n_dst + 4 * GK - 1) / (4 * GK));       // at least four LDS stages of depth per block
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- forward projection as a stream-K launch (gemm_streamk.h) ----
constexpr int SK_MAX_GRID = 256;                     // one persistent block per CU of an MI355X

static bool streamk_shape_ok(int64_t M, int32_t K0, int32_t K1, int32_t N) {
    if ((K0 & 3) || (K1 & 3) || M <= 0 || N <= 0) return false;
    const long long tiles = ((M + SK_TM - 1) / SK_TM) * ((N + SK_TN - 1) / SK_TN);
    const long long S = (K0 + SK_GK - 1) / SK_GK + (K1 + SK_GK - 1) / SK_GK;
    return tiles * S >= 4ll * SK_MAX_GRID;           // enough work (in depth-32 units) for every block to amortise its partial tiles
}

// Stage depth of the loader-wave kernels: 32 (three 40 KB buffers, two stages in flight); POPE_KNOB_GEMM_TILE = 6 selects 64 (two
// 80 KB buffers, one stage in flight, half the stage boundaries).  Measured on the layer-0 forward call (tools/gemm_fwd_ab.py,
// round 3): 89.1-93.2 us against 90.2-95.7 at a depth of 2 x 756, 72.0-73.2 against 68.6-69.0 at 2 x 532 (a depth of 532 pads to
// 576 in stages of 64, to 544 in stages of 32): the ~700 cycles a stage spends outside its MFMAs are not mostly its boundary.
static int skl_stage_depth() { return g_gemm_force_tile == 6 ? 64 : 32; }

static int device_cu_count(int *out) {
    static int cus[64];
    int dev = 0;
    POPE_HIP(hipGetDevice(&dev));
    POPE_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!cus[dev]) POPE_HIP(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
    *out = cus[dev];
    return POPE_OK;
}

// out = A0 * B0^T + A1 * B1^T + bias through k_gemm_streamk + k_streamk_fixup; *used = false if the operands do not
// qualify (alignment, size, slab too small) and nothing was launched.
static int gemm_streamk(const float *A0, const float *B0, int K0, const float *A1, const float *B1, int K1, long long lda, long long ldb,
                        int M, int N, const float *bias, float *C, long long ldc, void *slab, size_t slab_bytes, hipStream_t stream,
                        bool *used, const int *m_dev = nullptr) {
    *used = false;
    if (g_gemm_force_tile != 0 && g_gemm_force_tile < 4) return POPE_OK;
    if (!((g_gemm_force_tile >= 4 && g_gemm_force_tile != 6 && g_gemm_force_tile != 7) || streamk_shape_ok(M, K0, K1, N))) return POPE_OK;
    if ((long long)M * lda * 4 >= (1ll << 32) || (long long)N * ldb * 4 >= (1ll << 32)) return POPE_OK;   // the loaders' 32-bit byte offsets
    if (!sk_operand_ok(A0, lda, K0) || !sk_operand_ok(B0, ldb, K0) || (K1 > 0 && (!sk_operand_ok(A1, lda, K1) || !sk_operand_ok(B1, ldb, K1))))
        return POPE_OK;
    int cus = 0, rc;
    if ((rc = device_cu_count(&cus))) return rc;
    SkArgs a;
    a.p[0] = SkProduct{A0, B0, lda, ldb, K0};
    a.p[1] = SkProduct{K1 > 0 ? A1 : A0, K1 > 0 ? B1 : B0, lda, ldb, K1};
    a.M = M; a.N = N; a.bias = bias; a.C = C; a.ldc = ldc; a.slab = (float *)slab;
    a.tiles_m = (M + SK_TM - 1) / SK_TM; a.tiles_n = (N + SK_TN - 1) / SK_TN;
    const bool diag = g_gemm_force_tile == 4 || g_gemm_force_tile == 5 || g_gemm_force_tile == 8 || g_gemm_force_tile == 9;   // depth-32 kernels without loader waves
    const int gk = diag ? SK_GK : skl_stage_depth();
    a.S0 = (K0 + gk - 1) / gk; a.S1 = (K1 + gk - 1) / gk;
    a.m_dev = m_dev;
    const long long T = (long long)a.tiles_m * a.tiles_n * (a.S0 + a.S1);
    long long grid = cus < SK_MAX_GRID ? cus : SK_MAX_GRID;
    if (grid > T) grid = T;
    if (!slab || slab_bytes < sk_slab_bytes((int)grid)) return POPE_OK;
    static LdsOptIn opt_in;
    static const float *zero_page[64];
    int dev = 0;
    POPE_HIP(hipGetDevice(&dev));
    if (!opt_in.done()) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_streamk_ld<32>, hipFuncAttributeMaxDynamicSharedMemorySize, SkStage<32>::LDS_BYTES));
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_streamk_ld<64>, hipFuncAttributeMaxDynamicSharedMemorySize, SkStage<64>::LDS_BYTES));
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_streamk<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES));
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_streamk<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES));
#ifdef POPE_STAMP
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_streamk<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES));
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_streamk<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES));
#endif
        opt_in.mark();
    }
    if (!zero_page[dev]) POPE_HIP(hipGetSymbolAddress((void **)&zero_page[dev], HIP_SYMBOL(g_sk_zero)));
    a.zero = zero_page[dev];
    if (g_gemm_force_tile == 4)                  // A/B: every wave stages its own share and computes (one wave per SIMD)
        hipLaunchKernelGGL((k_gemm_streamk<4, 0>), dim3((unsigned)grid), dim3(256), SK_LDS_BYTES, stream, a);
    else if (g_gemm_force_tile == 5)             // A/B: the same with two waves per SIMD
        hipLaunchKernelGGL((k_gemm_streamk<8, 0>), dim3((unsigned)grid), dim3(512), SK_LDS_BYTES, stream, a);
#ifdef POPE_STAMP
    else if (g_gemm_force_tile == 8)             // diagnostic: no DMA in the loop
        hipLaunchKernelGGL((k_gemm_streamk<4, 3>), dim3((unsigned)grid), dim3(256), SK_LDS_BYTES, stream, a);
    else if (g_gemm_force_tile == 9)             // diagnostic: no MFMA
        hipLaunchKernelGGL((k_gemm_streamk<4, 4>), dim3((unsigned)grid), dim3(256), SK_LDS_BYTES, stream, a);
#endif
    else if (gk == 32)                           // 4 MFMA waves + 4 loader waves, stages of 32 (rounds 1-2)
        hipLaunchKernelGGL(k_gemm_streamk_ld<32>, dim3((unsigned)grid), dim3(SKL_THREADS), SkStage<32>::LDS_BYTES, stream, a);
    else                                         // default: the same with stages of 64 in two 80 KB buffers
        hipLaunchKernelGGL(k_gemm_streamk_ld<64>, dim3((unsigned)grid), dim3(SKL_THREADS), SkStage<64>::LDS_BYTES, stream, a);
    hipLaunchKernelGGL(k_streamk_fixup, dim3(a.tiles_m * a.tiles_n, SK_FIX_PARTS), dim3(256), 0, stream, a, (int)grid);
    POPE_HIP(hipGetLastError());
    *used = true;
    return POPE_OK;
}

// ---- forward projection as whole tiles fitted to the chip (gemm_tile16.h): no partial tiles, no fix-up ----
template <int RB, int NBUF>
static int launch_tile16_as(const T16Args &a, int grid, hipStream_t stream) {
    static LdsOptIn opt_in;
    constexpr int lds = T16Shape<RB, NBUF>::LDS_BYTES;
    if (!opt_in.done()) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_tile16<RB, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        opt_in.mark();
    }
    hipLaunchKernelGGL((k_gemm_tile16<RB, NBUF>), dim3((unsigned)grid), dim3(T16_THREADS), lds, stream, a);
    return POPE_OK;
}

// POPE_KNOB_GEMM_TILE16_BUFFERS: 3 or 4 stage buffers (gemm_tile16.h: one or two stage times to hide a request).
template <int RB>
static int launch_tile16(const T16Args &a, int grid, hipStream_t stream) {
    return g_gemm_tile16_buffers == 4 ? launch_tile16_as<RB, 4>(a, grid, stream) : launch_tile16_as<RB, 3>(a, grid, stream);
}

// A whole-tile product ready to launch: arguments, tile height and grid; ok = false if the shape does not fit the chip well
// enough or the operands do not qualify.
struct T16Plan {
    T16Args a;
    int rb = 0, grid = 0;
    bool ok = false;
};

static int t16_plan(const float *A0, const float *B0, int K0, const float *A1, const float *B1, int K1, long long lda, long long ldb, int M, int N,
                    const float *bias, float *C, long long ldc, const int *m_dev, T16Plan *plan, long long a0_rows = 0, bool small = false) {
    plan->ok = false;
    if (g_gemm_force_tile != 0) return POPE_OK;                      // any forced variant: not this kernel (7 = "stream-K as in round 2")
    if ((K0 & 3) || (K1 & 3) || M <= 0 || N <= 0 || K0 <= 0) return POPE_OK;
    if ((long long)std::max<long long>(M, a0_rows) * lda * 4 >= (1ll << 32) || (long long)N * ldb * 4 >= (1ll << 32)) return POPE_OK;
    if (!sk_operand_ok(A0, lda, K0) || !sk_operand_ok(B0, ldb, K0) || (K1 > 0 && (!sk_operand_ok(A1, lda, K1) || !sk_operand_ok(B1, ldb, K1))))
        return POPE_OK;
    int cus = 0, rc;
    if ((rc = device_cu_count(&cus))) return rc;
    // small = a product that cannot fill the chip whatever the tile: the shortest tiles (16 or 32 rows) that give the most blocks
    const int rb = small ? t16_pick_rb(M, N, cus, 0.3, 1) : t16_pick_rb(M, N, cus);
    if (rb == 0) return POPE_OK;
    static const float *zero_page[64];
    int dev = 0;
    POPE_HIP(hipGetDevice(&dev));
    if (!zero_page[dev]) POPE_HIP(hipGetSymbolAddress((void **)&zero_page[dev], HIP_SYMBOL(g_sk_zero)));
    T16Args &a = plan->a;
    a.p[0] = SkProduct{A0, B0, lda, ldb, K0};
    a.p[1] = SkProduct{K1 > 0 ? A1 : A0, K1 > 0 ? B1 : B0, lda, ldb, K1};
    a.M = M; a.N = N; a.bias = bias; a.C = C; a.ldc = ldc; a.zero = zero_page[dev]; a.m_dev = m_dev;
    a.rows = nullptr; a.accumulate = 0; a.stat_a = a.stat_b = nullptr;
    a.tiles_m = (M + 16 * rb - 1) / (16 * rb); a.tiles_n = (N + T16_TN - 1) / T16_TN;
    a.S0 = (K0 + T16_GK - 1) / T16_GK; a.S1 = (K1 + T16_GK - 1) / T16_GK;
    plan->grid = a.tiles_n == 2 ? (a.tiles_m + 7) / 8 * 16 : a.tiles_m * a.tiles_n;
    if (m_dev) plan->grid = std::min(plan->grid, a.tiles_n == 2 ? (cus + 15) / 16 * 16 : cus);   // capacity rows: one block per CU walks the true tiles
    plan->rb = rb;
    plan->ok = true;
    return POPE_OK;
}

static int t16_launch(const T16Plan &plan, hipStream_t stream) {
    int rc;
    switch (plan.rb) {
    case 1: rc = launch_tile16<1>(plan.a, plan.grid, stream); break;
    case 2: rc = launch_tile16<2>(plan.a, plan.grid, stream); break;
    case 3: rc = launch_tile16<3>(plan.a, plan.grid, stream); break;
    case 4: rc = launch_tile16<4>(plan.a, plan.grid, stream); break;
    case 5: rc = launch_tile16<5>(plan.a, plan.grid, stream); break;
    case 6: rc = launch_tile16<6>(plan.a, plan.grid, stream); break;
    case 7: rc = launch_tile16<7>(plan.a, plan.grid, stream); break;
    default: rc = launch_tile16<8>(plan.a, plan.grid, stream); break;
    }
    if (rc) return rc;
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

// A caller's wish for the first stage of the BatchNorm statistics of the layer's output (main.py:207) out of the projection's epilogue:
// pa / pb [parts_cap, c_out] doubles.  Filled in by the whole-tile kernels only; parts = 0 says "not produced" (the caller then runs
// the statistics pass of sage_bn_relu_dropout_forward as before).
struct BnStatsOut {
    double *pa = nullptr, *pb = nullptr;
    int parts_cap = 0;
    int parts = 0, rows_per_part = 0;      // out
};

static bool stats_fit(const BnStatsOut *st, const T16Plan &plan) { return st && st->pa && st->pb && plan.a.tiles_m <= st->parts_cap; }

// *used = false if nothing was launched (see T16Plan).
static int gemm_tile16(const float *A0, const float *B0, int K0, const float *A1, const float *B1, int K1, long long lda, long long ldb,
                       int M, int N, const float *bias, float *C, long long ldc, hipStream_t stream, bool *used, const int *m_dev = nullptr,
                       BnStatsOut *stats = nullptr) {
    *used = false;
    const bool small = !streamk_shape_ok(M, K0, K1, N);
    if (small && (g_gemm_small_tile16 == 0 || (long long)M * N < 64 * 1024)) return POPE_OK;   // (tiny products stay on the plain tile kernel)
    T16Plan plan;
    int rc = t16_plan(A0, B0, K0, A1, B1, K1, lda, ldb, M, N, bias, C, ldc, m_dev, &plan, 0, small);
    if (rc || !plan.ok) return rc;
    if (stats_fit(stats, plan)) {
        plan.a.stat_a = stats->pa; plan.a.stat_b = stats->pb;
        stats->parts = plan.a.tiles_m; stats->rows_per_part = 16 * plan.rb;
    }
    if ((rc = t16_launch(plan, stream))) return rc;
    *used = true;
    return POPE_OK;
}

// ---- layer forward as TWO launches that overlap the gather with half of the projection (round 3) ----
// The gather is HBM-bound (64 us inside the step), the projection MFMA-bound (78 us), and they ran one after the other.
// out = x_dst W_r^T + b does not need the gather: launch 1 carries that product (reading the destination rows straight from
// the source / feature matrix, through n_id in indexed mode) in its first blocks -- one per CU, dispatched first -- and the
// gather in the remaining blocks, which share the CUs with them; launch 2 adds agg W_l^T into out.  No block waits for
// another block.  Every block of launch 1 carries the GEMM role's LDS reservation, so only one or two gather blocks fit a
// CU: the gather role keeps twelve 16-byte loads in flight per lane (k_gather_mean: four) to make up for it.
// (Measured and not kept: x_dst written by the GEMM role out of LDS instead of read + written by the gather role -- by the consumer
//  waves (their copy doubled the fused kernel's spills) or by the loader waves: the launch took 84 us instead of 79 and the step
//  0.332 ms instead of 0.313 either way (128-byte pieces of 3 KB rows, stage by stage, are poor stores); and no x_dst at all, the weight-gradient
//  kernel reading the rows through n_id (sage_conv_backward_indexed): its 1 KB random reads cost 29 us more than the copy.)
struct GatherArgs {
    const int *rowptr, *col;
    int n_dst;
    const float *x;
    int C;
    float *agg;
    const long long *n_id;
    float *x_dst;
    const int *n_dst_dev;
};

// Only one or two of these blocks fit a CU beside the GEMM role's LDS, so what k_gather_mean hides behind occupancy is
// hidden here by a pipeline over the rows of a wave: while row j's data is in flight the wave loads rowptr of row j + 3, the
// neighbour ids of row j + 2 (one coalesced load, a lane per neighbour) and their n_id entries for row j + 1; the data
// phase takes its row addresses from registers (shuffles) and keeps twelve 16-byte loads in flight per lane.  Sums are
// formed in k_gather_mean's order (a short last group re-reads the last neighbour with weight 0: v * 1 and s + v * 0 are
// exact), so agg is bit for bit the same.  Rows with more than 64 neighbours take the plain loop.  (s_setprio 3 for this role:
// measured, no difference.)
__device__ __forceinline__ void gather_role(const GatherArgs &g, const int block, const int nblocks) {
    const int lane = threadIdx.x & 63;
    const int wave = block * (T16_THREADS / 64) + (threadIdx.x >> 6), nwaves = nblocks * (T16_THREADS / 64);
    const int n_dst = dyn_extent(g.n_dst_dev, g.n_dst);
    const int C = g.C, C4 = C >> 2;
    // pipeline registers: row r3 has its rowptr pair requested, r2 its neighbour ids, r1 their source rows, r0 is being summed
    int b1 = 0, e1 = 0, b2 = 0, e2 = 0, c1 = 0;
    long long id0 = 0, own0 = 0;
    int b0 = 0, e0 = 0;
    auto rowptr_pair = [&](int i, int &b, int &e) {
        if (i < n_dst) { b = g.rowptr[i]; e = g.rowptr[i + 1]; } else { b = e = 0; }
    };
    auto neighbour = [&](int b, int e) { return b + lane < e && lane < 64 ? g.col[b + lane] : 0; };
    auto source = [&](int c) { return g.n_id ? g.n_id[c] : (long long)c; };
    // prologue: fill the pipeline for the wave's first three rows
    rowptr_pair(wave, b0, e0);
    rowptr_pair(wave + nwaves, b1, e1);
    rowptr_pair(wave + 2 * nwaves, b2, e2);
    {
        const int c0 = neighbour(b0, e0);
        c1 = neighbour(b1, e1);
        id0 = source(c0);
        own0 = g.x_dst && wave < n_dst ? g.n_id[wave] : 0;
    }
    for (int i = wave; i < n_dst; i += nwaves) {
        // requests for the rows behind this one (nothing here depends on a load of this iteration)
        int b3, e3;
        rowptr_pair(i + 3 * nwaves, b3, e3);
        const int c2 = neighbour(b2, e2);
        const long long id1 = source(c1);
        const long long own1 = g.x_dst && i + nwaves < n_dst ? g.n_id[i + nwaves] : 0;
        // the data of row i
        const int beg = b0, end = e0, deg = end - beg;
        const float inv = deg > 0 ? 1.0f / (float)deg : 0.0f;
        for (int q0 = 0; q0 < C4; q0 += 192) {
            int q[3];
            bool ok[3];
            float4 s[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                q[k] = q0 + 64 * k + lane;
                ok[k] = q[k] < C4;
                s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            for (int p = 0; p < deg; p += 4) {                             // four neighbour rows x three pieces in flight
                long long row[4];
                float w[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int pp = min(p + t, deg - 1);
                    w[t] = p + t < deg ? 1.0f : 0.0f;
                    if (deg <= 64) {
                        const int lo = __shfl((int)(unsigned)(id0 & 0xffffffffll), pp), hi = __shfl((int)(id0 >> 32), pp);
                        row[t] = ((long long)hi << 32) | (unsigned)lo;
                    } else {
                        row[t] = source(g.col[beg + pp]);                  // a long row: ids straight from memory
                    }
                }
                float4 v[4][3];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 *r = reinterpret_cast<const float4 *>(g.x + (size_t)row[t] * C);
#pragma unroll
                    for (int k = 0; k < 3; ++k) v[t][k] = r[ok[k] ? q[k] : 0];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        s[k].x += v[t][k].x * w[t]; s[k].y += v[t][k].y * w[t]; s[k].z += v[t][k].z * w[t]; s[k].w += v[t][k].w * w[t];
                    }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (!ok[k]) continue;
                s[k].x *= inv; s[k].y *= inv; s[k].z *= inv; s[k].w *= inv;
                reinterpret_cast<float4 *>(g.agg + (size_t)i * C)[q[k]] = s[k];
            }
            if (g.x_dst) {                                                 // the destination's own row: requested after the neighbour loop, so that
                const float4 *own = reinterpret_cast<const float4 *>(g.x + (size_t)own0 * C);      // it does not hold twelve registers across it
                float4 *dst = reinterpret_cast<float4 *>(g.x_dst + (size_t)i * C);
                const float4 o0 = own[ok[0] ? q[0] : 0], o1 = own[ok[1] ? q[1] : 0], o2 = own[ok[2] ? q[2] : 0];
                if (ok[0]) dst[q[0]] = o0;
                if (ok[1]) dst[q[1]] = o1;
                if (ok[2]) dst[q[2]] = o2;
            }
        }
        // advance the pipeline
        b0 = b1; e0 = e1; id0 = id1; own0 = own1;
        b1 = b2; e1 = e2; c1 = c2;
        b2 = b3; e2 = e3;
    }
}

// (registers are allotted per kernel, not per role: at the GEMM role's 138 a gather block's two extra waves per SIMD would not fit
//  beside a GEMM block -- 4 x 144 > 512 -- and the roles ran one after the other: 62 + 45 us.  Hence four waves per SIMD.)
template <int RB>
__global__ __launch_bounds__(T16_THREADS, 4) void k_gather_beside_gemm(T16Args a, GatherArgs g, int gemm_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < gemm_blocks) t16_block_loop<RB>(a, smem, (int)blockIdx.x, gemm_blocks);
    else gather_role(g, (int)blockIdx.x - gemm_blocks, (int)gridDim.x - gemm_blocks);
}

// (Round 4 built the layer as ONE launch -- the projection's tiles waiting, inside the launch, for their rows of the aggregate: 124-127 us
//  against 114-119 for these two launches; removed in round 5.)
template <int RB>
static int launch_gather_beside_gemm(const T16Args &a, const GatherArgs &g, int gemm_blocks, int gather_blocks, hipStream_t stream) {
    static LdsOptIn opt_in;
    if (!opt_in.done()) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_gather_beside_gemm<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, T16Shape<RB>::LDS_BYTES));
        opt_in.mark();
    }
    hipLaunchKernelGGL((k_gather_beside_gemm<RB>), dim3((unsigned)(gemm_blocks + gather_blocks)), dim3(T16_THREADS), T16Shape<RB>::LDS_BYTES, stream, a, g,
                       gemm_blocks);
    return POPE_OK;
}

// x: the matrix the rows are gathered from (the block's own sources, or the whole feature matrix with n_id); x_rows its
// row count.  *used = false: nothing launched, the caller runs gather + one-pass projection as before.
static int forward_overlapped(const int32_t *rowptr, const int32_t *col, int64_t n_dst, const float *x, int64_t x_rows, int32_t c_in,
                              const long long *n_id, float *x_dst, float *agg, const float *w_l, const float *b_l, const float *w_r,
                              int32_t c_out, float *out, const int32_t *dims, hipStream_t stream, bool *used, BnStatsOut *stats = nullptr) {
    *used = false;
    if (g_sage_forward_overlap == 0) return POPE_OK;
    if (!streamk_shape_ok(n_dst, c_in, c_in, c_out) || !aligned16(x) || !aligned16(agg) || (x_dst && !aligned16(x_dst))) return POPE_OK;
    T16Plan first, second;
    int rc = t16_plan(x, w_r, c_in, nullptr, nullptr, 0, c_in, c_in, (int)n_dst, c_out, b_l, out, c_out, dims, &first, x_rows);
    if (rc || !first.ok) return rc;
    if ((rc = t16_plan(agg, w_l, c_in, nullptr, nullptr, 0, c_in, c_in, (int)n_dst, c_out, nullptr, out, c_out, dims, &second))) return rc;
    if (!second.ok || second.rb != first.rb) return POPE_OK;
    if (first.rb > 5) return POPE_OK;                               // taller tiles do not fit the fused kernel's 128 registers (they spill 140-430 bytes per lane)
    first.a.rows = n_id;                                            // nullptr: destination i is row i of x
    second.a.accumulate = 1;
    if (stats_fit(stats, second)) {                                 // the launch that leaves the layer's final values in `out`
        second.a.stat_a = stats->pa; second.a.stat_b = stats->pb;
        stats->parts = second.a.tiles_m; stats->rows_per_part = 16 * second.rb;
    }
    int cus = 0;
    if ((rc = device_cu_count(&cus))) return rc;
    const int gemm_blocks = first.a.tiles_n == 2 ? std::min(first.grid, (cus + 15) / 16 * 16) : std::min(first.grid, cus);
    const int gather_blocks = cus;                                  // one per CU, beside its GEMM block (2-6 per CU measured: 2-10 us slower -- the row pipeline wants rows)
    const GatherArgs g{rowptr, col, (int)n_dst, x, c_in, agg, n_id, x_dst, dims};
    switch (first.rb) {
    case 3: rc = launch_gather_beside_gemm<3>(first.a, g, gemm_blocks, gather_blocks, stream); break;
    case 4: rc = launch_gather_beside_gemm<4>(first.a, g, gemm_blocks, gather_blocks, stream); break;
    default: rc = launch_gather_beside_gemm<5>(first.a, g, gemm_blocks, gather_blocks, stream); break;
    }
    if (rc) return rc;
    POPE_HIP(hipGetLastError());
    if ((rc = t16_launch(second, stream))) return rc;
    *used = true;
    return POPE_OK;
}

// ---- both weight gradients as one stream-K launch (gemm_streamk_tn.h) ----
static bool streamk_tn_shape_ok(int64_t depth, int32_t M, int32_t Nb) {
    if ((M & 3) || (Nb & 3) || M < 4 || Nb < 4 || depth <= 0) return false;
    const long long tiles = 2ll * ((M + SK_TM - 1) / SK_TM) * ((Nb + SK_TN - 1) / SK_TN);
    return tiles * ((depth + SK_GK - 1) / SK_GK) >= 4ll * SK_MAX_GRID;
}

// C0 = G^T * B0, C1 = G^T * B1 (G [depth, M], B_q [depth, Nb], C_q [M, Nb]); *used = false if the operands do not qualify.
static int gemm_streamk_tn(const float *G, const float *B0, const float *B1, int64_t depth, int M, int Nb, float *C0, float *C1,
                           void *slab, size_t slab_bytes, hipStream_t stream, bool *used, const int *depth_dev = nullptr,
                           float *cs_part = nullptr, float *cs_out = nullptr, bool cs_vec = false, const long long *rows1 = nullptr) {
    *used = false;
    if (g_gemm_force_tile != 0 && g_gemm_force_tile < 4) return POPE_OK;
    if (!streamk_tn_shape_ok(depth, M, Nb)) return POPE_OK;
    if (!aligned16(G) || !aligned16(B0) || !aligned16(B1) || depth >= INT32_MAX) return POPE_OK;
    int cus = 0, rc;
    if ((rc = device_cu_count(&cus))) return rc;
    SkTnArgs a;
    a.G = G; a.ldg = M; a.B[0] = B0; a.B[1] = B1; a.ldb = Nb; a.C[0] = C0; a.C[1] = C1; a.ldc = Nb;
    a.M = M; a.Nb = Nb; a.depth = (int)depth; a.slab = (float *)slab;
    const int gk = skl_stage_depth();
    a.tiles_m = (M + SK_TM - 1) / SK_TM; a.tiles_nb = (Nb + SK_TN - 1) / SK_TN; a.S = (int)((depth + gk - 1) / gk);
    a.depth_dev = depth_dev;
    a.rows1 = rows1;
    // the bias gradient = column sums of G: its partial sums are a launch of their own in front of the GEMM (they read G and
    // nothing else), its final stage rides in the fix-up launch behind it
    a.cs_part = cs_part; a.cs_out = cs_out; a.cs_splits = COLSUM_SPLITS; a.cs_C = M;
    const int cs_C = M;
    const unsigned fix_blocks = 2u * a.tiles_m * a.tiles_nb + (cs_part ? (unsigned)(cs_C + 15) / 16 : 0u);
    const long long T = 2ll * a.tiles_m * a.tiles_nb * a.S;
    long long grid = cus < SK_MAX_GRID ? cus : SK_MAX_GRID;
    if (grid > T) grid = T;
    if (!slab || slab_bytes < sk_slab_bytes((int)grid)) return POPE_OK;
    a.xcd = g_streamk_xcd && grid % 8 == 0 && (grid / 8) % a.tiles_m == 0 && grid / a.tiles_m <= 2ll * a.tiles_nb * a.S;
    static LdsOptIn opt_in;
    static const float *zero_page[64];
    int dev = 0;
    POPE_HIP(hipGetDevice(&dev));
    if (!opt_in.done()) {
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_streamk_tn<32>, hipFuncAttributeMaxDynamicSharedMemorySize, SkStage<32>::LDS_BYTES));
        POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_streamk_tn<64>, hipFuncAttributeMaxDynamicSharedMemorySize, SkStage<64>::LDS_BYTES));
        opt_in.mark();
    }
    if (!zero_page[dev]) POPE_HIP(hipGetSymbolAddress((void **)&zero_page[dev], HIP_SYMBOL(g_sk_zero)));
    a.zero = zero_page[dev];
    if (cs_part) {
        if (cs_vec)
            hipLaunchKernelGGL(k_colsum_partial<true>, dim3((M + 255) / 256, COLSUM_SPLITS), dim3(256), 0, stream, G, (int)depth, M, cs_part, depth_dev);
        else
            hipLaunchKernelGGL(k_colsum_partial<false>, dim3((M + 63) / 64, COLSUM_SPLITS), dim3(256), 0, stream, G, (int)depth, M, cs_part, depth_dev);
    }
    if (gk == 32) {
        hipLaunchKernelGGL(k_gemm_streamk_tn<32>, dim3((unsigned)grid), dim3(SKL_THREADS), SkStage<32>::LDS_BYTES, stream, a);
        hipLaunchKernelGGL(k_streamk_tn_fixup<32>, dim3(fix_blocks, SK_TN_FIX_PARTS), dim3(256), 0, stream, a, (int)grid);
    } else {
        hipLaunchKernelGGL(k_gemm_streamk_tn<64>, dim3((unsigned)grid), dim3(SKL_THREADS), SkStage<64>::LDS_BYTES, stream, a);
        hipLaunchKernelGGL(k_streamk_tn_fixup<64>, dim3(fix_blocks, SK_TN_FIX_PARTS), dim3(256), 0, stream, a, (int)grid);
    }
    POPE_HIP(hipGetLastError());
    *used = true;
    return POPE_OK;
}

}  // namespace pope

using namespace pope;

#ifdef POPE_STAMP
extern "C" int pope_debug_read_streamk_stamps(unsigned long long *host, int count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sk_stamps), (size_t)count * sizeof(unsigned long long));
}
extern "C" int pope_debug_read_t16_trace(unsigned long long *host, int count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_t16_trace), (size_t)count * sizeof(unsigned long long));
}
extern "C" int pope_debug_read_gemm_stamps(unsigned long long *host, int count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_stamps), (size_t)count * sizeof(unsigned long long));
}
#endif

// dst[i, :] = x[rows[i], :], i < n (a device extent may shorten n): the destination rows of a sampled block as a matrix of
// their own, for the kernel paths that cannot read them through n_id.
__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ x, const long long *__restrict__ rows, int n, int C,
                                                     float *__restrict__ dst, const int *__restrict__ n_dev) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    n = dyn_extent(n_dev, n);
    for (int i = wave; i < n; i += nwaves) {
        const float *src = x + (size_t)rows[i] * C;
        float *d = dst + (size_t)i * C;
        if ((C & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
            for (int q = lane; q < (C >> 2); q += 64) reinterpret_cast<float4 *>(d)[q] = reinterpret_cast<const float4 *>(src)[q];
        } else {
            for (int c = lane; c < C; c += 64) d[c] = src[c];
        }
    }
}

static size_t rows_matrix_bytes(int64_t n_dst, int32_t c_in) { return align_up((size_t)n_dst * c_in * sizeof(float), 256); }

extern "C" size_t sage_conv_scratch_bytes(int64_t n_src, int64_t n_dst, int64_t nnz, int32_t c_in, int32_t c_out) {
    (void)n_src; (void)nnz;
    if (n_dst <= 0 || c_in <= 0 || c_out <= 0) return 0;
    // backward: grad_agg [n_dst, c_in] | split-K / stream-K slabs of the two weight gradients (one twin launch) | the partial
    // column sums of the bias gradient (a region of its own: they are computed beside the weight gradients)
    const size_t gagg = align_up((size_t)n_dst * c_in * sizeof(float), 256);
    size_t slabs = 2 * (size_t)weight_grad_splits(n_dst, c_in, c_out) * c_out * c_in * sizeof(float);
    if (streamk_tn_shape_ok(n_dst, c_out, c_in) && slabs < sk_slab_bytes(SK_MAX_GRID)) slabs = sk_slab_bytes(SK_MAX_GRID);
    slabs = align_up(slabs, 256);
    const size_t colsum = align_up((size_t)COLSUM_SPLITS * c_out * sizeof(float), 256);
    return gagg + slabs + colsum;
}

extern "C" size_t sage_conv_forward_scratch_bytes(int64_t n_dst, int32_t c_in, int32_t c_out) {
    if (n_dst <= 0 || c_in <= 0 || c_out <= 0) return 0;
    return streamk_shape_ok(n_dst, c_in, c_in, c_out) ? sk_slab_bytes(SK_MAX_GRID) : 0;
}

// The indexed entry points may be called without an x_dst matrix (the destination rows are then read through n_id); the
// kernel paths that cannot do that build the matrix in the tail of the scratch buffer.
extern "C" size_t sage_conv_forward_indexed_scratch_bytes(int64_t n_dst, int32_t c_in, int32_t c_out) {
    if (n_dst <= 0 || c_in <= 0 || c_out <= 0) return 0;
    return align_up(sage_conv_forward_scratch_bytes(n_dst, c_in, c_out), 256) + rows_matrix_bytes(n_dst, c_in);
}

extern "C" size_t sage_conv_backward_indexed_scratch_bytes(int64_t n_src, int64_t n_dst, int64_t nnz, int32_t c_in, int32_t c_out) {
    if (n_dst <= 0 || c_in <= 0 || c_out <= 0) return 0;
    return sage_conv_scratch_bytes(n_src, n_dst, nnz, c_in, c_out) + rows_matrix_bytes(n_dst, c_in);
}

static void enqueue_gather_mean(const int32_t *rowptr, const int32_t *col, int64_t n_dst, const float *x_src, int32_t c_in,
                                float *agg, hipStream_t stream, const int64_t *n_id = nullptr, float *x_dst = nullptr,
                                const int32_t *n_dst_dev = nullptr) {
    dim3 grid(capped_grid((size_t)n_dst * 64, 256));
    if ((c_in & 3) == 0 && aligned16(x_src) && aligned16(agg) && (!x_dst || aligned16(x_dst)))
        hipLaunchKernelGGL(k_gather_mean<true>, grid, dim3(256), 0, stream, rowptr, col, (int)n_dst, x_src, c_in, agg,
                           (const long long *)n_id, x_dst, n_dst_dev);
    else
        hipLaunchKernelGGL(k_gather_mean<false>, grid, dim3(256), 0, stream, rowptr, col, (int)n_dst, x_src, c_in, agg,
                           (const long long *)n_id, x_dst, n_dst_dev);
}

extern "C" int sage_gather_mean(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz,
                                const float *x_src, int32_t c_in, float *agg, void *stream_) {
    clear_error();
    POPE_REQUIRE(rowptr && (col || nnz == 0) && x_src && agg, "sage_gather_mean: null pointer");
    POPE_REQUIRE(n_dst > 0 && n_src > 0 && n_src < INT32_MAX && nnz >= 0 && nnz < INT32_MAX && c_in > 0, "sage_gather_mean: bad size");
    enqueue_gather_mean(rowptr, col, n_dst, x_src, c_in, agg, (hipStream_t)stream_);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

// `dims` (device int32 [4] = {n_dst, n_src, nnz, 0}, or NULL): see include/graphpope_hip.h, "Device extents".
static int conv_forward_impl(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz,
                             const float *x_src, int32_t c_in, const float *w_l, const float *b_l, const float *w_r,
                             int32_t c_out, float *agg, float *out, void *scratch, size_t scratch_bytes, const int32_t *dims,
                             BnStatsOut *stats, hipStream_t stream) {
    POPE_REQUIRE(rowptr && (col || nnz == 0) && x_src && w_l && w_r && agg && out, "sage_conv_forward: null pointer");
    POPE_REQUIRE(n_dst > 0 && n_dst <= n_src && n_src < INT32_MAX && nnz >= 0 && nnz < INT32_MAX && c_in > 0 && c_out > 0,
                 "sage_conv_forward: bad size (destinations must be the first n_dst sources)");
    bool used = false;
    int rc = forward_overlapped(rowptr, col, n_dst, x_src, n_src, c_in, nullptr, nullptr, agg, w_l, b_l, w_r, c_out, out, dims, stream, &used, stats);
    if (rc || used) return rc;
    enqueue_gather_mean(rowptr, col, n_dst, x_src, c_in, agg, stream, nullptr, nullptr, dims);
    // out = agg * w_l^T + b_l + x_dst * w_r^T in one pass
    if ((rc = gemm_tile16(agg, w_l, c_in, x_src, w_r, c_in, c_in, c_in, (int)n_dst, c_out, b_l, out, c_out, stream, &used, dims, stats))) return rc;
    if (used) return POPE_OK;
    rc = gemm_streamk(agg, w_l, c_in, x_src, w_r, c_in, c_in, c_in, (int)n_dst, c_out, b_l, out, c_out, scratch, scratch_bytes, stream, &used, dims);
    if (rc || used) return rc;
    const Operand A0{agg, c_in, 1}, B0{w_l, c_in, 1}, A1{x_src, c_in, 1}, B1{w_r, c_in, 1};
    GemmDyn dyn;
    dyn.m = dims;
    return gemm(A0, B0, c_in, A1, B1, c_in, (int)n_dst, c_out, b_l, out, c_out, 1, nullptr, stream, Twin{Operand{nullptr, 0, 0}, nullptr, 0}, dyn);
}

// The same layer on rows of the resident feature matrix: source j of the block is feats[n_id[j]].  Replaces
// Batch.x = data.x[n_id] (main.py:118-123) + the layer: x[n_id] is never materialised, only the n_dst destination rows
// (x_dst, needed again by the backward pass) are.
static int conv_forward_indexed_impl(const int32_t *rowptr, const int32_t *col, const int64_t *n_id, int64_t n_src, int64_t n_dst,
                                     int64_t nnz, const float *feats, int64_t n_rows, int32_t c_in, const float *w_l,
                                     const float *b_l, const float *w_r, int32_t c_out, float *agg, float *x_dst, float *out,
                                     void *scratch, size_t scratch_bytes, const int32_t *dims, BnStatsOut *stats, hipStream_t stream) {
    POPE_REQUIRE(rowptr && (col || nnz == 0) && n_id && feats && w_l && w_r && agg && out, "sage_conv_forward_indexed: null pointer");
    POPE_REQUIRE(n_dst > 0 && n_dst <= n_src && n_src < INT32_MAX && n_rows > 0 && nnz >= 0 && nnz < INT32_MAX && c_in > 0 && c_out > 0,
                 "sage_conv_forward_indexed: bad size (destinations must be the first n_dst entries of n_id)");
    bool used = false;
    int rc = forward_overlapped(rowptr, col, n_dst, feats, n_rows, c_in, (const long long *)n_id, x_dst, agg, w_l, b_l, w_r, c_out, out, dims, stream,
                                &used, stats);
    if (rc || used) return rc;
    if (!x_dst) {                                                    // no matrix of the destination rows from the caller: the kernels below want one
        const size_t at = align_up(sage_conv_forward_scratch_bytes(n_dst, c_in, c_out), 256);
        if (!scratch || scratch_bytes < at + rows_matrix_bytes(n_dst, c_in)) {
            set_error("sage_conv_forward_indexed: without x_dst the scratch must hold sage_conv_forward_indexed_scratch_bytes (%zu), got %zu",
                      at + rows_matrix_bytes(n_dst, c_in), scratch_bytes);
            return POPE_ERR_WORKSPACE;
        }
        x_dst = (float *)((char *)scratch + at);
        scratch_bytes = at;
    }
    enqueue_gather_mean(rowptr, col, n_dst, feats, c_in, agg, stream, n_id, x_dst, dims);
    if ((rc = gemm_tile16(agg, w_l, c_in, x_dst, w_r, c_in, c_in, c_in, (int)n_dst, c_out, b_l, out, c_out, stream, &used, dims, stats))) return rc;
    if (used) return POPE_OK;
    rc = gemm_streamk(agg, w_l, c_in, x_dst, w_r, c_in, c_in, c_in, (int)n_dst, c_out, b_l, out, c_out, scratch, scratch_bytes, stream, &used, dims);
    if (rc || used) return rc;
    const Operand A0{agg, c_in, 1}, B0{w_l, c_in, 1}, A1{x_dst, c_in, 1}, B1{w_r, c_in, 1};
    GemmDyn dyn;
    dyn.m = dims;
    return gemm(A0, B0, c_in, A1, B1, c_in, (int)n_dst, c_out, b_l, out, c_out, 1, nullptr, stream, Twin{Operand{nullptr, 0, 0}, nullptr, 0}, dyn);
}

extern "C" int sage_conv_forward(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz,
                                 const float *x_src, int32_t c_in, const float *w_l, const float *b_l, const float *w_r,
                                 int32_t c_out, float *agg, float *out, void *scratch, size_t scratch_bytes, const int32_t *dims,
                                 void *stream_) {
    clear_error();
    return conv_forward_impl(rowptr, col, n_src, n_dst, nnz, x_src, c_in, w_l, b_l, w_r, c_out, agg, out, scratch, scratch_bytes, dims, nullptr,
                             (hipStream_t)stream_);
}

extern "C" int sage_conv_forward_indexed(const int32_t *rowptr, const int32_t *col, const int64_t *n_id, int64_t n_src, int64_t n_dst,
                                         int64_t nnz, const float *feats, int64_t n_rows, int32_t c_in, const float *w_l,
                                         const float *b_l, const float *w_r, int32_t c_out, float *agg, float *x_dst, float *out,
                                         void *scratch, size_t scratch_bytes, const int32_t *dims, void *stream_) {
    clear_error();
    return conv_forward_indexed_impl(rowptr, col, n_id, n_src, n_dst, nnz, feats, n_rows, c_in, w_l, b_l, w_r, c_out, agg, x_dst, out, scratch,
                                     scratch_bytes, dims, nullptr, (hipStream_t)stream_);
}

// The same two calls for a layer whose output goes into BatchNorm (main.py:206-207: x = convs[i](...); x = bns[i](x)): the projection's
// epilogue also leaves the column sums of `out` and of its squares per row tile in bn_pa / bn_pb ([bn_parts_cap, c_out] doubles each;
// bn_parts_cap >= ceil(n_dst / 16) always suffices) -- the first stage of the BatchNorm statistics, which was a launch of its own
// re-reading `out`.  bn_info[0] = row tiles written, bn_info[1] = rows per tile (host ints); bn_info[0] = 0: this shape's kernel
// path produces no statistics, run sage_bn_relu_dropout_forward; else sage_bn_relu_dropout_forward_stats.
extern "C" int sage_conv_forward_stats(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz,
                                       const float *x_src, int32_t c_in, const float *w_l, const float *b_l, const float *w_r,
                                       int32_t c_out, float *agg, float *out, void *scratch, size_t scratch_bytes, const int32_t *dims,
                                       double *bn_pa, double *bn_pb, int32_t bn_parts_cap, int32_t *bn_info, void *stream_) {
    clear_error();
    POPE_REQUIRE(bn_pa && bn_pb && bn_info && bn_parts_cap > 0, "sage_conv_forward_stats: null pointer");
    BnStatsOut st;
    st.pa = bn_pa; st.pb = bn_pb; st.parts_cap = bn_parts_cap;
    const int rc = conv_forward_impl(rowptr, col, n_src, n_dst, nnz, x_src, c_in, w_l, b_l, w_r, c_out, agg, out, scratch, scratch_bytes, dims, &st,
                                     (hipStream_t)stream_);
    bn_info[0] = st.parts; bn_info[1] = st.rows_per_part;
    return rc;
}

extern "C" int sage_conv_forward_indexed_stats(const int32_t *rowptr, const int32_t *col, const int64_t *n_id, int64_t n_src, int64_t n_dst,
                                               int64_t nnz, const float *feats, int64_t n_rows, int32_t c_in, const float *w_l,
                                               const float *b_l, const float *w_r, int32_t c_out, float *agg, float *x_dst, float *out,
                                               void *scratch, size_t scratch_bytes, const int32_t *dims, double *bn_pa, double *bn_pb,
                                               int32_t bn_parts_cap, int32_t *bn_info, void *stream_) {
    clear_error();
    POPE_REQUIRE(bn_pa && bn_pb && bn_info && bn_parts_cap > 0, "sage_conv_forward_indexed_stats: null pointer");
    BnStatsOut st;
    st.pa = bn_pa; st.pb = bn_pb; st.parts_cap = bn_parts_cap;
    const int rc = conv_forward_indexed_impl(rowptr, col, n_id, n_src, n_dst, nnz, feats, n_rows, c_in, w_l, b_l, w_r, c_out, agg, x_dst, out,
                                             scratch, scratch_bytes, dims, &st, (hipStream_t)stream_);
    bn_info[0] = st.parts; bn_info[1] = st.rows_per_part;
    return rc;
}

// (Round 3 measured the bias gradient and the grad_x chain on side streams beside the weight gradients: 0.449 ms against 0.378 ms on
//  one stream -- every fork / join is two cross-stream dependencies and a side kernel on a CU delays the statically dealt stream-K
//  launch; removed in round 5.)
// x_rows == nullptr: x_src holds the block's source rows (the destinations first).  Otherwise the destination rows are
// x_src[x_rows[i]] (the resident feature matrix read through n_id) and x_tmp is room for them as a matrix, used only by the
// kernel paths that cannot follow the index.
static int conv_backward_impl(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz, const float *x_src,
                              const long long *x_rows, float *x_tmp, const float *agg, int32_t c_in, const float *w_l, const float *w_r,
                              int32_t c_out, const float *grad_out, float *grad_x, float *grad_w_l, float *grad_b_l, float *grad_w_r,
                              void *scratch, size_t scratch_bytes, const int32_t *dims, hipStream_t stream) {
    const size_t gagg_bytes = align_up((size_t)n_dst * c_in * sizeof(float), 256);
    const size_t colsum_bytes = align_up((size_t)COLSUM_SPLITS * c_out * sizeof(float), 256);
    float *gagg = (float *)scratch;
    float *slab = (float *)((char *)scratch + gagg_bytes);
    const size_t slab_bytes = scratch_bytes - gagg_bytes - colsum_bytes;
    float *colsum = (float *)((char *)scratch + scratch_bytes - colsum_bytes);
    const int32_t *n_dst_dev = dims, *n_src_dev = dims ? dims + 1 : nullptr;
    const int splits = weight_grad_splits(n_dst, c_in, c_out);
    const Operand none{nullptr, 0, 0};
    int rc;

    hipStream_t s_bias = stream, s_x = stream;

    // grad_w_l[o, c] = sum_i grad_out[i, o] * agg[i, c];  grad_w_r likewise with x_dst   (depth = rows i)
    bool used = false;
    const bool colsum_vec = (c_out & 3) == 0 && aligned16(grad_out);
    const bool bias_with_gemm = grad_b_l != nullptr;             // the bias gradient's two stages travel with the stream-K launches
    if ((rc = gemm_streamk_tn(grad_out, agg, x_src, n_dst, c_out, c_in, grad_w_l, grad_w_r, slab, slab_bytes, stream, &used, n_dst_dev,
                              bias_with_gemm ? colsum : nullptr, grad_b_l, colsum_vec, x_rows))) return rc;
    const bool bias_done = used && bias_with_gemm;
    if (!used && x_rows) {                                           // the other kernels want the destination rows as a matrix
        hipLaunchKernelGGL(k_gather_rows, dim3(capped_grid((size_t)n_dst * 64, 256)), dim3(256), 0, stream, x_src, x_rows, (int)n_dst, c_in, x_tmp,
                           n_dst_dev);
        x_src = x_tmp;
    }
    const Operand Gt{grad_out, 1, c_out};                       // (outer o, depth i) -> grad_out[i * c_out + o]
    const Operand AggT{agg, 1, c_in}, XdT{x_src, 1, c_in};      // (outer c, depth i)
    const Operand G{grad_out, c_out, 1};                        // (outer i, depth o)
    const Operand WrT{w_r, 1, c_in}, WlT{w_l, 1, c_in};         // (outer c, depth o) -> w[o * c_in + c]

    // Small layer (the weight gradients did not qualify for the stream-K kernel) with an input gradient: the two twin GEMMs,
    // the zeroing of grad_x's scatter-only rows and the bias gradient's partial sums all read grad_out and nothing of one
    // another -- one launch (k_gemm_dual), then one launch for both reductions, then the scatter.
    const bool dual = !used && grad_x && g_gemm_force_tile == 0 && splits > 1 && grad_b_l && colsum_vec &&
                      pick_layout(G, (int)n_dst, c_out) == LAYOUT_KC_VEC && pick_layout(WrT, c_in, c_out) == LAYOUT_OC_VEC &&
                      pick_layout(WlT, c_in, c_out) == LAYOUT_OC_VEC && pick_layout(Gt, c_out, (int)n_dst) == LAYOUT_OC_VEC &&
                      pick_layout(AggT, c_in, (int)n_dst) == LAYOUT_OC_VEC && pick_layout(XdT, c_in, (int)n_dst) == LAYOUT_OC_VEC &&
                      tiles((int)n_dst, c_in, 64, 128) * 2 < 384 && tiles(c_out, c_in, 64, 128) * splits * 2 < 384;   // both would take 64 x 64 tiles
    if (dual) {
        GemmDyn dx, dw;
        dx.m = n_dst_dev;
        dw.k0 = n_dst_dev;
        const GemmArgs p0 = gemm_args(G, WrT, c_out, none, none, 0, (int)n_dst, c_in, nullptr, grad_x, c_in, 1, nullptr, Twin{WlT, gagg, 0}, dx, 64, 64);
        const GemmArgs p1 = gemm_args(Gt, AggT, (int)n_dst, none, none, 0, c_out, c_in, nullptr, grad_w_l, c_in, splits, slab, Twin{XdT, grad_w_r, 0}, dw, 64, 64);
        DualAux aux;
        aux.zero_x = n_src > n_dst ? grad_x : nullptr;
        aux.zero_C = c_in; aux.zero_r0 = (int)n_dst; aux.zero_r1 = (int)n_src; aux.zero_r0_dev = n_dst_dev; aux.zero_r1_dev = n_src_dev;
        aux.zero_blocks = aux.zero_x ? (int)capped_grid((size_t)n_src * c_in / 4 / 8 + 1, 256, 256) : 0;
        aux.cs_g = grad_out; aux.cs_rows = (int)n_dst; aux.cs_C = c_out; aux.cs_part = colsum; aux.cs_rows_dev = n_dst_dev;
        aux.cs_gx = (c_out + 255) / 256; aux.cs_gy = COLSUM_SPLITS;
        const size_t lds = tile_lds_bytes<64, 64>();
        static LdsOptIn opt_in;
        if (!opt_in.done()) {
            POPE_HIP(hipFuncSetAttribute((const void *)k_gemm_dual<64, 64, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            opt_in.mark();
        }
        const int blocks = p0.gx * p0.gy * p0.gz + p1.gx * p1.gy * p1.gz + aux.zero_blocks + aux.cs_gx * aux.cs_gy;
        hipLaunchKernelGGL((k_gemm_dual<64, 64, 2, 2>), dim3(blocks), dim3(256), lds, stream, p0, p1, aux);
        const int reduce_blocks = (int)capped_grid((size_t)c_out * c_in * 2, 256);
        const BwdFinals fin{slab, splits, (size_t)c_out * c_in, c_in, grad_w_l, grad_w_r, (long long)c_in, reduce_blocks, colsum, COLSUM_SPLITS, c_out, grad_b_l};
        const int finals_blocks = reduce_blocks + (c_out + 15) / 16;
        if (nnz > 0)
            hipLaunchKernelGGL(k_scatter_and_finals, dim3(finals_blocks + capped_grid((size_t)n_dst * ((c_in + 255) / 256) * 64, 256)), dim3(256), 0, stream,
                               rowptr, col, (int)n_dst, gagg, c_in, grad_x, n_dst_dev, fin, finals_blocks);
        else
            hipLaunchKernelGGL(k_bwd_finals, dim3(finals_blocks), dim3(256), 0, stream, fin);
        POPE_HIP(hipGetLastError());
        return POPE_OK;
    }
    if (!used) {
        GemmDyn dyn;
        dyn.k0 = n_dst_dev;
        if ((rc = gemm(Gt, AggT, (int)n_dst, none, none, 0, c_out, c_in, nullptr, grad_w_l, c_in, splits, slab, stream,
                       Twin{XdT, grad_w_r, 0}, dyn))) return rc;
    }
    if (grad_b_l && !bias_done) {
        if (colsum_vec)
            hipLaunchKernelGGL(k_colsum_partial<true>, dim3((c_out + 255) / 256, COLSUM_SPLITS), dim3(256), 0, s_bias, grad_out, (int)n_dst, c_out, colsum, n_dst_dev);
        else
            hipLaunchKernelGGL(k_colsum_partial<false>, dim3((c_out + 63) / 64, COLSUM_SPLITS), dim3(256), 0, s_bias, grad_out, (int)n_dst, c_out, colsum, n_dst_dev);
        hipLaunchKernelGGL(k_colsum_final, dim3((c_out + 15) / 16), dim3(256), 0, s_bias, colsum, COLSUM_SPLITS, c_out, grad_b_l);
    }
    if (grad_x) {
        // grad_x[:n_dst] = grad_out * w_r ; rows >= n_dst start at zero; then scatter grad_agg = grad_out * w_l
        if (n_src > n_dst)
            hipLaunchKernelGGL(k_zero_rows, dim3(capped_grid((size_t)(n_src - (dims ? 0 : n_dst)) * c_in / 4 + 1, 256)), dim3(256), 0, s_x, grad_x, c_in,
                               (int)n_dst, (int)n_src, n_dst_dev, n_src_dev);
        GemmDyn dyn;
        dyn.m = n_dst_dev;
        if ((rc = gemm(G, WrT, c_out, none, none, 0, (int)n_dst, c_in, nullptr, grad_x, c_in, 1, nullptr, s_x,
                       Twin{WlT, gagg, 0}, dyn))) return rc;
        if (nnz > 0)
            hipLaunchKernelGGL(k_scatter_mean, dim3(capped_grid((size_t)n_dst * ((c_in + 255) / 256) * 64, 256)), dim3(256), 0, s_x, rowptr, col,
                               (int)n_dst, gagg, c_in, grad_x, n_dst_dev);
    }
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int sage_conv_backward(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz,
                                  const float *x_src, const float *agg, int32_t c_in, const float *w_l, const float *w_r,
                                  int32_t c_out, const float *grad_out, float *grad_x, float *grad_w_l, float *grad_b_l,
                                  float *grad_w_r, void *scratch, size_t scratch_bytes, const int32_t *dims, void *stream_) {
    clear_error();
    POPE_REQUIRE(rowptr && (col || nnz == 0) && x_src && agg && w_l && w_r && grad_out && grad_w_l && grad_w_r && scratch,
                 "sage_conv_backward: null pointer");
    POPE_REQUIRE(n_dst > 0 && n_dst <= n_src && n_src < INT32_MAX && nnz >= 0 && c_in > 0 && c_out > 0, "sage_conv_backward: bad size");
    if (scratch_bytes < sage_conv_scratch_bytes(n_src, n_dst, nnz, c_in, c_out)) {
        set_error("sage_conv_backward: scratch %zu < %zu bytes", scratch_bytes, sage_conv_scratch_bytes(n_src, n_dst, nnz, c_in, c_out));
        return POPE_ERR_WORKSPACE;
    }
    return conv_backward_impl(rowptr, col, n_src, n_dst, nnz, x_src, nullptr, nullptr, agg, c_in, w_l, w_r, c_out, grad_out, grad_x, grad_w_l, grad_b_l,
                              grad_w_r, scratch, scratch_bytes, dims, (hipStream_t)stream_);
}

// The backward pass of sage_conv_forward_indexed without a matrix of the destination rows: grad_w_r = grad_out^T x_dst reads
// x_dst[i] = feats[n_id[i]] through n_id in the weight-gradient kernel's loader (gemm_streamk_tn.h, rows1).  The input
// features have no gradient (layer 0), so there is no grad_x.  Scratch: sage_conv_backward_indexed_scratch_bytes.
extern "C" int sage_conv_backward_indexed(const int32_t *rowptr, const int32_t *col, const int64_t *n_id, int64_t n_src, int64_t n_dst,
                                          int64_t nnz, const float *feats, int64_t n_rows, const float *agg, int32_t c_in, const float *w_l,
                                          const float *w_r, int32_t c_out, const float *grad_out, float *grad_w_l, float *grad_b_l,
                                          float *grad_w_r, void *scratch, size_t scratch_bytes, const int32_t *dims, void *stream_) {
    clear_error();
    POPE_REQUIRE(rowptr && (col || nnz == 0) && n_id && feats && agg && w_l && w_r && grad_out && grad_w_l && grad_w_r && scratch,
                 "sage_conv_backward_indexed: null pointer");
    POPE_REQUIRE(n_dst > 0 && n_dst <= n_src && n_src < INT32_MAX && n_rows > 0 && nnz >= 0 && c_in > 0 && c_out > 0,
                 "sage_conv_backward_indexed: bad size");
    const size_t base = sage_conv_scratch_bytes(n_src, n_dst, nnz, c_in, c_out);
    if (scratch_bytes < base + rows_matrix_bytes(n_dst, c_in)) {
        set_error("sage_conv_backward_indexed: scratch %zu < %zu bytes", scratch_bytes, base + rows_matrix_bytes(n_dst, c_in));
        return POPE_ERR_WORKSPACE;
    }
    return conv_backward_impl(rowptr, col, n_src, n_dst, nnz, feats, (const long long *)n_id, (float *)((char *)scratch + base), agg, c_in, w_l, w_r,
                              c_out, grad_out, nullptr, grad_w_l, grad_b_l, grad_w_r, scratch, base, dims, (hipStream_t)stream_);
}
