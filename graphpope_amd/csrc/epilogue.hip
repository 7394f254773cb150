// BatchNorm1d + ReLU + dropout of the GraphSAGE hidden layers as ONE op, forward and backward
// (/root/reference/main.py:207-209: x = self.bns[i](x); x = x.relu_(); x = F.dropout(x, p, training)).
//
// The matrix is [M, C] float32 row-major (M = destination nodes of the hop, C = hidden_channels), a few MB: every
// kernel is a streaming pass, so the op is three launches per direction instead of torch's seven:
//   forward   k_bn_partial (column sums of x, x^2 over row slabs, f64)  ->  k_bn_final (mean, rstd, running stats)
//             ->  k_bn_apply (normalise, scale/shift, ReLU, dropout mask from a counter hash: no mask tensor)
//   backward  k_bn_bwd_partial (sums of g and g*xhat, g = dy through dropout and ReLU, recomputed from x)
//             ->  k_bn_bwd_final  ->  k_bn_bwd_apply (dx)
// Statistics follow torch.nn.BatchNorm1d: biased variance for normalisation, unbiased for running_var, momentum
// update, eps inside the square root.  Sums are accumulated in f64 (torch: f32 Welford); results agree to ~1e-6.
#include "common.h"

namespace pope {

constexpr int BN_MAX_PARTS = 256;

__device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// Dropout decision of element `idx`: a 24-bit uniform from a counter hash of (seed, idx), kept if >= p * 2^24.
__device__ __forceinline__ bool keep_element(unsigned long long seed, unsigned long long idx, unsigned threshold) {
    unsigned h = fmix32((unsigned)idx ^ (unsigned)seed);
    h = fmix32(h ^ ((unsigned)(idx >> 32) * 0x9e3779b1u) ^ (unsigned)(seed >> 32));
    return (h >> 8) >= threshold;
}

template <int VEC> struct Pack { float v[VEC]; };

template <int VEC>
__device__ __forceinline__ Pack<VEC> load_pack(const float *p) {
    Pack<VEC> r;
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
        r.v[0] = p[0];
    }
    return r;
}

template <int VEC>
__device__ __forceinline__ void store_pack(float *p, const Pack<VEC> &r) {
    if constexpr (VEC == 4) *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    else p[0] = r.v[0];
}

// A block = 4 waves; a wave walks rows r0 + wave, r0 + wave + 4, ... of its slab, lane l owns VEC columns.
// BWD: accumulate (g, g * xhat) instead of (x, x^2).
template <int VEC, bool BWD>
__global__ __launch_bounds__(256) void k_bn_partial(const float *__restrict__ x, const float *__restrict__ dy, int M, int C,
                                                    int rows_per_part, const float *__restrict__ mean,
                                                    const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, unsigned long long seed,
                                                    unsigned threshold, float inv_keep, double *__restrict__ pa,
                                                    double *__restrict__ pb, const int *__restrict__ m_dev,
                                                    const unsigned long long *__restrict__ seed_dev, double *__restrict__ pc) {
    // pc (BWD only, or nullptr): a third column sum, of xhat itself -- with it k_bn_bwd_final can give the column sums of dx, i.e. the
    // bias gradient of the layer that produced x (sage_bn_relu_dropout_backward_bias), without anybody reading dx back
    __shared__ double sh[3][4][64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.y * 64 + lane) * VEC;
    const bool valid = c0 < C;
    if (m_dev) {                                                   // device extent: the grid was sized for the capacity M
        M = dyn_extent(m_dev, M);
        rows_per_part = (M + (int)gridDim.x - 1) / (int)gridDim.x;
    }
    if (seed_dev) seed += *seed_dev;
    const int r0 = blockIdx.x * rows_per_part, r1 = min(M, r0 + rows_per_part);
    double a[VEC], b[VEC], cc[VEC];
    float mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        a[i] = b[i] = cc[i] = 0.0;
        mu[i] = rs[i] = ga[i] = be[i] = 0.f;
    }
    if (BWD && valid) {
        const Pack<VEC> m4 = load_pack<VEC>(mean + c0), r4 = load_pack<VEC>(rstd + c0), g4 = load_pack<VEC>(gamma + c0),
                        b4 = load_pack<VEC>(beta + c0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) { mu[i] = m4.v[i]; rs[i] = r4.v[i]; ga[i] = g4.v[i]; be[i] = b4.v[i]; }
    }
    if (valid) {
        // four rows of a wave's stride in flight (their loads issued before the first is used), accumulated in row order: a wave
        // has ~10 rows and one load pair in flight per row made the kernel ten dependent latencies long (9.5 us for 20 MB)
        auto accumulate = [&](const Pack<VEC> &xv, const Pack<VEC> &gv, size_t off) {
            if constexpr (!BWD) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    a[i] += (double)xv.v[i];
                    b[i] += (double)xv.v[i] * (double)xv.v[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float xhat = (xv.v[i] - mu[i]) * rs[i];
                    const bool on = xhat * ga[i] + be[i] > 0.f && keep_element(seed, off + i, threshold);
                    const float g = on ? gv.v[i] * inv_keep : 0.f;
                    a[i] += (double)g;
                    b[i] += (double)g * (double)xhat;
                    if (pc) cc[i] += (double)xhat;
                }
            }
        };
        int r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            Pack<VEC> xv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t off = (size_t)(r + 4 * u) * C + c0;
                xv[u] = load_pack<VEC>(x + off);
                if constexpr (BWD) gv[u] = load_pack<VEC>(dy + off);
                else gv[u] = xv[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) accumulate(xv[u], gv[u], (size_t)(r + 4 * u) * C + c0);
        }
        for (; r < r1; r += 4) {
            const size_t off = (size_t)r * C + c0;
            const Pack<VEC> xv = load_pack<VEC>(x + off);
            Pack<VEC> gv = xv;
            if constexpr (BWD) gv = load_pack<VEC>(dy + off);
            accumulate(xv, gv, off);
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sh[0][wave][lane * VEC + i] = a[i];
        sh[1][wave][lane * VEC + i] = b[i];
        sh[2][wave][lane * VEC + i] = cc[i];
    }
    __syncthreads();
    if (wave == 0 && valid) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int j = lane * VEC + i;
            pa[(size_t)blockIdx.x * C + c0 + i] = sh[0][0][j] + sh[0][1][j] + sh[0][2][j] + sh[0][3][j];
            pb[(size_t)blockIdx.x * C + c0 + i] = sh[1][0][j] + sh[1][1][j] + sh[1][2][j] + sh[1][3][j];
            if (BWD && pc) pc[(size_t)blockIdx.x * C + c0 + i] = sh[2][0][j] + sh[2][1][j] + sh[2][2][j] + sh[2][3][j];
        }
    }
}

// Column totals of the per-slab partials.  A block is 16 columns x 16 slab-groups: a thread adds every 16th slab, LDS
// folds the groups in a fixed order (one thread per column walking all slabs was 2 x 256 dependent-latency loads: 69 us).
template <bool THIRD = false>
__device__ __forceinline__ void fold_parts(const double *__restrict__ pa, const double *__restrict__ pb, int parts, int C, int c,
                                           double &s, double &q, const double *__restrict__ pc = nullptr, double *t = nullptr) {
    __shared__ double red[3][16][17];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    double a = 0.0, b = 0.0, d = 0.0;
    if (c < C)
#pragma unroll 4
        for (int p = grp; p < parts; p += 16) {                  // (THIRD as a template argument: a run-time test of pc in here cost the loop its unrolling, +3.5 us)
            a += pa[(size_t)p * C + c];
            b += pb[(size_t)p * C + c];
            if constexpr (THIRD) d += pc[(size_t)p * C + c];
        }
    red[0][grp][cl] = a;
    red[1][grp][cl] = b;
    red[2][grp][cl] = d;
    __syncthreads();
    s = q = 0.0;
    double tt = 0.0;
    if (grp == 0)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            s += red[0][g][cl];
            q += red[1][g][cl];
            tt += red[2][g][cl];
        }
    if (t) *t = tt;
}

// training: batch statistics + running-stat update; else: the running statistics.  Launch: 256 threads, C/16 blocks.
// rows_per_part > 0: the partials come from the projection's epilogue, one per row tile of that height (sage_conv_forward_stats):
// with a device extent only the tiles in front of the true row count were written.
__global__ __launch_bounds__(256) void k_bn_final(const double *__restrict__ pa, const double *__restrict__ pb, int parts, int M, int C, int training,
                           float momentum, float eps, float *__restrict__ running_mean, float *__restrict__ running_var,
                           float *__restrict__ mean, float *__restrict__ rstd, long long *num_batches_tracked,
                           const int *__restrict__ m_dev, int rows_per_part) {
    M = dyn_extent(m_dev, M);
    if (rows_per_part > 0) parts = min(parts, (M + rows_per_part - 1) / rows_per_part);
    // nn.BatchNorm1d's step counter (`num_batches_tracked += 1` in training mode): as a torch op it is a launch of its own
    if (training && num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const bool owner = (threadIdx.x >> 4) == 0 && c < C;
    if (!training) {
        if (owner) {
            mean[c] = running_mean[c];
            rstd[c] = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
        }
        return;
    }
    double s, q;
    fold_parts(pa, pb, parts, C, c, s, q);
    if (!owner) return;
    const double mu = s / M;
    double var = q / M - mu * mu;                                  // biased
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mu);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

// dx_colsum (or nullptr; needs pc): the column sums of dx = gamma * rstd * (g - c1 - xhat * c2), i.e. gamma * rstd * (sum g - M c1 - c2 sum xhat),
// from the float64 sums -- the bias gradient of the layer whose output x is (in training mode it is zero up to rounding: BatchNorm
// removes whatever a bias in front of it adds; summing the float32 dx matrix, as a launch of its own did, gives rounding noise too).
__global__ __launch_bounds__(256) void k_bn_bwd_final(const double *__restrict__ pa, const double *__restrict__ pb, int parts, int M, int C,
                               int training, float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ c1,
                               float *__restrict__ c2, const int *__restrict__ m_dev, const double *__restrict__ pc,
                               const float *__restrict__ gamma, const float *__restrict__ rstd, float *__restrict__ dx_colsum) {
    M = dyn_extent(m_dev, M);
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    double s, q, t = 0.0;
    if (pc) fold_parts<true>(pa, pb, parts, C, c, s, q, pc, &t);          // (block-uniform)
    else fold_parts<false>(pa, pb, parts, C, c, s, q);
    if ((threadIdx.x >> 4) != 0 || c >= C) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
    const double k1 = training ? s / M : 0.0, k2 = training ? q / M : 0.0;
    c1[c] = (float)k1;                                             // eval: the statistics are constants
    c2[c] = (float)k2;
    if (dx_colsum) dx_colsum[c] = (float)((double)gamma[c] * (double)rstd[c] * (s - (double)M * (double)(float)k1 - (double)(float)k2 * t));
}

// y = dropout(relu((x - mean) * rstd * gamma + beta));  BWD: dx = gamma * rstd * (g - c1 - xhat * c2)
template <int VEC, bool BWD>
__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ x, const float *__restrict__ dy, size_t total, int C,
                                                  const float *__restrict__ mean, const float *__restrict__ rstd,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  const float *__restrict__ c1, const float *__restrict__ c2,
                                                  unsigned long long seed, unsigned threshold, float inv_keep,
                                                  float *__restrict__ out, const int *__restrict__ m_dev,
                                                  const unsigned long long *__restrict__ seed_dev) {
    if (m_dev) total = (size_t)dyn_extent(m_dev, (int)(total / C)) * C;
    if (seed_dev) seed += *seed_dev;
    const size_t stride = (size_t)gridDim.x * blockDim.x * VEC;
    for (size_t off = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; off < total; off += stride) {
        const int c0 = (int)(off % C);                             // VEC == 4 only when C % 4 == 0: one row per pack
        const Pack<VEC> xv = load_pack<VEC>(x + off);
        Pack<VEC> gv, r;
        if constexpr (BWD) gv = load_pack<VEC>(dy + off);
        // per-column parameters as packs too: 6 vector loads per thread instead of 24 scalar ones (the kernel is
        // bound by the number of memory instructions per 16 bytes of output, not by bytes)
        const Pack<VEC> mu = load_pack<VEC>(mean + c0), rs = load_pack<VEC>(rstd + c0), ga = load_pack<VEC>(gamma + c0),
                        be = load_pack<VEC>(beta + c0);
        Pack<VEC> k1, k2;
        if constexpr (BWD) { k1 = load_pack<VEC>(c1 + c0); k2 = load_pack<VEC>(c2 + c0); }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float xhat = (xv.v[i] - mu.v[i]) * rs.v[i];
            const float z = xhat * ga.v[i] + be.v[i];
            const bool on = z > 0.f && keep_element(seed, off + i, threshold);
            if constexpr (!BWD) {
                r.v[i] = on ? z * inv_keep : 0.f;
            } else {
                const float g = on ? gv.v[i] * inv_keep : 0.f;
                r.v[i] = ga.v[i] * rs.v[i] * (g - k1.v[i] - xhat * k2.v[i]);
            }
        }
        store_pack<VEC>(out + off, r);
    }
}

struct BnPlan {
    int parts, rows_per_part, vec;
    double *pa, *pb, *pc;
    float *c1, *c2;
};

static size_t bn_scratch(int C) { return 3 * (size_t)BN_MAX_PARTS * C * sizeof(double) + 2 * align_up((size_t)C * sizeof(float), 256); }

static BnPlan bn_plan(int64_t M, int C, void *scratch, const void *p0, const void *p1, const void *p2, const void *p3,
                      const void *p4, const void *p5, const void *p6) {
    BnPlan p;
    p.parts = (int)((M + 15) / 16);
    if (p.parts > BN_MAX_PARTS) p.parts = BN_MAX_PARTS;
    if (p.parts < 1) p.parts = 1;
    p.rows_per_part = (int)((M + p.parts - 1) / p.parts);
    const bool aligned = (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2 | (uintptr_t)p3 | (uintptr_t)p4 | (uintptr_t)p5 |
                           (uintptr_t)p6) & 15) == 0;          // matrices AND per-column vectors: all read as 16-byte packs
    p.vec = (C % 4 == 0 && aligned) ? 4 : 1;
    p.pa = (double *)scratch;
    p.pb = p.pa + (size_t)BN_MAX_PARTS * C;
    p.pc = p.pb + (size_t)BN_MAX_PARTS * C;
    p.c1 = (float *)(p.pc + (size_t)BN_MAX_PARTS * C);
    p.c2 = (float *)((char *)p.c1 + align_up((size_t)C * sizeof(float), 256));
    return p;
}

static unsigned drop_threshold(float p) {
    if (!(p > 0.f)) return 0u;
    if (p >= 1.f) return 1u << 24;                                  // above every 24-bit draw: nothing kept
    return (unsigned)((double)p * 16777216.0);
}

}  // namespace pope

using namespace pope;

extern "C" size_t sage_bn_scratch_bytes(int32_t C) { return C <= 0 ? 0 : bn_scratch(C); }

static int bn_forward_impl(const float *x, int64_t M, int32_t C, const float *gamma, const float *beta,
                           float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum, float eps,
                           int32_t training, float p, uint64_t seed, float *y, float *save_mean,
                           float *save_rstd, void *scratch, size_t scratch_bytes, const int32_t *rows_dev,
                           const uint64_t *seed_dev_, const double *ext_pa, const double *ext_pb, int ext_parts, int ext_rows_per_part,
                           hipStream_t stream) {
    POPE_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && scratch, "sage_bn_relu_dropout_forward: null pointer");
    POPE_REQUIRE(M > 0 && M < INT32_MAX && C > 0 && (size_t)M * C < ((size_t)1 << 40), "sage_bn_relu_dropout_forward: bad size");
    POPE_REQUIRE(training || (running_mean && running_var), "sage_bn_relu_dropout_forward: eval mode needs running statistics");
    POPE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sage_bn_relu_dropout_forward: running_mean/var go together");
    POPE_REQUIRE(p >= 0.f && p <= 1.f && eps > 0.f, "sage_bn_relu_dropout_forward: need 0 <= p <= 1, eps > 0");
    if (scratch_bytes < bn_scratch(C)) {
        set_error("sage_bn_relu_dropout_forward: scratch %zu < %zu bytes", scratch_bytes, bn_scratch(C));
        return POPE_ERR_WORKSPACE;
    }
    const BnPlan pl = bn_plan(M, C, scratch, x, y, nullptr, gamma, beta, save_mean, save_rstd);
    const unsigned long long *seed_dev = (const unsigned long long *)seed_dev_;
    const bool external = training && ext_pa && ext_pb && ext_parts > 0 && ext_rows_per_part > 0;     // the first stage was done by the projection's epilogue
    if (training && !external) {
        const dim3 grid(pl.parts, (C + 64 * pl.vec - 1) / (64 * pl.vec));
        if (pl.vec == 4)
            hipLaunchKernelGGL((k_bn_partial<4, false>), grid, dim3(256), 0, stream, x, nullptr, (int)M, C, pl.rows_per_part,
                               nullptr, nullptr, nullptr, nullptr, 0ull, 0u, 1.f, pl.pa, pl.pb, rows_dev, nullptr, nullptr);
        else
            hipLaunchKernelGGL((k_bn_partial<1, false>), grid, dim3(256), 0, stream, x, nullptr, (int)M, C, pl.rows_per_part,
                               nullptr, nullptr, nullptr, nullptr, 0ull, 0u, 1.f, pl.pa, pl.pb, rows_dev, nullptr, nullptr);
    }
    hipLaunchKernelGGL(k_bn_final, dim3((C + 15) / 16), dim3(256), 0, stream, external ? ext_pa : pl.pa, external ? ext_pb : pl.pb,
                       external ? ext_parts : pl.parts, (int)M, C, training, momentum, eps, running_mean, running_var, save_mean, save_rstd,
                       (long long *)num_batches_tracked, rows_dev, external ? ext_rows_per_part : 0);
    const unsigned thr = training ? drop_threshold(p) : 0u;
    const float inv_keep = (training && p > 0.f && p < 1.f) ? (float)(1.0 / (1.0 - (double)p)) : 1.f;
    const size_t total = (size_t)M * C;
    const unsigned blocks = capped_grid(total / pl.vec, 256);
    if (pl.vec == 4)
        hipLaunchKernelGGL((k_bn_apply<4, false>), dim3(blocks), dim3(256), 0, stream, x, nullptr, total, C, save_mean, save_rstd,
                           gamma, beta, nullptr, nullptr, (unsigned long long)seed, thr, inv_keep, y, rows_dev, seed_dev);
    else
        hipLaunchKernelGGL((k_bn_apply<1, false>), dim3(blocks), dim3(256), 0, stream, x, nullptr, total, C, save_mean, save_rstd,
                           gamma, beta, nullptr, nullptr, (unsigned long long)seed, thr, inv_keep, y, rows_dev, seed_dev);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int sage_bn_relu_dropout_forward(const float *x, int64_t M, int32_t C, const float *gamma, const float *beta,
                                            float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum, float eps,
                                            int32_t training, float p, uint64_t seed, float *y, float *save_mean,
                                            float *save_rstd, void *scratch, size_t scratch_bytes, const int32_t *rows_dev,
                                            const uint64_t *seed_dev_, void *stream_) {
    clear_error();
    return bn_forward_impl(x, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, training, p, seed, y, save_mean,
                           save_rstd, scratch, scratch_bytes, rows_dev, seed_dev_, nullptr, nullptr, 0, 0, (hipStream_t)stream_);
}

// The same with the first stage of the statistics -- per-row-tile column sums of x and of x^2 in float64 -- already done by the
// kernel that wrote x (sage_conv_forward_stats: pa / pb [parts, C], one part per `rows_per_part` rows): two launches instead of three.
extern "C" int sage_bn_relu_dropout_forward_stats(const float *x, int64_t M, int32_t C, const float *gamma, const float *beta,
                                                  float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum, float eps,
                                                  int32_t training, float p, uint64_t seed, float *y, float *save_mean, float *save_rstd,
                                                  void *scratch, size_t scratch_bytes, const int32_t *rows_dev, const uint64_t *seed_dev_,
                                                  const double *pa, const double *pb, int32_t parts, int32_t rows_per_part, void *stream_) {
    clear_error();
    POPE_REQUIRE(pa && pb && parts > 0 && rows_per_part > 0 && (int64_t)parts * rows_per_part >= M, "sage_bn_relu_dropout_forward_stats: the partials do not cover the rows");
    return bn_forward_impl(x, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, training, p, seed, y, save_mean,
                           save_rstd, scratch, scratch_bytes, rows_dev, seed_dev_, pa, pb, parts, rows_per_part, (hipStream_t)stream_);
}

static int bn_backward_impl(const float *x, const float *grad_y, int64_t M, int32_t C, const float *gamma,
                            const float *beta, const float *save_mean, const float *save_rstd,
                            int32_t training, float p, uint64_t seed, float *grad_x, float *grad_gamma,
                            float *grad_beta, void *scratch, size_t scratch_bytes, const int32_t *rows_dev,
                            const uint64_t *seed_dev_, float *grad_x_colsum, hipStream_t stream) {
    POPE_REQUIRE(x && grad_y && gamma && beta && save_mean && save_rstd && grad_x && scratch,
                 "sage_bn_relu_dropout_backward: null pointer");
    POPE_REQUIRE(M > 0 && M < INT32_MAX && C > 0 && (size_t)M * C < ((size_t)1 << 40), "sage_bn_relu_dropout_backward: bad size");
    POPE_REQUIRE(p >= 0.f && p <= 1.f, "sage_bn_relu_dropout_backward: need 0 <= p <= 1");
    if (scratch_bytes < bn_scratch(C)) {
        set_error("sage_bn_relu_dropout_backward: scratch %zu < %zu bytes", scratch_bytes, bn_scratch(C));
        return POPE_ERR_WORKSPACE;
    }
    const BnPlan pl = bn_plan(M, C, scratch, x, grad_y, grad_x, gamma, beta, save_mean, save_rstd);
    const unsigned long long *seed_dev = (const unsigned long long *)seed_dev_;
    const unsigned thr = training ? drop_threshold(p) : 0u;
    const float inv_keep = (training && p > 0.f && p < 1.f) ? (float)(1.0 / (1.0 - (double)p)) : 1.f;
    const dim3 grid(pl.parts, (C + 64 * pl.vec - 1) / (64 * pl.vec));
    if (pl.vec == 4)
        hipLaunchKernelGGL((k_bn_partial<4, true>), grid, dim3(256), 0, stream, x, grad_y, (int)M, C, pl.rows_per_part, save_mean,
                           save_rstd, gamma, beta, (unsigned long long)seed, thr, inv_keep, pl.pa, pl.pb, rows_dev, seed_dev, grad_x_colsum ? pl.pc : nullptr);
    else
        hipLaunchKernelGGL((k_bn_partial<1, true>), grid, dim3(256), 0, stream, x, grad_y, (int)M, C, pl.rows_per_part, save_mean,
                           save_rstd, gamma, beta, (unsigned long long)seed, thr, inv_keep, pl.pa, pl.pb, rows_dev, seed_dev, grad_x_colsum ? pl.pc : nullptr);
    hipLaunchKernelGGL(k_bn_bwd_final, dim3((C + 15) / 16), dim3(256), 0, stream, pl.pa, pl.pb, pl.parts, (int)M, C, training,
                       grad_gamma, grad_beta, pl.c1, pl.c2, rows_dev, grad_x_colsum ? pl.pc : nullptr, gamma, save_rstd, grad_x_colsum);
    const size_t total = (size_t)M * C;
    const unsigned blocks = capped_grid(total / pl.vec, 256);
    if (pl.vec == 4)
        hipLaunchKernelGGL((k_bn_apply<4, true>), dim3(blocks), dim3(256), 0, stream, x, grad_y, total, C, save_mean, save_rstd,
                           gamma, beta, pl.c1, pl.c2, (unsigned long long)seed, thr, inv_keep, grad_x, rows_dev, seed_dev);
    else
        hipLaunchKernelGGL((k_bn_apply<1, true>), dim3(blocks), dim3(256), 0, stream, x, grad_y, total, C, save_mean, save_rstd,
                           gamma, beta, pl.c1, pl.c2, (unsigned long long)seed, thr, inv_keep, grad_x, rows_dev, seed_dev);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int sage_bn_relu_dropout_backward(const float *x, const float *grad_y, int64_t M, int32_t C, const float *gamma,
                                             const float *beta, const float *save_mean, const float *save_rstd,
                                             int32_t training, float p, uint64_t seed, float *grad_x, float *grad_gamma,
                                             float *grad_beta, void *scratch, size_t scratch_bytes, const int32_t *rows_dev,
                                             const uint64_t *seed_dev_, void *stream_) {
    clear_error();
    return bn_backward_impl(x, grad_y, M, C, gamma, beta, save_mean, save_rstd, training, p, seed, grad_x, grad_gamma, grad_beta, scratch,
                            scratch_bytes, rows_dev, seed_dev_, nullptr, (hipStream_t)stream_);
}

// The same, and grad_x_colsum[c] = sum over the rows of grad_x[:, c] (float32 [C]) from the float64 sums of the statistics pass -- the
// bias gradient of the layer that produced x (main.py:206-207: x = convs[i](...) feeds bns[i]), which sage_conv_backward otherwise
// computes by reading grad_x back in a launch of its own (pass it grad_b_l = NULL then).
extern "C" int sage_bn_relu_dropout_backward_bias(const float *x, const float *grad_y, int64_t M, int32_t C, const float *gamma,
                                                  const float *beta, const float *save_mean, const float *save_rstd,
                                                  int32_t training, float p, uint64_t seed, float *grad_x, float *grad_gamma,
                                                  float *grad_beta, void *scratch, size_t scratch_bytes, const int32_t *rows_dev,
                                                  const uint64_t *seed_dev_, float *grad_x_colsum, void *stream_) {
    clear_error();
    POPE_REQUIRE(grad_x_colsum, "sage_bn_relu_dropout_backward_bias: null pointer");
    return bn_backward_impl(x, grad_y, M, C, gamma, beta, save_mean, save_rstd, training, p, seed, grad_x, grad_gamma, grad_beta, scratch,
                            scratch_bytes, rows_dev, seed_dev_, grad_x_colsum, (hipStream_t)stream_);
}

namespace pope {
__global__ void k_xent_final(const float *__restrict__ row_loss, int N, float *__restrict__ loss, float *__restrict__ inv_count);
}

// ------------------------------------------------------------------------------------------------
// Adam step over every parameter tensor in ONE launch (main.py:244 torch.optim.Adam(self.parameters(), lr=args.lr)).
// torch's default implementation is eight foreach launches plus ~80 us of Python per step; the training step is
// launch-bound, so the optimiser is one kernel whose argument block carries up to ADAM_MAX_TENSORS tensor descriptors.
// Update rule = torch.optim.Adam (amsgrad=False, maximize=False):
//   g' = g + wd * p;  m = m + (g' - m) * (1 - b1);  v = b2 * v + (1 - b2) * g'^2;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// ------------------------------------------------------------------------------------------------
namespace pope {

constexpr int ADAM_MAX_TENSORS = 24;
constexpr int ADAM_CHUNK = 4096;               // elements per block

struct AdamTable {
    float *p[ADAM_MAX_TENSORS];
    const float *g[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    int first_block[ADAM_MAX_TENSORS + 1];     // prefix sum of ceil(n / ADAM_CHUNK)
    int count;
};

// The last stage of the cross-entropy (csrc below: k_xent_rows leaves one loss per row): loss = mean over the counted rows, and
// 1 / count.  One block.  A launch of its own (k_xent_final), or -- round 5 -- the extra block of the optimiser's launch: the scalar
// is only read by the host after the step, so it need not hold up the backward pass (sage_adam_step_loss).
__device__ __forceinline__ void xent_final_block(const float *__restrict__ row_loss, int N, float *__restrict__ loss,
                                                 float *__restrict__ inv_count) {
    __shared__ double ssum[256];
    __shared__ int scnt[256];
    double s = 0.0;
    int n = 0;
    for (int base = 0; base < N; base += 8 * 256) {             // eight row losses per thread in flight (same order of additions)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float t = row_loss[min(base + (int)threadIdx.x + 256 * u, N - 1)];
            v[u] = base + (int)threadIdx.x + 256 * u < N ? t : -1.f;         // a select, not a branch: -1 marks "not counted"
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u] >= 0.f) { s += (double)v[u]; ++n; }
    }
    ssum[threadIdx.x] = s;
    scnt[threadIdx.x] = n;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { ssum[threadIdx.x] += ssum[threadIdx.x + off]; scnt[threadIdx.x] += scnt[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *loss = scnt[0] > 0 ? (float)(ssum[0] / scnt[0]) : __builtin_nanf("");      // torch: nan when every label is ignored
        *inv_count = scnt[0] > 0 ? 1.f / (float)scnt[0] : 0.f;
    }
}

// one_minus_b1 / one_minus_b2 are formed in double on the host, as Python does for torch (1 - 0.999f != float(1 - 0.999)).
// step_dev != nullptr: the 1-based step count lives on the device (a replayed HIP graph cannot take it as an argument);
// lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t) are then formed here, in double like on the host.
__global__ __launch_bounds__(256) void k_adam(AdamTable t, float step_size, float one_minus_b1, float beta2, float one_minus_b2,
                                              float eps, float weight_decay, float inv_bc2_sqrt, const long long *__restrict__ step_dev,
                                              double lr, double beta1_d, double beta2_d, const float *__restrict__ xent_rows, int xent_n,
                                              float *__restrict__ xent_out) {
    if ((int)blockIdx.x == t.first_block[t.count]) {                 // the one block behind the parameter chunks: the loss scalar of this step
        xent_final_block(xent_rows, xent_n, xent_out, xent_out + 1);
        return;
    }
    if (step_dev) {
        const double step = (double)*step_dev;
        step_size = (float)(lr / (1.0 - pow(beta1_d, step)));
        inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow(beta2_d, step)));
    }
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
    const long long base = (long long)((int)blockIdx.x - t.first_block[k]) * ADAM_CHUNK;
    const long long end = min(t.n[k], base + ADAM_CHUNK);
    float *__restrict__ p = t.p[k];
    const float *__restrict__ g = t.g[k];
    float *__restrict__ m = t.m[k];
    float *__restrict__ v = t.v[k];
    auto update = [&](float &pi, float gi, float &mi, float &vi) {
        if (weight_decay != 0.f) gi += weight_decay * pi;
        mi = mi + (gi - mi) * one_minus_b1;
        vi = beta2 * vi + one_minus_b2 * gi * gi;
        pi = pi - step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
    };
    long long i = base + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
          reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        // 16-byte packs: 7 memory instructions per 4 parameters instead of 28 (chunks start at multiples of 4096)
        const long long end4 = base + ((end - base) & ~3ll);
        for (long long q = base + 4ll * threadIdx.x; q < end4; q += 4ll * blockDim.x) {
            float4 p4 = *reinterpret_cast<float4 *>(p + q), m4 = *reinterpret_cast<float4 *>(m + q), v4 = *reinterpret_cast<float4 *>(v + q);
            const float4 g4 = *reinterpret_cast<const float4 *>(g + q);
            update(p4.x, g4.x, m4.x, v4.x);
            update(p4.y, g4.y, m4.y, v4.y);
            update(p4.z, g4.z, m4.z, v4.z);
            update(p4.w, g4.w, m4.w, v4.w);
            *reinterpret_cast<float4 *>(m + q) = m4;
            *reinterpret_cast<float4 *>(v + q) = v4;
            *reinterpret_cast<float4 *>(p + q) = p4;
        }
        i = end4 + threadIdx.x;                                    // scalar tail (< 4 elements)
    }
    for (; i < end; i += blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i];
        update(pi, g[i], mi, vi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
    }
}

}  // namespace pope

static int adam_step_impl(int32_t n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                          float *const *exp_avg_sq, const int64_t *numel, double lr, double beta1, double beta2, double eps,
                          double weight_decay, int64_t step, const int64_t *step_dev, const float *xent_rows, int64_t xent_n, float *xent_out,
                          hipStream_t stream) {
    POPE_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && numel)), "sage_adam_step: null pointer");
    if (step_dev) step = 1;                                         // the device word is the step count; `step` is ignored
    POPE_REQUIRE(step >= 1 && lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0,
                 "sage_adam_step: need step >= 1, lr >= 0, 0 <= beta < 1, eps >= 0");
    // scalars are formed in double, as Python does for torch.optim.Adam, and rounded to float once
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    for (int t0 = 0; t0 < n_tensors; t0 += ADAM_MAX_TENSORS) {
        AdamTable tab;
        tab.count = 0;
        int blocks = 0;
        for (int t = t0; t < n_tensors && tab.count < ADAM_MAX_TENSORS; ++t) {
            POPE_REQUIRE(numel[t] >= 0 && (numel[t] == 0 || (params[t] && grads[t] && exp_avg[t] && exp_avg_sq[t])),
                         "sage_adam_step: tensor %d has a null pointer", t);
            if (numel[t] == 0) continue;
            const int c = tab.count++;
            tab.p[c] = params[t]; tab.g[c] = grads[t]; tab.m[c] = exp_avg[t]; tab.v[c] = exp_avg_sq[t]; tab.n[c] = numel[t];
            tab.first_block[c] = blocks;
            blocks += (int)((numel[t] + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        if (tab.count == 0) continue;
        tab.first_block[tab.count] = blocks;
        const bool with_loss = xent_rows != nullptr;              // the first launch carries the loss block
        hipLaunchKernelGGL(k_adam, dim3(blocks + (with_loss ? 1 : 0)), dim3(256), 0, stream, tab, step_size, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), (float)eps, (float)weight_decay, inv_bc2_sqrt, (const long long *)step_dev, lr, beta1, beta2,
                           xent_rows, (int)xent_n, xent_out);
        xent_rows = nullptr;
    }
    if (xent_rows)                                                   // no parameter had anything to update: the loss still has to be finished
        hipLaunchKernelGGL(k_xent_final, dim3(1), dim3(256), 0, stream, xent_rows, (int)xent_n, xent_out, xent_out + 1);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int sage_adam_step(int32_t n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                              float *const *exp_avg_sq, const int64_t *numel, double lr, double beta1, double beta2, double eps,
                              double weight_decay, int64_t step, const int64_t *step_dev, void *stream_) {
    clear_error();
    return adam_step_impl(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, step, step_dev, nullptr, 0,
                          nullptr, (hipStream_t)stream_);
}

// The same step, and in the same launch the last stage of this step's cross-entropy (sage_cross_entropy_forward with fused = 2 left
// it out): loss_out[0] = mean row loss, loss_out[1] = 1 / count.  main.py:216 + 244: the loss is a number the host logs after the step.
extern "C" int sage_adam_step_loss(int32_t n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                                   float *const *exp_avg_sq, const int64_t *numel, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, int64_t step, const int64_t *step_dev, const float *xent_rows, int64_t xent_n,
                                   float *loss_out, void *stream_) {
    clear_error();
    POPE_REQUIRE(xent_rows && loss_out && xent_n > 0 && xent_n < INT32_MAX, "sage_adam_step_loss: null pointer or bad row count");
    return adam_step_impl(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, step, step_dev, xent_rows,
                          xent_n, loss_out, (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------------------------
// Cross-entropy of the logits against integer labels, mean over the rows whose label is not `ignore_index`
// (main.py:216 F.cross_entropy(y_hat, y)): loss AND the gradient of the mean loss in two launches
// (torch: log_softmax + nll_loss forward, then two more for the backward).
//   k_xent_rows   one wave per row: max, sum of exponentials, loss_i = logsumexp - logit[y]; grad_i = softmax - onehot
//   k_xent_final  one block: loss = sum loss_i / count, grad scale 1 / count (both stay on the device)
// The backward pass is grad_logits = grad * (upstream * 1/count): one more elementwise launch.
// ------------------------------------------------------------------------------------------------
namespace pope {

// FUSED ("pre-scaled"): the gradient is scaled by 1 / count here -- every block counts the valid labels itself (N labels, a
// few loads per thread) -- so that a backward pass seeded with 1 needs no launch.  (Measured and rejected: also folding
// k_xent_final in, by letting the block that arrives last add the row losses behind an agent-scope release / ticket /
// acquire: 16.7 us for the one launch against 5.2 + 4.7 for two -- 388 blocks each pay the release fence.)
template <bool FUSED>
__global__ __launch_bounds__(256) void k_xent_rows(const float *__restrict__ logits, const long long *__restrict__ target, int N,
                                                   int C, long long ignore_index, float *__restrict__ grad,
                                                   float *__restrict__ row_loss, int *__restrict__ bad_label) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    __shared__ int s_cnt[256];
    float gscale = 1.f;
    if constexpr (FUSED) {
        int n = 0;
        // eight labels per thread requested at once (round 4: as a plain loop this compiled to load - wait - add per label, seven
        // serial round trips per block for 1 550 labels); indices past N are clamped and not counted
        for (int base = 0; base < N; base += 8 * 256) {
            long long y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) y[u] = target[min(base + (int)threadIdx.x + 256 * u, N - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                n += (base + (int)threadIdx.x + 256 * u < N && y[u] != ignore_index && y[u] >= 0 && y[u] < C) ? 1 : 0;
        }
        s_cnt[threadIdx.x] = n;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) s_cnt[threadIdx.x] += s_cnt[threadIdx.x + off];
            __syncthreads();
        }
        gscale = s_cnt[0] > 0 ? 1.f / (float)s_cnt[0] : 0.f;
    }
    for (int i = wave; i < N; i += nwaves) {
        const float *row = logits + (size_t)i * C;
        float *g = grad + (size_t)i * C;
        const long long y = target[i];
        if (y == ignore_index || y < 0 || y >= C) {
            if (y != ignore_index && lane == 0) *bad_label = 1;
            for (int c = lane; c < C; c += 64) g[c] = 0.f;
            if (lane == 0) row_loss[i] = -1.f;                    // marks "not counted" (a loss is never negative)
            continue;
        }
        if (C <= 512) {
            // the row lives in registers: ONE round trip instead of three passes of C / 64 serial loads (same operations in the
            // same order, so the same bits)
            // Columns past C read the row's last element and become -inf through a SELECT (no branch: a load whose only use sits
            // under an `if` is sunk into it and waited for there); -inf is neutral for the maximum and adds exp(-inf) = +0 to the sum.
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float t = row[min(lane + 64 * k, C - 1)];
                v[k] = lane + 64 * k < C ? t : -__builtin_huge_valf();
            }
            const float ly = row[y];
            float mx = -__builtin_huge_valf();
#pragma unroll
            for (int k = 0; k < 8; ++k) mx = fmaxf(mx, v[k]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += expf(v[k] - mx);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
            const float lse = logf(sum) + mx, inv = 1.f / sum;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = lane + 64 * k;
                if (c < C) {
                    const float w = expf(v[k] - mx) * inv - (c == (int)y ? 1.f : 0.f);
                    g[c] = FUSED ? w * gscale : w;
                }
            }
            const float loss_i = lse - ly;                     // computed by every lane: a use outside the `if` keeps the load of row[y] up with the others
            if (lane == 0) row_loss[i] = loss_i;
            continue;
        }
        float mx = -__builtin_huge_valf();
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) sum += expf(row[c] - mx);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        const float lse = logf(sum) + mx, inv = 1.f / sum;
        for (int c = lane; c < C; c += 64) {
            const float v = expf(row[c] - mx) * inv - (c == (int)y ? 1.f : 0.f);
            g[c] = FUSED ? v * gscale : v;
        }
        if (lane == 0) row_loss[i] = lse - row[y];
    }
}

__global__ __launch_bounds__(256) void k_xent_final(const float *__restrict__ row_loss, int N, float *__restrict__ loss,
                                                    float *__restrict__ inv_count) {
    xent_final_block(row_loss, N, loss, inv_count);
}

__global__ __launch_bounds__(256) void k_xent_scale(const float *__restrict__ g, size_t n, const float *__restrict__ upstream,
                                                    const float *__restrict__ inv_count, float *__restrict__ out) {
    const float k = *upstream * *inv_count;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = g[i] * k;
}

}  // namespace pope

extern "C" int sage_cross_entropy_forward(const float *logits, const int64_t *target, int64_t N, int32_t C, int64_t ignore_index,
                                          float *loss, float *grad_unscaled, float *inv_count, float *row_scratch,
                                          int32_t *bad_label, int32_t fused, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(logits && target && loss && grad_unscaled && inv_count && row_scratch && bad_label, "sage_cross_entropy_forward: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && C > 0, "sage_cross_entropy_forward: bad size");
    if (fused)
        hipLaunchKernelGGL(k_xent_rows<true>, dim3(capped_grid((size_t)N * 64, 256)), dim3(256), 0, stream, logits, (const long long *)target,
                           (int)N, C, (long long)ignore_index, grad_unscaled, row_scratch, bad_label);
    else
        hipLaunchKernelGGL(k_xent_rows<false>, dim3(capped_grid((size_t)N * 64, 256)), dim3(256), 0, stream, logits, (const long long *)target,
                           (int)N, C, (long long)ignore_index, grad_unscaled, row_scratch, bad_label);
    if (fused != 2)                                                  // 2: the caller folds the last stage into the optimiser's launch (sage_adam_step_loss)
        hipLaunchKernelGGL(k_xent_final, dim3(1), dim3(256), 0, stream, row_scratch, (int)N, loss, inv_count);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int sage_cross_entropy_backward(const float *grad_unscaled, int64_t N, int32_t C, const float *grad_loss,
                                           const float *inv_count, float *grad_logits, void *stream_) {
    clear_error();
    POPE_REQUIRE(grad_unscaled && grad_loss && inv_count && grad_logits, "sage_cross_entropy_backward: null pointer");
    POPE_REQUIRE(N > 0 && C > 0, "sage_cross_entropy_backward: bad size");
    hipLaunchKernelGGL(k_xent_scale, dim3(capped_grid((size_t)N * C, 256)), dim3(256), 0, (hipStream_t)stream_, grad_unscaled,
                       (size_t)N * C, grad_loss, inv_count, grad_logits);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

// ------------------------------------------------------------------------------------------------
// Plumbing of a training step that is replayed as a HIP graph (no host in the loop): the step's seeds and the optimiser's
// step count live in device words that one tiny launch advances, and a batch taken from a pre-sampled pool is loaded
// into the step's fixed buffers by ONE launch.
// ------------------------------------------------------------------------------------------------
namespace pope {

constexpr int COUNTERS_MAX = 8, SEGMENTS_MAX = 12;
struct CounterIncs { long long v[COUNTERS_MAX]; };
__global__ void k_advance_counters(long long *__restrict__ c, CounterIncs inc, int n) {
    if ((int)threadIdx.x < n && inc.v[threadIdx.x] != 0) c[threadIdx.x] += inc.v[threadIdx.x];      // a zero increment leaves the word alone (another stream may own it)
}

struct SegmentTable {
    char *dst[SEGMENTS_MAX];
    const char *src[SEGMENTS_MAX];
    long long bytes[SEGMENTS_MAX];
    int first_block[SEGMENTS_MAX + 1];
    int count;
};
constexpr int SEG_CHUNK = 64 * 1024;          // bytes per block

__global__ __launch_bounds__(256) void k_copy_segments(SegmentTable t) {
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
    const long long base = (long long)((int)blockIdx.x - t.first_block[k]) * SEG_CHUNK;
    const long long end = min(t.bytes[k], base + SEG_CHUNK);
    char *__restrict__ d = t.dst[k];
    const char *__restrict__ s = t.src[k];
    if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
        const long long end16 = base + ((end - base) & ~15ll);
        for (long long q = base + 16ll * threadIdx.x; q < end16; q += 16ll * blockDim.x)
            *reinterpret_cast<float4 *>(d + q) = *reinterpret_cast<const float4 *>(s + q);
        for (long long q = end16 + threadIdx.x; q < end; q += blockDim.x) d[q] = s[q];
    } else {
        for (long long q = base + threadIdx.x; q < end; q += blockDim.x) d[q] = s[q];
    }
}

}  // namespace pope

extern "C" int sage_advance_counters(int64_t *counters, const int64_t *increments_host, int32_t n, void *stream_) {
    clear_error();
    POPE_REQUIRE(counters && increments_host && n > 0 && n <= COUNTERS_MAX, "sage_advance_counters: null pointer or more than %d counters", COUNTERS_MAX);
    CounterIncs inc;
    for (int i = 0; i < COUNTERS_MAX; ++i) inc.v[i] = i < n ? (long long)increments_host[i] : 0;
    hipLaunchKernelGGL(k_advance_counters, dim3(1), dim3(64), 0, (hipStream_t)stream_, (long long *)counters, inc, n);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int sage_copy_segments(int32_t n, void *const *dst, const void *const *src, const int64_t *bytes, void *stream_) {
    clear_error();
    POPE_REQUIRE(n >= 0 && (n == 0 || (dst && src && bytes)), "sage_copy_segments: null pointer");
    for (int t0 = 0; t0 < n; t0 += SEGMENTS_MAX) {
        SegmentTable tab;
        tab.count = 0;
        int blocks = 0;
        for (int t = t0; t < n && tab.count < SEGMENTS_MAX; ++t) {
            POPE_REQUIRE(bytes[t] >= 0 && (bytes[t] == 0 || (dst[t] && src[t])), "sage_copy_segments: segment %d has a null pointer", t);
            if (bytes[t] == 0) continue;
            const int c = tab.count++;
            tab.dst[c] = (char *)dst[t]; tab.src[c] = (const char *)src[t]; tab.bytes[c] = bytes[t];
            tab.first_block[c] = blocks;
            blocks += (int)((bytes[t] + SEG_CHUNK - 1) / SEG_CHUNK);
        }
        if (tab.count == 0) continue;
        tab.first_block[tab.count] = blocks;
        hipLaunchKernelGGL(k_copy_segments, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, tab);
    }
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}
