// Stream-K f32 MFMA GEMM for the SAGEConv weight gradients on MI355X (gfx950): the "TN" twin of gemm_streamk.h.
//
//     C_q[M, Nb] = G^T * B_q       q = 0, 1         G [depth, M], B_q [depth, Nb]: both operands row-major over the DEPTH
//
// grad_w_l = grad_out^T * agg and grad_w_r = grad_out^T * x_dst (sage_conv_backward): M = c_out = 256, Nb = c_in = 756, the
// reduction runs over the n_dst ~ 10 000 destination rows.  The two products share G; together they are 24 output tiles of
// 64 x 256 over a depth of 313 stages of 32 rows -- the same 7 500 units as the forward projection, dealt out the same way
// (one persistent block per CU, partial tiles to per-block slabs, k_streamk_tn_fixup adds them in block order).
// Round 1 computed them as whole 64 x 128 tiles with split-K slabs: 324 MB fetched for 70 MB of operands
// (profiles/r02_sage_counters.json: every tile column re-read grad_out, every tile row re-read agg / x_dst).
//
// Here both operands are "k-major" in memory (for one depth row the 64 / 256 outer elements are contiguous), so the LDS
// image is simply [depth row][outer]: a DMA wave-instruction moves one 1 KiB depth row of B (or four 256-byte depth rows of
// G), nothing is swizzled, and a fragment read is 32 consecutive dwords per half-wave (ds_read_b32, conflict-free).
// Roles as in k_gemm_streamk_ld: waves 0-3 read fragments and issue MFMAs (32 x 128 each), waves 4-7 issue the DMAs
// (SADDR form: no vector arithmetic) and run one stage ahead.
#pragma once

#include "gemm_streamk.h"

namespace pope {

struct SkTnArgs {
    const float *G;              // [depth, M]:  A(m, k) = G[k * ldg + m]
    long long ldg;
    const float *B[2];           // [depth, Nb]: B_q(n, k) = B[q][k * ldb + n]
    long long ldb;
    float *C[2];                 // [M, Nb]
    long long ldc;
    int M, Nb, depth;
    float *slab;                 // [gridDim.x][2][SK_TM][SK_TN]
    const float *zero;
    int tiles_m, tiles_nb, S;    // tiles along M, along Nb (per product), depth stages
    const int *depth_dev;        // device extent: the true depth (<= depth, which then is the capacity), or null
    const long long *rows1;      // null, or: depth row k of the SECOND product's B operand is B[1] + rows1[k] * ldb (the destination rows
                                 // of a sampled block read straight from the resident feature matrix through n_id: no x_dst copy)
    // a second role of the fix-up launch (blocks behind the tiles'): out[c] = sum over z of part[z * C + c], the final stage of the
    // bias gradient's column sums, whose partials an earlier launch wrote -- one launch less in the backward pass
    const float *cs_part;
    float *cs_out;
    int cs_splits, cs_C;
    // 1 = XCD-aware deal (round 5, POPE_KNOB_STREAMK_XCD): physical block p sits on XCD p % 8.  The tiles_m tiles of one (product, column
    // panel) "group" share their B rows; the blocks that work on the same depth range of such a group are made neighbours IN ONE XCD
    // (slots 4 j .. 4 j + 3 of it), so the 32 KB B slice of a stage crosses the fabric once per XCD instead of once per block.  Units
    // are then (group, stage) pairs dealt to grid / tiles_m quads; a quad's blocks each take their own m-tile of the quad's span.
    int xcd;
};

// The span of units block `block` of `grid` owns, and how a unit maps to a tile.
struct SkTnSpan { long long u, u_end; int tm; };          // tm < 0: unit = tile * S + stage; else unit = group * S + stage, tile = group * tiles_m + tm
__device__ __forceinline__ void sk_tn_quads(const SkTnArgs &a, int grid, int &per, long long &Tq, int &Q) {
    per = (grid >> 3) / a.tiles_m;                         // quads per XCD
    Tq = (long long)a.tiles_nb * 2 * a.S;
    Q = grid / a.tiles_m;
}
__device__ __forceinline__ SkTnSpan sk_tn_span(const SkTnArgs &a, int block, int grid) {
    SkTnSpan w;
    if (!a.xcd) {
        const long long T = (long long)a.tiles_m * a.tiles_nb * 2 * a.S;
        w.u = sk_lo(block, T, grid);
        w.u_end = sk_lo(block + 1, T, grid);
        w.tm = -1;
    } else {
        int per, Q;
        long long Tq;
        sk_tn_quads(a, grid, per, Tq, Q);
        const int xcd = block & 7, slot = block >> 3;
        const int quad = xcd * per + slot / a.tiles_m;
        w.tm = slot % a.tiles_m;
        w.u = sk_lo(quad, Tq, Q);
        w.u_end = sk_lo(quad + 1, Tq, Q);
    }
    return w;
}

template <int GK>
__device__ __forceinline__ void sk_tn_resolve(SkTnArgs &a) {
    if (a.depth_dev) {
        a.depth = dyn_extent(a.depth_dev, a.depth);
        a.S = (a.depth + GK - 1) / GK;
    }
}

__device__ __forceinline__ void sk_tn_tile(const SkTnArgs &a, int tile, int &q, int &m0, int &n0) {
    const int tm = tile % a.tiles_m, rest = tile / a.tiles_m;
    q = rest / a.tiles_nb;
    m0 = tm * SK_TM;
    n0 = (rest - q * a.tiles_nb) * SK_TN;
}

// GK: depth rows per stage, as in k_gemm_streamk_ld (32: three buffers, two stages in flight; 64: two 80 KB buffers, one).
template <int GK>
__global__ __launch_bounds__(SKL_THREADS) void k_gemm_streamk_tn(SkTnArgs a) {
    sk_tn_resolve<GK>(a);
    constexpr int NT = 4;                               // 32 x 32 accumulator tiles per consumer wave: 32 rows x 128 columns
    constexpr int A_BYTES = SK_TM * GK * 4, STAGE_BYTES = (SK_TM + SK_TN) * GK * 4;
    constexpr int NBUF = GK == 32 ? 3 : 2;
    constexpr int A_INSTR = GK / 4, INSTR = GK / 4 + GK;   // A: four 256-byte depth rows per DMA instruction; B: one 1 KiB depth row
    constexpr int ND = INSTR / SKL_LOADERS;             // DMA wave-instructions per loader wave and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem);
    const int S = a.S;
    const SkTnSpan span = sk_tn_span(a, (int)blockIdx.x, (int)gridDim.x);
    long long u = span.u;
    const long long u_end = span.u_end;

    if (wave >= SKL_CONSUMERS) {
        // ---------------- loader waves ----------------
        __builtin_amdgcn_s_setprio(3);
        const int lw = wave - SKL_CONSUMERS;
        while (u < u_end) {
            const int base = (int)(u / S);
            const int s_begin = (int)(u - (long long)base * S);
            const int tile = span.tm < 0 ? base : base * a.tiles_m + span.tm;
            const long long left = u_end - u;
            const int s_end = (long long)(S - s_begin) <= left ? S : s_begin + (int)left;
            int q, m0, n0;
            sk_tn_tile(a, tile, q, m0, n0);
            const float *Bq = q ? a.B[1] : a.B[0];
            // per-lane byte offsets from the stage's first depth row; instruction i < A_INSTR: four depth rows of G (16 lanes
            // each), else depth row i - A_INSTR of B.  Outer indices past the matrix are clamped into it (their products land in
            // output rows / columns that are never stored); depth rows past the end are handled in the last stage only.
            unsigned off[ND];
            int drow[ND];
            const bool indexed = q == 1 && a.rows1 != nullptr;         // B rows through an index: the row offset is added per stage
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int instr = lw * ND + d;
                if (instr < A_INSTR) {
                    drow[d] = instr * 4 + (lane >> 4);
                    const int m = min(m0 + (lane & 15) * 4, a.M - 4);
                    off[d] = (unsigned)(((long long)drow[d] * a.ldg + m) * 4);
                } else {
                    drow[d] = instr - A_INSTR;
                    const int n = min(n0 + lane * 4, a.Nb - 4);
                    off[d] = indexed ? (unsigned)(n * 4) : (unsigned)(((long long)drow[d] * a.ldb + n) * 4);
                }
            }
            long long rowv = 0;
            if (indexed) {
                rowv = a.rows1[min(s_begin * GK + (lane & (GK - 1)), a.depth - 1)];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            auto issue_all = [&](int s, int buf) {
                const int k0 = s * GK;
                const float *Ga = a.G + (long long)k0 * a.ldg, *Ba = Bq + (long long)k0 * a.ldb;        // SGPR pairs
                if (indexed) {
                    // G as below; every B instruction is one depth row.  Lane r of the wave holds the place of the stage's depth
                    // row r in the feature matrix (rowv: loaded one stage ahead, behind the previous stage's DMAs, so the wait in
                    // front of the stage barrier covers it); a depth row past the end re-reads the last one: its G row is zero.
                    const long long cur = rowv;
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        const int instr = lw * ND + d;                 // wave-uniform
                        if (instr < A_INSTR) {
                            const bool past = k0 + drow[d] >= a.depth;
                            if (k0 + GK <= a.depth) sk_glds16_saddr(Ga, off[d], lds0 + buf * STAGE_BYTES + instr * 1024);
                            else sk_glds16(past ? a.zero : (const float *)((const char *)Ga + off[d]), lds0 + buf * STAGE_BYTES + instr * 1024);
                        } else {
                            const int r = instr - A_INSTR;
                            const unsigned lo = __builtin_amdgcn_readlane((unsigned)cur, r), hi = __builtin_amdgcn_readlane((unsigned)(cur >> 32), r);
                            const long long row = (long long)(((unsigned long long)hi << 32) | lo);
                            sk_glds16_saddr(Bq + row * a.ldb, off[d], lds0 + buf * STAGE_BYTES + instr * 1024);
                        }
                    }
                    rowv = a.rows1[min(k0 + GK + (lane & (GK - 1)), a.depth - 1)];        // for the next stage (stages are issued in order)
                    return;
                }
                if (k0 + GK <= a.depth) {
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        sk_glds16_saddr(lw * ND + d < A_INSTR ? Ga : Ba, off[d], lds0 + buf * STAGE_BYTES + (lw * ND + d) * 1024);
                } else {                                               // the last stage: depth rows past the end
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        const bool is_a = lw * ND + d < A_INSTR;
                        const bool past = k0 + drow[d] >= a.depth;
                        const float *src;
                        if (is_a) src = past ? a.zero : (const float *)((const char *)Ga + off[d]);           // zero kills the product
                        else src = (const float *)((const char *)Ba + off[d]) - (past ? (long long)(k0 + drow[d] - (a.depth - 1)) * a.ldb : 0);
                        sk_glds16(src, lds0 + buf * STAGE_BYTES + (lw * ND + d) * 1024);
                    }
                }
            };
            __syncthreads();
            issue_all(s_begin, 0);
            if (NBUF == 3 && s_begin + 1 < s_end) issue_all(s_begin + 1, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // B_(s_begin)
            int buf = 0;
            for (int s = s_begin; s < s_end; ++s) {
                if (NBUF == 3) {
                    if (s + 2 < s_end) issue_all(s + 2, buf == 0 ? 2 : buf - 1);
                } else {
                    if (s + 1 < s_end) issue_all(s + 1, buf ^ 1);
                }
                if (s + 1 < s_end) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                      // B_(s+1)
                }
                buf = NBUF == 3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
            }
            u += s_end - s_begin;
        }
        return;
    }

    // ---------------- consumer waves ----------------
    const int wm = wave & 1, wn = wave >> 1;
    const int g = lane >> 5, l31 = lane & 31;
    while (u < u_end) {
        const int base = (int)(u / S);
        const int s_begin = (int)(u - (long long)base * S);
        const int tile = span.tm < 0 ? base : base * a.tiles_m + span.tm;
        const long long left = u_end - u;
        const int s_end = (long long)(S - s_begin) <= left ? S : s_begin + (int)left;
        int q, m0, n0;
        sk_tn_tile(a, tile, q, m0, n0);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        __syncthreads();
        __builtin_amdgcn_s_barrier();          // B_(s_begin): stage s_begin (and s_begin + 1 with three buffers) has landed
        asm volatile("" ::: "memory");
        // fragments of depth step j: A(m, 2j + g), B(n, 2j + g) -- 32 consecutive dwords per half-wave
        float fa[2], fb[NT][2];
        auto read_step = [&](int b, int j, int set) {
            const float *As = reinterpret_cast<const float *>(smem + b * STAGE_BYTES);
            const float *Bs = reinterpret_cast<const float *>(smem + b * STAGE_BYTES + A_BYTES);
            const int k = 2 * j + g;
            fa[set] = As[k * SK_TM + wm * 32 + l31];
#pragma unroll
            for (int t = 0; t < NT; ++t) fb[t][set] = Bs[k * SK_TN + wn * 128 + t * 32 + l31];
        };
        int buf = 0;
        read_step(0, 0, 0);
        for (int s = s_begin; s < s_end; ++s) {
            const bool next = s + 1 < s_end;
            const int nbuf = NBUF == 3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
#pragma unroll
            for (int j = 0; j < GK / 2; ++j) {
                if (j + 1 < GK / 2) read_step(buf, j + 1, (j + 1) & 1);
                // the next stage's first step: landed since B_s (three buffers only).  Unconditional (behind the last stage it reads a stale
                // buffer, unused): behind `if (next)` hipcc waits lgkmcnt(0) in front of this step's MFMAs, i.e. for these very reads
                else if (NBUF == 3) read_step(nbuf, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j & 1], fb[t][j & 1], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (next) {
                __builtin_amdgcn_s_barrier();                          // B_(s+1)
                asm volatile("" ::: "memory");
                if (NBUF == 2) read_step(nbuf, 0, 0);                  // two buffers: stage s + 1 is complete only now
            }
            buf = nbuf;
        }
        // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        const bool complete = s_begin == 0 && s_end == S;
        if (complete) {
            float *C = q ? a.C[1] : a.C[0];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = n0 + wn * 128 + t * 32 + l31;
                if (n >= a.Nb) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < a.M) C[(size_t)m * a.ldc + n] = acc[t][r];
                }
            }
        } else {
            float *dst = a.slab + ((size_t)blockIdx.x * 2 + (s_begin > 0 ? 0 : 1)) * SK_SLAB_FLOATS;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = wn * 128 + t * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    dst[(size_t)m * SK_TN + n] = acc[t][r];
                }
            }
        }
        u += s_end - s_begin;
    }
}

// 16 columns x 16 split-groups per block of 256 threads: every thread adds splits / 16 partials, LDS folds the 16 groups in a
// fixed order (k_colsum_final in sage.hip is this block as a kernel of its own).
__device__ __forceinline__ void colsum_final_block(const float *__restrict__ part, int splits, int C, float *__restrict__ out, int block) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c = block * 16 + cl;
    float s = 0.f;
    if (c < C)
#pragma unroll 4
        for (int z = grp; z < splits; z += 16) s += part[(size_t)z * C + c];
    red[grp][cl] = s;
    __syncthreads();
    if (grp == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][cl];
        out[c] = t;
    }
}

// SK_TN_FIX_PARTS blocks per tile, as k_streamk_fixup: 24 tiles only, so more parts per tile (384 blocks, one piece per thread).
constexpr int SK_TN_FIX_PARTS = 16;
template <int GK>
__global__ __launch_bounds__(256) void k_streamk_tn_fixup(SkTnArgs a, int G) {
    sk_tn_resolve<GK>(a);
    const int S = a.S;
    const long long T = (long long)a.tiles_m * a.tiles_nb * 2 * S;
    const int tile = blockIdx.x;
    if (tile >= 2 * a.tiles_m * a.tiles_nb) {                                             // the column-sum role (sixteen columns per block)
        if (blockIdx.y == 0 && a.cs_part) colsum_final_block(a.cs_part, a.cs_splits, a.cs_C, a.cs_out, tile - 2 * a.tiles_m * a.tiles_nb);
        return;
    }
    if (T <= 0) return;                                                                   // (a device extent of 0 rows: nothing was computed)
    // the contributors of this tile, in depth order: blocks b_lo .. b_hi of the plain deal, or -- XCD-aware deal -- the quads b_lo .. b_hi
    // of the tile's group, of each the block that takes this tile's m (its physical index: slab_of)
    int per = 1, Q = G, tm = 0;
    long long Td = T;
    long long u0 = (long long)tile * S;
    if (a.xcd) {
        sk_tn_quads(a, G, per, Td, Q);
        tm = tile % a.tiles_m;
        u0 = (long long)(tile / a.tiles_m) * S;
    }
    const long long u1 = u0 + S;
    const int b_lo = (int)(((u0 + 1) * Q - 1) / Td), b_hi = (int)((u1 * Q - 1) / Td);
    if (b_lo == b_hi) return;
    auto slab_of = [&](int b) { return a.xcd ? (b / per) + 8 * ((b % per) * a.tiles_m + tm) : b; };
    int q, m0, n0;
    sk_tn_tile(a, tile, q, m0, n0);
    float *C = q ? a.C[1] : a.C[0];
    constexpr int PER = SK_TM * SK_TN / 4 / SK_TN_FIX_PARTS;                              // 16-byte pieces per block
    for (int p = blockIdx.y * PER + threadIdx.x; p < (blockIdx.y + 1) * PER; p += blockDim.x) {
        const int m = p / (SK_TN / 4), n = (p % (SK_TN / 4)) * 4;
        if (m0 + m >= a.M || n0 + n >= a.Nb) continue;
        const size_t off = (size_t)m * SK_TN + n;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // the partial slabs of this tile, blocks b_lo .. b_hi in order, TWELVE loads in flight at a time: a tile of the weight
        // gradients has about eleven partials, and with four in flight and two pieces per thread the kernel was six load
        // latencies deep (9.4 us for 18 MB); a block with no unit of the tile (possible only when there are fewer units than
        // blocks) contributes nothing
        constexpr int INFLIGHT = 12;
        for (int b0 = b_lo; b0 <= b_hi; b0 += INFLIGHT) {
            float4 v[INFLIGHT];
#pragma unroll
            for (int i = 0; i < INFLIGHT; ++i) {
                const int b = b0 + i;
                const long long lo = sk_lo(b, Td, Q), hi = sk_lo(b + 1, Td, Q);
                const long long sb = lo > u0 ? lo : u0, se = hi < u1 ? hi : u1;
                const bool has = b <= b_hi && sb < se;
                v[i] = has ? *reinterpret_cast<const float4 *>(a.slab + ((size_t)slab_of(b) * 2 + (sb > u0 ? 0 : 1)) * SK_SLAB_FLOATS + off)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < INFLIGHT; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
        }
        float *c = C + (size_t)(m0 + m) * a.ldc + n0 + n;
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n0 + n + i < a.Nb) c[i] = sv[i];
    }
}

}  // namespace pope
