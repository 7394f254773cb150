// Whole-tile f32 MFMA GEMM whose tile COUNT is fitted to the chip: the layer-0 forward projection without a fix-up pass.
//
//     C[M, N] = A0[M, K0] * B0[N, K0]^T + A1[M, K1] * B1[N, K1]^T (+ bias[n])        all operands depth-contiguous
//
// The stream-K kernel (gemm_streamk.h) balances the 157 row tiles of the 9 988 x 256 product over 256 CUs perfectly, but
// every one of its blocks ends inside a tile: 33 MB of partial slabs and a 10 us fix-up launch behind a 77 us kernel -- and
// the vendor library does the product in 82 us with whole tiles of 256 x 48, 234 of them (profiles/r02_kernel_stats.csv).
// The same idea, taken further: tiles of (16 RB) x 128 on v_mfma_f32_16x16x4_f32, RB chosen on the host so that
// ceil(M / 16 RB) * ceil(N / 128) is just under a multiple of the CU count -- 80 x 128 gives 125 x 2 = 250 tiles for
// M = 9 988: one tile per CU, the whole depth inside the block, every output written once with its bias, no slabs.
//   4 consumer waves: wave w owns columns [32 w, 32 w + 32) of the tile and all its rows: RB x 2 accumulators of 16 x 16.
//     Lane (r = l & 15, g = l >> 4) reads the 16-byte chunk 4 p + g of row r in pass p; its four floats feed four
//     consecutive MFMAs as k-slot g -- A and B use the same (MFMA, slot) -> depth map, so the sum is complete.
//   4 loader waves (one per SIMD, so that their issue slots are taken evenly from the four MFMA waves): LDS-DMA (SADDR form) of the next stages, three stage buffers, one barrier per stage -- the rhythm of
//     k_gemm_streamk_ld.  LDS image [row][8 chunks of 16 B], chunk XOR (row >> 1) & 7: conflict-free for this read too.
// Host-known M only (the tile height is fitted to it); device-extent launches stay on stream-K.
#pragma once

#include "gemm_streamk.h"

namespace pope {

typedef float f32x4acc __attribute__((ext_vector_type(4)));

constexpr int T16_TN = 128, T16_GK = 32, T16_CONSUMERS = 4, T16_LOADERS = 4, T16_THREADS = (T16_CONSUMERS + T16_LOADERS) * 64;

struct T16Args {
    SkProduct p[2];
    int M, N;
    const float *bias;
    float *C;
    long long ldc;
    int tiles_m, tiles_n, S0, S1;
    const float *zero;
    const int *m_dev;          // device extent: the true row count (<= M, the capacity the grid was sized for), or nullptr
    const long long *rows;     // nullptr, or: row m of the FIRST product's A operand is p[0].A + rows[m] * p[0].lda (the resident
                               // feature matrix read through n_id; byte offsets must fit 32 bits)
    int accumulate;            // 1: C += product (no bias): the second half of a layer whose first half another launch wrote
    // Round 5: the column sums of what this launch leaves in C and of its squares, per row tile, in float64 -- the first stage of the
    // BatchNorm statistics of the layer's output (main.py:207: bns[i](x)), which was a launch of its own reading C back
    // (k_bn_partial).  stat_a / stat_b: [tiles_m, N] doubles each, or nullptr.  Only the launch that writes C's FINAL value gets them.
    double *stat_a, *stat_b;
};

#ifdef POPE_STAMP
#define T16_STAMP(slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_gemm_stamps[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// per-stage trace of the kernel: [block][who: 0 loader wave 0 arrives at B_(s+1), 1 consumer wave 0 arrives, 2 consumer wave 0 leaves][stage]
__device__ unsigned long long g_t16_trace[256 * 4 * 64];                  // (row 3: the shader-clock counter at row 2's instants)
#define T16_TRACE(who, s) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && (s) < 64) g_t16_trace[(blockIdx.x * 4 + (who)) * 64 + (s)] = (who) == 3 ? __builtin_amdgcn_s_memtime() : __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define T16_STAMP(slot) do { } while (0)
#define T16_TRACE(who, s) do { } while (0)
#endif

template <int RB, int NBUF = 3> struct T16Shape {
    static constexpr int TM = 16 * RB;
    static constexpr int A_BYTES = TM * T16_GK * 4, STAGE_BYTES = (TM + T16_TN) * T16_GK * 4, LDS_BYTES = NBUF * STAGE_BYTES;
    static constexpr int A_INSTR = TM / 8, INSTR = (TM + T16_TN) / 8;      // DMA wave-instructions per stage: 8 rows x 8 chunks each
    static constexpr int ND = (INSTR + T16_LOADERS - 1) / T16_LOADERS;     // per loader wave (the last one may have fewer)
};

// One output tile: rows [tm * TM, ...) x columns [tn * 128, ...).  All eight waves of the block call it together.
// Exact f32 products on v_mfma_f32_16x16x4_f32.  (Round 3's opt-in split-bf16 arithmetic -- every f32 operand as three bf16 terms, six
// v_mfma_f32_16x16x32_bf16 per product -- measured 1.13 x, bound by the conversion in the consumer waves, and was removed in round 5:
// DESIGN.md 7h keeps the figures.)
// s_waitcnt vmcnt(n) with a wave-uniform run-time n <= 8 (the instruction takes an immediate)
__device__ __forceinline__ void t16_wait_vmcnt(int n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

// NBUF = 3: the loader waves wait for everything they issued before each barrier -- stage s + 2 is requested behind B_s and must
// have landed by B_(s+1): ONE stage of MFMA time hides the request.  NBUF = 4: stage s + 3 is requested behind B_s and only stage
// s + 2 must have landed by B_(s+1) (s_waitcnt vmcnt(the newest stage's instruction count): loads return in order), so a
// request has TWO stage times.  Stage t lives in buffer t % NBUF either way.
template <int RB, int NBUF = 3>
__device__ __forceinline__ void t16_tile(const T16Args &a, const int M, const int tm, const int tn, char *smem, const unsigned lds0,
                                         const int lane, const int wave) {
    using Sh = T16Shape<RB, NBUF>;
    const int m0 = tm * Sh::TM, n0 = tn * T16_TN;
    const int S = a.S0 + a.S1;

    if (wave >= T16_CONSUMERS) {
        // ---------------- loader waves ----------------
        __builtin_amdgcn_s_setprio(3);
        const int lw = wave - T16_CONSUMERS;
        const int sub = lane >> 3, cp = lane & 7;
        unsigned off[2][Sh::ND];
        int koff[Sh::ND];
#pragma unroll
        for (int d = 0; d < Sh::ND; ++d) {
            const int instr = lw * Sh::ND + d;
            const int row = (instr < Sh::A_INSTR ? instr : instr - Sh::A_INSTR) * 8 + sub;
            koff[d] = (cp ^ ((row >> 1) & 7)) * 4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                long long arow = min(m0 + row, M - 1);
                if (q == 0 && a.rows && instr < Sh::A_INSTR) arow = a.rows[arow];
                off[q][d] = instr < Sh::A_INSTR ? (unsigned)((arow * a.p[q].lda + koff[d]) * 4)
                                               : (unsigned)(((long long)min(n0 + row, a.N - 1) * a.p[q].ldb + koff[d]) * 4);
            }
        }
        auto issue_all = [&](int s, int buf) {
            const bool second = s >= a.S0;                             // wave-uniform
            const int PK = second ? a.p[1].K : a.p[0].K;
            const int k0 = (second ? s - a.S0 : s) * T16_GK;
            const float *A = (second ? a.p[1].A : a.p[0].A) + k0, *B = (second ? a.p[1].B : a.p[0].B) + k0;   // SGPR pairs
            if (k0 + T16_GK <= PK) {
#pragma unroll
                for (int d = 0; d < Sh::ND; ++d)
                    if (lw * Sh::ND + d < Sh::INSTR)
                        sk_glds16_saddr(lw * Sh::ND + d < Sh::A_INSTR ? A : B, second ? off[1][d] : off[0][d],
                                        lds0 + buf * Sh::STAGE_BYTES + (lw * Sh::ND + d) * 1024);
            } else {                                                   // depth padding: lanes past the depth read the zero page
#pragma unroll
                for (int d = 0; d < Sh::ND; ++d)
                    if (lw * Sh::ND + d < Sh::INSTR) {
                        const float *src = (const float *)((const char *)(lw * Sh::ND + d < Sh::A_INSTR ? A : B) + (second ? off[1][d] : off[0][d]));
                        if (k0 + koff[d] >= PK) src = a.zero;
                        sk_glds16(src, lds0 + buf * Sh::STAGE_BYTES + (lw * Sh::ND + d) * 1024);
                    }
            }
        };
        int mine = 0;                                              // DMA instructions of this wave per stage
#pragma unroll
        for (int d = 0; d < Sh::ND; ++d) mine += lw * Sh::ND + d < Sh::INSTR ? 1 : 0;
        issue_all(0, 0);
        if (1 < S) issue_all(1, 1);
        if (NBUF == 4 && 2 < S) issue_all(2, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // B_0: the stages issued so far have landed
        int buf = 0;                           // buffer of stage s
        for (int s = 0; s < S; ++s) {
            const int ahead = NBUF - 1;
            const bool more = s + ahead < S;
            if (more) issue_all(s + ahead, buf == 0 ? NBUF - 1 : buf - 1);   // the buffer of stage s - 1: its readers left it before B_s
            if (s + 1 < S) {
                if (NBUF == 4) t16_wait_vmcnt(more ? mine : 0);        // everything but the stage just requested
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lw == 0) T16_TRACE(0, s);
                __builtin_amdgcn_s_barrier();                          // B_(s+1)
            }
            buf = buf == NBUF - 1 ? 0 : buf + 1;
        }
        return;
    }

    // ---------------- consumer waves ----------------
    const int r15 = lane & 15, g = lane >> 4;
    // fragment addresses: row i * 16 + r15 of A, row wave * 32 + t * 16 + r15 of B, 128 bytes per row.  The swizzle of a row is
    // (row >> 1) & 7 -- the multiples of 16 drop out, so ONE value serves every fragment -- and the row offsets differ by
    // compile-time constants: one register each instead of 2 (RB + 2).
    const int swz = (r15 >> 1) & 7;
    const int fa_base = r15 * 128, fb_base = Sh::A_BYTES + (wave * 32 + r15) * 128;
    f32x4acc acc[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] = 0.0f;
    __builtin_amdgcn_s_barrier();              // B_0
    asm volatile("" ::: "memory");
    {
        float4 fa[2][RB], fb[2][2];
        auto read_pass = [&](int b, int p, int set) {
            const char *base = smem + b * Sh::STAGE_BYTES;
            const int c = 4 * p + g;
    #pragma unroll
            for (int i = 0; i < RB; ++i) fa[set][i] = *reinterpret_cast<const float4 *>(base + fa_base + i * 2048 + ((c ^ swz) << 4));
    #pragma unroll
            for (int t = 0; t < 2; ++t) fb[set][t] = *reinterpret_cast<const float4 *>(base + fb_base + t * 2048 + ((c ^ swz) << 4));
        };
        int buf = 0;
        read_pass(0, 0, 0);
        for (int s = 0; s < S; ++s) {
            const bool next = s + 1 < S;
            const int nbuf = buf == NBUF - 1 ? 0 : buf + 1;
    #pragma unroll
            for (int p = 0; p < 2; ++p) {                                  // 16 depth values per pass
                // pass 0 of stage s + 1 (landed since B_s).  Unconditional: behind an `if (next)` the two paths merge in front of this
                // pass's MFMAs and hipcc waits there for the smaller of their counters -- lgkmcnt(0), i.e. for the reads just issued
                // (ISA, round 4): the whole LDS latency once per stage.  Behind the last stage the read fetches a stale buffer, unused.
                if (p == 0) read_pass(buf, 1, 1);
                else read_pass(nbuf, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int i = 0; i < RB; ++i)
    #pragma unroll
                    for (int t = 0; t < 2; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p][i].x, fb[p][t].x, acc[i][t], 0, 0, 0);
    #pragma unroll
                for (int i = 0; i < RB; ++i)
    #pragma unroll
                    for (int t = 0; t < 2; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p][i].y, fb[p][t].y, acc[i][t], 0, 0, 0);
    #pragma unroll
                for (int i = 0; i < RB; ++i)
    #pragma unroll
                    for (int t = 0; t < 2; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p][i].z, fb[p][t].z, acc[i][t], 0, 0, 0);
    #pragma unroll
                for (int i = 0; i < RB; ++i)
    #pragma unroll
                    for (int t = 0; t < 2; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p][i].w, fb[p][t].w, acc[i][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (next) {
                if (wave == 0) T16_TRACE(1, s);
                __builtin_amdgcn_s_barrier();                              // B_(s+1)
                asm volatile("" ::: "memory");
                if (wave == 0) T16_TRACE(2, s);
                if (wave == 0) T16_TRACE(3, s);
            }
            buf = nbuf;
        }
    }
    // C/D layout of the 16 x 16 forms: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + wave * 32 + t * 16 + r15;
        if (n >= a.N) continue;
        float b = a.bias ? a.bias[n] : 0.0f;
        asm volatile("" : "+v"(b));
        float add[RB][4];                                              // what the product is added to: the bias, or (accumulate) what C holds
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + 4 * g + r;
                add[i][r] = a.accumulate ? a.C[(size_t)min(m, M - 1) * a.ldc + n] : b;      // all loads issued before the first store
            }
        double cs = 0.0, cq = 0.0;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + 4 * g + r;
                const float val = acc[i][t][r] + add[i][r];
                if (m < M) a.C[(size_t)m * a.ldc + n] = val;
                if (a.stat_a && m < M) {                               // (wave-uniform pointer test: no cost without statistics)
                    cs += (double)val;
                    cq += (double)val * (double)val;
                }
            }
        if (a.stat_a) {                                                // the four lanes that share column n (g = 0 .. 3), in a fixed order
            cs += __shfl_xor(cs, 16); cq += __shfl_xor(cq, 16);
            cs += __shfl_xor(cs, 32); cq += __shfl_xor(cq, 32);
            if (g == 0) {
                a.stat_a[(size_t)tm * a.N + n] = cs;
                a.stat_b[(size_t)tm * a.N + n] = cq;
            }
        }
    }
}

// The tiles `first`, `first + stride`, ... of the product, one after the other, by the calling block (all eight waves).
template <int RB, int NBUF = 3>
__device__ __forceinline__ void t16_block_loop(const T16Args &a, char *smem, const int first, const int stride) {
    using Sh = T16Shape<RB, NBUF>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem);
    int M = a.M;
    if (a.m_dev) M = min(M, __builtin_amdgcn_readfirstlane(*a.m_dev));
    const int tiles_m = (M + Sh::TM - 1) / Sh::TM;
    bool again = false;
    for (int b = first;; b += stride) {
        // tiles b and b + 8 sit on one XCD (round-robin dispatch, grids are multiples of 16 here): they take the two column
        // tiles of the same rows, so the A rows they both stream come out of that XCD's L2 the second time (speed only)
        int tm, tn;
        if (a.tiles_n == 2) {
            tn = (b >> 3) & 1;
            tm = (b & 7) + 8 * (b >> 4);
            if (8 * (b >> 4) >= tiles_m) break;
        } else {
            tm = b / a.tiles_n;
            tn = b - tm * a.tiles_n;
        }
        if (tm >= tiles_m) {
            if (a.tiles_n == 2) continue;                         // (the paired mapping leaves holes in the last group of 16)
            break;
        }
        if (again) __syncthreads();                               // the previous tile's last stage has been read: its buffers are free
        t16_tile<RB, NBUF>(a, M, tm, tn, smem, lds0, lane, wave);
        again = true;
    }
}

template <int RB, int NBUF = 3>
__global__ __launch_bounds__(T16_THREADS) void k_gemm_tile16(T16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    t16_block_loop<RB, NBUF>(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// The tile height (in 16-row blocks, 3 .. 8) that wastes the least of the chip for this M x N, or 0 if no choice reaches
// `min_util` (the caller then uses stream-K).  Cost model: rounds of `cus` tiles, each as long as its height.
inline int t16_pick_rb(long long M, int N, int cus, double min_util = 0.85, int rb_min = 3) {
    const long long tiles_n = (N + T16_TN - 1) / T16_TN;
    int best = 0;
    double best_util = 0.0;
    for (int rb = rb_min; rb <= 8; ++rb) {
        const long long tiles = ((M + 16 * rb - 1) / (16 * rb)) * tiles_n;
        const long long rounds = (tiles + cus - 1) / cus;
        const double util = (double)M * N / ((double)rounds * cus * 16 * rb * T16_TN);      // useful outputs / outputs the rounds could produce
        // prefer taller tiles at equal utilisation: fewer re-reads of B, longer MFMA runs per fragment
        if (util > best_util + 1e-9 || (util > best_util - 1e-9 && rb > best)) {
            best_util = util;
            best = rb;
        }
    }
    return best_util >= min_util ? best : 0;
}

}  // namespace pope
