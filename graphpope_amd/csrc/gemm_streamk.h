// Stream-K f32 MFMA GEMM for the skinny SAGEConv forward product on MI355X (gfx950).
//
//     C[M, N] = A0[M, K0] * B0[N, K0]^T + A1[M, K1] * B1[N, K1]^T (+ bias[n])        all operands depth-contiguous
//
// The forward projection of a sampled block is 9 988 x 256 with a depth of 2 x 756: 157 row tiles of 64 for 256 CUs.
// Any whole-tile decomposition either leaves 40 % of the chip idle (64 x 256 tiles) or re-reads the A operand once
// per tile column (64 x 64 tiles: 3.2x the algorithmic traffic, profiles/r01_sage_counters.json).  Here the work is
// cut into UNITS of (one 64 x 256 tile, one depth stage of 32) and the units -- 157 x 48 of them -- are dealt out
// evenly and contiguously to one persistent block per CU: every CU runs the same number of MFMAs (+-1 stage) and every
// A element is read exactly once.  A block that starts or stops in the middle of a tile writes its partial
// accumulators to a slab of its own; k_streamk_fixup adds the (two or three) partials of each cut tile in block order
// (deterministic, no atomics, no inter-block waiting inside the launch).
//
// Staging is direct-to-LDS (global_load_lds_dwordx4: no staging registers, no LDS store instructions) into three
// rotating buffers (two stages in flight), one barrier per stage.  The LDS image is [row][8 chunks of 16 B] with the chunk index XOR-swizzled by
// (row >> 1) & 7 -- applied on the SOURCE address of the DMA (its destination is lane-linear) and again on the read --
// which makes every ds_read_b128 of the fragment loads conflict-free.  A lane's four depth values per 16-byte read feed
// four consecutive MFMAs: lanes 0-31 own depth chunk 2p, lanes 32-63 chunk 2p + 1 (the two k-slots of
// v_mfma_f32_32x32x2_f32), so one pass p covers 8 consecutive depth values.
#pragma once

#include "gemm_tile.h"
#include "lds_dma.h"

namespace pope {

constexpr int SK_TM = 64, SK_TN = 256, SK_GK = 32;
constexpr int SK_A_BYTES = SK_TM * SK_GK * 4, SK_B_BYTES = SK_TN * SK_GK * 4, SK_STAGE_BYTES = SK_A_BYTES + SK_B_BYTES;
constexpr int SK_NBUF = 3;                                      // 120 KB of LDS: exactly one block per CU, two stages in flight
constexpr int SK_LDS_BYTES = SK_NBUF * SK_STAGE_BYTES;
constexpr size_t SK_SLAB_FLOATS = (size_t)SK_TM * SK_TN;        // one partial tile


#ifdef POPE_STAMP
// Diagnostic build only (make stamp, tools/stamp_streamk.py): shader-clock stamps of one mid-segment stage, waves 0 and 4
// of every block: [block][wave][slot].
__device__ unsigned long long g_sk_stamps[256 * 8 * 8];
#define SK_STAMP(slot)                                                                                  \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if (s == s_begin + 6 && lane == 0 && wave < 8 && blockIdx.x < 256)                              \
            g_sk_stamps[(blockIdx.x * 8 + wave) * 8 + (slot)] = __builtin_amdgcn_s_memtime();          \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#else
#define SK_STAMP(slot) do { } while (0)
#endif

struct SkProduct {
    const float *A, *B;          // A[m * lda + k], B[n * ldb + k]
    long long lda, ldb;
    int K;
};

struct SkArgs {
    SkProduct p[2];
    int M, N;
    const float *bias;
    float *C;
    long long ldc;
    float *slab;                 // [gridDim.x][2][SK_TM][SK_TN]
    int tiles_m, tiles_n, S0, S1;
    const float *zero;           // 16 zero bytes in device memory (&g_sk_zero): as an argument it sits in SGPRs from the start
    const int *m_dev;            // device extent: the true M (<= M, which then is the capacity the launch was sized for), or null
};

// Device extents: a kernel's first statement.  M and tiles_m become the true values; every block derives the same unit range.
__device__ __forceinline__ void sk_resolve(SkArgs &a) {
    if (a.m_dev) {
        a.M = dyn_extent(a.m_dev, a.M);
        a.tiles_m = (a.M + SK_TM - 1) / SK_TM;
    }
}

// First unit of block b when T units are dealt to G blocks: block b owns [sk_lo(b), sk_lo(b + 1)).
__host__ __device__ __forceinline__ long long sk_lo(long long b, long long T, long long G) { return b * T / G; }

// WAVES = 4: one wave per SIMD, each 32 rows x 128 columns (four accumulator tiles);  WAVES = 8: two per SIMD, 32 x 64 each.
// SCHED: 0 = the kernel; 3 / 4 are diagnostics of the stamp build (wrong results): 3 = no DMA inside the stage loop, 4 = no MFMA.
// (Measured, no gain: leaving the instruction order to the compiler, a sched_group_barrier interleave of one MFMA / one DS
//  read / a few VALU, one DMA per MFMA group instead of two or three per pass: 5 600-5 850 cycles per stage every time.)
template <int WAVES, int SCHED>
__global__ __launch_bounds__(WAVES * 64) void k_gemm_streamk(SkArgs a) {
    sk_resolve(a);
    constexpr int WN = WAVES / 2;                       // waves along N (2 along M)
    constexpr int NT = SK_TN / WN / 32;                 // 32 x 32 accumulator tiles per wave
    constexpr int NA = 8 / WAVES, NB = 32 / WAVES;      // DMA wave-instructions per wave and stage: A rows, B rows
    constexpr int ND = NA + NB;
    static_assert(WAVES == 4 || WAVES == 8, "wave layout");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    // Two waves per SIMD (w and w + 4): the matrix pipe is arbitrated by priority, then age, so the younger half only gets
    // what the older half leaves and finishes each stage ~1 000 cycles later; static priority for it evens the two out.
    if constexpr (WAVES == 8) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem);   // LDS byte address of the staging buffers
    const int S = a.S0 + a.S1;
    const long long T = (long long)a.tiles_m * a.tiles_n * S, G = gridDim.x;
    long long u = sk_lo(blockIdx.x, T, G);
    const long long u_end = sk_lo(blockIdx.x + 1, T, G);

    // staging geometry of this lane: one DMA wave-instruction moves 8 rows x 8 chunks of 16 bytes; DMA d of wave w is
    // instruction w * ND + d of the stage's 40: the first 8 instructions carry the A rows, the other 32 the B rows
    const int sub = lane >> 3, cp = lane & 7;
    int d_row[ND], d_koff[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int instr = d < NA ? wave * NA + d : 8 + wave * NB + (d - NA);
        d_row[d] = (d < NA ? instr : instr - 8) * 8 + sub;           // row inside the A tile / the B tile
        d_koff[d] = (cp ^ ((d_row[d] >> 1) & 7)) * 4;
    }
    // fragment geometry
    const int g = lane >> 5;
    const int fa_row = wm * 32 + (lane & 31), fa_swz = (fa_row >> 1) & 7;
    int fb_row[NT], fb_swz[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        fb_row[t] = wn * (SK_TN / WN) + t * 32 + (lane & 31);
        fb_swz[t] = (fb_row[t] >> 1) & 7;
    }

    while (u < u_end) {
        const int tile = (int)(u / S);
        const int s_begin = (int)(u - (long long)tile * S);
        const long long left = u_end - u;
        const int s_end = (long long)(S - s_begin) <= left ? S : s_begin + (int)left;
        const int m0 = (tile % a.tiles_m) * SK_TM, n0 = (tile / a.tiles_m) * SK_TN;

        // Row bases of this lane's DMA sources for both products, once per segment: the per-stage address is then one select
        // and one 64-bit add per DMA (a full row * ld multiply in front of every DMA cost ~500 VALU cycles per stage and wave).
        const float *rbase[2][ND];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int d = 0; d < ND; ++d)
                rbase[q][d] = d < NA ? a.p[q].A + (long long)min(m0 + d_row[d], a.M - 1) * a.p[q].lda + d_koff[d]
                                     : a.p[q].B + (long long)min(n0 + d_row[d], a.N - 1) * a.p[q].ldb + d_koff[d];
        // DMA d of stage s into buffer buf.  Kept as cheap as it can be made -- a select, a 64-bit add, the M0 write and the
        // DMA -- because a wave starts no MFMA while it issues one; the depth-padding test runs only in a product's last stage.
        auto issue = [&](int s, int buf, int d) {
            const bool second = s >= a.S0;                             // wave-uniform
            const int PK = second ? a.p[1].K : a.p[0].K;
            const int k0 = (second ? s - a.S0 : s) * SK_GK;
            const int instr = d < NA ? wave * NA + d : 8 + wave * NB + (d - NA);
            const unsigned dst = lds0 + buf * SK_STAGE_BYTES + instr * 1024;   // uniform: lds0, buf and wave live in SGPRs
            const float *src = (second ? rbase[1][d] : rbase[0][d]) + k0;
            if (k0 + SK_GK > PK) {                                     // uniform branch: only a product's last stage can run past its depth
                if (k0 + d_koff[d] >= PK) src = a.zero;
            }
            sk_glds16(src, dst);
        };
        auto issue_all = [&](int s, int buf) {
#pragma unroll
            for (int d = 0; d < ND; ++d) issue(s, buf, d);
        };

        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

        // Three LDS buffers in rotation, two stages of DMA in flight, ONE barrier per stage:
        //   top of stage s:  wait until this wave's share of stage s has landed (its ND younger DMAs of stage s + 1 may
        //                    still fly: vmcnt(ND)); barrier -> stage s is complete for everybody AND nobody still reads
        //                    stage s - 1, whose buffer is refilled with stage s + 2 -- the DMAs are issued a few at a time
        //                    in the shadow of the passes' MFMAs, not in a burst behind the barrier.
        // (Measured and rejected: waiting for stage s + 1 one barrier early so that its first fragments can be read before
        //  the barrier -- that leaves only ONE stage of DMA in flight and the wait at the barrier grew from 18 % to 23 %.)
        __syncthreads();                       // the previous segment's fragment reads are done: the buffers may be refilled
        issue_all(s_begin, 0);
        if (s_begin + 1 < s_end) issue_all(s_begin + 1, 1);
        int buf = 0;
        for (int s = s_begin; s < s_end; ++s) {
            SK_STAMP(0);
            if (s + 1 < s_end) {
                if constexpr (ND == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            SK_STAMP(1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            SK_STAMP(2);
            const bool more = s + 2 < s_end;
            const int nbuf = buf == 0 ? 2 : buf - 1;                   // (buf + 2) % 3
            const char *base = smem + buf * SK_STAGE_BYTES;
            // fragments of pass p + 1 are read before the MFMAs of pass p are issued (two register sets)
            float4 fa[2], fb[NT][2];
            auto read_pass = [&](int p, int set) {
                const int c = 2 * p + g;
                fa[set] = *reinterpret_cast<const float4 *>(base + (fa_row * 8 + (c ^ fa_swz)) * 16);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    fb[t][set] = *reinterpret_cast<const float4 *>(base + SK_A_BYTES + (fb_row[t] * 8 + (c ^ fb_swz[t])) * 16);
            };
            read_pass(0, 0);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (p + 1 < 4) read_pass(p + 1, (p + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);                     // the next pass's LDS reads are issued above this line
                // 16 groups of NT MFMAs per stage (4 passes x the 4 depth values of a 16-byte fragment)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float av = c == 0 ? fa[p & 1].x : c == 1 ? fa[p & 1].y : c == 2 ? fa[p & 1].z : fa[p & 1].w;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float bv = c == 0 ? fb[t][p & 1].x : c == 1 ? fb[t][p & 1].y : c == 2 ? fb[t][p & 1].z : fb[t][p & 1].w;
                        if constexpr (SCHED != 4) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                        else acc[t][0] += av * bv;                      // diagnostic: no matrix work, the staging alone
                    }
                    if (p == 0 && c == 0) SK_STAMP(3);
                    if (c == 0) {                                      // this pass's DMAs, behind its first group of MFMAs
                        if constexpr (WAVES == 8) {
                            // two waves per SIMD (w and w + 4) take turns: waves 0-3 issue in passes 0-1, waves 4-7 in 2-3
                            if (more && (p >> 1) == (wave >> 2)) {
                                issue(s + 2, nbuf, 2 * (p & 1));
                                issue(s + 2, nbuf, 2 * (p & 1) + 1);
                                if ((p & 1) == 1) issue(s + 2, nbuf, 4);
                            }
                        } else if (more && SCHED != 3) {               // one wave per SIMD: 10 DMAs over the 4 passes (3: diagnostic, none)
                            issue(s + 2, nbuf, p);
                            issue(s + 2, nbuf, 4 + p);
                            if (p < 2) issue(s + 2, nbuf, 8 + p);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            SK_STAMP(4);
            buf = buf == 2 ? 0 : buf + 1;
        }

        // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        const bool complete = s_begin == 0 && s_end == S;
        if (complete) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = n0 + wn * (SK_TN / WN) + t * 32 + (lane & 31);
                if (n >= a.N) continue;
                float b = a.bias ? a.bias[n] : 0.0f;
                asm volatile("" : "+v"(b));      // the bias load is waited for HERE, once: left to the compiler, the wait sinks into
#pragma unroll                                   // every masked store branch below, where vmcnt(0) also waits for the stores before it
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < a.M) a.C[(size_t)m * a.ldc + n] = acc[t][r] + b;
                }
            }
        } else {
            // first segment of the block (the tile began in an earlier block): slot 0; a tile this block begins: slot 1
            float *dst = a.slab + ((size_t)blockIdx.x * 2 + (s_begin > 0 ? 0 : 1)) * SK_SLAB_FLOATS;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = wn * (SK_TN / WN) + t * 32 + (lane & 31);
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

// The same stream-K decomposition with the roles split (in-kernel stamps, tools/stamp_streamk.py: a stage of the
// kernel above takes 5 640 cycles -- 4 100 for its 64 MFMAs per SIMD, 680 at the stage boundary and 850 for the ten
// DMAs each wave issues, which an in-order wave cannot overlap with its own MFMAs): waves 0-3 only read fragments and
// issue MFMAs (one per SIMD, 32 x 128 each), waves 4-7 only move data and join the same one-barrier-per-stage rhythm.
//
// GK = depth of a stage.  32 (the default): three 40 KB buffers, two stages of DMA in flight, 4 790 cycles per stage for
// 4 096 cycles of MFMA.  64 (round 3 experiment, POPE_KNOB_GEMM_TILE = 6): TWO 80 KB buffers (the CU's whole 160 KB), ONE
// stage in flight -- the same prefetch distance in time, since a stage lasts twice as long -- and half as many stage
// boundaries per unit of depth: within 1 % at a depth of 756, slower where the depth pads badly (sage.hip, skl_stage_depth).  A unit of the stream-K deal is one stage
// of one tile either way (SkArgs::S0 / S1 are counted in stages of the kernel's GK; the fix-up only sees units).
constexpr int SKL_CONSUMERS = 4, SKL_LOADERS = 4, SKL_THREADS = (SKL_CONSUMERS + SKL_LOADERS) * 64;

template <int GK> struct SkStage {
    static constexpr int CPR = GK / 4;                                   // 16-byte chunks per image row
    static constexpr int RPI = 64 / CPR;                                 // image rows per DMA wave-instruction (1 KiB)
    static constexpr int A_BYTES = SK_TM * GK * 4, B_BYTES = SK_TN * GK * 4, BYTES = A_BYTES + B_BYTES;
    static constexpr int NBUF = GK == 32 ? 3 : 2;
    static constexpr int LDS_BYTES = NBUF * BYTES;                       // 120 KB / 160 KB
    static constexpr int A_INSTR = SK_TM / RPI, INSTR = (SK_TM + SK_TN) / RPI;   // 8 + 32 = 40 / 16 + 64 = 80 per stage
    static constexpr int PASSES = GK / 8;                                // 8 depth values per fragment pass
    // chunk index XOR that makes every ds_read_b128 of a fragment conflict-free (16 lanes of a read group sit in 16 rows)
    __device__ static __forceinline__ int swz(int row) { return GK == 32 ? ((row >> 1) & 7) : (row & 15); }
};

template <int GK>
__global__ __launch_bounds__(SKL_THREADS) void k_gemm_streamk_ld(SkArgs a) {
    sk_resolve(a);
    using St = SkStage<GK>;
    constexpr int NT = 4;                               // 32 x 32 accumulator tiles per consumer wave: 32 rows x 128 columns
    constexpr int ND = St::INSTR / SKL_LOADERS;         // DMA wave-instructions per loader wave and stage
    constexpr int NBUF = St::NBUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= SKL_CONSUMERS;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem);
    const int S = a.S0 + a.S1;
    const long long T = (long long)a.tiles_m * a.tiles_n * S, G = gridDim.x;
    long long u = sk_lo(blockIdx.x, T, G);
    const long long u_end = sk_lo(blockIdx.x + 1, T, G);

    if (loader) {
        // ---------------- loader waves ----------------
        // A loader shares its SIMD with a consumer whose MFMA stream would otherwise take every issue slot first (issue is
        // arbitrated by priority, then age: stamps showed 570 cycles per DMA); its few instructions go ahead of the MFMAs.
        __builtin_amdgcn_s_setprio(3);
        const int lw = wave - SKL_CONSUMERS;
        const int sub = lane / St::CPR, cp = lane % St::CPR;
        while (u < u_end) {
            const int tile = (int)(u / S);
            const int s_begin = (int)(u - (long long)tile * S);
            const long long left = u_end - u;
            const int s_end = (long long)(S - s_begin) <= left ? S : s_begin + (int)left;
            const int m0 = (tile % a.tiles_m) * SK_TM, n0 = (tile / a.tiles_m) * SK_TN;
            // Byte offsets of this lane's ND sources from the tile's A / B origin, for both products (instruction i < A_INSTR:
            // A rows RPI * i .., else B rows); 32 bits are enough (checked on the host: rows * ld * 4 < 2^32).
            unsigned off[2][ND];
            int koff[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int instr = lw * ND + d;
                const int row = (instr < St::A_INSTR ? instr : instr - St::A_INSTR) * St::RPI + sub;
                koff[d] = (cp ^ St::swz(row)) * 4;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    off[q][d] = instr < St::A_INSTR ? (unsigned)(((long long)min(m0 + row, a.M - 1) * a.p[q].lda + koff[d]) * 4)
                                                   : (unsigned)(((long long)min(n0 + row, a.N - 1) * a.p[q].ldb + koff[d]) * 4);
            }
            auto issue_all = [&](int s, int buf) {
                const bool second = s >= a.S0;                         // wave-uniform
                const int PK = second ? a.p[1].K : a.p[0].K;
                const int k0 = (second ? s - a.S0 : s) * GK;
                const float *A = (second ? a.p[1].A : a.p[0].A) + k0, *B = (second ? a.p[1].B : a.p[0].B) + k0;   // SGPR pairs
                if (k0 + GK <= PK) {                                   // every stage but a product's last: no vector arithmetic
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        sk_glds16_saddr(lw * ND + d < St::A_INSTR ? A : B, second ? off[1][d] : off[0][d], lds0 + buf * St::BYTES + (lw * ND + d) * 1024);
                } else {                                               // depth padding: lanes past the depth read the zero page
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        const float *src = (const float *)((const char *)(lw * ND + d < St::A_INSTR ? A : B) + (second ? off[1][d] : off[0][d]));
                        if (k0 + koff[d] >= PK) src = a.zero;
                        sk_glds16(src, lds0 + buf * St::BYTES + (lw * ND + d) * 1024);
                    }
                }
            };
            // Rhythm (one barrier per stage, B_s at the top of stage s).  Three buffers: the loader arrives at B_s once stages
            // <= s + 1 have landed and refills the buffer of stage s - 1 with stage s + 2 behind it.  Two buffers: it arrives once
            // stage s has landed and fills the other buffer with stage s + 1 behind B_s (its readers left it before B_s).
            __syncthreads();                   // the previous segment's fragment reads are done: the buffers may be refilled
            issue_all(s_begin, 0);
            if (NBUF == 3 && s_begin + 1 < s_end) issue_all(s_begin + 1, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // B_(s_begin)
            int buf = 0;
            for (int s = s_begin; s < s_end; ++s) {
                SK_STAMP(0);
                if (NBUF == 3) {
                    if (s + 2 < s_end) issue_all(s + 2, buf == 0 ? 2 : buf - 1);
                } else {
                    if (s + 1 < s_end) issue_all(s + 1, buf ^ 1);
                }
                SK_STAMP(1);
                if (s + 1 < s_end) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    SK_STAMP(2);
                    __builtin_amdgcn_s_barrier();                      // B_(s+1)
                }
                SK_STAMP(3);
                SK_STAMP(4);
                buf = NBUF == 3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
            }
            u += s_end - s_begin;
        }
        return;
    }

    // ---------------- consumer waves: fragments + MFMAs only ----------------
    const int wm = wave & 1, wn = wave >> 1;
    const int g = lane >> 5;
    const int fa_row = wm * 32 + (lane & 31), fa_swz = St::swz(fa_row);
    int fb_row[NT], fb_swz[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        fb_row[t] = wn * 128 + t * 32 + (lane & 31);
        fb_swz[t] = St::swz(fb_row[t]);
    }
    while (u < u_end) {
        const int tile = (int)(u / S);
        const int s_begin = (int)(u - (long long)tile * S);
        const long long left = u_end - u;
        const int s_end = (long long)(S - s_begin) <= left ? S : s_begin + (int)left;
        const int m0 = (tile % a.tiles_m) * SK_TM, n0 = (tile / a.tiles_m) * SK_TN;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        __syncthreads();
        __builtin_amdgcn_s_barrier();          // B_(s_begin): stage s_begin (and s_begin + 1 with three buffers) has landed
        asm volatile("" ::: "memory");
        float4 fa[2], fb[NT][2];
        auto read_pass = [&](int b, int p, int set) {
            const char *base = smem + b * St::BYTES;
            const int c = 2 * p + g;
            fa[set] = *reinterpret_cast<const float4 *>(base + (fa_row * St::CPR + (c ^ fa_swz)) * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                fb[t][set] = *reinterpret_cast<const float4 *>(base + St::A_BYTES + (fb_row[t] * St::CPR + (c ^ fb_swz[t])) * 16);
        };
        int buf = 0;
        read_pass(0, 0, 0);
        for (int s = s_begin; s < s_end; ++s) {
            SK_STAMP(0);
            const bool next = s + 1 < s_end;
            const int nbuf = NBUF == 3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
#pragma unroll
            for (int p = 0; p < St::PASSES; ++p) {
                if (p + 1 < St::PASSES) read_pass(buf, p + 1, (p + 1) & 1);
                else if (NBUF == 3) read_pass(nbuf, 0, 0);             // pass 0 of stage s + 1: landed since B_s (three buffers only); unconditional, see gemm_tile16.h
                __builtin_amdgcn_sched_barrier(0);                     // the next pass's LDS reads are issued above this line
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p & 1].x, fb[t][p & 1].x, acc[t], 0, 0, 0);
                if (p == 0) SK_STAMP(1);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p & 1].y, fb[t][p & 1].y, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p & 1].z, fb[t][p & 1].z, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p & 1].w, fb[t][p & 1].w, acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            SK_STAMP(2);
            if (next) {
                __builtin_amdgcn_s_barrier();                          // B_(s+1)
                asm volatile("" ::: "memory");
                if (NBUF == 2) read_pass(nbuf, 0, 0);                  // two buffers: stage s + 1 is complete only now
            }
            SK_STAMP(3);
            SK_STAMP(4);
            buf = nbuf;
        }
        // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        const bool complete = s_begin == 0 && s_end == S;
        if (complete) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = n0 + wn * 128 + t * 32 + (lane & 31);
                if (n >= a.N) continue;
                float b = a.bias ? a.bias[n] : 0.0f;
                asm volatile("" : "+v"(b));      // the bias load is waited for HERE, once (see k_gemm_streamk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < a.M) a.C[(size_t)m * a.ldc + n] = acc[t][r] + b;
                }
            }
        } else {
            float *dst = a.slab + ((size_t)blockIdx.x * 2 + (s_begin > 0 ? 0 : 1)) * SK_SLAB_FLOATS;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = wn * 128 + t * 32 + (lane & 31);
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

// SK_FIX_PARTS blocks per tile: a tile whose units went to several blocks is the sum of their partial slabs, in block
// order.  Each thread owns two 16-byte pieces and has the loads of all partials in flight together.
constexpr int SK_FIX_PARTS = 8;       // (16 parts, one piece per thread: 10.7 us against 9.7)

__global__ __launch_bounds__(256) void k_streamk_fixup(SkArgs a, int G) {
    sk_resolve(a);
    const int S = a.S0 + a.S1;
    const long long T = (long long)a.tiles_m * a.tiles_n * S;
    const int tile = blockIdx.x;
    if (tile >= a.tiles_m * a.tiles_n || T <= 0) return;                                  // device extents: the grid covers the capacity
    const long long u0 = (long long)tile * S, u1 = u0 + S;
    const int b_lo = (int)(((u0 + 1) * G - 1) / T), b_hi = (int)((u1 * G - 1) / T);       // owners of the first and the last unit
    if (b_lo == b_hi) return;                                                             // written whole by its one block
    const int m0 = (tile % a.tiles_m) * SK_TM, n0 = (tile / a.tiles_m) * SK_TN;
    // the partial slabs of this tile (wave-uniform): blocks b_lo .. b_hi, in practice two or three of them; a block with
    // no unit of the tile (possible only when there are fewer units than blocks) contributes nothing
    constexpr int PER = SK_TM * SK_TN / 4 / SK_FIX_PARTS;                                 // 16-byte pieces per block
    for (int p = blockIdx.y * PER + threadIdx.x; p < (blockIdx.y + 1) * PER; p += blockDim.x) {
        const int m = p / (SK_TN / 4), n = (p % (SK_TN / 4)) * 4;
        if (m0 + m >= a.M || n0 + n >= a.N) continue;
        const size_t off = (size_t)m * SK_TN + n;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // the partial slabs of this tile, blocks b_lo .. b_hi in order, FOUR loads in flight at a time (two or three partials
        // for the forward projection, about eleven per tile for the weight gradients); a block with no unit of the tile
        // (possible only when there are fewer units than blocks) contributes nothing
        for (int b0 = b_lo; b0 <= b_hi; b0 += 4) {
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0 + i;
                const long long lo = sk_lo(b, T, G), hi = sk_lo(b + 1, T, G);
                const long long sb = lo > u0 ? lo : u0, se = hi < u1 ? hi : u1;
                const bool has = b <= b_hi && sb < se;
                v[i] = has ? *reinterpret_cast<const float4 *>(a.slab + ((size_t)b * 2 + (sb > u0 ? 0 : 1)) * SK_SLAB_FLOATS + off)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
        }
        float *c = a.C + (size_t)(m0 + m) * a.ldc + n0 + n;
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n0 + n + i < a.N) c[i] = sv[i] + (a.bias ? a.bias[n0 + n + i] : 0.0f);
    }
}

// Can the stream-K kernel take this product?  (16-byte chunks: aligned bases, leading dimensions and depths multiples of 4.)
inline bool sk_operand_ok(const float *p, long long ld, int K) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0 && (K & 3) == 0;
}

inline size_t sk_slab_bytes(int grid) { return (size_t)grid * 2 * SK_SLAB_FLOATS * sizeof(float); }

}  // namespace pope
