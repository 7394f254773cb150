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
// Staging is direct-to-LDS (global_load_lds_dwordx4: no staging registers, no LDS store instructions), double
// buffered, one barrier per stage.  The LDS image is [row][8 chunks of 16 B] with the chunk index XOR-swizzled by
// (row >> 1) & 7 -- applied on the SOURCE address of the DMA (its destination is lane-linear) and again on the read --
// which makes every ds_read_b128 of the fragment loads conflict-free.  A lane's four depth values per 16-byte read feed
// four consecutive MFMAs: lanes 0-31 own depth chunk 2p, lanes 32-63 chunk 2p + 1 (the two k-slots of
// v_mfma_f32_32x32x2_f32), so one pass p covers 8 consecutive depth values.
#pragma once

#include "gemm_tile.h"

namespace pope {

constexpr int SK_TM = 64, SK_TN = 256, SK_GK = 32, SK_THREADS = 512;
constexpr int SK_A_BYTES = SK_TM * SK_GK * 4, SK_B_BYTES = SK_TN * SK_GK * 4, SK_STAGE_BYTES = SK_A_BYTES + SK_B_BYTES;
constexpr int SK_LDS_BYTES = 2 * SK_STAGE_BYTES;
constexpr size_t SK_SLAB_FLOATS = (size_t)SK_TM * SK_TN;        // one partial tile

static __device__ __attribute__((aligned(16))) float g_sk_zero[4];   // never written: the source of depth padding

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
};

// One LDS-DMA wave-instruction: lane l's 16 bytes at `src` land at LDS byte address lds_wave_base + 16 * l.
// Inline asm on purpose: beside a __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) waits vmcnt(0) in front of EVERY
// ds_read -- it cannot tell the buffer being filled from the buffer being read -- which serialises the prefetch with the
// MFMAs of the current stage.  The asm form is invisible to that bookkeeping; the stage loop waits for it itself
// (s_waitcnt vmcnt(0) + barrier, one stage later).  M0 holds the LDS base and is compiler-reserved: saved, written and
// restored inside the one statement (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void sk_glds16(const float *src, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_wave_base) : "memory");
}

// First unit of block b when T units are dealt to G blocks: block b owns [sk_lo(b), sk_lo(b + 1)).
__host__ __device__ __forceinline__ long long sk_lo(long long b, long long T, long long G) { return b * T / G; }

__global__ __launch_bounds__(SK_THREADS) void k_gemm_streamk(SkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;                           // 2 x 4 waves: rows wm * 32, columns wn * 64
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;   // LDS byte address of the staging buffers
    const int S = a.S0 + a.S1;
    const long long T = (long long)a.tiles_m * a.tiles_n * S, G = gridDim.x;
    long long u = sk_lo(blockIdx.x, T, G);
    const long long u_end = sk_lo(blockIdx.x + 1, T, G);

    // staging geometry of this lane: one DMA wave-instruction moves 8 rows x 8 chunks; wave w issues the A rows
    // [8w, 8w + 8) and the B rows [32w, 32w + 32)
    const int sub = lane >> 3, cp = lane & 7;
    const int a_row = wave * 8 + sub;
    const int a_koff = (cp ^ ((a_row >> 1) & 7)) * 4;
    int b_row[4], b_koff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b_row[j] = (wave * 4 + j) * 8 + sub;
        b_koff[j] = (cp ^ ((b_row[j] >> 1) & 7)) * 4;
    }
    // fragment geometry
    const int g = lane >> 5;
    const int fa_row = wm * 32 + (lane & 31), fa_swz = (fa_row >> 1) & 7;
    int fb_row[2], fb_swz[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        fb_row[t] = wn * 64 + t * 32 + (lane & 31);
        fb_swz[t] = (fb_row[t] >> 1) & 7;
    }

    while (u < u_end) {
        const int tile = (int)(u / S);
        const int s_begin = (int)(u - (long long)tile * S);
        const long long left = u_end - u;
        const int s_end = (long long)(S - s_begin) <= left ? S : s_begin + (int)left;
        const int m0 = (tile % a.tiles_m) * SK_TM, n0 = (tile / a.tiles_m) * SK_TN;

        auto issue = [&](int s, int buf) {
            const bool second = s >= a.S0;                             // wave-uniform: scalar selects, no indexed struct
            const float *PA = second ? a.p[1].A : a.p[0].A, *PB = second ? a.p[1].B : a.p[0].B;
            const long long lda = second ? a.p[1].lda : a.p[0].lda, ldb = second ? a.p[1].ldb : a.p[0].ldb;
            const int PK = second ? a.p[1].K : a.p[0].K;
            const int k0 = (second ? s - a.S0 : s) * SK_GK;
            const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + buf * SK_STAGE_BYTES + wave * 1024);
            {
                const int r = min(m0 + a_row, a.M - 1), k = k0 + a_koff;
                sk_glds16(k < PK ? PA + (long long)r * lda + k : g_sk_zero, base);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = min(n0 + b_row[j], a.N - 1), k = k0 + b_koff[j];
                sk_glds16(k < PK ? PB + (long long)r * ldb + k : g_sk_zero, base + SK_A_BYTES + (wave * 3 + j) * 1024);
            }
        };

        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

        __syncthreads();                       // the previous segment's fragment reads are done: buffer 0 may be refilled
        issue(s_begin, 0);
        for (int s = s_begin; s < s_end; ++s) {
            const int buf = (s - s_begin) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of stage s has landed ...
            __syncthreads();                                           // ... and everybody's; buffer buf ^ 1 is no longer read
            if (s + 1 < s_end) issue(s + 1, buf ^ 1);                  // in flight underneath this stage's MFMAs
            const char *base = smem + buf * SK_STAGE_BYTES;
            float4 fa[4], fb[2][4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int c = 2 * p + g;
                fa[p] = *reinterpret_cast<const float4 *>(base + (fa_row * 8 + (c ^ fa_swz)) * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    fb[t][p] = *reinterpret_cast<const float4 *>(base + SK_A_BYTES + (fb_row[t] * 8 + (c ^ fb_swz[t])) * 16);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p].x, fb[t][p].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p].y, fb[t][p].y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p].z, fb[t][p].z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p].w, fb[t][p].w, acc[t], 0, 0, 0);
                }
            }
        }

        // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        const bool complete = s_begin == 0 && s_end == S;
        if (complete) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = n0 + wn * 64 + t * 32 + (lane & 31);
                if (n >= a.N) continue;
                const float b = a.bias ? a.bias[n] : 0.0f;
                float v[16];                                         // bias added ahead of the masked stores: a wait for the bias
#pragma unroll                                                       // load inside every store branch would also wait for the stores
                for (int r = 0; r < 16; ++r) v[r] = acc[t][r] + b;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < a.M) a.C[(size_t)m * a.ldc + n] = v[r];
                }
            }
        } else {
            // first segment of the block (the tile began in an earlier block): slot 0; a tile this block begins: slot 1
            float *dst = a.slab + ((size_t)blockIdx.x * 2 + (s_begin > 0 ? 0 : 1)) * SK_SLAB_FLOATS;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = wn * 64 + t * 32 + (lane & 31);
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
constexpr int SK_FIX_PARTS = 8;

__global__ __launch_bounds__(256) void k_streamk_fixup(SkArgs a, int G) {
    const int S = a.S0 + a.S1;
    const long long T = (long long)a.tiles_m * a.tiles_n * S;
    const int tile = blockIdx.x;
    const long long u0 = (long long)tile * S, u1 = u0 + S;
    const int b_lo = (int)(((u0 + 1) * G - 1) / T), b_hi = (int)((u1 * G - 1) / T);       // owners of the first and the last unit
    if (b_lo == b_hi) return;                                                             // written whole by its one block
    const int m0 = (tile % a.tiles_m) * SK_TM, n0 = (tile / a.tiles_m) * SK_TN;
    // the partial slabs of this tile (wave-uniform): blocks b_lo .. b_hi, in practice two or three of them; a block with
    // no unit of the tile (possible only when there are fewer units than blocks) contributes nothing
    const float *part[4];
    bool has[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = b_lo + i;
        const long long lo = sk_lo(b, T, G), hi = sk_lo(b + 1, T, G);
        const long long sb = lo > u0 ? lo : u0, se = hi < u1 ? hi : u1;
        has[i] = b <= b_hi && sb < se;
        part[i] = a.slab + ((size_t)b * 2 + (sb > u0 ? 0 : 1)) * SK_SLAB_FLOATS;
    }
    const bool few = b_hi - b_lo < 4;
    constexpr int PER = SK_TM * SK_TN / 4 / SK_FIX_PARTS;                                 // 16-byte pieces per block
    for (int q = blockIdx.y * PER + threadIdx.x; q < (blockIdx.y + 1) * PER; q += blockDim.x) {
        const int m = q / (SK_TN / 4), n = (q % (SK_TN / 4)) * 4;
        if (m0 + m >= a.M || n0 + n >= a.N) continue;
        const size_t off = (size_t)m * SK_TN + n;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (few) {
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = has[i] ? *reinterpret_cast<const float4 *>(part[i] + off) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
        } else {                                                                          // a tile cut into many pieces (tiny grids only)
            for (int b = b_lo; b <= b_hi; ++b) {
                const long long lo = sk_lo(b, T, G), hi = sk_lo(b + 1, T, G);
                const long long sb = lo > u0 ? lo : u0, se = hi < u1 ? hi : u1;
                if (sb >= se) continue;
                const float4 v = *reinterpret_cast<const float4 *>(a.slab + ((size_t)b * 2 + (sb > u0 ? 0 : 1)) * SK_SLAB_FLOATS + off);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
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
