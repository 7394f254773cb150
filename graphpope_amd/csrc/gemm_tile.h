// Exact-f32 MFMA tile machinery shared by the SAGEConv projections (sage.hip) and the pairwise kernel (pairwise.hip).
//
// v_mfma_f32_32x32x2_f32: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the accumulator tile
// has col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  A block of 4 waves (WM x WN) owns a
// TM x TN output tile; a wave owns 32 rows x (TN / WN) columns = NT accumulators.  The depth is walked in stages of
// GK = 64: the next stage's global loads are issued before the current stage's MFMAs and written to LDS after them
// (register double buffering, one LDS image); LDS fragments are read in batches ahead of the MFMAs that use them.
#pragma once

#include "common.h"

namespace pope {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef POPE_STAMP
// Diagnostic build only: per-wave timestamps (100 MHz ticks) around stage 2 of mfma_accumulate.
__device__ unsigned long long g_gemm_stamps[8192 * 8];
#define GSTAMP(slot)                                                                                     \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (s == 2 && (threadIdx.x & 63) == 0) {                                                         \
            const int gw_ = ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + (threadIdx.x >> 6)); \
            if (gw_ < 8192) g_gemm_stamps[gw_ * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();         \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define GSTAMP(slot) do { } while (0)
#endif

constexpr int GK = 64, GLD = GK + 1;          // depth per LDS stage, padded leading dimension

struct Operand {            // element (outer index i, depth k) lives at p[i * s_outer + k * s_k]
    const float *p;
    long long s_outer, s_k;
};

// A [ROWS x GK] slice of an operand (outer range [o0, o0 + ROWS), depth [k0, k0 + GK)) travels global -> registers ->
// LDS in two steps so that the global loads of stage s+1 are in flight while the MFMAs of stage s run.
// LDS image, chosen by which index is contiguous in memory:
//   depth contiguous  (s_k == 1):      [row][GK + 1]    scalar stores, fragment reads conflict-free
//   outer contiguous  (s_outer == 1):  [k][ROWS + 4]    one 16-byte store per load, fragment reads conflict-free
// How an operand slice is fetched.  The choice is a TEMPLATE parameter: a load inside a run-time if/else makes the
// compiler wait for it at the join (before the next load is issued), which serialises the prefetch (measured: 2.5 us
// per stage instead of one latency).
enum Layout {
    LAYOUT_GENERIC = 0,   // any strides / alignment: clamped scalar loads, image chosen at run time
    LAYOUT_KC_VEC = 1,    // depth contiguous, 16-byte loads along the depth       -> [row][GK + 1] image
    LAYOUT_OC_VEC = 2,    // outer index contiguous, 16-byte loads along the outer  -> [k][ROWS + 4] image
};

// Can `op` (outer extent o_end, depth K) use the vector layouts?
inline Layout pick_layout(const Operand &op, int o_end, int K) {
    const bool aligned = (reinterpret_cast<uintptr_t>(op.p) & 15) == 0;
    if (aligned && op.s_k == 1 && (op.s_outer & 3) == 0 && (K & 3) == 0) return LAYOUT_KC_VEC;
    if (aligned && op.s_outer == 1 && (op.s_k & 3) == 0 && (o_end & 3) == 0) return LAYOUT_OC_VEC;
    return LAYOUT_GENERIC;
}

template <int ROWS>
struct Tile {
    static constexpr int NL = ROWS * GK / 4 / 256;             // float4 loads per thread
    static constexpr int LDK = ROWS + 4;                        // leading dimension of the k-major image
    static constexpr int FLOATS = (ROWS * GLD > GK * LDK) ? ROWS * GLD : GK * LDK;

    // Loads are UNCONDITIONAL (addresses clamped into the operand) so that all of a stage's loads are in flight at once.
    // Rows / columns beyond the matrix read a clamped (duplicate) row whose results the epilogue discards; depth beyond
    // k_end is zeroed in store(), after the MFMAs the loads were hidden behind.
    template <int LAYOUT>
    __device__ static __forceinline__ void load(float4 (&reg)[NL], const Operand &op, int o0, int o_end, int k0, int k_end,
                                                int tid) {
        if constexpr (LAYOUT == LAYOUT_KC_VEC) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int idx = tid + 256 * i, r = idx / (GK / 4), kq = (idx % (GK / 4)) * 4;
                const float *row = op.p + (size_t)min(o0 + r, o_end - 1) * op.s_outer;
                reg[i] = *reinterpret_cast<const float4 *>(row + (k0 + kq < k_end ? k0 + kq : k0));
            }
        } else if constexpr (LAYOUT == LAYOUT_OC_VEC) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int idx = tid + 256 * i, k = idx / (ROWS / 4), rq = (idx % (ROWS / 4)) * 4;
                const float *line = op.p + (size_t)min(k0 + k, k_end - 1) * op.s_k;
                reg[i] = *reinterpret_cast<const float4 *>(line + (o0 + rq < o_end ? o0 + rq : o0));
            }
        } else if (op.s_k == 1) {                               // generic, depth contiguous
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int idx = tid + 256 * i, r = idx / (GK / 4), kq = (idx % (GK / 4)) * 4;
                const float *row = op.p + (size_t)min(o0 + r, o_end - 1) * op.s_outer;
                reg[i].x = row[min(k0 + kq, k_end - 1)];
                reg[i].y = row[min(k0 + kq + 1, k_end - 1)];
                reg[i].z = row[min(k0 + kq + 2, k_end - 1)];
                reg[i].w = row[min(k0 + kq + 3, k_end - 1)];
            }
        } else {                                                // generic, read along the outer index
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int idx = tid + 256 * i, k = idx / (ROWS / 4), rq = (idx % (ROWS / 4)) * 4;
                const float *line = op.p + (size_t)min(k0 + k, k_end - 1) * op.s_k;
                reg[i].x = line[(size_t)min(o0 + rq, o_end - 1) * op.s_outer];
                reg[i].y = line[(size_t)min(o0 + rq + 1, o_end - 1) * op.s_outer];
                reg[i].z = line[(size_t)min(o0 + rq + 2, o_end - 1) * op.s_outer];
                reg[i].w = line[(size_t)min(o0 + rq + 3, o_end - 1) * op.s_outer];
            }
        }
    }

    __device__ static __forceinline__ void store(float *__restrict__ lds, const float4 (&reg)[NL], bool k_contig, int k0,
                                                 int k_end, int tid) {
        if (k_contig) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int idx = tid + 256 * i, r = idx / (GK / 4), kq = (idx % (GK / 4)) * 4;
                float *d = lds + r * GLD + kq;
                d[0] = k0 + kq < k_end ? reg[i].x : 0.0f;
                d[1] = k0 + kq + 1 < k_end ? reg[i].y : 0.0f;
                d[2] = k0 + kq + 2 < k_end ? reg[i].z : 0.0f;
                d[3] = k0 + kq + 3 < k_end ? reg[i].w : 0.0f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int idx = tid + 256 * i, k = idx / (ROWS / 4), rq = (idx % (ROWS / 4)) * 4;
                const bool in = k0 + k < k_end;               // component-wise: a ternary on the float4 struct spills to scratch
                float4 v;
                v.x = in ? reg[i].x : 0.0f;
                v.y = in ? reg[i].y : 0.0f;
                v.z = in ? reg[i].z : 0.0f;
                v.w = in ? reg[i].w : 0.0f;
                *reinterpret_cast<float4 *>(lds + k * LDK + rq) = v;
            }
        }
    }
};

// acc[t] += sum over the depth slices [kb0, ke0) of product 0 and [kb1, ke1) of product 1 of A_p[m0.., :] * B_p[n0.., :]^T.
// LA / LB: Layout of the A and B operands of BOTH products (LAYOUT_GENERIC accepts anything).
template <int TM, int TN, int WM, int WN, int LA, int LB>
__device__ __forceinline__ void mfma_accumulate(f32x16 (&acc)[TN / WN / 32], const Operand &A0, const Operand &B0, int kb0,
                                                int ke0, const Operand &A1, const Operand &B1, int kb1, int ke1, int m0,
                                                int n0, int M, int N, float *__restrict__ As, float *__restrict__ Bs) {
    static_assert(WM * WN == 4 && TM == WM * 32 && TN % (WN * 32) == 0, "tile shape");
    constexpr int NT = TN / WN / 32;
    using TA = Tile<TM>;
    using TB = Tile<TN>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int S0 = ke0 > kb0 ? (ke0 - kb0 + GK - 1) / GK : 0, S1 = ke1 > kb1 ? (ke1 - kb1 + GK - 1) / GK : 0;
    const int S = S0 + S1;
    float4 ra[TA::NL], rb[TB::NL];
    auto fetch = [&](int s) {                                   // global -> registers for stage s
        const bool second = s >= S0;
        const Operand &A = second ? A1 : A0;
        const Operand &B = second ? B1 : B0;
        const int k0 = second ? kb1 + (s - S0) * GK : kb0 + s * GK, ke = second ? ke1 : ke0;
        TA::template load<LA>(ra, A, m0, M, k0, ke, tid);
        TB::template load<LB>(rb, B, n0, N, k0, ke, tid);
    };
    if (S > 0) fetch(0);
    for (int s = 0; s < S; ++s) {
        const bool second = s >= S0;
        const bool a_kc = LA == LAYOUT_KC_VEC || (LA == LAYOUT_GENERIC && (second ? A1 : A0).s_k == 1);
        const bool b_kc = LB == LAYOUT_KC_VEC || (LB == LAYOUT_GENERIC && (second ? B1 : B0).s_k == 1);
        const int sk0 = second ? kb1 + (s - S0) * GK : kb0 + s * GK, ske = second ? ke1 : ke0;
        GSTAMP(0);
        __syncthreads();                                        // previous stage's fragment reads are done
        GSTAMP(1);
        TA::store(As, ra, a_kc, sk0, ske, tid);
        TB::store(Bs, rb, b_kc, sk0, ske, tid);
        GSTAMP(2);
        __syncthreads();
        GSTAMP(3);
        if (s + 1 < S) fetch(s + 1);                            // in flight while the MFMAs below run
        __builtin_amdgcn_sched_barrier(0);
        GSTAMP(4);
        const float *xa = a_kc ? As + (wm * 32 + (lane & 31)) * GLD + (lane >> 5)
                               : As + (lane >> 5) * TA::LDK + wm * 32 + (lane & 31);
        const float *xb = b_kc ? Bs + (wn * (TN / WN) + (lane & 31)) * GLD + (lane >> 5)
                               : Bs + (lane >> 5) * TB::LDK + wn * (TN / WN) + (lane & 31);
        const int ask = a_kc ? 1 : TA::LDK, bsk = b_kc ? 1 : TB::LDK, bst = b_kc ? 32 * GLD : 32;
        // Fragments are read from LDS in batches of KB k-steps into registers BEFORE the MFMAs that use them, and the
        // reads of batch b+1 are issued before the MFMAs of batch b (two register sets): with one wave per SIMD nothing
        // else hides the LDS latency, and a read-wait-MFMA sequence per batch idles the matrix pipe for that latency.
        constexpr int KB = NT <= 2 ? 8 : 4;
        constexpr int NB = GK / 2 / KB;
        float fa[2][KB], fb[2][NT][KB];
        auto read_batch = [&](int b, int set) {
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                fa[set][j] = xa[(b * KB + j) * 2 * ask];
#pragma unroll
                for (int t = 0; t < NT; ++t) fb[set][t][j] = xb[t * bst + (b * KB + j) * 2 * bsk];
            }
        };
        read_batch(0, 0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) read_batch(b + 1, (b + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);                  // the next batch's LDS reads are issued above this line
#pragma unroll
            for (int j = 0; j < KB; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[b & 1][j], fb[b & 1][t][j], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // keep the masking / LDS stores of the prefetched registers BELOW the MFMAs: hoisted above them they wait for
        // the loads that the MFMAs are supposed to hide
        __builtin_amdgcn_sched_barrier(0);
        GSTAMP(5);
    }
}

template <int TM, int TN>
constexpr size_t tile_lds_bytes() { return (size_t)(Tile<TM>::FLOATS + Tile<TN>::FLOATS) * sizeof(float); }

}  // namespace pope
