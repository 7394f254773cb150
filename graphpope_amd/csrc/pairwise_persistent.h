// Persistent tile kernel of the node2vec-space embedding for depths up to 128 (the node2vec table is [N, 128]).
//
// One block per CU, for the whole call.  The 256 anchor rows of a column group stay RESIDENT in LDS (256 x 128 floats =
// 128 KB, loaded once); the node2vec table streams through a 2 x 16 KB double buffer in tiles of 32 rows; the block's
// LDS is the CU's full 160 KB.  Roles as in the stream-K GEMMs (gemm_streamk.h):
//   waves 0-7  two sets of four consumer waves; wave w of a set owns anchor columns [64 w, 64 w + 64) of the set's tiles:
//              fragments + MFMAs (two 32 x 32 accumulators, 128 MFMAs per tile), then the metric epilogue, the raw stores
//              and the running column min / max -- kept in registers for the whole kernel (a lane always owns the same
//              two columns), written once at the end.  The sets alternate: while one runs the MFMAs of tile i the other
//              runs the epilogue of tile i - 1, so the matrix cores do not idle through the epilogues (one set alone:
//              10 300 cycles of MFMA phase + 5 000 of epilogue and loop top per tile);
//   waves 8-9  LDS-DMA of the next tile (8 wave-instructions each).
// One barrier per step (= one tile's MFMA phase): the DMA waves arrive once the next tile has landed, the MFMA set once it
// has read the last fragment of its tile, the other set once its epilogue is done.
// The feature copy out[:, :F] = x (utils.py:177) is NOT in this kernel: k_copy_features runs beside it, on a side stream,
// in the wave slots and registers this kernel leaves free on every CU (pairwise.hip).  Measured first as two and as four
// extra waves of this block, coupled to the tile cadence by the barrier: 64 KB in flight per CU, 0.9-1.3 TB/s each way,
// the kernel 208 us instead of 83 -- a copy needs more bytes in flight than a block that is sized for MFMA can hold.
// Everything a row contributes to the epilogue beyond the dot product -- (float)|x|^2 and 1 / |x| -- is computed ONCE per
// row, by the DMA wave that brought the row in, and handed over as one float2 through global memory (L2) a step ahead; the
// anchors' norms come from the resident image in the prologue.  As inline arithmetic in the consumers (f64 -> f32, IEEE sqrt
// and division for the 16 rows a lane owns, every wave again) it cost as much as the tile's 128 MFMAs (stamps: 8 400 of
// 17 400 cycles); as a kernel of its own (k_sqnorm, still used by the round-1 kernel) a 45 MB pass per call.
// LDS images are [row][32 chunks of 16 B], chunk index XOR (row & 15): every ds_read_b128 of a fragment is conflict-free
// (16 lanes of a read group sit in 16 different rows: 16 different low chunk bits).  Rows shorter than 128 floats are
// padded with zeros from a zero page (DMA source addresses are per lane).
// Round 1's kernel (one 64 x 128 tile per block, both operands restaged by every block) took 105 us for this product and
// 170 us with the copy inside; it stays as the fallback for depths above 128.
#pragma once

#include "lds_dma.h"   // sk_glds16: the LDS-DMA wave-instruction

// included by pairwise.hip after f32x4v and wave_sqdist

namespace pope {

constexpr int PP_ROWS = 32, PP_COLS = 256, PP_DMAX = 128, PP_CHUNKS = PP_DMAX / 4;
constexpr int PP_B_BYTES = PP_COLS * PP_DMAX * 4, PP_A_BYTES = PP_ROWS * PP_DMAX * 4;
constexpr int PP_LDS_BYTES = PP_B_BYTES + 2 * PP_A_BYTES;                 // 160 KB
constexpr int PP_THREADS = 640, PP_MAX_GRID = 1024;

#ifdef POPE_STAMP
// Diagnostic build only (make stamp, tools/stamp_pairwise.py): shader-clock stamps of every block's sixth tile: [block][wave][slot].
__device__ unsigned long long g_pp_stamps[256 * 8 * 8];
#define PP_STAMP(slot)                                                                                  \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if ((tile == (int)blockIdx.x + 5 * (int)gridDim.x || tile == (int)blockIdx.x + 6 * (int)gridDim.x) && lane == 0 && blockIdx.x < 256 && (wave < 4 || wave >= 8) == (tile == (int)blockIdx.x + 6 * (int)gridDim.x)) \
            g_pp_stamps[(blockIdx.x * 8 + (wave & 7)) * 8 + (slot)] = __builtin_amdgcn_s_memtime();    \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#else
#define PP_STAMP(slot) do { } while (0)
#endif

struct PpArgs {
    const float *X;              // [N, D] node2vec table
    const float *A;              // [K, D] anchor rows -- or, with anchor_rows, the table itself: anchor j is row anchor_rows[j] of it
    const long long *anchor_rows;
    int N, D, K, metric;
    float2 *xn;                  // [N] scratch: per table row {(float)|row|^2, 1 / |row| (1 for a zero row)}, written by the DMA waves
    float *out;                  // [N, out_cols]; embedding columns start at c0; N * out_cols * 4 < 2^32 (32-bit store offsets)
    unsigned out_cols;
    int c0;
    float *part_min, *part_max;  // [sets * gridDim.x][Kpad]: one row per block and consumer set
    int Kpad;
    const float *zero;
    int sets;                    // consumer sets: 2, or 1 when the feature copy runs beside this kernel (its waves need the registers)
};

// Anchor j is row ids[j] of the table (device int64 supplied by the caller, engine._anchor_ids checks ids that come from the
// host).  An id outside [0, N) is clamped into the table: the result for that column is meaningless, but the kernel never
// reads outside the caller's allocation (a faulting access can take the whole node down).
__device__ __forceinline__ size_t anchor_row(const long long *ids, int j, int N) {
    const long long r = ids[j];
    return (size_t)(r < 0 ? 0 : (r >= N ? N - 1 : r));
}

__global__ __launch_bounds__(PP_THREADS) void k_pairwise_persistent(PpArgs args) {
    // plain locals: a lambda that captured the argument struct by reference would force a copy of it into scratch memory
    const float *const X = args.X, *const A = args.A, *const zero = args.zero;
    const long long *const arows = args.anchor_rows;                  // utils.py:167 embedding[anchor_nodes]: read through the ids, no gathered copy
    const int N = args.N, D = args.D, K = args.K, metric = args.metric, c0 = args.c0, Kpad = args.Kpad;
    float2 *const xn = args.xn;
    float *const out = args.out, *const part_min = args.part_min, *const part_max = args.part_max;
    const unsigned out_cols = args.out_cols;
    const int sets = args.sets;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem);
    const int col_base = blockIdx.y * PP_COLS;
    const int n_tiles = (N + PP_ROWS - 1) / PP_ROWS;
    const int passes = (D + 7) / 8;                                   // 8 depth values per pass
#ifdef POPE_STAMP
    if (lane == 0 && wave == 0 && blockIdx.x < 256) g_pp_stamps[(blockIdx.x * 8) * 8 + 6] = __builtin_amdgcn_s_memtime();     // kernel entry
#endif

    // ---- the anchor image, once: 256 rows x 32 chunks = 128 DMA wave-instructions ----
    for (int i = wave; i < PP_COLS * PP_CHUNKS / 64; i += PP_THREADS / 64) {
        const int row = i * 2 + (lane >> 5), cslot = lane & 31;
        const int c = cslot ^ (row & 15);
        const int col = col_base + row;
        const float *src = (col < K && c * 4 < D) ? A + (arows ? anchor_row(arows, col, N) : (size_t)col) * D + c * 4 : zero;
        sk_glds16(src, lds0 + i * 1024);
    }

    const int my_tiles = blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;     // this block's tiles
    const int steps = sets == 2 ? my_tiles + 1 : my_tiles;          // barriers every wave passes after the first one
    if (wave >= 8) {
        // ---------------- DMA waves: in step s the tile of step s + 1 ----------------
        const int lw = wave - 8;
        auto issue_tile = [&](int tile, int buf) {
            const int row0 = tile * PP_ROWS;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int instr = lw * 8 + d;                            // rows 2 instr, 2 instr + 1
                const int row = instr * 2 + (lane >> 5), cslot = lane & 31;
                const int c = cslot ^ (row & 15);
                const int grow = min(row0 + row, N - 1);                 // rows past N: a valid row, masked in the epilogue
                const float *src = c * 4 < D ? X + (size_t)grow * D + c * 4 : zero;
                sk_glds16(src, lds0 + PP_B_BYTES + buf * PP_A_BYTES + instr * 1024);
            }
        };
        // The norms of a landed tile's rows (sklearn row_norms on the upcast chunk: f64 accumulation), read back from LDS by the
        // wave that brought the tile in: 4 lanes per row, 8 chunks each.  A pass of its own over the table (k_sqnorm) was 9 us
        // and 45 MB per call; here it is ~100 instructions per tile in waves that otherwise wait.
        auto tile_norms = [&](int tile, int buf) {
            const int row = lw * 16 + (lane >> 2), sub = lane & 3;
            const char *base = smem + PP_B_BYTES + buf * PP_A_BYTES + row * (PP_CHUNKS * 16);
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 v = *reinterpret_cast<const float4 *>(base + (((sub * 8 + k) ^ (row & 15)) << 4));
                acc += (double)v.x * (double)v.x;
                acc += (double)v.y * (double)v.y;
                acc += (double)v.z * (double)v.z;
                acc += (double)v.w * (double)v.w;
            }
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            const int grow = tile * PP_ROWS + row;
            if (sub == 0 && grow < N) xn[grow] = make_float2((float)acc, acc == 0.0 ? 1.0f : 1.0f / sqrtf((float)acc));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // in L2 before the barrier that lets a consumer set load them
        };
        int tile = blockIdx.x, buf = 0;
        if (tile < n_tiles) issue_tile(tile, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the anchor image and the first tile have landed
        if (tile < n_tiles) tile_norms(tile, 0);
        __builtin_amdgcn_s_barrier();                                    // B_0
        for (int s = 0; s < steps; ++s, tile += gridDim.x) {
            const int next = tile + gridDim.x;
            PP_STAMP(0);
            if (next < n_tiles) issue_tile(next, buf ^ 1);               // the buffer of the tile of step s - 1: its readers left it before that step's barrier
            PP_STAMP(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the next tile has landed
            if (next < n_tiles) tile_norms(next, buf ^ 1);
            PP_STAMP(2);
            __builtin_amdgcn_s_barrier();                                // end of step s
            PP_STAMP(3);
            buf ^= 1;
        }
        return;
    }

    // ---------------- consumer waves: set 0 (waves 0-3) takes the block's even tiles, set 1 (waves 4-7) the odd ones ----------------
    const int set = wave >> 2, cw = wave & 3;
    if (set >= sets) return;                                             // one set only: these waves' registers go to the copy kernel beside this one
    const int g = lane >> 5, l31 = lane & 31;
    const float inf = __builtin_huge_valf();
    // this lane's two columns for the whole kernel
    int col[2];
    bool col_ok[2];
    float a2f[2], rna[2], cmin[2], cmax[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        col[t] = col_base + cw * 64 + t * 32 + l31;
        col_ok[t] = col[t] < K;
        cmin[t] = inf;
        cmax[t] = -inf;
    }
    const bool cols_full = col_base + cw * 64 + 64 <= K;
    const char *Bimg = smem;
    const int fa_off = (l31 * PP_CHUNKS) * 16, fa_swz = l31 & 15;        // row l31 of the tile
    int fb_off[2], fb_swz[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int brow = cw * 64 + t * 32 + l31;
        fb_off[t] = brow * PP_CHUNKS * 16;
        fb_swz[t] = brow & 15;
    }
    const unsigned pitch = out_cols * 4u;                                // bytes per output row
    const float *const xn_sel = reinterpret_cast<const float *>(xn) + (metric == POPE_METRIC_EUCLIDEAN ? 0 : 1);   // written by this block's DMA waves one step ahead
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this wave's share of the anchor image
    __builtin_amdgcn_s_barrier();                                        // B_0
    asm volatile("" ::: "memory");
#ifdef POPE_STAMP
    if (lane == 0 && wave == 0 && blockIdx.x < 256) g_pp_stamps[(blockIdx.x * 8) * 8 + 7] = __builtin_amdgcn_s_memtime();     // past B_0
#endif
    // the norms of this lane's two anchor rows, from the resident image: its half of the chunks (the ones its fragments use), f64
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        double acc = 0.0;
        for (int p = 0; p < passes; ++p) {
            const float4 v = *reinterpret_cast<const float4 *>(Bimg + fb_off[t] + (((2 * p + g) ^ fb_swz[t]) << 4));
            acc += (double)v.x * (double)v.x;
            acc += (double)v.y * (double)v.y;
            acc += (double)v.z * (double)v.z;
            acc += (double)v.w * (double)v.w;
        }
        acc += __shfl_xor(acc, 32);
        a2f[t] = (float)acc;
        rna[t] = acc == 0.0 ? 1.0f : 1.0f / sqrtf((float)acc);     // columns past K are zero rows of the image: (0, 1)
    }
    if (col_base + cw * 64 >= K) {                                       // K <= 192 within this group: nothing to compute, keep the barriers
        for (int s = 0; s < steps; ++s) __builtin_amdgcn_s_barrier();
        return;
    }
    // Step s: the set s & 1 runs the MFMAs of the block's s-th tile (LDS buffer s & 1) while the other set runs the epilogue
    // of tile s - 1 -- the matrix cores never wait for an epilogue.  Every wave passes one barrier per step, my_tiles + 1 in all.
    // (One set: MFMA phase, barrier, epilogue, tile after tile -- one barrier per tile.)
    int steps_left = steps;
    if (set == 1) {                                                      // step 0 belongs to set 0
        __builtin_amdgcn_s_barrier();
        --steps_left;
    }
    int buf = set;
    for (int tile = blockIdx.x + set * gridDim.x; tile < n_tiles; tile += sets * gridDim.x) {
        const int row0 = tile * PP_ROWS;
        const char *const Abuf = smem + PP_B_BYTES + buf * PP_A_BYTES;
        if (sets == 1) buf ^= 1;
        PP_STAMP(0);
        // the norms of the 16 rows this lane holds outputs of: issued ahead of the MFMAs, consumed after them
        // (one load per tile row and 16 lane permutes in the epilogue instead: the MFMA phase 300 cycles shorter, the epilogue 4 000 longer)
        float xr[16];                                                    // |x|^2 (euclidean) or 1 / |x| (cosine): the component the metric uses
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = xn_sel[2 * min(row0 + (r & 3) + 8 * (r >> 2) + 4 * g, N - 1)];
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        float4 fa[2], fb[2][2];
        auto read_pass = [&](int p, int set) {
            const int c = 2 * p + g;
            fa[set] = *reinterpret_cast<const float4 *>(Abuf + fa_off + ((c ^ fa_swz) << 4));
#pragma unroll
            for (int t = 0; t < 2; ++t) fb[t][set] = *reinterpret_cast<const float4 *>(Bimg + fb_off[t] + ((c ^ fb_swz[t]) << 4));
        };
        read_pass(0, 0);
        for (int p = 0; p < passes; p += 2) {                            // two passes per trip: the register sets alternate statically
            read_pass(p + 1 < passes ? p + 1 : p, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0].x, fb[t][0].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0].y, fb[t][0].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0].z, fb[t][0].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0].w, fb[t][0].w, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p + 1 < passes) {
                read_pass(p + 2 < passes ? p + 2 : p + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1].x, fb[t][1].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1].y, fb[t][1].y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1].z, fb[t][1].z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1].w, fb[t][1].w, acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the last fragment of this tile has been read: the DMA waves may refill this buffer in the step after next
        PP_STAMP(1);
        __builtin_amdgcn_s_barrier();                                    // end of this tile's MFMA step
        PP_STAMP(2);
        asm volatile("" ::: "memory");

        // ---- metric epilogue (the arithmetic of k_pairwise, see there), raw stores, running column min / max.
        // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  FULL: all 32 rows and this
        // wave's 64 columns exist -- no per-output predicate (every tile but the last when K is a multiple of 64).
        auto epilogue = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float e[16];
                if (metric == POPE_METRIC_EUCLIDEAN) {
                    float margin = inf;                                   // min over the outputs of d2 - 1e-2 (|x|^2 + |a|^2): negative = digits lost
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float norms = xr[r] + a2f[t];
                        e[r] = fmaf(-2.0f, acc[t][r], norms);             // squared distance
                        const bool live = FULL || (row0 + (r & 3) + 8 * (r >> 2) + 4 * g < N && col_ok[t]);
                        if (live) margin = fminf(margin, fmaf(-1e-2f, norms, e[r]));
                    }
                    // cancellation (an anchor against its own row, coincident rows): recompute exactly as a sum of squared
                    // differences.  Rare: ONE wave-wide test per 32 x 32 tile, then the whole wave does each flagged output together.
                    if (__any(margin < 0.0f)) {
                        for (int r = 0; r < 16; ++r) {
                            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * g;
                            float er = 0.0f, nr = 0.0f;
#pragma unroll
                            for (int q = 0; q < 16; ++q)
                                if (q == r) { er = e[q]; nr = xr[q] + a2f[t]; }
                            unsigned long long fix = __ballot((FULL || (row < N && col_ok[t])) && er < 1e-2f * nr);
                            while (fix) {
                                const int src = __ffsll((long long)fix) - 1;
                                fix &= fix - 1;
                                const int acol = __shfl(col[t], src);
                                const float v = wave_sqdist(X + (size_t)__shfl(row, src) * D, A + (arows ? anchor_row(arows, acol, N) : (size_t)acol) * D, D, lane);
                                if (lane == src) {
#pragma unroll
                                    for (int q = 0; q < 16; ++q)
                                        if (q == r) e[q] = v;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_sqrtf(fmaxf(e[r], 0.0f));   // v_sqrt_f32 (1 ulp)
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float sim = acc[t][r] * xr[r] * rna[t];
                        e[r] = metric == POPE_METRIC_COSINE_SIMILARITY ? sim : fminf(fmaxf(1.0f - sim, 0.0f), 2.0f);
                    }
                }
                const unsigned lane_off = ((unsigned)(row0 + 4 * g) * out_cols + (unsigned)(c0 + col[t])) * 4u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2);
                    if (FULL || (row0 + m + 4 * g < N && col_ok[t])) {
                        *reinterpret_cast<float *>(reinterpret_cast<char *>(out) + (lane_off + (unsigned)m * pitch)) = e[r];
                        cmin[t] = fminf(cmin[t], e[r]);
                        cmax[t] = fmaxf(cmax[t], e[r]);
                    }
                }
                PP_STAMP(3 + t);
            }
        };
        // The other set's MFMA wave on this SIMD needs one issue slot per 64 cycles, this epilogue ~1 000 of them: ahead of the
        // MFMA stream in priority it still leaves that stream full.
        __builtin_amdgcn_s_setprio(3);
        if (cols_full && row0 + PP_ROWS <= N) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        __builtin_amdgcn_s_setprio(0);
        if (sets == 2) __builtin_amdgcn_s_barrier();                     // end of this tile's epilogue step
        asm volatile("" ::: "memory");
        steps_left -= sets;
    }
    for (; steps_left > 0; --steps_left) __builtin_amdgcn_s_barrier();
#ifdef POPE_STAMP
    if (lane == 0 && wave == 0 && blockIdx.x < 256) g_pp_stamps[(blockIdx.x * 8) * 8 + 5] = __builtin_amdgcn_s_memtime();     // all steps done
#endif
    // this block's column minima / maxima: lanes l and l + 32 hold the same column
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float mn = fminf(cmin[t], __shfl_xor(cmin[t], 32)), mx = fmaxf(cmax[t], __shfl_xor(cmax[t], 32));
        if (lane < 32 && col_ok[t]) {
            part_min[(size_t)(blockIdx.x * sets + set) * Kpad + col[t]] = mn;
            part_max[(size_t)(blockIdx.x * sets + set) * Kpad + col[t]] = mx;
        }
    }
}

}  // namespace pope
