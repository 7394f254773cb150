// Geodesic GraphPOPE embedding on MI355X (gfx950): CSR build, bit-parallel multi-source BFS, finalise.
//
// Replaces /root/reference/utils.py:64-135 (one NetworkX bidirectional BFS per (node, anchor) pair on a
// multiprocessing pool, then 1/len(path), tensor conversion and torch.cat).  Design:
//
//  * anchors are packed 64 per uint64 word; every node carries W words, so one pass over the CSR advances
//    the BFS of all K anchors by one level ("MS-BFS");
//  * the level kernel is BOTTOM-UP (pull): node v ORs the frontier words of its out-neighbours, because
//    hop(v -> anchor) = 1 + min over edges v -> u of hop(u -> anchor).  It is edge-parallel (256 CSR slots per wave);
//    OR is the only combining operation, so the result is independent of scheduling; the only atomics are ORs for the
//    rows that span several waves' chunks.  A one-bit-per-node "live" table (staged in LDS) skips quiet neighbours;
//  * hop counts are stored bit-sliced: plane b gets `new` OR-ed in when bit b of the level is set.  State is
//    a few N*W*8-byte planes that live in L2 / Infinity Cache; the 4*N*K-byte float matrix is written once,
//    coalesced, by the finalise kernel straight into the [N, F+K] output (no transpose, no torch.cat).
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "common.h"
#include "side_copy.h"

// Level-kernel look-ahead, as macros so that tools/ab_variants.sh can build the alternatives side by side (level_expand):
#ifndef POPE_AHEAD
#define POPE_AHEAD 2                  // 0 no look-ahead, 1 indices and live bits of the next chunk, 2 its indices only   (graphs with LIVE >= 2)
#endif
#ifndef POPE_TILE_PREFETCH
#define POPE_TILE_PREFETCH 2          // 0 = the next tile's gathers go out when the tile is reached, 1 = behind this tile's mask loads,
                                      // 2 = tiles in pairs, both gathers of a pair back to back (one fetch of the 128-byte line they share)
#endif
#ifndef POPE_WT8
#define POPE_WT8 1                    // 1 = tiles of 8 words (a node's whole 64-byte row in one gather) where the frontier comes from HBM
#endif
#ifndef POPE_WT8_L2
#define POPE_WT8_L2 1                 // 1 = 8-word tiles (a wave each) on graphs that live in L2 too
#endif
#ifndef POPE_WT8_WAVES
#define POPE_WT8_WAVES 3              // waves per SIMD the compiler must fit the 8-word kernel into (1: its own choice)
#endif
#ifndef POPE_WT8_LOOP_WAVES
#define POPE_WT8_LOOP_WAVES 1         // the same for the 8-word kernel that walks several tiles
#endif
#ifndef POPE_WT8_PREFETCH
#define POPE_WT8_PREFETCH 0           // POPE_TILE_PREFETCH of the 8-word tiles
#endif
#ifndef POPE_NT_INDEX
#define POPE_NT_INDEX 1               // 1 = graphs with LIVE >= 2 read the erow / col index streams with the non-temporal hint
#endif
#ifndef POPE_NT_PLANES
#define POPE_NT_PLANES 0              // 1 = ... and the reachability / hop-bit planes (row masks, the housekeeping's commit) likewise
#endif

namespace pope {

// ------------------------------------------------------------------------------------------------
// CSR build
// ------------------------------------------------------------------------------------------------
enum { CSR_FLAG_BAD_INDEX = 1, CSR_FLAG_UNSORTED = 2 };
constexpr size_t CTL_BYTES = 4096;             // BfsCtl at the start of the BFS scratch
enum { AUX_FLAGS = 2, AUX_HEADER = 16 };
constexpr int SLOTS = 4;                      // CSR slots per lane in the BFS expand kernel
constexpr int CHUNK_SHIFT = 8, CHUNK = 1 << CHUNK_SHIFT;   // slots per wave pass = 64 lanes x SLOTS

// Anchor j starts its BFS at node a: bit j of a's words in the reachability plane and the level-0 frontier, and a's
// live bit.  Atomics: duplicate anchors share a node (distinct bits of the same words).
__device__ __forceinline__ void seed_anchor(long long a, int j, int Wp, u64 *seen, u64 *front, unsigned *live) {
    const size_t idx = (size_t)a * Wp + (j >> 6);
    const u64 bit = 1ull << (j & 63);
    atomicOr(&seen[idx], bit);
    atomicOr(&front[idx], bit);
    atomicOr(&live[a >> 5], 1u << (a & 31));
}

// Fast path, speculative: PyG stores edge_index grouped by source (coalesced), so slot e of the CSR is edge e
// and rowptr is where the source changes.  One streaming pass, no atomics, no scan.  If a pair is out of
// order the flag is raised and the counting path redoes the build.
// The BFS walks the CSR in chunks of CHUNK = 256 slots; a row that spans several chunks is accumulated with atomics
// and committed one level late (k_bfs_level).  aux = header | mrow[chunk]: the row that first continues INTO that
// chunk, or -1 -- written here by the thread that owns the chunk's first slot (fixed position: no counter, no atomics).
// pope_geodesic_run also seeds the BFS from the last block (K > 0): one launch less; the planes were zeroed by the
// launch before this one.
__device__ __forceinline__ int csr_sorted_edge(int e, long long s, long long d, long long prev, const long long *__restrict__ src, int E, int N,
                                               int *__restrict__ rowptr, int *aux) {
    if (s < 0 || s >= N || d < 0 || d >= N) return CSR_FLAG_BAD_INDEX;
    int flags = 0;
    if (prev > s) {
        flags = CSR_FLAG_UNSORTED;
    } else if (prev >= -1 && prev < s) {
        for (long long r = prev + 1; r <= s; ++r) rowptr[r] = e;      // rows prev+1 .. s start here
    }
    if (e == E - 1)
        for (long long r = s + 1; r <= N; ++r) rowptr[r] = E;
    if ((e & (CHUNK - 1)) == 0) {
        const int c = e >> CHUNK_SHIFT;
        // row s runs in from chunk c-1 and its first slot lies there (not further back)
        const bool first_continuation = c > 0 && prev == s && (c == 1 || src[e - CHUNK - 1] != s);
        aux[AUX_HEADER + c] = first_continuation ? (int)s : -1;
    }
    return flags;
}

// PAIRS: a thread takes two consecutive edges with 16-byte loads and one 8-byte store per output array (E even, 16-byte
// aligned halves of edge_index): half the memory instructions of the one-edge form for the same 22 MB.
template <bool PAIRS>
__device__ __forceinline__ int csr_sorted_role(const long long *__restrict__ src, const long long *__restrict__ dst, int E, int N,
                                               int *__restrict__ rowptr, int *__restrict__ col, int *__restrict__ erow, int *aux,
                                               const int bid, const int nblk) {
    int flags = 0;
    if (PAIRS) {
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        for (int t = bid * blockDim.x + threadIdx.x; 2 * t < E; t += nblk * blockDim.x) {
            const int e = 2 * t;
            const ll2 s2 = reinterpret_cast<const ll2 *>(src)[t], d2 = reinterpret_cast<const ll2 *>(dst)[t];
            const long long prev = e > 0 ? src[e - 1] : -1;
            const int f0 = csr_sorted_edge(e, s2.x, d2.x, prev, src, E, N, rowptr, aux);
            const int f1 = csr_sorted_edge(e + 1, s2.y, d2.y, s2.x, src, E, N, rowptr, aux);
            flags |= f0 | f1;
            // a bad id is flagged and the call fails: what lands in its slot does not matter, the pair is stored as one
            reinterpret_cast<int2 *>(col)[t] = make_int2((int)d2.x, (int)d2.y);
            reinterpret_cast<int2 *>(erow)[t] = make_int2((int)s2.x, (int)s2.y);
        }
    } else {
        for (int e = bid * blockDim.x + threadIdx.x; e < E; e += nblk * blockDim.x) {
            const long long s = src[e], d = dst[e];
            const int f = csr_sorted_edge(e, s, d, e > 0 ? src[e - 1] : -1, src, E, N, rowptr, aux);
            flags |= f;
            if (f & CSR_FLAG_BAD_INDEX) continue;
            col[e] = (int)d;
            erow[e] = (int)s;
        }
    }
    return flags;
}

template <bool PAIRS>
__global__ __launch_bounds__(256) void k_csr_sorted(const long long *__restrict__ src,
                                                    const long long *__restrict__ dst, int E, int N,
                                                    int *__restrict__ rowptr, int *__restrict__ col,
                                                    int *__restrict__ erow, int *aux,
                                                    const long long *__restrict__ anchors, int K, int Wp, u64 *seen,
                                                    u64 *front, unsigned *live) {
    if (K > 0 && blockIdx.x == gridDim.x - 1)
        for (int j = threadIdx.x; j < K; j += blockDim.x) seed_anchor(anchors[j], j, Wp, seen, front, live);
    const int flags = csr_sorted_role<PAIRS>(src, dst, E, N, rowptr, col, erow, aux, (int)blockIdx.x, (int)gridDim.x);
    if (flags) atomicOr(&aux[AUX_FLAGS], flags);
}

// General path for edge lists in arbitrary order: histogram, scan, scatter.
__global__ __launch_bounds__(256) void k_csr_count(const long long *__restrict__ src, int E, int *__restrict__ cnt) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x)
        atomicAdd(&cnt[src[e]], 1);
}

__global__ __launch_bounds__(256) void k_csr_scatter(const long long *__restrict__ src,
                                                     const long long *__restrict__ dst, int E,
                                                     const int *__restrict__ rowptr, int *__restrict__ cursor,
                                                     int *__restrict__ col, int *__restrict__ erow) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const int s = (int)src[e];
        const int pos = rowptr[s] + atomicAdd(&cursor[s], 1);
        col[pos] = (int)dst[e];
        erow[pos] = s;
    }
}

__global__ __launch_bounds__(256) void k_index_check(const long long *__restrict__ ids, long long count, long long N, int *flag) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
        bad |= ids[i] < 0 || ids[i] >= N;
    if (bad) atomicOr(flag, 1);
}

// mrow[chunk] for a CSR built by the counting path (same format as k_csr_sorted writes): one thread per chunk.
__global__ __launch_bounds__(256) void k_csr_lists(const int *__restrict__ rowptr, const int *__restrict__ erow,
                                                   int E, int *aux) {
    const int nchunks = (E + CHUNK - 1) >> CHUNK_SHIFT;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += gridDim.x * blockDim.x) {
        int m = -1;
        if (c > 0) {
            const int v = erow[c * CHUNK];
            if (erow[c * CHUNK - 1] == v && (rowptr[v] >> CHUNK_SHIFT) == c - 1) m = v;
        }
        aux[AUX_HEADER + c] = m;
    }
}

static size_t scan_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (int *)nullptr, (int *)nullptr, 0, n, rocprim::plus<int>());
    return bytes;
}

// Row-wise sort of the scattered targets (general CSR path): k_csr_scatter places the edges of a row in the order its
// atomic cursor happened to hand out, which differs from run to run; sorted by target the CSR is a pure function of the
// edge SET (duplicates stay, adjacent), so everything that reads neighbours by position (the fan-out sampler) is repeatable.
static size_t rowsort_temp_bytes(size_t E, size_t N) {
    size_t bytes = 0;
    (void)rocprim::segmented_radix_sort_keys(nullptr, bytes, (const int *)nullptr, (int *)nullptr, (unsigned)E, (unsigned)N,
                                            (const int *)nullptr, (const int *)nullptr);
    return bytes;
}

// ------------------------------------------------------------------------------------------------
// BFS
// ------------------------------------------------------------------------------------------------
struct BfsCtl {          // device control block at the start of the BFS scratch (CTL_BYTES, zeroed by the first launch of a BFS)
    int last_active;     // highest level at which some (node, anchor) pair was newly reached
    int pad0[3];
    unsigned flag_epoch; // the tag the CSR status word must carry to count (csr_flags); 0 = the zeroed word of the separate launches
    int pad[27];
};

// The CSR status word aux[AUX_FLAGS] = (tag << 3) | flags.  The separate launches zero it and OR flags into it (tag 0); the merged
// prepare launch (k_prepare) cannot zero it in front of the blocks that may raise a flag, so those write it whole with the call's
// tag and the readers ignore a word whose tag is not the one the launch left in the control block (an older call's, or whatever
// an uninitialised workspace held).
static_assert(offsetof(BfsCtl, last_active) == 0, "write_report finds the control block through the address of last_active");
__device__ __forceinline__ int csr_flags(const BfsCtl *ctl, const int *aux) {
    const unsigned w = (unsigned)aux[AUX_FLAGS];
    return (w >> 3) == ctl->flag_epoch ? (int)(w & 7u) : 0;
}

__device__ __forceinline__ void csr_raise(int *aux, int flags, unsigned epoch) {
    if (epoch == 0) {
        atomicOr(&aux[AUX_FLAGS], flags);
        return;
    }
    unsigned *p = reinterpret_cast<unsigned *>(&aux[AUX_FLAGS]);
    unsigned old = *p;
    for (int tries = 0; tries < 1 << 20; ++tries) {                  // (bounded: the word is contended by the raising blocks only)
        const unsigned want = (old >> 3) == epoch ? old | (unsigned)flags : (epoch << 3) | (unsigned)flags;
        const unsigned seen = atomicCAS(p, old, want);
        if (seen == old) return;
        old = seen;
    }
}

__device__ __forceinline__ bool bfs_over(const BfsCtl *ctl, const int *aux, int level) {
    // The previous level reached nothing new (every later launch is a no-op), or the CSR is not usable.
    // Plain loads: both words were last written by EARLIER launches (a wave of this launch may be raising
    // last_active to `level` meanwhile, which does not change the verdict).
    // (plain, wave-uniform loads: scalar loads, which do not occupy the vector memory counter the index loads wait on)
    return ctl->last_active < level - 1 || csr_flags(ctl, aux) != 0;
}

// pope_geodesic_run, one launch in front of the levels instead of two (round 4): blocks [0, zero_blocks) clear the BFS state and
// seed it, the others build the speculative CSR.  The two roles share nothing:
//  * the anchors come by value (at most PREP_MAX_ANCHORS), and a seeded word is written by the block that zeroed it -- behind its
//    own stores and a block barrier -- so no seed can meet a later zero;
//  * the CSR status word is not zeroed but tagged (csr_raise / csr_flags); the clear role leaves the tag in the control block.
constexpr int PREP_MAX_ANCHORS = 256;
struct PrepSeeds { int a[PREP_MAX_ANCHORS]; };

template <bool PAIRS>
__global__ __launch_bounds__(256) void k_prepare(const long long *__restrict__ src, const long long *__restrict__ dst, int E, int N,
                                                 int *__restrict__ rowptr, int *__restrict__ col, int *__restrict__ erow, int *aux,
                                                 uint4 *za, size_t na, uint4 *zb, size_t nb, u64 *zb_tail, int zero_blocks, unsigned epoch,
                                                 PrepSeeds seeds, int K, int Wp, u64 *seen, u64 *front, unsigned *live) {
    if ((int)blockIdx.x >= zero_blocks) {
        const int flags = csr_sorted_role<PAIRS>(src, dst, E, N, rowptr, col, erow, aux, (int)blockIdx.x - zero_blocks, (int)gridDim.x - zero_blocks);
        if (flags) csr_raise(aux, flags, epoch);
        return;
    }
    const uint4 z = make_uint4(0, 0, 0, 0);
    const size_t stride = (size_t)zero_blocks * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // region a starts with the control block: its second 16-byte unit holds flag_epoch in its first word
    static_assert(offsetof(BfsCtl, flag_epoch) == 16, "k_prepare writes the tag as the first word of the control block's second unit");
    for (size_t i = first; i < na; i += stride) za[i] = i == 1 ? make_uint4(epoch, 0, 0, 0) : z;
    for (size_t i = first; i < nb; i += stride) zb[i] = z;
    // region b is (1 + eager) planes of N * W words: an odd word count leaves one 8-byte word behind the last 16-byte unit (the last
    // node's word of the last eager hop-bit plane; never a seeded word: seeds go to plane 0)
    if (zb_tail && first == 0) *zb_tail = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this thread's stores are in L2
    __syncthreads();
    // the block's own seeds: unit u of a region was zeroed by thread u % stride, i.e. by block (u % stride) / blockDim.x
    auto mine = [&](const void *word, const void *region) {
        const size_t u = (size_t)((const char *)word - (const char *)region) >> 4;
        return (u % stride) / blockDim.x == blockIdx.x;
    };
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        const long long a = seeds.a[j];
        const size_t idx = (size_t)a * Wp + (j >> 6);
        const u64 bit = 1ull << (j & 63);
        if (mine(&seen[idx], zb)) atomicOr(&seen[idx], bit);
        if (mine(&front[idx], za)) atomicOr(&front[idx], bit);
        if (mine(&live[a >> 5], za)) atomicOr(&live[a >> 5], 1u << (a & 31));
    }
}

// Same-address device-scope stores serialise at the memory side (tens of ns each): a wave stores only while
// the flag still shows an older level.
__device__ __forceinline__ void raise_level(BfsCtl *ctl, int level) {
    if (__hip_atomic_load(&ctl->last_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != level)
        __hip_atomic_store(&ctl->last_active, level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Zero `n16` 16-byte units at each of up to 3 regions + the control block, then nothing else: one launch
// instead of a string of hipMemsetAsync calls (each is its own ~4 us fill kernel).
// rows != nullptr: also the Wp words of row rows[j] of `row_base`, j < n_rows (the anchors' rows of the level-0 frontier, when the
// frontier buffers themselves are not cleared: bfs_enqueue_clear).
__global__ __launch_bounds__(256) void k_zero(uint4 *a, size_t na, uint4 *b, size_t nb, uint4 *c, size_t nc, u64 *b_tail, uint4 *d, size_t nd,
                                              const long long *__restrict__ rows, int n_rows, u64 *row_base, int Wp) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (b_tail && blockIdx.x == 0 && threadIdx.x == 0) *b_tail = 0;       // the 8-byte word behind region b's last 16-byte unit (odd word count)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += stride) d[i] = z;
    if (rows)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n_rows * Wp; i += stride) row_base[(size_t)rows[i / Wp] * Wp + i % Wp] = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) a[i] = z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) b[i] = z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += stride) c[i] = z;
}

// A run of 8-byte words that need not start on a 16-byte boundary (hop-bit planes of an odd N * W: deep levels only).
__global__ __launch_bounds__(256) void k_zero_words(u64 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0;
}

// The words of the rows that span chunks, in one frontier buffer (the accumulator of level 1 when the buffers are not cleared wholesale:
// their pieces are OR-ed in with atomics; later levels' accumulators are cleared by the housekeeping blocks two levels ahead).
__global__ __launch_bounds__(256) void k_clear_spanning_rows(const int *__restrict__ aux, int nchunks, u64 *__restrict__ buf, int Wp) {
    const int *mrows = aux + AUX_HEADER;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += gridDim.x * blockDim.x) {
        const int mv = mrows[i];
        if (mv >= 0)
            for (int w = 0; w < Wp; ++w) buf[(size_t)mv * Wp + w] = 0;
    }
}

__global__ void k_bfs_seed(const long long *__restrict__ anchors, int K, int Wp, u64 *seen, u64 *front, unsigned *live) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < K) seed_anchor(anchors[j], j, Wp, seen, front, live);
}

#ifdef POPE_STAMP
// Diagnostic build only (make stamp): per-wave phase timestamps of k_bfs_level in 100 MHz real-time ticks.
__device__ unsigned long long g_stamps[16384 * 8];
__device__ int g_stamp_level;
#define STAMP(slot)                                                                         \
    do {                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (lane == 0 && wave < 16384 && level == g_stamp_level) g_stamps[wave * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

template <int WT> struct Words { u64 w[WT]; };

typedef float f32x4 __attribute__((ext_vector_type(4)));     // native vector: what the non-temporal builtins accept

// Cross-lane moves on the vector ALUs (DPP) instead of the LDS crossbar (ds_bpermute, which sixteen waves of a CU share): shifts
// inside rows of 16 lanes, the row broadcasts (lane 15 of a row to the next row, lane 31 to rows 2 and 3) and whole-wave shifts by
// one lane.  A lane without a source reads 0 (bound_ctrl).
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142,
              DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ u64 dpp_mov64(u64 x) {
    const unsigned lo = (unsigned)dpp_mov<CTRL>((int)(unsigned)x), hi = (unsigned)dpp_mov<CTRL>((int)(unsigned)(x >> 32));
    return ((u64)hi << 32) | lo;
}

template <int WT>
__device__ __forceinline__ Words<WT> load_words(const u64 *__restrict__ p) {
    Words<WT> r;
    if constexpr (WT == 1) {
        r.w[0] = p[0];
    } else {
#pragma unroll
        for (int i = 0; i < WT; i += 2) {                      // 16-byte loads (rows of 16 / 32 bytes, aligned)
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p + i);
            r.w[i] = v.x;
            r.w[i + 1] = v.y;
        }
    }
    return r;
}

// Frontier gathers go through L1 like any load: reading them with the non-temporal hint was measured 57 % slower
// (BFS 349 us against 223 us, tools/ab_lib.py) -- the rows of hubs are gathered again and again and L1 serves them.
// The same accesses with the non-temporal hint: streams that are read or written once per level and should not displace the frontier
// rows (the gathers' table) from L2 and the Infinity Cache on graphs whose frontier does not fit beside them.
typedef unsigned long long u64x2v __attribute__((ext_vector_type(2)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
template <int WT, bool NT>
__device__ __forceinline__ Words<WT> load_words_hint(const u64 *__restrict__ p) {
    if constexpr (!NT) return load_words<WT>(p);
    Words<WT> r;
    if constexpr (WT == 1) {
        r.w[0] = __builtin_nontemporal_load(p);
    } else {
#pragma unroll
        for (int i = 0; i < WT; i += 2) {
            const u64x2v v = __builtin_nontemporal_load(reinterpret_cast<const u64x2v *>(p + i));
            r.w[i] = v.x;
            r.w[i + 1] = v.y;
        }
    }
    return r;
}

template <int WT>
__device__ __forceinline__ Words<WT> gather_words(const u64 *__restrict__ p) { return load_words<WT>(p); }

template <int WT>
__device__ __forceinline__ void store_words(u64 *__restrict__ p, const Words<WT> &r) {
    if constexpr (WT == 1) {
        p[0] = r.w[0];
    } else {
#pragma unroll
        for (int i = 0; i < WT; i += 2) *reinterpret_cast<ulonglong2 *>(p + i) = make_ulonglong2(r.w[i], r.w[i + 1]);
    }
}

template <int WT, bool NT>
__device__ __forceinline__ void store_words_hint(u64 *__restrict__ p, const Words<WT> &r) {
    if constexpr (!NT) {
        store_words<WT>(p, r);
    } else if constexpr (WT == 1) {
        __builtin_nontemporal_store(r.w[0], p);
    } else {
#pragma unroll
        for (int i = 0; i < WT; i += 2) {
            const u64x2v v = {r.w[i], r.w[i + 1]};
            __builtin_nontemporal_store(v, reinterpret_cast<u64x2v *>(p + i));
        }
    }
}

template <int WT>
__device__ __forceinline__ u64 any_bits(const Words<WT> &r) {
    u64 a = 0;
#pragma unroll
    for (int i = 0; i < WT; ++i) a |= r.w[i];
    return a;
}

// Newly reached anchors of node slot idx at `level`: reachability plane and hop-bit planes (bit-sliced count).
// All plane loads are issued before the first store, so the read-modify-writes cost ONE memory round trip
// instead of one per set bit of the level.
template <int WT, bool NT = false>
__device__ __forceinline__ void commit_words(const Words<WT> &fresh, const Words<WT> &seen_old, size_t idx,
                                             u64 *__restrict__ seen, u64 *__restrict__ hop_planes,
                                             size_t plane_elems, int level) {
    Words<WT> s;
#pragma unroll
    for (int i = 0; i < WT; ++i) s.w[i] = seen_old.w[i] | fresh.w[i];
    store_words_hint<WT, NT>(seen + idx, s);
    Words<WT> h[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        h[b] = fresh;
        if ((level >> b) & 1) h[b] = load_words_hint<WT, NT>(hop_planes + (size_t)b * plane_elems + idx);
    }
#pragma unroll
    for (int b = 0; b < 5; ++b)
        if ((level >> b) & 1) {
#pragma unroll
            for (int i = 0; i < WT; ++i) h[b].w[i] |= fresh.w[i];
            store_words_hint<WT, NT>(hop_planes + (size_t)b * plane_elems + idx, h[b]);
        }
    for (int b = 5, l = level >> 5; l; ++b, l >>= 1)              // levels >= 32: rare, one at a time
        if (l & 1) {
            u64 *p = hop_planes + (size_t)b * plane_elems + idx;
            Words<WT> g = load_words_hint<WT, NT>(p);
#pragma unroll
            for (int i = 0; i < WT; ++i) g.w[i] |= fresh.w[i];
            store_words_hint<WT, NT>(p, g);
        }
}

// One BFS level, bottom-up and EDGE-parallel: a lane owns SLOTS = 4 consecutive CSR slots e = (v -> u), a wave
// pass covers a chunk of 256 slots.
//   cand = front[u] & ~seen[v]           anchors that reach v through u and had not reached v before
// Slots are sorted by v, so a row is a run of consecutive slots.  Runs are combined in two steps: serially
// inside the lane, then ONE 6-step segmented OR-scan across the 64 lanes on each lane's last run (a lane whose
// four slots share one row is "transparent" and passes the carry on).  Work per wave is 256 edges whatever the
// degree distribution (no long rows, no dependent pointer chase: erow/col are coalesced 16-byte streams).
//   * A row that lies inside this chunk is complete: its words are stored to acc[v] (the next frontier, zeros
//     included unless the live table makes them unnecessary, so acc needs no clearing).
//   * A row that spans chunks ("multi-chunk": every hub) receives one piece per chunk, OR-ed into acc[v] with a
//     device-scope atomic (a few thousand per level, distinct addresses); the housekeeping blocks clear those words
//     in the idle third buffer, which launch l+1 will accumulate into.
//   * Nobody commits level l inside launch l.  The COMMIT (reachability plane, bit-sliced hop planes) of level l-1 is
//     done by the housekeeping blocks of launch l, one thread per node with a non-zero frontier row, beside the expand
//     waves; every row masks its candidates with seen[v] | front[v] -- front[v] is exactly what level l-1 added -- so a
//     commit that has or has not landed yet gives the same result.  The expand waves' dependent chain therefore ends at
//     the frontier store (round 1 ended it with a plane read-modify-write: ~2.7 of a wave's ~13 us).  The launch after
//     the last productive level finds nothing and commits that level: the BFS always runs it (it also proves the end).
//     One launch per level, no second pass, no inter-block hand-off inside a launch.
// Three frontier buffers rotate: front = level l-1 (read), acc = level l (written), idle = level l+1 (cleared).
// Beside each goes a "live" table, one BIT per node: set when the node's frontier row is not all zero.  It is N/8
// bytes (11 KB for Flickr) and every block copies it into LDS first (LIVE = 1; graphs up to LIVE_MAX_NODES), so a lane looks
// its four neighbours up there and gathers the 8*W-byte frontier row -- a random 128-byte line from L2 -- only for
// live ones.  The first and the last levels of a BFS have few live nodes: their launches skip most gathers, and a
// chunk with no live neighbour skips its mask loads too.  (Looking the bits up in global memory instead was measured
// slower than no table at all FOR FLICKR: each chunk's 256 gathered lines flush the 32 KB L1, so the lookups went to L2
// as well.  Beyond LIVE_MAX_NODES the table is read from global memory (LIVE = 2): there the frontier rows come from the
// Infinity Cache or HBM while the table still sits in L2 -- R-MAT scale 22 runs 20 % faster with it than without.)
// WT = words per tile (1, 2 or 4); a node with more words (K > 256) has several tiles (TILES, see level_expand).
// The live table (one bit per node, at most LIVE_MAX_NODES / 8 = 32 KB) into LDS: every load of a thread is requested before its
// first write (round 4: as `for (i ...) lds[i] = src[i]` the loop compiled to load - s_waitcnt vmcnt(0) - ds_write per trip, three
// serial round trips for Flickr's 11 KB in front of the barrier every expand wave waits at).
__device__ __forceinline__ void stage_live_table(const unsigned *__restrict__ live, int live_words, uint4 *live_lds4) {
    const uint4 *src = reinterpret_cast<const uint4 *>(live);                       // tables are padded to 256 bytes
    const int n4 = (live_words + 3) / 4;
    for (int base = 0; base < n4; base += 4 * 256) {                                // one trip up to 131 072 nodes
        // Branch-free on purpose: indices past the table are clamped to its last piece (loaded and written again by several threads,
        // the same 16 bytes).  A load under an `if` is waited for at the join, and loads whose only use sits under an `if` are sunk
        // into it by the optimiser -- either way one load in flight.
        uint4 t[4];
        int idx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) idx[j] = min(base + (int)threadIdx.x + 256 * j, n4 - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = src[idx[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j) live_lds4[idx[j]] = t[j];
    }
}

// The summary of a live table for LIVE = 3: bit w of it says that table word w is non-zero (one bit per 32 nodes: 16 KB for the 4.2 M
// nodes of R-MAT scale 22, where the table itself is 512 KB and cannot be staged).  A launch of its own between two level launches
// (~4 us against levels of 0.3-2.5 ms): the table of the level just finished is complete, nobody else writes the summary.
// Tables are padded to 256 bytes with zeros, so a wave may read its 64 words unconditionally.
__global__ __launch_bounds__(256) void k_live_summary(const unsigned *__restrict__ live, int padded_words, unsigned *__restrict__ sum) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned v = w < padded_words ? live[w] : 0u;
    const unsigned long long b = __ballot(v != 0u);
    if ((threadIdx.x & 63) == 0 && w < padded_words) {
        sum[(w >> 5)] = (unsigned)b;
        sum[(w >> 5) + 1] = (unsigned)(b >> 32);
    }
}

// Housekeeping share of one level (see k_bfs_level): thread t0 of tstride threads.  (1) clears two levels ahead -- the live
// table and the accumulator words of the rows that span chunks; (2) commits level - 1 for every node whose
// frontier row is non-zero.  A node's W = tiles * WT words are walked tile by tile.
template <int WT, int LIVE, int TILES>
__device__ __forceinline__ void level_housekeeping(int E, int N, int Wp, int tiles_arg, const u64 *__restrict__ front, u64 *__restrict__ seen,
                                                   u64 *__restrict__ idle, u64 *__restrict__ hop_planes, size_t plane_elems, int level,
                                                   const int *aux, const unsigned *__restrict__ live, unsigned *__restrict__ live_idle,
                                                   int live_words, int t0, int tstride) {
    const int tiles = TILES ? tiles_arg : 1;
    const int n = (E + CHUNK - 1) >> CHUNK_SHIFT;              // one slot per chunk, -1 = no row continues into it
    const int *mrows = aux + AUX_HEADER;
    for (int i = t0; i < live_words; i += tstride) live_idle[i] = 0u;
    Words<WT> zero;
#pragma unroll
    for (int i = 0; i < WT; ++i) zero.w[i] = 0;
    for (int i = t0; i < n; i += tstride) {
        const int mv = mrows[i];
        if (mv >= 0)
            for (int t = 0; t < tiles; ++t) store_words<WT>(idle + (size_t)mv * Wp + t * WT, zero);
    }
    if (level > 1) {
        for (int v = t0; v < N; v += tstride) {
            if (!((live[v >> 5] >> (v & 31)) & 1u)) continue;                      // frontier row all zero: nothing gained
            constexpr bool NT = LIVE >= 2 && POPE_NT_PLANES != 0;
            for (int t = 0; t < tiles; ++t) {
                const size_t idx = (size_t)v * Wp + t * WT;
                const Words<WT> fresh = load_words<WT>(front + idx);
                if (any_bits<WT>(fresh))
                    commit_words<WT, NT>(fresh, load_words_hint<WT, NT>(seen + idx), idx, seen, hop_planes, plane_elems, level - 1);
            }
        }
    }
}

// erow of the slot in front of chunk `chunk` (.x, -1: none) and of the slot behind it (.y, -2: none).
__device__ __forceinline__ int2 chunk_edge_rows(const int *__restrict__ erow, int chunk, int E) {
    int2 r = make_int2(-1, -2);
    if (chunk > 0 && chunk * CHUNK - 1 < E) r.x = erow[chunk * CHUNK - 1];
    if ((chunk + 1) * CHUNK < E) r.y = erow[(chunk + 1) * CHUNK];
    return r;
}

// Expand share of one level (see k_bfs_level): this wave walks chunks wave, wave + nwaves, ...; (vr, ur) hold the first chunk's
// slots, loaded by the caller before it staged the live table.  Returns whether this lane emitted a non-zero row.
// A node with more than 256 anchors has several WT-word tiles (TILES != 0).  Two ways to walk them, chosen by the size of the graph:
//  TILES = 1 (round 5, graphs whose frontier lives in HBM: LIVE >= 2): INSIDE the wave -- the chunk's index loads, live look-ups and row
//    structure (which slots end a run, which rows span chunks, who connects to whom in the scan) are computed once and the gather /
//    mask / scan / store part runs once per tile, the next tile's gathers requested behind this tile's mask loads.  R-MAT scale 22 with
//    512 anchors: 11.95 -> 9.3 ms for the nine levels (one pass over the 522 MB index stream and over the live look-ups instead of two).
//  TILES = 2 (rounds 2-4, graphs that live in L2: LIVE = 1): every tile of a chunk is a wave of its own, adjacent waves of one block, so
//    the 128-byte frontier line they all gather from is fetched from L2 once.  These levels are latency-bound and want the waves: with
//    the tiles inside the wave the Flickr-shaped graph with 1 024 anchors ran 0.616 ms instead of 0.565.
template <int WT, int LIVE, int TILES>
__device__ __forceinline__ bool level_expand(const int *__restrict__ erow, const int *__restrict__ col, int E, int Wp, int tile_begin, int tile_end,
                                             const u64 *__restrict__ front, u64 *__restrict__ seen, u64 *__restrict__ acc,
                                             const unsigned *__restrict__ live, unsigned *__restrict__ live_acc,
                                             const unsigned *live_lds, int level, int lane, int wave, int nwaves, int nchunks,
                                             int4 vr, int4 ur, unsigned *wave_words) {
    auto load_idx = [&](const int *p) {
        if constexpr (LIVE >= 2 && POPE_NT_INDEX != 0) {
            const i32x4v t = __builtin_nontemporal_load(reinterpret_cast<const i32x4v *>(p));
            return make_int4(t.x, t.y, t.z, t.w);
        } else {
            return *reinterpret_cast<const int4 *>(p);
        }
    };
    constexpr bool LOOP = TILES == 1;                          // only then a wave sees more than one tile
    constexpr int TILE_AHEAD = LOOP ? (WT == 8 ? POPE_WT8_PREFETCH : POPE_TILE_PREFETCH) : 0;   // ... and carries the prefetch registers
    if (!TILES) { tile_begin = 0; tile_end = 1; }
    bool found = false;
    STAMP(0);
    // One bit per node: is the frontier row of node u non-zero?  LIVE = 3: a two-level table -- the summary in LDS says whether the
    // node's table word holds any bit at all, and only then the word itself matters (from L2); lanes whose summary bit is clear read
    // word 0 instead (one address, served by a broadcast), so the four look-ups of a lane are four loads in flight at once and a
    // sparse level's waves stream the indices and touch little else.  (As `if (!summary) return false; return word` every look-up was
    // a branch with a load inside: four serial L2 round trips per chunk, 4.2 us of a chunk's 21.8 on R-MAT scale 22.)
    auto live4 = [&](int u0, int u1, int u2, int u3, bool &q0, bool &q1, bool &q2, bool &q3) {
        if constexpr (LIVE == 3) {
            const bool s0 = (live_lds[u0 >> 10] >> ((u0 >> 5) & 31)) & 1u, s1 = (live_lds[u1 >> 10] >> ((u1 >> 5) & 31)) & 1u,
                       s2 = (live_lds[u2 >> 10] >> ((u2 >> 5) & 31)) & 1u, s3 = (live_lds[u3 >> 10] >> ((u3 >> 5) & 31)) & 1u;
            const unsigned w0 = live[s0 ? u0 >> 5 : 0], w1 = live[s1 ? u1 >> 5 : 0], w2 = live[s2 ? u2 >> 5 : 0], w3 = live[s3 ? u3 >> 5 : 0];
            q0 = s0 & ((w0 >> (u0 & 31)) & 1u); q1 = s1 & ((w1 >> (u1 & 31)) & 1u);
            q2 = s2 & ((w2 >> (u2 & 31)) & 1u); q3 = s3 & ((w3 >> (u3 & 31)) & 1u);
        } else {
            const unsigned *t = LIVE == 1 ? live_lds : live;
            const unsigned w0 = t[u0 >> 5], w1 = t[u1 >> 5], w2 = t[u2 >> 5], w3 = t[u3 >> 5];
            q0 = (w0 >> (u0 & 31)) & 1u; q1 = (w1 >> (u1 & 31)) & 1u; q2 = (w2 >> (u2 & 31)) & 1u; q3 = (w3 >> (u3 & 31)) & 1u;
        }
    };
    // Graphs whose waves walk many chunks (LIVE >= 2): the NEXT chunk's indices are requested before this chunk is worked on and its
    // live look-ups go out behind this chunk's gathers -- memory instructions retire in order, so neither waits for the gathers -- and
    // the chain index load -> live look-up -> gather of a chunk no longer starts from nothing (round 5: 4 of a chunk's ~22 us).
    constexpr bool AHEAD = LIVE >= 2 && POPE_AHEAD != 0;
    constexpr bool AHEAD_LIVE = AHEAD && POPE_AHEAD == 1;
    auto slots_of = [&](int chunk, const int4 &vr_, const int4 &ur_, int &v0, int &v1, int &v2, int &v3, int &u0, int &u1, int &u2, int &u3) {
        const int base = chunk * CHUNK + lane * SLOTS;
        v0 = v1 = v2 = v3 = -1;
        u0 = u1 = u2 = u3 = 0;
        if (base < E) {                       // arrays are padded to a multiple of 4 entries: the 16-byte load is in bounds
            v0 = vr_.x; u0 = ur_.x;
            if (base + 1 < E) { v1 = vr_.y; u1 = ur_.y; }
            if (base + 2 < E) { v2 = vr_.z; u2 = ur_.z; }
            if (base + 3 < E) { v3 = vr_.w; u3 = ur_.w; }
        }
    };
    bool q0 = false, q1 = false, q2 = false, q3 = false;       // AHEAD_LIVE: the live bits of the chunk about to be worked on
    if (AHEAD_LIVE && wave < nchunks) {
        int a0, a1, a2, a3, b0, b1, b2, b3;
        slots_of(wave, vr, ur, a0, a1, a2, a3, b0, b1, b2, b3);
        live4(b0, b1, b2, b3, q0, q1, q2, q3);
    }
    for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
        STAMP(7);                                              // (slots 1-5 and 7 hold the wave's LAST chunk; 2-4 its first tile)
        if (!AHEAD && chunk != wave && chunk * CHUNK + lane * SLOTS < E) {
            vr = load_idx(erow + chunk * CHUNK + lane * SLOTS);
            ur = load_idx(col + chunk * CHUNK + lane * SLOTS);
        }
        int v0, v1, v2, v3, u0, u1, u2, u3;
        slots_of(chunk, vr, ur, v0, v1, v2, v3, u0, u1, u2, u3);
        // the next chunk's indices: requested now, looked at behind this chunk's gathers (next_live)
        int4 vr_n = make_int4(-1, -1, -1, -1), ur_n = make_int4(0, 0, 0, 0);
        bool qn0 = false, qn1 = false, qn2 = false, qn3 = false;
        const int chunk_n = chunk + nwaves;
        if (AHEAD && chunk_n < nchunks && chunk_n * CHUNK + lane * SLOTS < E) {
            vr_n = load_idx(erow + chunk_n * CHUNK + lane * SLOTS);
            ur_n = load_idx(col + chunk_n * CHUNK + lane * SLOTS);
        }
        auto next_live = [&]() {
            if constexpr (AHEAD_LIVE) {
                if (chunk_n < nchunks) {
                    int a0, a1, a2, a3, b0, b1, b2, b3;
                    slots_of(chunk_n, vr_n, ur_n, a0, a1, a2, a3, b0, b1, b2, b3);
                    live4(b0, b1, b2, b3, qn0, qn1, qn2, qn3);
                }
            }
        };
        auto advance = [&]() {
            if constexpr (AHEAD) { vr = vr_n; ur = ur_n; }
            if constexpr (AHEAD_LIVE) { q0 = qn0; q1 = qn1; q2 = qn2; q3 = qn3; }
        };
        // the four look-ups first, unconditionally (u = 0 for an empty slot), then the tests: as `v >= 0 && is_live(u)` each look-up sat
        // behind a branch and was waited for on its own
        if constexpr (!AHEAD_LIVE) live4(u0, u1, u2, u3, q0, q1, q2, q3);
        const bool g0 = (v0 >= 0) & q0, g1 = (v1 >= 0) & q1, g2 = (v2 >= 0) & q2, g3 = (v3 >= 0) & q3;
        const int vc = __builtin_amdgcn_readlane(v0, 0);                               // row of the chunk's first slot
        const int vl = __builtin_amdgcn_readlane(v3, 63);                              // row of its last slot (-1: short chunk)
        STAMP(1);
        const bool work = __any(g0 || g1 || g2 || g3);                 // else: no live neighbour behind these 256 slots
        if (!work) {                                                   // nothing to gather, nothing to store (nobody gathers a row whose live bit is clear), nothing to mark
            next_live();
            advance();
            continue;
        }
        // the rows of the slots just outside the chunk: does its first row begin earlier, does its last row run on?
        const int2 er = chunk_edge_rows(erow, __builtin_amdgcn_readfirstlane(chunk), E);     // wave-uniform: scalar loads
        const bool head_multi = chunk > 0 && er.x == vc;                               // first row began in an earlier chunk
        const bool tail_multi = vl >= 0 && (chunk + 1) * CHUNK < E && er.y == vl;     // last row runs on
        // slots of a row that spans chunks (only the chunk's first and last row can)
        const bool x0 = (head_multi && v0 == vc) || (tail_multi && v0 == vl);
        const bool x1 = (head_multi && v1 == vc) || (tail_multi && v1 == vl);
        const bool x2 = (head_multi && v2 == vc) || (tail_multi && v2 == vl);
        const bool x3 = (head_multi && v3 == vc) || (tail_multi && v3 == vl);
        // a run ends in this lane where the slot after it belongs to another row, or the chunk ends
        const int nv0 = dpp_mov<DPP_WAVE_SHL1>(v0);
        const int after3 = lane == 63 ? -3 : nv0;
        const bool e0 = v0 >= 0 && v0 != v1, e1 = v1 >= 0 && v1 != v2, e2 = v2 >= 0 && v2 != v3, e3 = v3 >= 0 && v3 != after3;
        // across lanes: segmented scan over each lane's LAST run (row v3); a lane starts a new segment unless all
        // its slots share one row and that row is also the previous lane's last row
        const int pv3 = dpp_mov<DPP_WAVE_SHR1>(v3);
        const bool connects = lane > 0 && pv3 == v0 && v0 >= 0;
        const bool head0 = !(connects && v0 == v3);
        // the live bits of the rows themselves: row v's mask includes front[v] only when v is live
        bool lv0, lv1, lv2, lv3;                                       // (only used for v >= 0)
        live4(max(v0, 0), max(v1, 0), max(v2, 0), max(v3, 0), lv0, lv1, lv2, lv3);
        bool m0 = false, m1 = false, m2 = false, m3 = false;           // run ends here with something new, in any tile
        // With the live table an all-zero row need not be written: nobody gathers a row whose live bit is clear.
        // (Several tiles share one live bit per node: then zeros are written too, so a live row is exact in every tile.)
        const bool dense = TILES != 0;

        auto gather4 = [&](int woff, Words<WT> &c0, Words<WT> &c1, Words<WT> &c2, Words<WT> &c3) {
#pragma unroll
            for (int i = 0; i < WT; ++i) c0.w[i] = c1.w[i] = c2.w[i] = c3.w[i] = 0;
            if (g0) c0 = gather_words<WT>(front + (size_t)u0 * Wp + woff);
            if (g1) c1 = gather_words<WT>(front + (size_t)u1 * Wp + woff);
            if (g2) c2 = gather_words<WT>(front + (size_t)u2 * Wp + woff);
            if (g3) c3 = gather_words<WT>(front + (size_t)u3 * Wp + woff);
        };
        Words<WT> c0, c1, c2, c3, d0, d1, d2, d3;
        gather4(tile_begin * WT, c0, c1, c2, c3);
        // Tiles in pairs (TILE_AHEAD = 2): the two tiles' pieces of a node's row lie in one 128-byte line.  Requested a tile apart, the
        // second request came 1-3 us after the first, behind a tile's mask loads and their waits -- by then the ~8 MB of lines the waves
        // of one XCD have in flight had pushed the line out of its 4 MB L2 again: 103 raw bytes fetched per edge on the dense levels of
        // R-MAT scale 22 / 512 anchors where one line per edge and the streams make 75 (profiles/r05_config4_pmc.json).
        if (TILE_AHEAD == 2 && tile_begin + 1 < tile_end) gather4((tile_begin + 1) * WT, d0, d1, d2, d3);
        next_live();                                                   // behind the gathers: its loads wait for the NEXT chunk's indices only
        for (int tile = tile_begin; tile < tile_end; ++tile) {
            const int woff = tile * WT;
            // mask of row v: what reached it before this level = seen[v] | front[v].  front[v] (level - 1's gain) is committed to
            // seen by the housekeeping blocks of THIS launch: either order gives the same mask.  Rows whose live bit is clear
            // have an all-zero (possibly never written) frontier row: not loaded.
            // (Round 4, after the finalise kernel's lesson: this chain compiles to up to four serial load - wait rounds behind the gathers.
            //  Requesting the first and last row's masks with the gathers and the interior rows' in a second batch was built and
            //  A/B-ed as separate library builds, tools/ab_lib.py: BFS 205-212 us against 193-197 us for this chain; all four rows at
            //  once needs 142 registers, three waves per SIMD, every level 3-6 us slower.  Requesting the rows of the slots next to
            //  the chunk with the index loads made no measurable difference either.  profiles/r04_level_ab_libs.txt)
            {
                auto row_mask = [&](int v, bool lv) {
                    Words<WT> m = load_words_hint<WT, LIVE >= 2 && POPE_NT_PLANES != 0>(seen + (size_t)v * Wp + woff);
                    if (lv) {
                        const Words<WT> f = load_words<WT>(front + (size_t)v * Wp + woff);
#pragma unroll
                        for (int i = 0; i < WT; ++i) m.w[i] |= f.w[i];
                    }
                    return m;
                };
                Words<WT> s0, s1, s2, s3;
#pragma unroll
                for (int i = 0; i < WT; ++i) s0.w[i] = s1.w[i] = s2.w[i] = s3.w[i] = 0;
                if (v0 >= 0) s0 = row_mask(v0, lv0);
                if (v3 >= 0) s3 = v3 == v0 ? s0 : row_mask(v3, lv3);
                // an interior row (neither the lane's first nor last row)
                if (v1 >= 0) s1 = v1 == v0 ? s0 : (v1 == v3 ? s3 : row_mask(v1, lv1));
                if (v2 >= 0) s2 = v2 == v1 ? s1 : (v2 == v3 ? s3 : row_mask(v2, lv2));
#pragma unroll
                for (int i = 0; i < WT; ++i) {
                    c0.w[i] &= ~s0.w[i];
                    c1.w[i] &= ~s1.w[i];
                    c2.w[i] &= ~s2.w[i];
                    c3.w[i] &= ~s3.w[i];
                }
            }
            // the next tile's gathers go out behind this tile's mask loads (memory instructions retire in order: requested in front of
            // them they would be waited for first), and fly while this tile is scanned and stored
            if (TILE_AHEAD == 1 && tile + 1 < tile_end) gather4(woff + WT, d0, d1, d2, d3);
            const u64 any = any_bits<WT>(c0) | any_bits<WT>(c1) | any_bits<WT>(c2) | any_bits<WT>(c3);
            if (tile == tile_begin) STAMP(2);
            if (__any(any != 0)) {                                         // else: nothing new through these 256 edges
                // inclusive OR along the lane's own slots, restarting where the row changes
#pragma unroll
                for (int i = 0; i < WT; ++i) {
                    if (v1 == v0) c1.w[i] |= c0.w[i];
                    if (v2 == v1) c2.w[i] |= c1.w[i];
                    if (v3 == v2) c3.w[i] |= c2.w[i];
                }
                Words<WT> t = c3;
                bool head = head0;
                // Round 4: the scan network runs on DPP moves -- four shifts inside the rows of 16 lanes, then lane 15 of rows 0 / 2 to
                // rows 1 / 3 and lane 31 to rows 2 / 3 -- where rounds 1-3 shuffled through the LDS crossbar (9 ds_bpermute per step and
                // wave, sixteen waves of a CU queueing for it: 1.9 us of a wave's 13.8, tools/stamp_expand.py).  The operator on
                // (value, head) pairs is the same, so is the result.  A step nobody would take anything in is skipped: rows average ten
                // slots, so chunks without a hub row need two or three of the six.
                auto scan_step = [&](auto ctrl, bool valid) {
                    constexpr int CTRL = decltype(ctrl)::value;
                    if (!__any(valid && !head)) return;
                    const bool ph = dpp_mov<CTRL>((int)head) != 0;
                    const bool take = valid && !head;
#pragma unroll
                    for (int i = 0; i < WT; ++i) {
                        const u64 pt = dpp_mov64<CTRL>(t.w[i]);
                        if (take) t.w[i] |= pt;
                    }
                    if (take) head = ph;
                };
                const int in_row = lane & 15;
                scan_step(std::integral_constant<int, DPP_ROW_SHR1>{}, in_row >= 1);
                scan_step(std::integral_constant<int, DPP_ROW_SHR2>{}, in_row >= 2);
                scan_step(std::integral_constant<int, DPP_ROW_SHR4>{}, in_row >= 4);
                scan_step(std::integral_constant<int, DPP_ROW_SHR8>{}, in_row >= 8);
                scan_step(std::integral_constant<int, DPP_ROW_BCAST15>{}, ((lane >> 4) & 1) != 0);
                scan_step(std::integral_constant<int, DPP_ROW_BCAST31>{}, lane >= 32);
                // carry into this lane's first run = accumulated value of the previous lane's last run
#pragma unroll
                for (int i = 0; i < WT; ++i) {
                    u64 ci = dpp_mov64<DPP_WAVE_SHR1>(t.w[i]);
                    if (!connects) ci = 0;
                    c0.w[i] |= ci;
                    if (v1 == v0) c1.w[i] |= ci;
                    if (v2 == v0) c2.w[i] |= ci;
                    if (v3 == v0) c3.w[i] |= ci;
                }
            }
            if (tile == tile_begin) STAMP(3);
            // Emit every run that ends in this lane.
            const size_t i0 = (size_t)v0 * Wp + woff, i1 = (size_t)v1 * Wp + woff, i2 = (size_t)v2 * Wp + woff,
                         i3 = (size_t)v3 * Wp + woff;
            const bool n0 = e0 && any_bits<WT>(c0) != 0, n1 = e1 && any_bits<WT>(c1) != 0, n2 = e2 && any_bits<WT>(c2) != 0,
                       n3 = e3 && any_bits<WT>(c3) != 0;
            // rows that lie inside the chunk: plain stores
            if (e0 && !x0 && (n0 || dense)) store_words<WT>(acc + i0, c0);
            if (e1 && !x1 && (n1 || dense)) store_words<WT>(acc + i1, c1);
            if (e2 && !x2 && (n2 || dense)) store_words<WT>(acc + i2, c2);
            if (e3 && !x3 && (n3 || dense)) store_words<WT>(acc + i3, c3);
            // pieces of the (at most two) rows that span chunks: OR them in (their words were cleared two launches ago), commit later.
            // Wave-uniform guard, and no branch per word (round 4: ~28 divergent branch regions in this phase before).
            if (head_multi || tail_multi) {
                auto piece = [&](size_t idx, const Words<WT> &c) {
#pragma unroll
                    for (int i = 0; i < WT; ++i) atomicOr(&acc[idx + i], c.w[i]);
                };
                if (n0 && x0) piece(i0, c0);
                if (n1 && x1) piece(i1, c1);
                if (n2 && x2) piece(i2, c2);
                if (n3 && x3) piece(i3, c3);
            }
            if (tile == tile_begin) STAMP(4);
            m0 |= n0; m1 |= n1; m2 |= n2; m3 |= n3;
            if (TILE_AHEAD == 1 && tile + 1 < tile_end) { c0 = d0; c1 = d1; c2 = d2; c3 = d3; }
            else if (TILE_AHEAD == 2 && tile + 1 < tile_end) {
                if (!((tile - tile_begin) & 1)) { c0 = d0; c1 = d1; c2 = d2; c3 = d3; }          // second tile of the pair: already here
                else {                                                                         // the next pair
                    gather4(woff + WT, c0, c1, c2, c3);
                    if (tile + 2 < tile_end) gather4(woff + 2 * WT, d0, d1, d2, d3);
                }
            } else if (LOOP && tile + 1 < tile_end) gather4(woff + WT, c0, c1, c2, c3);
        }
        found |= m0 || m1 || m2 || m3;
        // Mark the rows that received something.  The chunk's rows are a short ascending run of node ids: build each
        // 32-bit table word with a wave-wide OR and let one lane publish it (per-row atomics -- ~30 to every word
        // from a few waves -- cost 14 us per dense level).
        if (__any(m0 || m1 || m2 || m3)) {
            const int wfirst = vc >> 5;
            // the row of the chunk's last slot: lane 63's last slot, except in the one short chunk at the end of the edge list
            const int last_row = vl >= 0 ? vl : erow[min((chunk + 1) * CHUNK, E) - 1];
            const int kmax = (last_row >> 5) - wfirst;
            if (kmax < 8) {
                // Round 4: every row of the chunk has exactly one emitting slot (the end of its run), so the set bits are distinct:
                // the emitting lanes OR them into eight LDS words of the wave (one ds_or each, no return), and lanes 0 .. kmax publish
                // the words.  (Rounds 2-3 built each word with a six-step wave-wide OR per word and read the last row from memory:
                // 1.36 us of a wave's 13.8, tools/stamp_expand.py.)  A wave's LDS instructions execute in order: no barrier.
                if (lane < 8) wave_words[lane] = 0u;
                if (m0) atomicOr(&wave_words[(v0 >> 5) - wfirst], 1u << (v0 & 31));
                if (m1) atomicOr(&wave_words[(v1 >> 5) - wfirst], 1u << (v1 & 31));
                if (m2) atomicOr(&wave_words[(v2 >> 5) - wfirst], 1u << (v2 & 31));
                if (m3) atomicOr(&wave_words[(v3 >> 5) - wfirst], 1u << (v3 & 31));
                __builtin_amdgcn_wave_barrier();
                if (lane <= kmax) {
                    const unsigned m = wave_words[lane];
                    if (m) atomicOr(&live_acc[wfirst + lane], m);
                }
            } else {                                       // a run with wide gaps (isolated nodes in between)
                if (m0) atomicOr(&live_acc[v0 >> 5], 1u << (v0 & 31));
                if (m1) atomicOr(&live_acc[v1 >> 5], 1u << (v1 & 31));
                if (m2) atomicOr(&live_acc[v2 >> 5], 1u << (v2 & 31));
                if (m3) atomicOr(&live_acc[v3 >> 5], 1u << (v3 & 31));
            }
        }
        STAMP(5);
        advance();
    }
    STAMP(6);
    return found;
}

// LIVE: 1 live table staged in LDS; 2 live table read from global memory (graphs beyond LIVE_MAX_NODES); 3 the same behind a summary
// in LDS (one bit per table word, built by k_live_summary between the launches).  TILES: 0 one WT-word tile per node; several tiles
// walked inside the wave (1) or dealt to adjacent waves (2), see level_expand.
template <int WT, int LIVE, int TILES>
__global__ __launch_bounds__(256, WT == 8 ? (TILES == 1 ? POPE_WT8_LOOP_WAVES : POPE_WT8_WAVES) : 1) void k_bfs_level(const int *__restrict__ erow, const int *__restrict__ col,
                                                   int E, int N, int Wp, int tiles, const u64 *__restrict__ front,
                                                   u64 *__restrict__ seen, u64 *__restrict__ acc,
                                                   u64 *__restrict__ idle, u64 *__restrict__ hop_planes,
                                                   size_t plane_elems, int level, BfsCtl *ctl, const int *aux,
                                                   int expand_blocks, const unsigned *__restrict__ live,
                                                   unsigned *__restrict__ live_acc, unsigned *__restrict__ live_idle,
                                                   int live_words, const unsigned *__restrict__ live_sum, int sum_words) {
    if (bfs_over(ctl, aux, level)) return;
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= expand_blocks) {
        // Housekeeping blocks (beside the expand waves, not on their critical path):
        //  (1) clear, two levels ahead: the live table and the accumulator words of the rows that span chunks;
        //  (2) COMMIT level - 1 for every node: a node whose frontier row is non-zero gained those anchors at level - 1
        //      -> reachability plane and hop-bit planes.  The expand waves never commit: they mask with seen[v] | front[v],
        //      which is the same whether this commit has landed or not (OR is idempotent), and their chain ends at the store
        //      of the next frontier instead of a plane read-modify-write behind it.
        const int hb = (int)gridDim.x - expand_blocks;
        level_housekeeping<WT, LIVE, TILES>(E, N, Wp, tiles, front, seen, idle, hop_planes, plane_elems, level, aux, live, live_idle, live_words,
                                     ((int)blockIdx.x - expand_blocks) * blockDim.x + threadIdx.x, hb * blockDim.x);
        return;
    }
    // which stream of chunks this wave walks, and which tiles of a node's words
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (expand_blocks * blockDim.x) >> 6, tile_begin = 0, tile_end = tiles;
    if constexpr (TILES == 2) {                                    // the tiles of one chunk go to adjacent waves of the same block
        const int wid = wave;
        wave = wid / tiles;
        tile_begin = wid - wave * tiles;
        tile_end = tile_begin + 1;
        nwaves /= tiles;
    }
    const int nchunks = (E + CHUNK - 1) >> CHUNK_SHIFT;
    // The first chunk's slot loads are issued before the live table is staged: they fly while LDS fills.
    int4 vr = make_int4(-1, -1, -1, -1), ur = make_int4(0, 0, 0, 0);
    if (wave < nchunks && wave * CHUNK + lane * SLOTS < E) {
        if constexpr (LIVE >= 2 && POPE_NT_INDEX != 0) {
            const i32x4v a = __builtin_nontemporal_load(reinterpret_cast<const i32x4v *>(erow + wave * CHUNK + lane * SLOTS));
            const i32x4v b = __builtin_nontemporal_load(reinterpret_cast<const i32x4v *>(col + wave * CHUNK + lane * SLOTS));
            vr = make_int4(a.x, a.y, a.z, a.w);
            ur = make_int4(b.x, b.y, b.z, b.w);
        } else {
            vr = *reinterpret_cast<const int4 *>(erow + wave * CHUNK + lane * SLOTS);
            ur = *reinterpret_cast<const int4 *>(col + wave * CHUNK + lane * SLOTS);
        }
    }
    extern __shared__ uint4 live_lds4[];
    const unsigned *live_lds = reinterpret_cast<const unsigned *>(live_lds4);
    if constexpr (LIVE == 1) {
        stage_live_table(live, live_words, live_lds4);
        __syncthreads();
    } else if constexpr (LIVE == 3) {
        stage_live_table(live_sum, sum_words, live_lds4);
        __syncthreads();
    }
    __shared__ unsigned wave_live_words[4][8];                     // per wave: the live-table words its chunk's rows fall into
    const bool found = level_expand<WT, LIVE, TILES>(erow, col, E, Wp, tile_begin, tile_end, front, seen, acc, live, live_acc, live_lds, level, lane, wave, nwaves,
                                              nchunks, vr, ur, wave_live_words[threadIdx.x >> 6]);
    if (__any(found) && lane == 0) raise_level(ctl, level);
}

// pope_geodesic_run: the finalise kernel doubles as the report (deepest active level, CSR flags) into pinned,
// device-mapped host memory, which the host reads after its one stream synchronisation.
// The ticket is stored last (system-scope release): a host thread spinning on it sees the verdict as soon as the
// kernel STARTS, i.e. when the BFS levels before it in the stream are done, not when the 100 us expansion ends.
__device__ __forceinline__ void write_report(const int *max_hop_dev, const int *aux, int *report, int ticket) {
    if (report && blockIdx.x == 0 && threadIdx.x == 0) {
        report[0] = *max_hop_dev;
        report[1] = csr_flags(reinterpret_cast<const BfsCtl *>(max_hop_dev), aux);        // (&ctl->last_active: the block's first word)
        __hip_atomic_store(&report[2], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// Finalise: hop planes -> 1/(h+1) float32 written next to the features (utils.py:73,125,129-135)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hop_value(const u64 *__restrict__ planes, size_t plane_elems, int n_hop_bits,
                                           size_t widx, int bit) {
    if (!((planes[widx] >> bit) & 1ull)) return 0.0f;             // unreachable (utils.py:75-76)
    int h = 0;
    for (int b = 0; b < n_hop_bits; ++b)
        h |= (int)((planes[(size_t)(b + 1) * plane_elems + widx] >> bit) & 1ull) << b;
    return 1.0f / (float)(h + 1);                                  // IEEE division, == f32(1.0 / (h + 1))
}

// One wave per row at a time.  VEC: 16-byte accesses (F, K, c0, out_cols multiples of 4, bases aligned).
template <bool VEC>
__global__ __launch_bounds__(256) void k_finalize(const u64 *__restrict__ planes, size_t plane_elems,
                                                  int n_hop_bits, const int *__restrict__ max_hop_dev, int N, int K,
                                                  int Wp, const float *__restrict__ x, int F,
                                                  float *__restrict__ out, long long out_cols, int c0,
                                                  const int *__restrict__ aux, int *report, int ticket) {
    if (max_hop_dev) {                        // enqueued before the host knew the depth: read it from the BFS control block
        const int m = *max_hop_dev;
        n_hop_bits = m > 0 ? 32 - __clz(m) : 0;
        write_report(max_hop_dev, aux, report, ticket);
    }
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int v = wave; v < N; v += nwaves) {
        float *orow = out + (size_t)v * out_cols;
        if (x) {
            const float *xrow = x + (size_t)v * F;
            if (VEC) {
                const float4 *xs = reinterpret_cast<const float4 *>(xrow);
                float4 *os = reinterpret_cast<float4 *>(orow);
                for (int q = lane; q < F / 4; q += 64) os[q] = xs[q];
            } else {
                for (int c = lane; c < F; c += 64) orow[c] = xrow[c];
            }
        }
        float *erow = orow + F + c0;
        const size_t wbase = (size_t)v * Wp;
        if (VEC) {
            for (int q = lane; q < K / 4; q += 64) {
                const int j = q * 4;                       // four anchors of one word: one load per plane
                const size_t widx = wbase + (j >> 6);
                const int bit = j & 63;
                const unsigned reach = (unsigned)(planes[widx] >> bit) & 15u;
                int h0 = 0, h1 = 0, h2 = 0, h3 = 0;
                for (int b = 0; b < n_hop_bits; ++b) {
                    const unsigned p = (unsigned)(planes[(size_t)(b + 1) * plane_elems + widx] >> bit) & 15u;
                    h0 |= (int)(p & 1u) << b;
                    h1 |= (int)((p >> 1) & 1u) << b;
                    h2 |= (int)((p >> 2) & 1u) << b;
                    h3 |= (int)((p >> 3) & 1u) << b;
                }
                float4 r;
                r.x = (reach & 1u) ? 1.0f / (float)(h0 + 1) : 0.0f;
                r.y = (reach & 2u) ? 1.0f / (float)(h1 + 1) : 0.0f;
                r.z = (reach & 4u) ? 1.0f / (float)(h2 + 1) : 0.0f;
                r.w = (reach & 8u) ? 1.0f / (float)(h3 + 1) : 0.0f;
                reinterpret_cast<float4 *>(erow)[q] = r;
            }
        } else {
            for (int j = lane; j < K; j += 64)
                erow[j] = hop_value(planes, plane_elems, n_hop_bits, wbase + (j >> 6), j & 63);
        }
    }
}

// Fast path of the finalise kernel: 16-byte accesses, at most 4 hop-bit planes (hops < 16: any small-world graph).
//  * every wave owns a CONTIGUOUS block of rows, so the cache lines that straddle two rows (row pitch 4*(F+K) bytes is
//    not a multiple of 128) are completed by the same wave;
//  * 1/(h+1) comes from a 16-entry table built once per block with the same IEEE division (bit-identical);
//  * the four hop counts of a lane are pulled out of the packed plane nibbles with one multiply each;
//  * x is read with non-temporal loads (read once); stores are plain -- non-temporal stores measured 23 % slower.
// Since round 4 the fallback of k_finalize_pipe / k_finalize_wide for shapes they have no instance for (F > 1024).
// n_shards > 1 (multi-GPU): `planes` holds the all-gathered shards back to back (shard_elems words apart, K anchors
// each); a row's columns of ALL shards are written in one pass, so the [N, F + shards*K] matrix is streamed once.
__global__ __launch_bounds__(256) void k_finalize_fast(const u64 *__restrict__ planes, size_t plane_elems,
                                                       int n_hop_bits, const int *__restrict__ max_hop_dev, int N, int K,
                                                       int Wp, const float *__restrict__ x, int F,
                                                       float *__restrict__ out, long long out_cols, int c0,
                                                       int n_shards, size_t shard_elems, const int *__restrict__ aux,
                                                       int *report, int ticket) {
    if (max_hop_dev) write_report(max_hop_dev, aux, report, ticket);
    __shared__ float inv[16];
    if (threadIdx.x < 16) inv[threadIdx.x] = 1.0f / (float)(threadIdx.x + 1);
    __syncthreads();
    if (max_hop_dev) {
        const int m = *max_hop_dev;
        n_hop_bits = m > 0 ? 32 - __clz(m) : 0;
    }
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int per = (N + nwaves - 1) / nwaves;
    const int v_begin = wave * per, v_end = min(N, v_begin + per);
    const int F4 = F >> 2, K4 = K >> 2;
    for (int v = v_begin; v < v_end; ++v) {
        f32x4 *orow = reinterpret_cast<f32x4 *>(out + (size_t)v * out_cols);
        if (x) {
            const f32x4 *xs = reinterpret_cast<const f32x4 *>(x + (size_t)v * F);
            for (int q = lane; q < F4; q += 64) {
                const f32x4 t = __builtin_nontemporal_load(xs + q);
                orow[q] = t;
            }
        }
        f32x4 *erow = reinterpret_cast<f32x4 *>(out + (size_t)v * out_cols + F + c0);
        const size_t wbase = (size_t)v * Wp;
        for (int q = lane; q < K4 * n_shards; q += 64) {
            const int shard = q / K4;
            const int j = (q - shard * K4) * 4;                // four anchors of one word of that shard
            const size_t widx = (size_t)shard * shard_elems + wbase + (j >> 6);
            const int bit = j & 63;
            const unsigned reach = (unsigned)(planes[widx] >> bit) & 15u;
            unsigned t = 0;                                    // nibble b = the four anchors' hop bit b
            for (int b = 0; b < n_hop_bits; ++b)
                t |= ((unsigned)(planes[(size_t)(b + 1) * plane_elems + widx] >> bit) & 15u) << (4 * b);
            // bits 0,4,8,12 of (t >> i) are anchor i's hop bits 0..3: the multiply gathers them into bits 12..15
            const unsigned h0 = (((t) & 0x1111u) * 0x1248u >> 12) & 15u;
            const unsigned h1 = (((t >> 1) & 0x1111u) * 0x1248u >> 12) & 15u;
            const unsigned h2 = (((t >> 2) & 0x1111u) * 0x1248u >> 12) & 15u;
            const unsigned h3 = (((t >> 3) & 0x1111u) * 0x1248u >> 12) & 15u;
            f32x4 r;
            r.x = (reach & 1u) ? inv[h0] : 0.0f;
            r.y = (reach & 2u) ? inv[h1] : 0.0f;
            r.z = (reach & 4u) ? inv[h2] : 0.0f;
            r.w = (reach & 8u) ? inv[h3] : 0.0f;
            erow[q] = r;
        }
    }
}

// k_finalize_fast with every load of a row in flight at once, the NEXT row's loads issued before this row's stores, and the rows
// dealt to the waves round-robin (round 4).  The ISA of k_finalize_fast shows why it runs at 4.6 TB/s: its loops compile to
// load - s_waitcnt vmcnt(0) - store per 16-byte piece and to one plane load per s_waitcnt in the hop-bit loop -- seven serial round
// trips per row and ONE load in flight per lane, the chip's 32 waves per CU being all that hides them.  Here a row's XP feature
// pieces and the 5 plane words of its EP embedding pieces are independent loads (no loops), held in registers for one
// iteration while the next row's are requested: 0.263 -> 0.254 ms per configs[1] step.  Row v goes to wave v mod nwaves, so
// the waves that run at the same time stream through ONE moving window of consecutive rows instead of 8 192 separate places
// (0.254 -> 0.247; with the old kernel's serial loops contiguous row blocks per wave were the faster choice), and the grid is one
// row per wave (22 313 blocks for Flickr: 0.2395 ms; profiles/r04_finalize_pipe*.txt): 463 MB in ~74 us = 6.25 TB/s, the measured
// copy rate of the part.  Shapes: F <= 256 XP (XP <= 4), any K * shards (rows wider than 1 024 columns are cut into segments, one
// work item each), at most four hop bits (others: k_finalize_fast).
template <int XP, int EP>
struct FinRow {
    f32x4 x[XP > 0 ? XP : 1];
    u64 w[EP][5];
};

template <int XP, int EP>
__device__ __forceinline__ FinRow<XP, EP> fin_load(const u64 *__restrict__ planes, size_t plane_elems, int n_hop_bits, int Wp, const float *__restrict__ x,
                                                   int F4, int v, int lane, bool copy_x, int K4, int n_emb, size_t shard_elems, int q0) {
    FinRow<XP, EP> r;
#pragma unroll
    for (int i = 0; i < (XP > 0 ? XP : 1); ++i) r.x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (XP > 0 && copy_x) {
        const f32x4 *xs = reinterpret_cast<const f32x4 *>(x) + (size_t)v * F4;
#pragma unroll
        for (int i = 0; i < XP; ++i)
            if (lane + 64 * i < F4) r.x[i] = __builtin_nontemporal_load(xs + lane + 64 * i);
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) {
#pragma unroll
        for (int b = 0; b < 5; ++b) r.w[e][b] = 0;
        const int q = q0 + lane + 64 * e;
        if (q < n_emb) {
            const int shard = q / K4, j = (q - shard * K4) * 4;
            const size_t widx = (size_t)shard * shard_elems + (size_t)v * Wp + (j >> 6);
            r.w[e][0] = planes[widx];
            if (n_hop_bits > 0) r.w[e][1] = planes[plane_elems + widx];
            if (n_hop_bits > 1) r.w[e][2] = planes[2 * plane_elems + widx];
            if (n_hop_bits > 2) r.w[e][3] = planes[3 * plane_elems + widx];
            if (n_hop_bits > 3) r.w[e][4] = planes[4 * plane_elems + widx];
        }
    }
    return r;
}

template <int XP, int EP>
__global__ __launch_bounds__(256) void k_finalize_pipe(const u64 *__restrict__ planes, size_t plane_elems, int n_hop_bits,
                                                       const int *__restrict__ max_hop_dev, int N, int K, int Wp,
                                                       const float *__restrict__ x, int F, float *__restrict__ out, long long out_cols,
                                                       int c0, int n_shards, size_t shard_elems, const int *__restrict__ aux, int *report,
                                                       int ticket) {
    if (max_hop_dev) write_report(max_hop_dev, aux, report, ticket);
    __shared__ float inv[16];
    if (threadIdx.x < 16) inv[threadIdx.x] = 1.0f / (float)(threadIdx.x + 1);
    __syncthreads();
    if (max_hop_dev) {
        const int m = *max_hop_dev;
        n_hop_bits = m > 0 ? 32 - __clz(m) : 0;
    }
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    // A work item is (row, segment): a row wider than 256 EP embedding columns (many shards) is cut into segments of 64 EP pieces,
    // each a work item of its own; segment 0 also copies the row's features.  Items are dealt to the waves round-robin
    const int F4 = F >> 2, K4 = K >> 2, n_emb = K4 * n_shards;
    const int n_seg = (n_emb + 64 * EP - 1) / (64 * EP);
    const int items = N * n_seg;                                         // < 2^31: checked on the host
    const int i_begin = wave, i_end = items, i_step = nwaves;
    if (i_begin >= i_end) return;
    auto row_of = [&](int i, int &seg) { const int v = (int)((unsigned)i / (unsigned)n_seg); seg = i - v * n_seg; return v; };
    int seg = 0, v = row_of(i_begin, seg);
    FinRow<XP, EP> cur = fin_load<XP, EP>(planes, plane_elems, n_hop_bits, Wp, x, F4, v, lane, x && seg == 0, K4, n_emb, shard_elems, seg * 64 * EP);
    for (int i = i_begin; i < i_end; i += i_step) {
        FinRow<XP, EP> nxt = cur;
        int seg_n = 0, v_n = 0;
        if (i + i_step < i_end) {
            v_n = row_of(i + i_step, seg_n);
            nxt = fin_load<XP, EP>(planes, plane_elems, n_hop_bits, Wp, x, F4, v_n, lane, x && seg_n == 0, K4, n_emb, shard_elems, seg_n * 64 * EP);
        }
        f32x4 *orow = reinterpret_cast<f32x4 *>(out + (size_t)v * out_cols);
        if (XP > 0 && x && seg == 0) {
#pragma unroll
            for (int p = 0; p < XP; ++p)
                if (lane + 64 * p < F4) orow[lane + 64 * p] = cur.x[p];
        }
        f32x4 *erow = reinterpret_cast<f32x4 *>(out + (size_t)v * out_cols + F + c0);
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            const int q = seg * 64 * EP + lane + 64 * e;
            if (q < n_emb) {
                const int shard = q / K4, j = (q - shard * K4) * 4, bit = j & 63;
                const unsigned reach = (unsigned)(cur.w[e][0] >> bit) & 15u;
                unsigned t = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) t |= ((unsigned)(cur.w[e][b + 1] >> bit) & 15u) << (4 * b);      // planes past n_hop_bits were loaded as 0
                const unsigned h0 = (((t) & 0x1111u) * 0x1248u >> 12) & 15u;
                const unsigned h1 = (((t >> 1) & 0x1111u) * 0x1248u >> 12) & 15u;
                const unsigned h2 = (((t >> 2) & 0x1111u) * 0x1248u >> 12) & 15u;
                const unsigned h3 = (((t >> 3) & 0x1111u) * 0x1248u >> 12) & 15u;
                f32x4 r;
                r.x = (reach & 1u) ? inv[h0] : 0.0f;
                r.y = (reach & 2u) ? inv[h1] : 0.0f;
                r.z = (reach & 4u) ? inv[h2] : 0.0f;
                r.w = (reach & 8u) ? inv[h3] : 0.0f;
                erow[q] = r;
            }
        }
        cur = nxt;
        v = v_n;
        seg = seg_n;
    }
}

// Wide rows (more than 256 embedding columns: several shards after the all-gather, or K > 256 on one GPU), K a multiple of 64.
// In k_finalize_pipe sixteen lanes load the same plane word, and an item of 256 pieces costs twenty narrow loads and ~100
// registers: at 8 x 256 anchors the plane loads alone took 162 us for 114 MB and the stores another 150 (profiles/
// r04_finalize_shards.txt).  Here a work item is (row, 16 words): lane l < 32 loads one 32-bit HALF of a word of each of the five
// planes -- five loads per item -- and every lane fetches the half-word of its four anchors from lane (piece >> 3) with ONE
// 32-bit shuffle per plane; five registers per item instead of forty, so the next item's loads fit beside this one's stores at full occupancy: 220 us against 382 at 8 x 256
// anchors.  (Four lanes per word and no shuffles -- each lane expanding pieces (l & 3) + 4 e of its own word -- makes every store
// instruction write sixteen 64-byte runs instead of whole lines: 285 us.)
template <int XP>
__global__ __launch_bounds__(256) void k_finalize_wide(const u64 *__restrict__ planes, size_t plane_elems, int n_hop_bits,
                                                       const int *__restrict__ max_hop_dev, int N, int K, int Wp,
                                                       const float *__restrict__ x, int F, float *__restrict__ out, long long out_cols,
                                                       int c0, int n_shards, size_t shard_elems, const int *__restrict__ aux, int *report,
                                                       int ticket) {
    if (max_hop_dev) write_report(max_hop_dev, aux, report, ticket);
    __shared__ float inv[16];
    if (threadIdx.x < 16) inv[threadIdx.x] = 1.0f / (float)(threadIdx.x + 1);
    __syncthreads();
    if (max_hop_dev) {
        const int m = *max_hop_dev;
        n_hop_bits = m > 0 ? 32 - __clz(m) : 0;
    }
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int F4 = F >> 2, K4 = K >> 2, n_emb = K4 * n_shards;          // K4 is a multiple of 16: a word never spans two shards
    const int wps = K4 >> 4;                                             // words per shard and node (not Wp: that one is padded to the tile width)
    const int n_words = n_emb >> 4, n_seg = (n_words + 15) >> 4;
    const int items = N * n_seg;                                         // < 2^31: checked on the host
    if (wave >= items) return;
    struct Item { f32x4 x[XP > 0 ? XP : 1]; unsigned w[5]; };      // w: one 32-bit HALF of a plane word per lane (lanes 0 .. 31)
    auto load = [&](int i, int &v, int &seg) {
        v = (int)((unsigned)i / (unsigned)n_seg);
        seg = i - v * n_seg;
        Item r;
#pragma unroll
        for (int p = 0; p < (XP > 0 ? XP : 1); ++p) r.x[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 5; ++b) r.w[b] = 0;
        if (XP > 0 && x && seg == 0) {
            const f32x4 *xs = reinterpret_cast<const f32x4 *>(x) + (size_t)v * F4;
#pragma unroll
            for (int p = 0; p < XP; ++p)
                if (lane + 64 * p < F4) r.x[p] = __builtin_nontemporal_load(xs + lane + 64 * p);
        }
        const int word = seg * 16 + (lane >> 1);                         // lanes 0 .. 31: half (lane & 1) of word lane >> 1 of the item
        if (lane < 32 && word < n_words) {
            const int shard = word / wps;
            const unsigned *p = reinterpret_cast<const unsigned *>(planes + ((size_t)shard * shard_elems + (size_t)v * Wp + (word - shard * wps))) + (lane & 1);
            r.w[0] = p[0];
            if (n_hop_bits > 0) r.w[1] = p[2 * plane_elems];
            if (n_hop_bits > 1) r.w[2] = p[4 * plane_elems];
            if (n_hop_bits > 2) r.w[3] = p[6 * plane_elems];
            if (n_hop_bits > 3) r.w[4] = p[8 * plane_elems];
        }
        return r;
    };
    int v = 0, seg = 0;
    Item cur = load(wave, v, seg);
    for (int i = wave; i < items; i += nwaves) {
        Item nxt = cur;
        int v_n = 0, seg_n = 0;
        if (i + nwaves < items) nxt = load(i + nwaves, v_n, seg_n);
        if (XP > 0 && x && seg == 0) {
            f32x4 *orow = reinterpret_cast<f32x4 *>(out + (size_t)v * out_cols);
#pragma unroll
            for (int p = 0; p < XP; ++p)
                if (lane + 64 * p < F4) orow[lane + 64 * p] = cur.x[p];
        }
        f32x4 *erow = reinterpret_cast<f32x4 *>(out + (size_t)v * out_cols + F + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = seg * 256 + lane + 64 * e;                     // piece: four anchors of half-word q >> 3, held by lane (q >> 3) - 32 seg
            const int src = (lane >> 3) + 8 * e, bit = (q & 7) * 4;
            unsigned nib[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) nib[b] = ((unsigned)__shfl((int)cur.w[b], src) >> bit) & 15u;
            if (q < n_emb) {
                const unsigned reach = nib[0];
                const unsigned t = nib[1] | (nib[2] << 4) | (nib[3] << 8) | (nib[4] << 12);
                const unsigned h0 = (((t) & 0x1111u) * 0x1248u >> 12) & 15u;
                const unsigned h1 = (((t >> 1) & 0x1111u) * 0x1248u >> 12) & 15u;
                const unsigned h2 = (((t >> 2) & 0x1111u) * 0x1248u >> 12) & 15u;
                const unsigned h3 = (((t >> 3) & 0x1111u) * 0x1248u >> 12) & 15u;
                f32x4 r;
                r.x = (reach & 1u) ? inv[h0] : 0.0f;
                r.y = (reach & 2u) ? inv[h1] : 0.0f;
                r.z = (reach & 4u) ? inv[h2] : 0.0f;
                r.w = (reach & 8u) ? inv[h3] : 0.0f;
                erow[q] = r;
            }
        }
        cur = nxt;
        v = v_n;
        seg = seg_n;
    }
}

// Wide rows without shuffles and with a tenth of the bit arithmetic (round 5).  k_finalize_wide spends ~50 vector instructions and
// five ds_bpermute per 16-byte store pulling nibbles out of bit-sliced planes (8.6 GB of [N, 512] columns for R-MAT scale 22 at
// 4.0 TB/s; 3.8 TB/s at 8 x 256 anchors).  Here a lane owns one 32-bit HALF of a plane word -- 32 anchors, 128 bytes of output -- and
// turns it into floats byte by byte through two tables in LDS:
//   spread[b][byte]   the byte's 8 bits moved to bit b of 8 nibbles (u32), so the OR over the four hop-bit planes is the 8 anchors'
//                     4-bit hop counts side by side;
//   pair[code]        code = two neighbouring hop nibbles + their two reachability bits (10 bits) -> float2{1 / (h + 1) or 0}, built
//                     per block with the same IEEE division as every other finalise kernel (bit-identical to f32(1.0 / (h + 1))).
// Per 8 anchors: four spread look-ups, three ORs, and per pair one field extract for the code, one for the reachability bits, one
// OR and one 8-byte look-up.  A wave takes 64 consecutive half-words of the flat (row, half-word) sequence (rows with few words do
// not leave lanes idle), prefetches the next batch's five plane dwords before it stores, and transposes its 8 KB through LDS so that
// every store instruction writes 1 KB of whole lines (a lane's own 128 bytes are 64 partial lines per instruction).  Shapes: K a
// multiple of 64, words per shard and number of shards powers of two (the half-words per row then are one: shifts, no divisions).
constexpr int FIN_LUT_LDS = 4 * 256 * 4 + 1024 * 8 + 4 * 8192;      // spread tables, pair table, one 8 KB transpose image per wave

__global__ __launch_bounds__(256) void k_finalize_lut(const u64 *__restrict__ planes, size_t plane_elems, int n_hop_bits,
                                                      const int *__restrict__ max_hop_dev, int N, int Wp, float *__restrict__ out,
                                                      long long out_cols, int col0, int hpr_shift, int wps_shift, int rows_shift, size_t shard_elems,
                                                      const int *__restrict__ aux, int *report, int ticket) {
    if (max_hop_dev) write_report(max_hop_dev, aux, report, ticket);
    extern __shared__ __attribute__((aligned(16))) char fin_lds[];
    unsigned *spread = reinterpret_cast<unsigned *>(fin_lds);                       // [4][256]
    float2 *pair = reinterpret_cast<float2 *>(fin_lds + 4 * 256 * 4);               // [1024]
    char *image = fin_lds + 4 * 256 * 4 + 1024 * 8 + (threadIdx.x >> 6) * 8192;     // this wave's transpose image
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int b = i >> 8, x = i & 255;
        unsigned y = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) y |= ((unsigned)(x >> k) & 1u) << (4 * k + b);
        spread[i] = y;
        const int h0 = i & 15, h1 = (i >> 4) & 15;
        pair[i] = make_float2((i & 256) ? 1.0f / (float)(h0 + 1) : 0.0f, (i & 512) ? 1.0f / (float)(h1 + 1) : 0.0f);
    }
    __syncthreads();
    if (max_hop_dev) {
        const int m = *max_hop_dev;
        n_hop_bits = m > 0 ? 32 - __clz(m) : 0;
    }
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const unsigned hpr_mask = (1u << hpr_shift) - 1u;
    const long long total = (long long)N << hpr_shift;                              // half-words in all (< 2^31: checked on the host)
    // Which half-word slot o (0 .. 63) of batch `batch` is: node v, half-word hw of the output row, and where its plane dwords lie.
    //  rows_shift < 0: the flat (row, half-word) sequence, 64 consecutive half-words a batch.
    //  rows_shift >= 0 (several shards whose rows are shorter than a batch): a batch is ONE shard's half-words of 2^rows_shift consecutive
    //    nodes -- 256 contiguous bytes of each plane, where the flat order reads eight 32-byte pieces from eight shards (R-MAT scale 22,
    //    8 shards x 64 anchors: 2.79 -> 1.94 ms; Flickr-shaped, 8 x 256: 196-211 -> 170 us); the batches of a row block in the other shards are the neighbouring waves'.
    const int hps_shift = wps_shift + 1, shards_shift = hpr_shift - hps_shift;
    const int batches = rows_shift < 0 ? (int)((total + 63) >> 6) : (((N + (1 << rows_shift) - 1) >> rows_shift) << shards_shift);
    const unsigned *planes32 = reinterpret_cast<const unsigned *>(planes);
    struct Slot { unsigned v, hw; size_t off; bool ok; };
    auto slot_of = [&](int batch, int o) {
        Slot t;
        if (rows_shift < 0) {
            const long long g = (long long)batch * 64 + o;
            t.ok = g < total;
            t.v = (unsigned)(g >> hpr_shift);
            t.hw = (unsigned)g & hpr_mask;
        } else {
            const unsigned shard = (unsigned)batch & ((1u << shards_shift) - 1u), rb = (unsigned)batch >> shards_shift;
            t.v = (rb << rows_shift) + ((unsigned)o >> hps_shift);
            t.hw = (shard << hps_shift) | ((unsigned)o & ((1u << hps_shift) - 1u));
            t.ok = t.v < (unsigned)N;
        }
        const unsigned word = t.hw >> 1, shard = word >> wps_shift, wl = word & ((1u << wps_shift) - 1u);
        t.off = (((size_t)shard * shard_elems + (size_t)t.v * Wp + wl) << 1) + (t.hw & 1u);
        return t;
    };
    struct Halves { unsigned w[5]; };
    auto load = [&](int batch) {
        Halves r;
#pragma unroll
        for (int b = 0; b < 5; ++b) r.w[b] = 0;
        const Slot t = slot_of(batch, lane);
        if (t.ok) {
            const unsigned *p = planes32 + t.off;
            r.w[0] = p[0];
            if (n_hop_bits > 0) r.w[1] = p[2 * plane_elems];
            if (n_hop_bits > 1) r.w[2] = p[4 * plane_elems];
            if (n_hop_bits > 2) r.w[3] = p[6 * plane_elems];
            if (n_hop_bits > 3) r.w[4] = p[8 * plane_elems];
        }
        return r;
    };
    if (wave >= batches) return;
    Halves cur = load(wave);
    for (int batch = wave; batch < batches; batch += nwaves) {
        Halves nxt = cur;
        if (batch + nwaves < batches) nxt = load(batch + nwaves);
        // this lane's 32 floats, as 8 pieces of 16 bytes, into the wave's image: piece j of lane l at l * 128 + ((j ^ (l & 7)) * 16)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned code = spread[(cur.w[1] >> (8 * k)) & 255u] | spread[256 + ((cur.w[2] >> (8 * k)) & 255u)] |
                                  spread[512 + ((cur.w[3] >> (8 * k)) & 255u)] | spread[768 + ((cur.w[4] >> (8 * k)) & 255u)];
            const unsigned reach = (cur.w[0] >> (8 * k)) & 255u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {                                            // two pairs = one 16-byte piece
                const float2 a = pair[((code >> (16 * q)) & 255u) | (((reach >> (4 * q)) & 3u) << 8)];
                const float2 b = pair[((code >> (16 * q + 8)) & 255u) | (((reach >> (4 * q + 2)) & 3u) << 8)];
                const int j = 2 * k + q;
                *reinterpret_cast<float4 *>(image + lane * 128 + ((j ^ (lane & 7)) << 4)) = make_float4(a.x, a.y, b.x, b.y);
            }
        }
        // store instruction e writes bytes [1024 e, 1024 e + 1024) of the batch's 8 KB: lane l takes piece l & 7 of owner 8 e + (l >> 3)
        // (a wave's LDS instructions execute in order: no barrier between the writes above and these reads)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int o = 8 * e + (lane >> 3), pc = lane & 7;
            const float4 val = *reinterpret_cast<const float4 *>(image + o * 128 + ((pc ^ (o & 7)) << 4));
            const Slot t = slot_of(batch, o);
            if (t.ok) *reinterpret_cast<float4 *>(out + (size_t)t.v * out_cols + col0 + t.hw * 32 + pc * 4) = val;
        }
        cur = nxt;
    }
}

__global__ __launch_bounds__(256) void k_hops(const u64 *__restrict__ planes, size_t plane_elems, int n_hop_bits,
                                              int N, int K, int Wp, int *__restrict__ hops) {
    const size_t total = (size_t)N * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i / K), j = (int)(i % K);
        const size_t widx = (size_t)v * Wp + (j >> 6);
        const int bit = j & 63;
        int h = -1;
        if ((planes[widx] >> bit) & 1ull) {
            h = 0;
            for (int b = 0; b < n_hop_bits; ++b)
                h |= (int)((planes[(size_t)(b + 1) * plane_elems + widx] >> bit) & 1ull) << b;
        }
        hops[i] = h;
    }
}

// Transport form of the embedding for the host -> host boundary: one byte per (node, anchor), 0 = no path, c = hops + 1
// otherwise (the caller has checked max hop <= 254), plus the 256 floats the bytes stand for -- lut[c] = 1 / c computed
// HERE with the finalise kernel's own expression, so the host only looks values up.  A quarter of the float matrix's bytes
// cross PCIe.  Wave-per-row-block like k_finalize_fast; a lane turns four anchors of one plane word into one 32-bit store.
__global__ __launch_bounds__(256) void k_hop_codes(const u64 *__restrict__ planes, size_t plane_elems, int n_hop_bits, int N, int K, int Wp,
                                                   unsigned char *__restrict__ codes, long long pitch, float *__restrict__ lut) {
    if (blockIdx.x == 0) lut[threadIdx.x] = threadIdx.x ? 1.0f / (float)threadIdx.x : 0.0f;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int per = (N + nwaves - 1) / nwaves;
    const int v_begin = wave * per, v_end = min(N, v_begin + per);
    const int K4 = (K + 3) >> 2;
    const bool words = (K & 3) == 0 && (pitch & 3) == 0;
    for (int v = v_begin; v < v_end; ++v) {
        unsigned char *row = codes + (size_t)v * pitch;
        const size_t wbase = (size_t)v * Wp;
        for (int q = lane; q < K4; q += 64) {
            const int j = q * 4;
            const size_t widx = wbase + (j >> 6);
            const int bit = j & 63;
            const unsigned reach = (unsigned)(planes[widx] >> bit) & 15u;
            unsigned c[4] = {0, 0, 0, 0};
            for (int b = 0; b < n_hop_bits; ++b) {
                const unsigned t = (unsigned)(planes[(size_t)(b + 1) * plane_elems + widx] >> bit) & 15u;
                c[0] |= (t & 1u) << b; c[1] |= ((t >> 1) & 1u) << b; c[2] |= ((t >> 2) & 1u) << b; c[3] |= ((t >> 3) & 1u) << b;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = ((reach >> i) & 1u) ? c[i] + 1u : 0u;
            if (words) {
                reinterpret_cast<unsigned *>(row)[q] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (j + i < K) row[j + i] = (unsigned char)c[i];
            }
        }
    }
}

// Per-anchor column statistics of the hop matrix straight from the planes: how many nodes reach anchor j and the sum of
// their hop counts (closeness centrality = inward distances, utils.py:50-54).  Thread t of a block owns anchor column
// tile * 256 + t and walks a slice of the rows; 64 threads share each plane word (one L1 line).  Two deterministic stages.
__global__ __launch_bounds__(256) void k_column_stats_partial(const u64 *__restrict__ planes, size_t plane_elems, int n_hop_bits,
                                                              int N, int K, int Wp, long long *__restrict__ part_sum,
                                                              long long *__restrict__ part_cnt) {
    const int j = blockIdx.y * 256 + threadIdx.x;
    const int per = (N + gridDim.x - 1) / gridDim.x;
    const int v0 = blockIdx.x * per, v1 = min(N, v0 + per);
    long long sum = 0, cnt = 0;
    if (j < K) {
        const int w = j >> 6, bit = j & 63;
        for (int v = v0; v < v1; ++v) {
            const size_t widx = (size_t)v * Wp + w;
            if ((planes[widx] >> bit) & 1ull) {
                int h = 0;
                for (int b = 0; b < n_hop_bits; ++b) h |= (int)((planes[(size_t)(b + 1) * plane_elems + widx] >> bit) & 1ull) << b;
                sum += h;
                ++cnt;
            }
        }
        part_sum[(size_t)blockIdx.x * K + j] = sum;
        part_cnt[(size_t)blockIdx.x * K + j] = cnt;
    }
}

__global__ __launch_bounds__(256) void k_column_stats_final(const long long *__restrict__ part_sum, const long long *__restrict__ part_cnt,
                                                            int parts, int K, long long *__restrict__ hop_sum, long long *__restrict__ reach) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= K) return;
    long long s = 0, c = 0;
    for (int p = 0; p < parts; ++p) {
        s += part_sum[(size_t)p * K + j];
        c += part_cnt[(size_t)p * K + j];
    }
    hop_sum[j] = s;
    reach[j] = c;
}

// out[v, 0:F] = x[v, :].  Every wave owns a contiguous block of rows (see k_finalize_fast).
__global__ __launch_bounds__(256) void k_concat(const float *__restrict__ x, int N, int F, float *__restrict__ out,
                                                long long out_cols, bool vec) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int per = (N + nwaves - 1) / nwaves;
    const int v_begin = wave * per, v_end = min(N, v_begin + per);
    for (int v = v_begin; v < v_end; ++v) {
        const float *xrow = x + (size_t)v * F;
        float *orow = out + (size_t)v * out_cols;
        if (vec) {
            const f32x4 *xs = reinterpret_cast<const f32x4 *>(xrow);
            f32x4 *os = reinterpret_cast<f32x4 *>(orow);
            for (int q = lane; q < F / 4; q += 64) os[q] = __builtin_nontemporal_load(xs + q);
        } else {
            for (int c = lane; c < F; c += 64) orow[c] = xrow[c];
        }
    }
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace pope

using namespace pope;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int aux_cap(int64_t E) { return (int)((E + CHUNK - 1) / CHUNK) + 1; }

extern "C" size_t pope_csr_aux_elems(int64_t E) { return E < 0 ? 0 : (size_t)AUX_HEADER + (size_t)aux_cap(E); }

struct CsrScratch {
    size_t cnt, scan_tmp, cols, sort_tmp, total;
};

static CsrScratch csr_scratch_layout(int64_t N, int64_t E) {
    // cnt[N + 1] | rocPRIM scan temp | unsorted targets [E] | rocPRIM segmented-sort temp   (only used for edge lists that
    // are not sorted by source)
    CsrScratch L;
    size_t o = 0;
    L.cnt = o;      o += align_up((size_t)(N + 1) * sizeof(int), 256);
    L.scan_tmp = o; o += align_up(scan_temp_bytes((size_t)N + 1), 256);
    L.cols = o;     o += align_up((size_t)(E > 0 ? E : 1) * sizeof(int), 256);
    L.sort_tmp = o; o += align_up(rowsort_temp_bytes((size_t)E, (size_t)N), 256);
    L.total = o;
    return L;
}

extern "C" size_t pope_csr_scratch_bytes(int64_t N, int64_t E) {
    if (N < 0 || E < 0) return 0;
    return csr_scratch_layout(N, E).total;
}

static int csr_fallback(const long long *src, const long long *dst, int E, int N, int *rowptr, int *col, int *erow,
                        int *aux, void *scratch, hipStream_t stream) {
    const CsrScratch L = csr_scratch_layout(N, E);
    int *cnt = (int *)((char *)scratch + L.cnt);
    void *scan_tmp = (char *)scratch + L.scan_tmp;
    int *cols_unsorted = (int *)((char *)scratch + L.cols);
    void *sort_tmp = (char *)scratch + L.sort_tmp;
    size_t scan_bytes = scan_temp_bytes((size_t)N + 1), sort_bytes = rowsort_temp_bytes((size_t)E, (size_t)N);
    POPE_HIP(hipMemsetAsync(cnt, 0, (size_t)(N + 1) * sizeof(int), stream));
    hipLaunchKernelGGL(k_csr_count, dim3(capped_grid(E, 256)), dim3(256), 0, stream, src, E, cnt);
    POPE_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, cnt, rowptr, 0, (size_t)N + 1, rocprim::plus<int>(), stream));
    POPE_HIP(hipMemsetAsync(cnt, 0, (size_t)(N + 1) * sizeof(int), stream));
    hipLaunchKernelGGL(k_csr_scatter, dim3(capped_grid(E, 256)), dim3(256), 0, stream, src, dst, E, rowptr, cnt, cols_unsorted, erow);
    // every row's targets in ascending order: the result no longer depends on which edge won which atomic cursor value
    POPE_HIP(rocprim::segmented_radix_sort_keys(sort_tmp, sort_bytes, (const int *)cols_unsorted, col, (unsigned)E, (unsigned)N,
                                                (const int *)rowptr, (const int *)rowptr + 1, 0, 32, stream));
    POPE_HIP(hipMemsetAsync(aux, 0, AUX_HEADER * sizeof(int), stream));
    hipLaunchKernelGGL(k_csr_lists, dim3(capped_grid((size_t)aux_cap(E), 256)), dim3(256), 0, stream, rowptr, erow, E, aux);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

struct SeedArgs {                         // pope_geodesic_run: seed the BFS from the CSR launch
    const long long *anchors = nullptr;
    int K = 0, Wp = 0;
    u64 *seen = nullptr, *front = nullptr;
    unsigned *live = nullptr;
};

static int csr_build(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr, int32_t *col, int32_t *erow,
                     int32_t *aux, void *scratch, size_t scratch_bytes, int32_t defer_check, const SeedArgs &seed,
                     hipStream_t stream);

extern "C" int pope_csr_build(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr, int32_t *col,
                              int32_t *erow, int32_t *aux, void *scratch, size_t scratch_bytes, int32_t defer_check,
                              void *stream_) {
    clear_error();
    return csr_build(edge_index, E, N, rowptr, col, erow, aux, scratch, scratch_bytes, defer_check, SeedArgs(), (hipStream_t)stream_);
}

// The canonical form regardless of the input order: counting sort by source, every row's targets ascending (repeated
// edges adjacent).  What the rankings need (distinct-neighbour counts, SciPy's accumulation order); synchronises once.
extern "C" int pope_csr_build_canonical(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr, int32_t *col, int32_t *erow,
                                        int32_t *aux, void *scratch, size_t scratch_bytes, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(N > 0 && N < INT32_MAX && E >= 0 && E < INT32_MAX, "pope_csr_build_canonical: need 0 < N < 2^31, 0 <= E < 2^31");
    POPE_REQUIRE(rowptr && aux && scratch && ((edge_index && col && erow) || E == 0), "pope_csr_build_canonical: null pointer");
    if (scratch_bytes < pope_csr_scratch_bytes(N, E)) {
        set_error("pope_csr_build_canonical: scratch %zu < %zu bytes", scratch_bytes, pope_csr_scratch_bytes(N, E));
        return POPE_ERR_WORKSPACE;
    }
    if (E == 0) {
        POPE_HIP(hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * sizeof(int), stream));
        POPE_HIP(hipMemsetAsync(aux, 0, AUX_HEADER * sizeof(int), stream));
        return POPE_OK;
    }
    // index check first: the counting pass indexes its histogram with the source ids
    int *flag = (int *)scratch;
    POPE_HIP(hipMemsetAsync(flag, 0, sizeof(int), stream));
    hipLaunchKernelGGL(k_index_check, dim3(capped_grid((size_t)2 * E, 256)), dim3(256), 0, stream, (const long long *)edge_index, 2 * E, (long long)N, flag);
    int bad = 0;
    POPE_HIP(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    POPE_HIP(hipStreamSynchronize(stream));
    if (bad) {
        set_error("pope_csr_build_canonical: edge_index holds a node id outside [0, %lld)", (long long)N);
        return POPE_ERR_INDEX;
    }
    return csr_fallback((const long long *)edge_index, (const long long *)edge_index + E, (int)E, (int)N, rowptr, col, erow, aux, scratch, stream);
}

static int csr_build(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr, int32_t *col, int32_t *erow,
                     int32_t *aux, void *scratch, size_t scratch_bytes, int32_t defer_check, const SeedArgs &seed,
                     hipStream_t stream) {
    POPE_REQUIRE(N >= 0 && N < INT32_MAX && E >= 0 && E < INT32_MAX, "pope_csr_build: need 0 <= N, E < 2^31 (N=%lld E=%lld)",
                 (long long)N, (long long)E);
    POPE_REQUIRE(rowptr && aux && scratch && ((edge_index && col && erow) || E == 0), "pope_csr_build: null pointer");
    if (scratch_bytes < pope_csr_scratch_bytes(N, E)) {
        set_error("pope_csr_build: scratch %zu < %zu bytes", scratch_bytes, pope_csr_scratch_bytes(N, E));
        return POPE_ERR_WORKSPACE;
    }
    const long long *src = (const long long *)edge_index, *dst = src + E;
    if (defer_check != 2)                                       // 2 (internal): the caller's clear kernel zeroed the header
        POPE_HIP(hipMemsetAsync(aux, 0, AUX_HEADER * sizeof(int), stream));
    if (E == 0) {
        POPE_HIP(hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * sizeof(int), stream));
        if (seed.K > 0)
            hipLaunchKernelGGL(k_bfs_seed, dim3((seed.K + 255) / 256), dim3(256), 0, stream, seed.anchors, seed.K, seed.Wp, seed.seen,
                               seed.front, seed.live);
        return POPE_OK;
    }
    if ((E & 1) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(col) | reinterpret_cast<uintptr_t>(erow)) & 15u) == 0)
        hipLaunchKernelGGL(k_csr_sorted<true>, dim3(capped_grid(E / 2, 256)), dim3(256), 0, stream, src, dst, (int)E, (int)N, rowptr, col, erow, aux,
                           seed.anchors, seed.K, seed.Wp, seed.seen, seed.front, seed.live);
    else
        hipLaunchKernelGGL(k_csr_sorted<false>, dim3(capped_grid(E, 256)), dim3(256), 0, stream, src, dst, (int)E, (int)N, rowptr, col, erow, aux,
                           seed.anchors, seed.K, seed.Wp, seed.seen, seed.front, seed.live);
    POPE_HIP(hipGetLastError());
    if (defer_check) return POPE_OK;                 // pope_geodesic_bfs reports what the speculative pass found
    int flags = 0;
    POPE_HIP(hipMemcpyAsync(&flags, aux + AUX_FLAGS, sizeof(int), hipMemcpyDeviceToHost, stream));
    POPE_HIP(hipStreamSynchronize(stream));
    if (flags & CSR_FLAG_BAD_INDEX) {
        set_error("pope_csr_build: edge_index holds a node id outside [0, %lld)", (long long)N);
        return POPE_ERR_INDEX;
    }
    if (flags & CSR_FLAG_UNSORTED) return csr_fallback(src, dst, (int)E, (int)N, rowptr, col, erow, aux, scratch, stream);
    return POPE_OK;
}

extern "C" int32_t pope_words(int32_t K) { return K <= 0 ? 0 : words_for(K); }

extern "C" size_t pope_plane_bytes(int64_t N, int32_t K) {
    if (N < 0 || K <= 0) return 0;
    return (size_t)N * words_for(K) * sizeof(u64);
}

static size_t live_bytes(int64_t N) { return align_up((size_t)((N + 31) / 32) * sizeof(unsigned), 256); }
// the live tables' summary (LIVE = 3): one bit per table word, written two words per 64 table words (k_live_summary), padded like the tables
static size_t live_sum_bytes(int64_t N) { return align_up(live_bytes(N) / 32, 256); }

extern "C" size_t pope_bfs_scratch_bytes(int64_t N, int64_t E, int32_t K) {
    if (N < 0 || E < 0 || K <= 0) return 0;
    (void)E;
    // control block | anchors[K] | three rotating frontier planes | their three live-bit tables | one summary of a live table
    return CTL_BYTES + align_up((size_t)K * sizeof(long long), 256) + 3 * align_up(pope_plane_bytes(N, K), 256) + 3 * live_bytes(N) + live_sum_bytes(N);
}

constexpr int LIVE_MAX_NODES = 256 * 1024;   // live table of 32 KB per block in LDS (4 blocks per CU); beyond: read from global
constexpr size_t LIVE_SUM_MAX_BYTES = 48 * 1024;   // ... behind a summary of at most this size in LDS (12.5 M nodes); beyond: the global table alone
constexpr int EAGER_PLANES = 4;      // hop-bit planes cleared up front (levels < 16); deeper ones when first needed

// Optional per-launch timing of the level kernels with HIP events on the launch stream (bench.py's roofline leg).
struct LevelProfile {
    bool enabled = false;
    bool span_only = false;              // mode 2: one event pair around each enqueued run of levels, not around every launch
    std::vector<hipEvent_t> ev;          // 2 per level: before and after the level kernel
    std::vector<int> level;
};
static LevelProfile g_profile;

static void profile_mark(hipStream_t stream, int level, int which, bool span = false) {
    if (!g_profile.enabled || g_profile.span_only != span) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, stream);
    g_profile.ev.push_back(e);
    if (which == 0) g_profile.level.push_back(level);
}

// Diagnostic knobs behind pope_debug_set() (include/graphpope_hip.h): process-global, not thread-safe, A/B tooling only.
static int g_live_mode = -1;            // -1: by graph size (LDS table up to LIVE_MAX_NODES, global table beyond); 2: the global table on a small graph too (tests)

// Which instantiation of k_bfs_level a graph of N nodes and Wp words per node gets (bfs_enqueue_levels launches it; bench.py and the
// profiles label it through pope_level_kernel_name).  LIVE: the live-bit table staged in LDS (1) up to LIVE_MAX_NODES, beyond that read
// from global memory behind a summary in LDS (3), or plainly (2) where even the summary does not fit.  Tiles: up to 4 words in one
// tile; more on a graph that lives in L2 (LIVE = 1): 4-word tiles, a wave each (TILES = 2); where the frontier rows come from HBM
// (LIVE >= 2): 8-word tiles if the row is made of them -- ONE gather takes everything the row has in its 128-byte line -- else 4-word
// tiles gathered in pairs, walked inside the wave (TILES = 1; a single 8-word tile: TILES = 0).
struct LevelChoice { int wt, live, tiles; };
static int live_mode_for(int64_t N) {
    const int64_t live_words = (N + 31) / 32;
    int mode = g_live_mode > 0 ? g_live_mode : (live_words <= LIVE_MAX_NODES / 32 ? 1 : 3);
    if (mode == 1 && live_words > LIVE_MAX_NODES / 32) mode = 3;
    if (mode == 3 && live_sum_bytes(N) > LIVE_SUM_MAX_BYTES) mode = 2;
    return mode;
}
static LevelChoice level_choice(int Wp, int live_mode) {
    if (Wp <= 4) return {Wp, live_mode, 0};
    if (live_mode == 1) return (POPE_WT8_L2 && Wp % 8 == 0) ? LevelChoice{8, 1, POPE_WT8_L2 == 2 && Wp > 8 ? 1 : 2} : LevelChoice{4, 1, 2};
    if (POPE_WT8 && Wp % 8 == 0) return {8, live_mode, Wp == 8 ? 0 : 1};
    return {4, live_mode, 1};
}
static int g_finalize_variant = 1;      // 1: pipelined / wide fast paths (default), 7: round 1-3 fast path, 0: generic kernel -- kept so the tests can compare their bits
static int g_finalize_blocks = 256 * 8;
static bool g_finalize_blocks_set = false;   // POPE_KNOB_FINALIZE_BLOCKS given: it also sizes the pipelined kernels (default: one work item per wave)
static int g_finalize_shard_batches = 1;   // k_finalize_lut over several short-rowed shards: 1 = a batch per (shard, block of rows), 0 = the flat order (POPE_KNOB_FINALIZE_VARIANT 11 / 12)
static int g_finalize_lut = 1;           // wide rows: 1 (default) k_finalize_lut for rows without features, k_finalize_wide with them (copy kernel + table kernel measured slower: Flickr / 1 024 anchors 0.619 against 0.562 ms); 2 always; 0 never -- POPE_KNOB_FINALIZE_VARIANT 8 / 9 / 10
static int g_prepare_merge = 1;          // POPE_KNOB_PREPARE_MERGE: 1 (default) = pope_geodesic_run clears, seeds and builds the CSR in ONE launch (k_prepare); 0 = two launches
namespace pope { int g_streamk_xcd = 1; int g_gemm_force_tile = 0, g_pairwise_kernel = 0, g_fail_host_register = 0, g_sage_forward_overlap = 1, g_gemm_small_tile16 = 1, g_gemm_tile16_buffers = 4; }

extern "C" int pope_debug_set(int32_t knob, int32_t value) {
    clear_error();
    switch (knob) {
    case POPE_KNOB_LIVE_MODE:        g_live_mode = value; break;
    case POPE_KNOB_FINALIZE_VARIANT:                                 // 8 / 9: the default kernels, but shapes with features keep k_finalize_wide (8) or not (9)
        if (value >= 8 && value <= 10) { g_finalize_variant = 1; g_finalize_lut = value == 9 ? 2 : value == 8 ? 1 : 0; }
        else if (value == 11 || value == 12) g_finalize_shard_batches = value == 11;     // k_finalize_lut's batch order over several shards
        else g_finalize_variant = value;
        break;
    case POPE_KNOB_FINALIZE_BLOCKS:  g_finalize_blocks = value > 0 ? value : 256 * 8; g_finalize_blocks_set = value > 0; break;
    case POPE_KNOB_GEMM_TILE:        pope::g_gemm_force_tile = value; break;
    case POPE_KNOB_PAIRWISE_KERNEL:  pope::g_pairwise_kernel = value; break;
    case POPE_KNOB_COPY_BATCHES:     pope::g_copy_batches_per_wave = value; break;
    case POPE_KNOB_FAIL_HOST_REGISTER: pope::g_fail_host_register = value; break;
    case POPE_KNOB_SAGE_FORWARD_OVERLAP: pope::g_sage_forward_overlap = value != 0; break;
    case POPE_KNOB_GEMM_TILE16_BUFFERS: pope::g_gemm_tile16_buffers = value == 4 ? 4 : 3; break;
    case POPE_KNOB_GEMM_SMALL_TILE16: pope::g_gemm_small_tile16 = value; break;
    case POPE_KNOB_PREPARE_MERGE:    g_prepare_merge = value; break;
    case POPE_KNOB_STREAMK_XCD:      pope::g_streamk_xcd = value != 0; break;
    default: set_error("pope_debug_set: unknown knob %d", knob); return POPE_ERR_INVALID;
    }
    return POPE_OK;
}

template <int WT, int TILES>
static void launch_level(int E, int N, int Wp, const int *col, const int *erow, const int *aux, const u64 *front, u64 *seen,
                         u64 *acc, u64 *idle, u64 *hop_planes, size_t plane_elems, int level, BfsCtl *ctl,
                         const unsigned *live, unsigned *live_acc, unsigned *live_idle, int live_words, int mode, unsigned *live_sum, hipStream_t stream) {
    const int nchunks = (E + CHUNK - 1) >> CHUNK_SHIFT;
    const int tiles = Wp / WT;
    int expand_blocks = (nchunks + 3) / 4;                           // one wave per chunk ...
    if (expand_blocks > 256 * 8) expand_blocks = 256 * 8;            // ... up to 8 blocks per CU, then waves loop
    if (TILES == 2) expand_blocks *= tiles;                          // ... and per tile (4 waves per block: the tiles of a chunk share a block for 1, 2 or 4 tiles)
    int house_blocks = (N + 255) / 256;                              // the commit of the previous level: one thread per node
    if (house_blocks > 1024) house_blocks = 1024;                    // (+ the clears: rows that span chunks, the live table)
    const int gx = expand_blocks + house_blocks;                     // the housekeeping blocks come last
    profile_mark(stream, level, 0);
    const int padded_words = (int)(align_up((size_t)live_words * sizeof(unsigned), 256) / sizeof(unsigned));
    const int sum_words = padded_words / 32;
    if (mode == 1) {
        hipLaunchKernelGGL((k_bfs_level<WT, 1, TILES>), dim3(gx), dim3(256), align_up((size_t)live_words * sizeof(unsigned), 16), stream, erow, col, E, N,
                           Wp, tiles, front, seen, acc, idle, hop_planes, plane_elems, level, ctl, aux, expand_blocks, live, live_acc, live_idle, live_words,
                           (const unsigned *)nullptr, 0);
    } else if (mode == 3) {
        // the summary of the table this launch reads (complete since the previous launch ended), then the level
        hipLaunchKernelGGL(k_live_summary, dim3((padded_words + 255) / 256), dim3(256), 0, stream, live, padded_words, live_sum);
        hipLaunchKernelGGL((k_bfs_level<WT, 3, TILES>), dim3(gx), dim3(256), align_up((size_t)sum_words * sizeof(unsigned), 16), stream, erow, col, E, N,
                           Wp, tiles, front, seen, acc, idle, hop_planes, plane_elems, level, ctl, aux, expand_blocks, live, live_acc, live_idle, live_words,
                           (const unsigned *)live_sum, sum_words);
    } else {
        hipLaunchKernelGGL((k_bfs_level<WT, 2, TILES>), dim3(gx), dim3(256), 0, stream, erow, col, E, N, Wp, tiles, front, seen, acc, idle, hop_planes,
                           plane_elems, level, ctl, aux, expand_blocks, live, live_acc, live_idle, live_words, (const unsigned *)nullptr, 0);
    }
    profile_mark(stream, level, 1);
}

// Per-device host-side context, created on first use (the only objects the library ever keeps): a ring of small
// pinned, device-mapped host SLOTS.  Every call takes a slot of its own: its anchors are staged there (the seed kernel
// reads them in place -- a pageable hipMemcpyAsync would be a synchronous staging copy) and its BFS verdict comes back
// there (the finalise / report kernel writes it straight into host memory: no copy kernel, one stream sync).  A slot is
// handed out again only after the event recorded behind its last device-side user has completed, so calls on other
// streams or from other host threads never share staging memory; the ring is guarded by a mutex.
// (Measured and rejected: running the feature copy out[:, :F] = x on a side stream underneath the BFS levels.  The
// streaming copy saturates the memory queues and the latency-bound level kernels run 2-4x slower beside it; the
// serial order is faster.)
struct Slot {
    int *report = nullptr;               // pinned host: [0] last_active, [1] csr flags, [2] ticket of the call that wrote them
    int *report_dev = nullptr;           // the same memory as seen from the device
    long long *anchors = nullptr;        // pinned, device-mapped host staging for the anchor ids
    long long *anchors_dev = nullptr;    // the same memory as seen from the device (the seed kernel reads it in place)
    size_t anchors_cap = 0;
    hipEvent_t ev = nullptr;             // recorded behind the last kernel that reads / writes this slot
    bool busy = false;
    int ticket = 0;
};
constexpr int N_SLOTS = 8;
struct DeviceCtx {
    std::mutex mu;
    Slot slots[N_SLOTS];
    unsigned next = 0;
};
static DeviceCtx g_ctx[64];

// Take the next slot of the current device's ring (waits for its previous user), sized for n_anchors ids.
static int slot_acquire(Slot **out, size_t n_anchors) {
    int dev = 0;
    const hipError_t de = hipGetDevice(&dev);
    if (de == hipErrorNoDevice || de == hipErrorInvalidDevice) {
        set_error("no gfx950 device visible (%s)", hipGetErrorString(de));
        return POPE_ERR_NO_DEVICE;
    }
    POPE_HIP(de);
    POPE_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    DeviceCtx &c = g_ctx[dev];
    std::lock_guard<std::mutex> lock(c.mu);
    Slot &s = c.slots[c.next++ % N_SLOTS];
    if (s.busy) {
        POPE_HIP(hipEventSynchronize(s.ev));
        s.busy = false;
    }
    if (!s.report) {
        POPE_HIP(hipHostMalloc((void **)&s.report, 256, hipHostMallocMapped | hipHostMallocCoherent));   // fine-grained: visible mid-kernel
        memset(s.report, 0, 256);
        POPE_HIP(hipHostGetDevicePointer((void **)&s.report_dev, s.report, 0));
        POPE_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    }
    if (n_anchors > s.anchors_cap) {
        if (s.anchors) POPE_HIP(hipHostFree(s.anchors));
        s.anchors = nullptr;
        s.anchors_cap = 0;
        const size_t cap = n_anchors < 1024 ? 1024 : n_anchors;
        POPE_HIP(hipHostMalloc((void **)&s.anchors, cap * sizeof(long long), hipHostMallocMapped));
        POPE_HIP(hipHostGetDevicePointer((void **)&s.anchors_dev, s.anchors, 0));
        s.anchors_cap = cap;
    }
    *out = &s;
    return POPE_OK;
}

// Everything enqueued on `stream` so far may use the slot; it becomes reusable once that work has completed.
static void slot_release(Slot *s, hipStream_t stream) {
    if (s && hipEventRecord(s->ev, stream) == hipSuccess) s->busy = true;
}

struct SlotGuard {                       // releases the call's slot on every return path
    Slot *const *slot;
    hipStream_t stream;
    // Set once the host has SEEN that the device is through with the slot (the verdict arrived, or the stream was
    // synchronised): no event then -- an event record at the end of every call put a ~7 us bubble in front of the next
    // call's first kernel.
    bool quiescent = false;
    ~SlotGuard() {
        if (!quiescent) slot_release(*slot, stream);
    }
};

// The BFS verdict (deepest active level, CSR status flags) written straight into pinned host memory.
__global__ void k_bfs_report(const BfsCtl *ctl, const int *aux, int *report) {
    report[0] = ctl->last_active;
    report[1] = csr_flags(ctl, aux);
    __threadfence_system();
}

// Everything one BFS needs, carved out of the caller's buffers.
struct Bfs {
    const int *rowptr, *col, *erow, *aux;
    int N, E, K, Wp, capacity;
    size_t plane_elems, plane_bytes, front_off;
    u64 *seen, *hop_planes, *front[3];
    unsigned *live[3];           // one bit per node beside each frontier buffer: row not all zero
    int live_words;
    bool frontiers_cleared = false;  // the merged prepare launch zeroed the three frontier buffers (else: k_clear_spanning_rows in front of level 1)
    unsigned *live_sum;          // LIVE = 3: summary of the table the next level launch reads (one bit per table word)
    int live_mode;               // 1 / 2 / 3 (k_bfs_level's LIVE)
    char *base;
    BfsCtl *ctl;
    long long *anchors_dev;
    long long level_limit;       // levels 1 .. limit-1 fit `capacity` hop bits
    Slot *slot;                  // this call's pinned staging (anchors, verdict)
};

constexpr double POPE_POLL_TIMEOUT_S = 30.0;   // wall-clock bound of the host spin on the verdict word
constexpr int LEVEL_BATCH = 12;     // levels enqueued between two polls of the device flag (hops <= 10: one poll)

// pope_geodesic_run's speculative window remembers how deep the previous call on the same device and the same sizes went:
// a level launch that only finds "the BFS is over" still costs 4.5 us (Flickr: two of the twelve).  A wrong guess is not an
// error: a deeper graph answers "not done" and the call continues on the general path, a shallower one runs spare launches.
// After a guess that was too shallow the window keeps one spare level for these sizes (anchor sets whose depth wanders by one).
struct DepthHint {
    std::mutex mu;
    int64_t N = -1, E = -1;
    int K = -1, last_active = 0, margin = 1;
};
constexpr int MAX_DEVICES = 64;        // like g_ctx: device indices beyond it simply get no hint
static DepthHint g_depth_hint[MAX_DEVICES];

static int speculative_window(int64_t N, int64_t E, int K) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return LEVEL_BATCH;
    std::lock_guard<std::mutex> lock(g_depth_hint[dev].mu);
    const DepthHint &h = g_depth_hint[dev];
    if (h.N == N && h.E == E && h.K == K) return std::min(LEVEL_BATCH, h.last_active + h.margin);   // margin 1: one level past the last one that found something
    return LEVEL_BATCH;
}

static void remember_depth(int64_t N, int64_t E, int K, int last_active, bool window_was_too_short = false) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_depth_hint[dev].mu);
    DepthHint &h = g_depth_hint[dev];
    const bool same = h.N == N && h.E == E && h.K == K;
    h.margin = same ? (window_was_too_short ? 2 : h.margin) : 1;
    h.N = N; h.E = E; h.K = K; h.last_active = last_active;
}

static int bfs_setup(Bfs &b, const int32_t *rowptr, const int32_t *col, const int32_t *erow, const int32_t *aux,
                     int64_t N, int64_t E, const int64_t *anchors_host, int32_t K, uint64_t *planes,
                     int32_t plane_capacity, void *scratch, size_t scratch_bytes) {
    POPE_REQUIRE(N > 0 && N < INT32_MAX && E >= 0 && E < INT32_MAX, "geodesic bfs: need 0 < N < 2^31, 0 <= E < 2^31");
    POPE_REQUIRE(K > 0 && plane_capacity >= 1 && plane_capacity <= 31, "geodesic bfs: need K > 0 and 1 <= plane_capacity <= 31");
    POPE_REQUIRE(rowptr && aux && ((erow && col) || E == 0) && anchors_host && planes && scratch, "geodesic bfs: null pointer");
    if (scratch_bytes < pope_bfs_scratch_bytes(N, E, K)) {
        set_error("geodesic bfs: scratch %zu < %zu bytes", scratch_bytes, pope_bfs_scratch_bytes(N, E, K));
        return POPE_ERR_WORKSPACE;
    }
    for (int j = 0; j < K; ++j)
        if (anchors_host[j] < 0 || anchors_host[j] >= N) {
            set_error("geodesic bfs: anchor %d = %lld outside [0, %lld)", j, (long long)anchors_host[j], (long long)N);
            return POPE_ERR_INDEX;
        }
    b.rowptr = rowptr; b.col = col; b.erow = erow; b.aux = aux;
    b.N = (int)N; b.E = (int)E; b.K = K; b.Wp = words_for(K); b.capacity = plane_capacity;
    b.plane_elems = (size_t)N * b.Wp;
    b.plane_bytes = b.plane_elems * sizeof(u64);
    b.seen = (u64 *)planes;
    b.hop_planes = b.seen + b.plane_elems;
    b.base = (char *)scratch;
    b.ctl = (BfsCtl *)b.base;
    static_assert(sizeof(BfsCtl) <= CTL_BYTES, "control block");
    b.front_off = CTL_BYTES + align_up((size_t)K * sizeof(long long), 256);
    b.anchors_dev = (long long *)(b.base + CTL_BYTES);
    b.front[0] = (u64 *)(b.base + b.front_off);
    b.front[1] = (u64 *)((char *)b.front[0] + align_up(b.plane_bytes, 256));
    b.front[2] = (u64 *)((char *)b.front[1] + align_up(b.plane_bytes, 256));
    b.live[0] = (unsigned *)((char *)b.front[2] + align_up(b.plane_bytes, 256));
    b.live[1] = (unsigned *)((char *)b.live[0] + live_bytes(N));
    b.live[2] = (unsigned *)((char *)b.live[1] + live_bytes(N));
    b.live_words = (int)((N + 31) / 32);
    b.live_sum = (unsigned *)((char *)b.live[2] + live_bytes(N));
    b.live_mode = live_mode_for(N);
    b.level_limit = 1ll << plane_capacity;
    return slot_acquire(&b.slot, (size_t)K);
}

// One launch clears the control block, the live tables, the reachability plane, the first hop planes and (pope_geodesic_run) the CSR
// status header.  The three frontier buffers (a third of the bytes: 0.8 of 2.15 GB for R-MAT scale 22 with 512 anchors) are NOT cleared
// (round 5): nobody reads a frontier row whose live bit is clear, so what has to be zero is only what is OR-ed into -- the anchors' rows
// of the level-0 buffer (cleared here, through the call's pinned anchor list, which the caller has filled) and the rows that span
// chunks in level 1's accumulator (k_clear_spanning_rows in front of level 1; later accumulators are cleared two levels ahead by the
// housekeeping blocks).  Tests run every entry point on workspaces filled with 0xFF.
static void bfs_enqueue_clear(const Bfs &b, int *aux_header, hipStream_t stream) {
    const int eager = b.capacity < EAGER_PLANES ? b.capacity : EAGER_PLANES;
    const size_t words = (size_t)(1 + eager) * b.plane_elems;        // odd (N * W odd, an even number of eager planes): one word behind the last 16-byte unit
    hipLaunchKernelGGL(k_zero, dim3(2048), dim3(256), 0, stream, (uint4 *)b.base, b.front_off / 16, (uint4 *)b.seen,
                       words / 2, (uint4 *)aux_header, aux_header ? (size_t)AUX_HEADER * sizeof(int) / 16 : (size_t)0,
                       (words & 1) ? b.seen + words - 1 : (u64 *)nullptr, (uint4 *)b.live[0], 3 * live_bytes(b.N) / 16,
                       (const long long *)b.slot->anchors_dev, b.K, b.front[0], b.Wp);
}

// Anchors go through pinned, device-mapped host memory and the seed kernel reads them in place: no copy kernel.
static int bfs_enqueue_seed(const Bfs &b, const int64_t *anchors_host, hipStream_t stream) {
    hipLaunchKernelGGL(k_bfs_seed, dim3((b.K + 255) / 256), dim3(256), 0, stream, b.slot->anchors_dev, b.K, b.Wp, b.seen, b.front[0], b.live[0]);
    return POPE_OK;
}

static int bfs_enqueue_init(const Bfs &b, const int64_t *anchors_host, hipStream_t stream) {
    memcpy(b.slot->anchors, anchors_host, (size_t)b.K * sizeof(long long));     // this call's pinned, device-mapped slot: the clear and the seed read it in place
    bfs_enqueue_clear(b, nullptr, stream);
    return bfs_enqueue_seed(b, anchors_host, stream);
}

// Enqueue levels [level, stop) (clipped to what the hop-bit capacity can represent); returns the next level.
static int bfs_enqueue_levels(const Bfs &b, int level, int stop, hipStream_t stream) {
    const int first = level;
    profile_mark(stream, 0, 0, true);                                // span mode: one event pair around the whole run
    for (; level < stop; ++level) {
        if (level >= b.level_limit || b.E == 0) break;
        if ((level & (level - 1)) == 0 && level >= (1 << EAGER_PLANES)) {   // first level with this hop bit
            int bit = 0;
            while ((1 << bit) < level) ++bit;
            // (8-byte stores: with an odd N * W every second plane starts 8 bytes off a 16-byte boundary)
            hipLaunchKernelGGL(k_zero_words, dim3(1024), dim3(256), 0, stream, b.hop_planes + (size_t)bit * b.plane_elems, b.plane_elems);
        }
        if (level == 1 && !b.frontiers_cleared)                  // (the frontier buffers are not cleared wholesale: bfs_enqueue_clear)
            hipLaunchKernelGGL(k_clear_spanning_rows, dim3(capped_grid((size_t)((b.E + CHUNK - 1) >> CHUNK_SHIFT), 256, 1024)), dim3(256), 0, stream, b.aux,
                               (b.E + CHUNK - 1) >> CHUNK_SHIFT, b.front[1], b.Wp);
        const u64 *prev = b.front[(level - 1) % 3];           // frontier of level - 1
        u64 *next = b.front[level % 3];                          // receives the frontier of this level
        u64 *idle = b.front[(level + 1) % 3];                    // next level's accumulator: rows spanning chunks cleared now
        const unsigned *lp = b.live[(level - 1) % 3];
        unsigned *ln = b.live[level % 3], *li = b.live[(level + 1) % 3];
        const LevelChoice lc = level_choice(b.Wp, b.live_mode);
#define POPE_LEVEL(WT, MULTI) launch_level<WT, MULTI>(b.E, b.N, b.Wp, b.col, b.erow, b.aux, prev, b.seen, next, idle, b.hop_planes, b.plane_elems, level, b.ctl, lp, ln, li, b.live_words, b.live_mode, b.live_sum, stream)
        if (lc.wt == 1)                        POPE_LEVEL(1, 0);
        else if (lc.wt == 2)                   POPE_LEVEL(2, 0);
        else if (lc.wt == 4 && lc.tiles == 0)  POPE_LEVEL(4, 0);
        else if (lc.wt == 4 && lc.tiles == 1)  POPE_LEVEL(4, 1);
        else if (lc.wt == 4)                   POPE_LEVEL(4, 2);
#if POPE_WT8
        else if (lc.tiles == 0)                POPE_LEVEL(8, 0);
        else if (lc.tiles == 1)                POPE_LEVEL(8, 1);
#endif
#if POPE_WT8_L2
        else                                   POPE_LEVEL(8, 2);
#endif
#undef POPE_LEVEL
    }
    profile_mark(stream, 0, 1, true);
    if (g_profile.enabled && g_profile.span_only && !g_profile.level.empty())
        g_profile.level.back() = -(level - first);                  // span entries carry minus the number of launches
    return level;
}

// Wait for the stream and read the verdicts.  Returns POPE_OK with *done set, or an error code.
// ticket != 0: the finalise kernel enqueued last writes the report when it STARTS; spin on the pinned ticket word instead
// of waiting for the stream to drain (the expansion keeps running; its output is complete in stream order).
static int bfs_poll(const Bfs &b, int next_level, int *last_active, bool *done, hipStream_t stream, int ticket = 0) {
    if (ticket) {
        // Bounded spin: a kernel that never finishes without faulting leaves hipStreamQuery at NotReady for ever, so the
        // wait is also limited by the wall clock (POPE_POLL_TIMEOUT_S seconds) and then reported, not sat out.
        const auto t_start = std::chrono::steady_clock::now();
        bool seen_ticket = false;
        for (long it = 0; !seen_ticket; ++it) {
            if (__atomic_load_n(&b.slot->report[2], __ATOMIC_ACQUIRE) == ticket) {
                seen_ticket = true;
            } else if ((it & 1023) == 1023) {
                const hipError_t q = hipStreamQuery(stream);           // a fault or a drained stream ends the spin
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
                if (waited > POPE_POLL_TIMEOUT_S) {
                    set_error("geodesic bfs: no verdict from the device after %.0f s (stream still busy): giving up the wait", waited);
                    return POPE_ERR_HIP;
                }
            }
        }
        if (!seen_ticket) {
            POPE_HIP(hipStreamSynchronize(stream));
            POPE_REQUIRE(__atomic_load_n(&b.slot->report[2], __ATOMIC_ACQUIRE) == ticket, "geodesic bfs: the report was not written");
        }
    } else {
        hipLaunchKernelGGL(k_bfs_report, dim3(1), dim3(1), 0, stream, b.ctl, b.aux, b.slot->report_dev);
        POPE_HIP(hipStreamSynchronize(stream));
    }
    POPE_HIP(hipGetLastError());
    *last_active = b.slot->report[0];
    const int flags = b.slot->report[1];
    if (flags & CSR_FLAG_BAD_INDEX) {
        set_error("geodesic bfs: edge_index holds a node id outside [0, %d)", b.N);
        return POPE_ERR_INDEX;
    }
    if (flags & CSR_FLAG_UNSORTED) {
        set_error("geodesic bfs: edge_index is not sorted by source; rebuild the CSR with defer_check = 0");
        return POPE_ERR_UNSORTED;
    }
    *done = *last_active < next_level - 1 || b.E == 0;          // some enqueued level found nothing
    if (!*done && next_level >= b.level_limit) {
        // the last representable level still discovered nodes: deeper levels may exist
        set_error("geodesic bfs: hop count needs more than %d bits", b.capacity);
        return POPE_ERR_HOP_OVERFLOW;
    }
    return POPE_OK;
}

static int hop_bits(int max_hop) {
    int bits = 0;
    while ((1 << bits) <= max_hop) ++bits;
    return bits;
}

#ifdef POPE_STAMP
extern "C" int pope_debug_set_stamp_level(int level) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_level), &level, sizeof(int));
}
extern "C" int pope_debug_read_stamps(unsigned long long *host, int count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)count * sizeof(unsigned long long));
}
#endif

extern "C" void pope_profile_levels(int32_t enable) {
    for (hipEvent_t e : g_profile.ev) (void)hipEventDestroy(e);
    g_profile.ev.clear();
    g_profile.level.clear();
    g_profile.enabled = enable != 0;
    g_profile.span_only = enable == 2;
}

extern "C" int32_t pope_profile_read(int32_t *levels, float *level_ms, int32_t capacity) {
    const int n = (int)g_profile.level.size();
    int written = 0;
    for (int i = 0; i < n && written < capacity; ++i) {
        if ((size_t)(2 * i + 1) >= g_profile.ev.size()) break;
        float a = 0.f;
        if (hipEventSynchronize(g_profile.ev[2 * i + 1]) != hipSuccess) break;
        (void)hipEventElapsedTime(&a, g_profile.ev[2 * i], g_profile.ev[2 * i + 1]);
        levels[written] = g_profile.level[i];
        level_ms[written] = a;
        ++written;
    }
    return written;
}

// The BFS in two halves, so that a caller can put other stream work (an all-gather, the finalise kernel) between the
// enqueue and the host synchronisation: begin = clears + seed + the first LEVEL_BATCH levels, nothing is waited for;
// finish = wait, read the verdict, keep going if the graph is deeper.  Both take the same arguments.
static int bfs_begin_impl(const int32_t *rowptr, const int32_t *col, const int32_t *erow, const int32_t *aux,
                          int64_t N, int64_t E, const int64_t *anchors_host, int32_t K, uint64_t *planes,
                          int32_t plane_capacity, void *scratch, size_t scratch_bytes, int window, hipStream_t stream) {
    Bfs b;
    b.slot = nullptr;
    SlotGuard guard{&b.slot, stream};
    int rc = bfs_setup(b, rowptr, col, erow, aux, N, E, anchors_host, K, planes, plane_capacity, scratch, scratch_bytes);
    if (rc) return rc;
    if ((rc = bfs_enqueue_init(b, anchors_host, stream))) return rc;
    bfs_enqueue_levels(b, 1, 1 + window, stream);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

static int bfs_finish_impl(const int32_t *rowptr, const int32_t *col, const int32_t *erow, const int32_t *aux,
                           int64_t N, int64_t E, const int64_t *anchors_host, int32_t K, uint64_t *planes,
                           int32_t plane_capacity, void *scratch, size_t scratch_bytes, int32_t *max_hop_host,
                           int32_t *n_hop_bits_host, int window, hipStream_t stream) {
    Bfs b;
    b.slot = nullptr;
    SlotGuard guard{&b.slot, stream};
    int rc = bfs_setup(b, rowptr, col, erow, aux, N, E, anchors_host, K, planes, plane_capacity, scratch, scratch_bytes);
    if (rc) return rc;
    int level = 1 + window, last_active = 0;                       // what begin enqueued
    if (level > b.level_limit) level = (int)b.level_limit;
    if (b.E == 0) level = 1;
    bool done = false;
    if ((rc = bfs_poll(b, level, &last_active, &done, stream))) return rc;
    while (!done) {
        level = bfs_enqueue_levels(b, level, level + LEVEL_BATCH, stream);
        if ((rc = bfs_poll(b, level, &last_active, &done, stream))) return rc;
    }
    guard.quiescent = true;                                        // every poll synchronised the stream
    if (max_hop_host) *max_hop_host = last_active;
    if (n_hop_bits_host) *n_hop_bits_host = hop_bits(last_active);
    return POPE_OK;
}

extern "C" int pope_geodesic_bfs_begin(const int32_t *rowptr, const int32_t *col, const int32_t *erow, const int32_t *aux,
                                       int64_t N, int64_t E, const int64_t *anchors_host, int32_t K, uint64_t *planes,
                                       int32_t plane_capacity, void *scratch, size_t scratch_bytes, void *stream_) {
    clear_error();
    return bfs_begin_impl(rowptr, col, erow, aux, N, E, anchors_host, K, planes, plane_capacity, scratch, scratch_bytes, LEVEL_BATCH,
                          (hipStream_t)stream_);
}

extern "C" int pope_geodesic_bfs_finish(const int32_t *rowptr, const int32_t *col, const int32_t *erow, const int32_t *aux,
                                        int64_t N, int64_t E, const int64_t *anchors_host, int32_t K, uint64_t *planes,
                                        int32_t plane_capacity, void *scratch, size_t scratch_bytes, int32_t *max_hop_host,
                                        int32_t *n_hop_bits_host, void *stream_) {
    clear_error();
    return bfs_finish_impl(rowptr, col, erow, aux, N, E, anchors_host, K, planes, plane_capacity, scratch, scratch_bytes, max_hop_host,
                           n_hop_bits_host, LEVEL_BATCH, (hipStream_t)stream_);
}

extern "C" int pope_geodesic_bfs(const int32_t *rowptr, const int32_t *col, const int32_t *erow, const int32_t *aux,
                                 int64_t N, int64_t E, const int64_t *anchors_host, int32_t K, uint64_t *planes,
                                 int32_t plane_capacity, void *scratch, size_t scratch_bytes, int32_t *max_hop_host,
                                 int32_t *n_hop_bits_host, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    const int window = speculative_window(N, E, K);        // both halves in one call: the run of levels can follow the previous depth
    int rc = bfs_begin_impl(rowptr, col, erow, aux, N, E, anchors_host, K, planes, plane_capacity, scratch, scratch_bytes, window, stream);
    if (rc) return rc;
    int32_t max_hop = 0;
    rc = bfs_finish_impl(rowptr, col, erow, aux, N, E, anchors_host, K, planes, plane_capacity, scratch, scratch_bytes, &max_hop,
                         n_hop_bits_host, window, stream);
    if (rc) return rc;
    remember_depth(N, E, K, max_hop, window < LEVEL_BATCH && max_hop >= window);
    if (max_hop_host) *max_hop_host = max_hop;
    return POPE_OK;
}

// Which finalise kernel a shape gets (bench.py labels its roofline entry with the same choice: pope_finalize_kernel_name).
enum FinKernel { FIN_GENERIC, FIN_FAST, FIN_PIPE, FIN_WIDE, FIN_LUT };
struct FinChoice { FinKernel kernel; int xp, ep; };

static FinChoice finalize_choice(int64_t N, int32_t K, bool has_x, int32_t F, int n_shards, bool vec, bool four_bits) {
    FinChoice c{FIN_GENERIC, 0, 0};
    if (!(vec && (g_finalize_variant > 0 || n_shards > 1) && four_bits)) return c;
    c.kernel = FIN_FAST;
    const int xp = !has_x ? 0 : (F <= 256 ? 1 : F <= 512 ? 2 : F <= 1024 ? 4 : -1);
    const int64_t ne = (int64_t)(K / 4) * n_shards;
    const int ep = ne <= 64 ? 1 : ne <= 128 ? 2 : 4;             // wider rows: segments of 256 pieces, one work item each
    const int64_t items = N * ((ne + 64 * ep - 1) / (64 * ep));
    const int64_t witems = N * ((ne / 16 + 15) / 16);
    c.xp = xp; c.ep = ep;
    if (g_finalize_variant != 1 || xp < 0) return c;
    const auto pow2 = [](int64_t x) { return x > 0 && (x & (x - 1)) == 0; };
    if (ne > 64 && (K & 63) == 0 && pow2(K / 64) && pow2(n_shards) && N * (ne / 8) < INT32_MAX && g_finalize_lut > (has_x ? 1 : 0))
        c.kernel = FIN_LUT;                                                                      // wide rows: a half-word per lane through the LDS tables
    else if (ne > 64 && (K & 63) == 0 && witems + 32768 * 4 < INT32_MAX) c.kernel = FIN_WIDE;      // wide rows: one load per plane half-word, shuffles to the lanes
    else if (items + 32768 * 4 < INT32_MAX) c.kernel = FIN_PIPE;
    return c;
}

static int finalize_launch(const FinChoice &ch, const u64 *planes, size_t plane_elems, int n_hop_bits, const int *max_hop_dev, int64_t N, int32_t K, int Wp,
                           const float *x, int32_t F, float *out, int64_t out_cols, int32_t c0, hipStream_t stream, int n_shards, size_t shard_elems,
                           const int *aux, int *report, int ticket);

static int finalize_enqueue(const u64 *planes, int n_hop_bits, const int *max_hop_dev, int64_t N, int32_t K,
                            const float *x, int32_t F, float *out, int64_t out_cols, int32_t c0, hipStream_t stream,
                            int n_shards = 1, size_t shard_elems = 0, const int *aux = nullptr, int *report = nullptr,
                            int ticket = 0) {
    const int Wp = words_for(K);
    const size_t plane_elems = (size_t)N * Wp;
    const bool vec = F % 4 == 0 && K % 4 == 0 && c0 % 4 == 0 && out_cols % 4 == 0 && aligned16(out) && (!x || aligned16(x));
    // The device-side depth (max_hop_dev) is only used by pope_geodesic_run, whose speculative window stops at
    // LEVEL_BATCH = 12 levels: at most 4 hop bits.  With a host-side count the fast paths need n_hop_bits <= 4.
    const bool four_bits = max_hop_dev || n_hop_bits <= 4;
    if (n_shards > 1 && !(vec && four_bits)) {                // generic kernel: one launch per shard
        for (int g = 0; g < n_shards; ++g) {
            int rc = finalize_enqueue(planes + (size_t)g * shard_elems, n_hop_bits, max_hop_dev, N, K, g == 0 ? x : nullptr, F, out,
                                      out_cols, c0 + g * K, stream);
            if (rc) return rc;
        }
        return POPE_OK;
    }
    const FinChoice ch = finalize_choice(N, K, x != nullptr, F, n_shards, vec, four_bits);
    if (ch.kernel == FIN_LUT && x && !SideCopy::eligible(x, F, out, out_cols, N)) {     // no separate feature copy for this shape: the shuffle kernel copies and expands
        FinChoice alt = ch;
        alt.kernel = FIN_WIDE;
        return finalize_launch(alt, planes, plane_elems, n_hop_bits, max_hop_dev, N, K, Wp, x, F, out, out_cols, c0, stream, n_shards, shard_elems, aux, report, ticket);
    }
    return finalize_launch(ch, planes, plane_elems, n_hop_bits, max_hop_dev, N, K, Wp, x, F, out, out_cols, c0, stream, n_shards, shard_elems, aux, report, ticket);
}

static int finalize_launch(const FinChoice &ch, const u64 *planes, size_t plane_elems, int n_hop_bits, const int *max_hop_dev, int64_t N, int32_t K, int Wp,
                           const float *x, int32_t F, float *out, int64_t out_cols, int32_t c0, hipStream_t stream, int n_shards, size_t shard_elems,
                           const int *aux, int *report, int ticket) {
    const bool vec = ch.kernel != FIN_GENERIC || (F % 4 == 0 && K % 4 == 0 && c0 % 4 == 0 && out_cols % 4 == 0 && aligned16(out) && (!x || aligned16(x)));
    dim3 grid(capped_grid((size_t)N * 64, 256)), block(256);
    const int64_t ne = (int64_t)(K / 4) * n_shards;
    if (ch.kernel == FIN_LUT) {
        // the feature columns first, by the copy kernel (5.9 TB/s alone); the verdict travels with the column kernel behind it
        if (x) { int rc = enqueue_copy_features(x, F, out, out_cols, N, stream); if (rc) return rc; }
        int hpr_shift = 0, wps_shift = 0;
        while ((1 << wps_shift) < K / 64) ++wps_shift;
        while ((1ll << hpr_shift) < (int64_t)(K / 64) * n_shards * 2) ++hpr_shift;
        // several shards with rows shorter than a batch: a batch per (shard, block of rows) -- see the kernel
        const int rows_shift = (n_shards > 1 && wps_shift + 1 < 6 && g_finalize_shard_batches) ? 6 - (wps_shift + 1) : -1;
        const int64_t batches = rows_shift < 0 ? ((N << hpr_shift) + 63) >> 6 : ((N + (1 << rows_shift) - 1) >> rows_shift) * n_shards;
        static LdsOptIn opt_in;
        if (!opt_in.done()) {
            POPE_HIP(hipFuncSetAttribute((const void *)k_finalize_lut, hipFuncAttributeMaxDynamicSharedMemorySize, FIN_LUT_LDS));
            opt_in.mark();
        }
        // blocks live for a few batches each: the tables cost a block ~1 us to build
        const unsigned blocks = g_finalize_blocks_set ? (unsigned)g_finalize_blocks : (unsigned)std::min<int64_t>(std::max<int64_t>((batches + 15) / 16, 1), 4096);
        hipLaunchKernelGGL(k_finalize_lut, dim3(blocks), block, FIN_LUT_LDS, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, Wp, out,
                           (long long)out_cols, F + c0, hpr_shift, wps_shift, rows_shift, shard_elems, aux, report, ticket);
    } else if (ch.kernel == FIN_WIDE) {
        const int64_t witems = N * ((ne / 16 + 15) / 16);
        dim3 wgrid(g_finalize_blocks_set ? g_finalize_blocks : (unsigned)std::min<int64_t>(std::max<int64_t>((witems + 3) / 4, 256), 32768));
#define POPE_FIN_WIDE(XP)                                                                                                                 \
    hipLaunchKernelGGL((k_finalize_wide<XP>), wgrid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, \
                       (long long)out_cols, c0, n_shards, shard_elems, aux, report, ticket)
        if (ch.xp == 0) POPE_FIN_WIDE(0); else if (ch.xp == 1) POPE_FIN_WIDE(1); else if (ch.xp == 2) POPE_FIN_WIDE(2); else POPE_FIN_WIDE(4);
#undef POPE_FIN_WIDE
    } else if (ch.kernel == FIN_PIPE) {
        // one row per wave by default (grid sweep, profiles/r04_finalize_pipe*.txt: 2 048 blocks 0.2479 ms, 4 096 0.2456, 8 192
        // 0.2416, 16 384 0.2394, one row per wave 0.2395, 32 768 0.2400): short-lived waves in row order
        const int64_t items = N * ((ne + 64 * ch.ep - 1) / (64 * ch.ep));
        dim3 pgrid(g_finalize_blocks_set ? g_finalize_blocks : (unsigned)std::min<int64_t>(std::max<int64_t>((items + 3) / 4, 256), 32768));
        const int xp = ch.xp, ep = ch.ep;
#define POPE_FIN_PIPE(XP, EP)                                                                                                             \
    hipLaunchKernelGGL((k_finalize_pipe<XP, EP>), pgrid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, \
                       (long long)out_cols, c0, n_shards, shard_elems, aux, report, ticket)
        if (xp == 0)      { if (ep == 1) POPE_FIN_PIPE(0, 1); else if (ep == 2) POPE_FIN_PIPE(0, 2); else POPE_FIN_PIPE(0, 4); }
        else if (xp == 1) { if (ep == 1) POPE_FIN_PIPE(1, 1); else if (ep == 2) POPE_FIN_PIPE(1, 2); else POPE_FIN_PIPE(1, 4); }
        else if (xp == 2) { if (ep == 1) POPE_FIN_PIPE(2, 1); else if (ep == 2) POPE_FIN_PIPE(2, 2); else POPE_FIN_PIPE(2, 4); }
        else              { if (ep == 1) POPE_FIN_PIPE(4, 1); else if (ep == 2) POPE_FIN_PIPE(4, 2); else POPE_FIN_PIPE(4, 4); }
#undef POPE_FIN_PIPE
    } else if (ch.kernel == FIN_FAST) {
        hipLaunchKernelGGL(k_finalize_fast, dim3(g_finalize_blocks), block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out,
                           (long long)out_cols, c0, n_shards, shard_elems, aux, report, ticket);
    } else if (vec) {
        hipLaunchKernelGGL(k_finalize<true>, grid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, (long long)out_cols, c0, aux, report, ticket);
    } else {
        hipLaunchKernelGGL(k_finalize<false>, grid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, (long long)out_cols, c0, aux, report, ticket);
    }
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

// The name of the level kernel a BFS over N nodes from K anchors launches (what a profile will show).
extern "C" int pope_level_kernel_name(int64_t N, int32_t K, char *name, size_t cap) {
    clear_error();
    POPE_REQUIRE(name && cap > 0 && N > 0 && K > 0, "pope_level_kernel_name: bad argument");
    const LevelChoice lc = level_choice(words_for(K), live_mode_for(N));
    snprintf(name, cap, "k_bfs_level<%d, %d, %d>", lc.wt, lc.live, lc.tiles);
    return POPE_OK;
}

// The name of the finalise kernel pope_geodesic_run / pope_geodesic_finalize(_shards) launches for a shape (what a profile will show).
extern "C" int pope_finalize_kernel_name(int64_t N, int32_t K, int32_t F, int32_t has_x, int32_t n_shards, char *name, size_t cap) {
    clear_error();
    POPE_REQUIRE(name && cap > 0 && N > 0 && K > 0 && F >= 0 && n_shards >= 1, "pope_finalize_kernel_name: bad argument");
    const bool vec = F % 4 == 0 && K % 4 == 0;                   // aligned bases and row pitches assumed (torch allocations)
    const FinChoice c = finalize_choice(N, K, has_x != 0, F, n_shards, vec, true);
    switch (c.kernel) {
    case FIN_LUT:  snprintf(name, cap, "k_finalize_lut"); break;
    case FIN_WIDE: snprintf(name, cap, "k_finalize_wide<%d>", c.xp); break;
    case FIN_PIPE: snprintf(name, cap, "k_finalize_pipe<%d, %d>", c.xp, c.ep); break;
    case FIN_FAST: snprintf(name, cap, "k_finalize_fast"); break;
    default:       snprintf(name, cap, vec ? "k_finalize<true>" : "k_finalize<false>"); break;
    }
    return POPE_OK;
}

extern "C" int pope_geodesic_finalize(const uint64_t *planes, int32_t n_hop_bits, int64_t N, int32_t K,
                                      const float *x, int32_t F, float *out, int64_t out_cols, int32_t c0,
                                      void *stream_) {
    clear_error();
    POPE_REQUIRE(planes && out, "pope_geodesic_finalize: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && K > 0 && F >= 0 && c0 >= 0 && n_hop_bits >= 0 && n_hop_bits <= 31,
                 "pope_geodesic_finalize: bad size");
    POPE_REQUIRE(out_cols >= (int64_t)F + c0 + K, "pope_geodesic_finalize: out_cols %lld < F + c0 + K = %lld",
                 (long long)out_cols, (long long)F + c0 + K);
    return finalize_enqueue((const u64 *)planes, n_hop_bits, nullptr, N, K, x, F, out, out_cols, c0, (hipStream_t)stream_);
}

// ---- the whole geodesic hot path in one call: edge_index -> [N, out_cols] features, one host synchronisation ----
struct RunLayout {
    size_t rowptr, col, erow, aux, csr_scratch, planes, bfs_scratch, total;
};

static RunLayout run_layout(int64_t N, int64_t E, int32_t K, int32_t capacity) {
    RunLayout L;
    size_t o = 0;
    L.rowptr = o;      o += align_up((size_t)(N + 1) * sizeof(int), 256);
    L.col = o;         o += align_up((size_t)(E > 0 ? E : 1) * sizeof(int), 256);
    L.erow = o;        o += align_up((size_t)(E > 0 ? E : 1) * sizeof(int), 256);
    L.aux = o;         o += align_up(pope_csr_aux_elems(E) * sizeof(int), 256);
    L.csr_scratch = o; o += align_up(pope_csr_scratch_bytes(N, E), 256);
    L.planes = o;      o += align_up((size_t)(capacity + 1) * pope_plane_bytes(N, K), 256);
    L.bfs_scratch = o; o += align_up(pope_bfs_scratch_bytes(N, E, K), 256);
    L.total = o;
    return L;
}

extern "C" size_t pope_geodesic_run_workspace_bytes(int64_t N, int64_t E, int32_t K, int32_t plane_capacity) {
    if (N < 0 || E < 0 || K <= 0 || plane_capacity < 1 || plane_capacity > 31) return 0;
    return run_layout(N, E, K, plane_capacity).total;
}

extern "C" uint64_t *pope_geodesic_run_planes(void *workspace, int64_t N, int64_t E, int32_t K, int32_t plane_capacity) {
    if (!workspace || N < 0 || E < 0 || K <= 0 || plane_capacity < 1 || plane_capacity > 31) return nullptr;
    return (uint64_t *)((char *)workspace + run_layout(N, E, K, plane_capacity).planes);
}

extern "C" int pope_geodesic_run(const int64_t *edge_index, int64_t E, int64_t N, const int64_t *anchors_host, int32_t K,
                                 const float *x, int32_t F, float *out, int64_t out_cols, int32_t plane_capacity,
                                 void *workspace, size_t workspace_bytes, int32_t *max_hop_host,
                                 int32_t *n_hop_bits_host, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(N > 0 && N < INT32_MAX && E >= 0 && E < INT32_MAX && K > 0 && F >= 0, "pope_geodesic_run: bad size");
    POPE_REQUIRE(plane_capacity >= 1 && plane_capacity <= 31, "pope_geodesic_run: need 1 <= plane_capacity <= 31");
    POPE_REQUIRE(workspace && (edge_index || E == 0) && anchors_host, "pope_geodesic_run: null pointer");
    POPE_REQUIRE(!out || out_cols >= (int64_t)F + K, "pope_geodesic_run: out_cols %lld < F + K", (long long)out_cols);
    const RunLayout L = run_layout(N, E, K, plane_capacity);
    if (workspace_bytes < L.total) {
        set_error("pope_geodesic_run: workspace %zu < %zu bytes", workspace_bytes, L.total);
        return POPE_ERR_WORKSPACE;
    }
    char *ws = (char *)workspace;
    int *rowptr = (int *)(ws + L.rowptr), *col = (int *)(ws + L.col), *erow = (int *)(ws + L.erow), *aux = (int *)(ws + L.aux);
    u64 *planes = (u64 *)(ws + L.planes);
    // speculative: sorted-CSR fast path, the first LEVEL_BATCH levels and the finalise kernel are all enqueued
    // before the host looks at anything; the finalise kernel reads the depth from the BFS control block.
    int rc;
    // (Measured and rejected, round 2: out[:, :F] = x on a side stream beside the CSR build and the BFS levels instead of inside
    //  the finalise kernel.  The finalise kernel drops from 98 to 26 us, but the dense levels are bound by the same L2 / fabric
    //  the copy streams through: levels 3-4 ran 43 us instead of 19 while it was in flight, and the step stayed at 0.275 ms
    //  with 2, 4 or 8 resident copy blocks per CU, plain or non-temporal stores.  side_copy.h serves the node2vec path only.)
    Bfs b;
    b.slot = nullptr;
    SlotGuard guard{&b.slot, stream};
    if ((rc = bfs_setup(b, rowptr, col, erow, aux, N, E, anchors_host, K, (uint64_t *)planes, plane_capacity,
                        ws + L.bfs_scratch, L.total - L.bfs_scratch))) return rc;
    const int window = speculative_window(N, E, K);
    memcpy(b.slot->anchors, anchors_host, (size_t)K * sizeof(long long));     // this call's pinned, device-mapped slot: read in place
    if (g_prepare_merge && K <= PREP_MAX_ANCHORS && E > 0) {
        // one launch: clear + seed role beside the speculative CSR role (k_prepare)
        // The tag of this call's CSR status word: 29 bits whose TOP bit is always set and whose lower 28 are a scrambled call count.
        // The word then reads as a NEGATIVE int32, and what a workspace holds from earlier use at that address -- node ids, row
        // offsets, chunk rows of another graph's CSR, or -1 (tag 0x1fffffff, never handed out) -- cannot carry it.  (Round 4 counted
        // 1, 2, 3, ...: the word (7 << 3) | 1 = 57 is also a node id, and a workspace reused across graphs of different sizes, or
        // fresh from an allocator that had held index arrays, reported "node id outside [0, N)" for a clean edge list --
        // tests/test_level_kernels_soak_gpu.py hit it on its fourth call; random and 0xFF fills, which round 4 soaked, could not.)
        static std::atomic<unsigned> epochs{0};
        unsigned epoch;
        do epoch = 0x10000000u | ((++epochs * 0x9E3779B1u) & 0x0fffffffu); while (epoch == 0x1fffffffu);
        PrepSeeds seeds;
        for (int j = 0; j < K; ++j) seeds.a[j] = (int)anchors_host[j];          // (validated by bfs_setup)
        const int eager = b.capacity < EAGER_PLANES ? b.capacity : EAGER_PLANES;
        const size_t zwords = (size_t)(1 + eager) * b.plane_elems;          // odd: one 8-byte word behind the last 16-byte unit (zb_tail)
        const size_t na = (b.front_off + 3 * align_up(b.plane_bytes, 256) + 3 * live_bytes(b.N)) / 16, nb = zwords / 2;
        u64 *zb_tail = (zwords & 1) ? b.seen + zwords - 1 : nullptr;
        const long long *src = (const long long *)edge_index, *dst = src + E;
        // (knob values above 1, for A/B: low 16 bits = the clear role's block count, high 16 bits = a cap on the CSR role's)
        const int zero_blocks = (g_prepare_merge & 0xffff) > 1 ? (g_prepare_merge & 0xffff) : 1024;
        const bool pairs = (E & 1) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(col) | reinterpret_cast<uintptr_t>(erow)) & 15u) == 0;
        unsigned csr_blocks = pairs ? capped_grid(E / 2, 256) : capped_grid(E, 256);
        if ((g_prepare_merge >> 16) > 0) csr_blocks = std::min<unsigned>(csr_blocks, (unsigned)(g_prepare_merge >> 16));
        if (pairs)
            hipLaunchKernelGGL(k_prepare<true>, dim3(zero_blocks + csr_blocks), dim3(256), 0, stream, src, dst, (int)E, (int)N, rowptr, col, erow, aux,
                               (uint4 *)b.base, na, (uint4 *)b.seen, nb, zb_tail, zero_blocks, epoch, seeds, K, b.Wp, b.seen, b.front[0], b.live[0]);
        else
            hipLaunchKernelGGL(k_prepare<false>, dim3(zero_blocks + csr_blocks), dim3(256), 0, stream, src, dst, (int)E, (int)N, rowptr, col, erow, aux,
                               (uint4 *)b.base, na, (uint4 *)b.seen, nb, zb_tail, zero_blocks, epoch, seeds, K, b.Wp, b.seen, b.front[0], b.live[0]);
        POPE_HIP(hipGetLastError());
        b.frontiers_cleared = true;
    } else {
        bfs_enqueue_clear(b, aux, stream);                    // BFS state and the CSR status header in one launch
        SeedArgs seed;
        seed.anchors = b.slot->anchors_dev; seed.K = K; seed.Wp = b.Wp; seed.seen = b.seen; seed.front = b.front[0]; seed.live = b.live[0];
        rc = csr_build(edge_index, E, N, rowptr, col, erow, aux, ws + L.csr_scratch, L.planes - L.csr_scratch, 2, seed, stream);
        if (rc) return rc;
    }
    int level = bfs_enqueue_levels(b, 1, 1 + window, stream);
    // The finalise kernel writes the verdict into the pinned report when it starts: no report launch, and the host
    // returns as soon as the BFS is known to be complete -- `out` is finished in stream order.
    int ticket = 0;
    if (out) {
        ticket = b.slot->ticket = b.slot->ticket == INT32_MAX ? 1 : b.slot->ticket + 1;
        if ((rc = finalize_enqueue(planes, 0, &b.ctl->last_active, N, K, x, F, out, out_cols, 0, stream, 1, 0, aux,
                                   b.slot->report_dev, ticket))) return rc;
    }
    int last_active = 0;
    bool done = false;
    rc = bfs_poll(b, level, &last_active, &done, stream, ticket);
    if (rc == POPE_ERR_UNSORTED) {                        // general path: counting sort, then start over
        clear_error();
        if ((rc = csr_fallback((const long long *)edge_index, (const long long *)edge_index + E, (int)E, (int)N, rowptr, col,
                               erow, aux, ws + L.csr_scratch, stream))) return rc;
        b.frontiers_cleared = false;
        if ((rc = bfs_enqueue_init(b, anchors_host, stream))) return rc;
        level = 1;
        done = false;
    } else if (rc) {
        return rc;
    } else if (done) {
        guard.quiescent = true;                           // the seed read the anchors long ago and the verdict has arrived: nobody on the device uses the slot any more
        remember_depth(N, E, K, last_active);
        if (max_hop_host) *max_hop_host = last_active;
        if (n_hop_bits_host) *n_hop_bits_host = hop_bits(last_active);
        return POPE_OK;
    }
    while (!done) {                                        // deep or re-sorted graph: keep going, then finalise again
        level = bfs_enqueue_levels(b, level, level + LEVEL_BATCH, stream);
        if ((rc = bfs_poll(b, level, &last_active, &done, stream))) return rc;
    }
    if (out && (rc = finalize_enqueue(planes, hop_bits(last_active), nullptr, N, K, x, F, out, out_cols, 0, stream))) return rc;
    guard.quiescent = true;                               // every poll of this path synchronised the stream; the late finalise kernel does not touch the slot
    remember_depth(N, E, K, last_active, window < LEVEL_BATCH);
    if (max_hop_host) *max_hop_host = last_active;
    if (n_hop_bits_host) *n_hop_bits_host = hop_bits(last_active);
    return POPE_OK;
}

extern "C" int pope_geodesic_finalize_shards(const uint64_t *planes, int32_t n_shards, int64_t shard_stride_words,
                                             int32_t n_hop_bits, int64_t N, int32_t K_shard, const float *x, int32_t F,
                                             float *out, int64_t out_cols, void *stream_) {
    clear_error();
    POPE_REQUIRE(planes && out, "pope_geodesic_finalize_shards: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && K_shard > 0 && F >= 0 && n_shards >= 1 && n_hop_bits >= 0 && n_hop_bits <= 31,
                 "pope_geodesic_finalize_shards: bad size");
    POPE_REQUIRE(shard_stride_words >= (int64_t)(1 + n_hop_bits) * N * words_for(K_shard), "pope_geodesic_finalize_shards: shard stride too small");
    POPE_REQUIRE(out_cols >= (int64_t)F + (int64_t)n_shards * K_shard, "pope_geodesic_finalize_shards: out_cols too small");
    return finalize_enqueue((const u64 *)planes, n_hop_bits, nullptr, N, K_shard, x, F, out, out_cols, 0, (hipStream_t)stream_,
                            n_shards, (size_t)shard_stride_words);
}

extern "C" int pope_geodesic_hops(const uint64_t *planes, int32_t n_hop_bits, int64_t N, int32_t K, int32_t *hops,
                                  void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(planes && hops, "pope_geodesic_hops: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && K > 0 && n_hop_bits >= 0 && n_hop_bits <= 31, "pope_geodesic_hops: bad size");
    const int Wp = words_for(K);
    hipLaunchKernelGGL(k_hops, dim3(capped_grid((size_t)N * K, 256)), dim3(256), 0, stream, (const u64 *)planes,
                       (size_t)N * Wp, n_hop_bits, (int)N, K, Wp, hops);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int pope_geodesic_hop_codes(const uint64_t *planes, int32_t n_hop_bits, int32_t max_hop, int64_t N, int32_t K, uint8_t *codes,
                                       int64_t codes_pitch_bytes, float *lut, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(planes && codes && lut, "pope_geodesic_hop_codes: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && K > 0 && codes_pitch_bytes >= K, "pope_geodesic_hop_codes: bad size");
    POPE_REQUIRE(n_hop_bits >= 0 && n_hop_bits <= 8 && max_hop >= 0 && max_hop <= 254 && max_hop < (1 << n_hop_bits),
                 "pope_geodesic_hop_codes: hop counts above 254 do not fit the byte code (use pope_geodesic_finalize)");
    const int Wp = words_for(K);
    const size_t waves = ((size_t)N + 7) / 8;                          // about eight rows per wave
    const int grid = (int)std::min<size_t>(std::max<size_t>((waves + 3) / 4, 1), 4096);
    hipLaunchKernelGGL(k_hop_codes, dim3(grid), dim3(256), 0, stream, (const u64 *)planes, (size_t)N * Wp, n_hop_bits, (int)N, K, Wp, codes,
                       (long long)codes_pitch_bytes, lut);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

constexpr int STATS_PARTS = 256;

extern "C" size_t pope_column_stats_scratch_bytes(int32_t K) { return K <= 0 ? 0 : 2 * (size_t)STATS_PARTS * K * sizeof(long long); }

extern "C" int pope_geodesic_column_stats(const uint64_t *planes, int32_t n_hop_bits, int64_t N, int32_t K, int64_t *hop_sum,
                                          int64_t *reach, void *scratch, size_t scratch_bytes, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(planes && hop_sum && reach && scratch, "pope_geodesic_column_stats: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && K > 0 && n_hop_bits >= 0 && n_hop_bits <= 31, "pope_geodesic_column_stats: bad size");
    if (scratch_bytes < pope_column_stats_scratch_bytes(K)) {
        set_error("pope_geodesic_column_stats: scratch %zu < %zu bytes", scratch_bytes, pope_column_stats_scratch_bytes(K));
        return POPE_ERR_WORKSPACE;
    }
    const int Wp = words_for(K);
    long long *ps = (long long *)scratch, *pc = ps + (size_t)STATS_PARTS * K;
    hipLaunchKernelGGL(k_column_stats_partial, dim3(STATS_PARTS, (K + 255) / 256), dim3(256), 0, stream, (const u64 *)planes,
                       (size_t)N * Wp, n_hop_bits, (int)N, K, Wp, ps, pc);
    hipLaunchKernelGGL(k_column_stats_final, dim3((K + 255) / 256), dim3(256), 0, stream, ps, pc, STATS_PARTS, K,
                       (long long *)hop_sum, (long long *)reach);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int pope_concat(const float *x, int64_t N, int32_t F, float *out, int64_t out_cols, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(x && out, "pope_concat: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && F > 0 && out_cols >= F, "pope_concat: bad size");
    // Round 4: side_copy.hip's kernel (16 pieces of 16 bytes in flight per lane: 5.9 TB/s) where its shapes allow -- k_concat's loop
    // compiles to load - wait - store per piece (4.8 TB/s); this is the feature copy the multi-GPU path runs underneath its all-gather.
    if (SideCopy::eligible(x, F, out, out_cols, N)) return enqueue_copy_features(x, F, out, out_cols, N, stream);
    const bool vec = F % 4 == 0 && out_cols % 4 == 0 && aligned16(out) && aligned16(x);
    hipLaunchKernelGGL(k_concat, dim3(256 * 8), dim3(256), 0, stream, x, (int)N, F, out,
                       (long long)out_cols, vec);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}
