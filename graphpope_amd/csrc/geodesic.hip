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

namespace pope {

// ------------------------------------------------------------------------------------------------
// CSR build
// ------------------------------------------------------------------------------------------------
enum { CSR_FLAG_BAD_INDEX = 1, CSR_FLAG_UNSORTED = 2, BFS_FLAG_TAIL_FAILED = 4 };
constexpr int TAIL_GROUPS = 16;                // k_tail_finalize: groups of the two-stage barrier among its BFS blocks
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
    int tail_done;       // k_tail_finalize: the ticket of the call once its BFS blocks are through (agent-scope release)
    unsigned tail_top;   // k_tail_finalize: second stage of the barrier of its BFS blocks (groups that have arrived)
    int tail_failed;     // k_tail_finalize: a bounded wait ran out (never expected; the host turns it into an error)
    unsigned flag_epoch; // the tag the CSR status word must carry to count (csr_flags); 0 = the zeroed word of the separate launches
    int pad[27];
    unsigned tail_group[TAIL_GROUPS * 32];      // first stage: one counter per group of BFS blocks, 128 bytes apart
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
                                                 uint4 *za, size_t na, uint4 *zb, size_t nb, int zero_blocks, unsigned epoch,
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
__global__ __launch_bounds__(256) void k_zero(uint4 *a, size_t na, uint4 *b, size_t nb, uint4 *c, size_t nc) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) a[i] = z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) b[i] = z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += stride) c[i] = z;
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

// COPY ROLE of a level launch (round 4).  out[:, :F] = x (utils.py:129-135) is two thirds of the finalise kernel's traffic and
// depends on nothing the BFS computes, while the level launches are bound by L2 line fills and latency and leave the HBM
// interface idle.  Every level launch of pope_geodesic_run therefore carries, BEHIND its expand and housekeeping blocks, a
// bounded number of short-lived blocks that copy a fixed slice of x's rows -- one batch of 16 pieces of 16 bytes per lane,
// every load issued before the first store (side_copy.hip's shape) -- and the finalise kernel starts its own copy at the
// first row no launch took.  Same launch, same stream: nothing forks, nothing delays the next level's launch (both side-stream
// forms of this were measured slower, DESIGN.md section 3 lessons 5, 11, 12).  x is dense [N, F]: piece p of the flat sequence of
// 16-byte pieces lies at x + 16 p and goes to row p / F4, piece p % F4 of out.
struct LevelCopy {
    const float *x = nullptr;
    float *out = nullptr;
    unsigned F4 = 0, opitch4 = 0;                 // row length of x and row pitch of out, in 16-byte pieces
    unsigned piece_begin = 0, piece_end = 0;      // flat pieces of x this launch copies
    int first_block = 0, blocks = 0;              // blockIdx.x of the role's first block (set by launch_level), number of its blocks
};
constexpr int LEVEL_COPY_PIECES = 16;             // per lane and batch
constexpr unsigned LEVEL_COPY_BLOCK_PIECES = 256u * LEVEL_COPY_PIECES;

__device__ __forceinline__ void level_copy_role(const LevelCopy &cp) {
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const unsigned base = cp.piece_begin + (((unsigned)blockIdx.x - (unsigned)cp.first_block) * 4u + w) * (64u * LEVEL_COPY_PIECES) + lane;
    if (base >= cp.piece_end) return;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(cp.x);
    f32x4 *dst = reinterpret_cast<f32x4 *>(cp.out);
    f32x4 v[LEVEL_COPY_PIECES];
#pragma unroll
    for (int j = 0; j < LEVEL_COPY_PIECES; ++j) {
        const unsigned p = base + 64u * j;
        if (p < cp.piece_end) v[j] = __builtin_nontemporal_load(src + p);
    }
    unsigned row = base / cp.F4, q = base - row * cp.F4;
#pragma unroll
    for (int j = 0; j < LEVEL_COPY_PIECES; ++j) {
        const unsigned p = base + 64u * j;
        if (p < cp.piece_end) dst[(size_t)row * cp.opitch4 + q] = v[j];
        q += 64u;
        while (q >= cp.F4) { q -= cp.F4; ++row; }
    }
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

typedef unsigned long long u64x2v __attribute__((ext_vector_type(2)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

// The same with the non-temporal hint (POPE_KNOB_LEVEL_VARIANT experiments: streams that should not evict the frontier).
template <int WT>
__device__ __forceinline__ Words<WT> load_words_nt(const u64 *__restrict__ p) {
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

// Frontier gathers go through L1 like any load: reading them with the non-temporal hint was measured 57 % slower
// (BFS 349 us against 223 us, tools/ab_lib.py) -- the rows of hubs are gathered again and again and L1 serves them.
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
template <int WT>
__device__ __forceinline__ void commit_words(const Words<WT> &fresh, const Words<WT> &seen_old, size_t idx,
                                             u64 *__restrict__ seen, u64 *__restrict__ hop_planes,
                                             size_t plane_elems, int level) {
    Words<WT> s;
#pragma unroll
    for (int i = 0; i < WT; ++i) s.w[i] = seen_old.w[i] | fresh.w[i];
    store_words<WT>(seen + idx, s);
    Words<WT> h[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        h[b] = fresh;
        if ((level >> b) & 1) h[b] = load_words<WT>(hop_planes + (size_t)b * plane_elems + idx);
    }
#pragma unroll
    for (int b = 0; b < 5; ++b)
        if ((level >> b) & 1) {
#pragma unroll
            for (int i = 0; i < WT; ++i) h[b].w[i] |= fresh.w[i];
            store_words<WT>(hop_planes + (size_t)b * plane_elems + idx, h[b]);
        }
    for (int b = 5, l = level >> 5; l; ++b, l >>= 1)              // levels >= 32: rare, one at a time
        if (l & 1) {
            u64 *p = hop_planes + (size_t)b * plane_elems + idx;
            Words<WT> g = load_words<WT>(p);
#pragma unroll
            for (int i = 0; i < WT; ++i) g.w[i] |= fresh.w[i];
            store_words<WT>(p, g);
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
// WT = words per tile (1, 2 or 4); blockIdx.y selects the tile of a node's W words.
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

// Housekeeping share of one level (see k_bfs_level): thread t0 of tstride threads.  (1) clears two levels ahead -- the live
// table (first_tile only) and the accumulator words of the rows that span chunks; (2) commits level - 1 for every node whose
// frontier row is non-zero.
template <int WT, int LIVE>
__device__ __forceinline__ void level_housekeeping(int E, int N, int Wp, const u64 *__restrict__ front, u64 *__restrict__ seen,
                                                   u64 *__restrict__ idle, u64 *__restrict__ hop_planes, size_t plane_elems, int level,
                                                   const int *aux, const unsigned *__restrict__ live, unsigned *__restrict__ live_idle,
                                                   int live_words, int t0, int tstride, int woff, bool first_tile) {
    const int n = (E + CHUNK - 1) >> CHUNK_SHIFT;              // one slot per chunk, -1 = no row continues into it
    const int *mrows = aux + AUX_HEADER;
    if (LIVE && first_tile)
        for (int i = t0; i < live_words; i += tstride) live_idle[i] = 0u;
    Words<WT> zero;
#pragma unroll
    for (int i = 0; i < WT; ++i) zero.w[i] = 0;
    if constexpr (LIVE == 0) {
        // No live table (A/B mode): the commit below reads EVERY frontier row, so a row that no chunk writes (a node
        // without out-edges) must not keep what the buffer held three levels ago: the whole buffer is cleared.
        for (int v = t0; v < N; v += tstride) store_words<WT>(idle + (size_t)v * Wp + woff, zero);
    } else {
        for (int i = t0; i < n; i += tstride) {
            const int mv = mrows[i];
            if (mv >= 0) store_words<WT>(idle + (size_t)mv * Wp + woff, zero);
        }
    }
    if (level > 1) {
        for (int v = t0; v < N; v += tstride) {
            if (LIVE && !((live[v >> 5] >> (v & 31)) & 1u)) continue;              // frontier row all zero: nothing gained
            const size_t idx = (size_t)v * Wp + woff;
            const Words<WT> fresh = load_words<WT>(front + idx);
            if (any_bits<WT>(fresh))
                commit_words<WT>(fresh, load_words<WT>(seen + idx), idx, seen, hop_planes, plane_elems, level - 1);
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

// Expand share of one level (see k_bfs_level): this wave walks chunks wave, wave + nwaves, ... of tile `woff`; (vr, ur) hold the
// first chunk's slots, loaded by the caller before it staged the live table.  Returns whether this lane emitted a non-zero row.
template <int WT, int LIVE>
__device__ __forceinline__ bool level_expand(const int *__restrict__ erow, const int *__restrict__ col, int E, int Wp,
                                             const u64 *__restrict__ front, u64 *__restrict__ seen, u64 *__restrict__ acc,
                                             const unsigned *__restrict__ live, unsigned *__restrict__ live_acc,
                                             const unsigned *live_lds, int variant, int level, int lane, int wave, int nwaves, int nchunks,
                                             int woff, int tiles, int4 vr, int4 ur, unsigned *wave_words) {
    auto load_idx = [&](const int *p) {
        if (variant & 1) {
            const i32x4v t = __builtin_nontemporal_load(reinterpret_cast<const i32x4v *>(p));
            return make_int4(t.x, t.y, t.z, t.w);
        }
        return *reinterpret_cast<const int4 *>(p);
    };
    bool found = false;
    STAMP(0);
    for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
        const int base = chunk * CHUNK + lane * SLOTS;
        int v0 = -1, v1 = -1, v2 = -1, v3 = -1, u0 = 0, u1 = 0, u2 = 0, u3 = 0;
        if (base < E) {                       // arrays are padded to a multiple of 4 entries: the 16-byte load is in bounds
            if (chunk != wave) {
                vr = load_idx(erow + base);
                ur = load_idx(col + base);
            }
            v0 = vr.x; u0 = ur.x;
            if (base + 1 < E) { v1 = vr.y; u1 = ur.y; }
            if (base + 2 < E) { v2 = vr.z; u2 = ur.z; }
            if (base + 3 < E) { v3 = vr.w; u3 = ur.w; }
        }
        auto is_live = [&](int u) {
            if constexpr (LIVE == 0) return true;
            const unsigned w = LIVE == 1 ? live_lds[u >> 5] : live[u >> 5];
            return ((w >> (u & 31)) & 1u) != 0;
        };
        // the four look-ups first, unconditionally (u = 0 for an empty slot), then the tests: as `v >= 0 && is_live(u)` each look-up sat
        // behind a branch and was waited for on its own
        const bool q0 = is_live(u0), q1 = is_live(u1), q2 = is_live(u2), q3 = is_live(u3);
        const bool g0 = (v0 >= 0) & q0, g1 = (v1 >= 0) & q1, g2 = (v2 >= 0) & q2, g3 = (v3 >= 0) & q3;
        const int vc = __builtin_amdgcn_readlane(v0, 0);                               // row of the chunk's first slot
        const int vl = __builtin_amdgcn_readlane(v3, 63);                              // row of its last slot (-1: short chunk)
        STAMP(1);
        // the rows of the slots just outside the chunk: does its first row begin earlier, does its last row run on?
        const int2 er = chunk_edge_rows(erow, __builtin_amdgcn_readfirstlane(chunk), E);     // wave-uniform: scalar loads
        const bool head_multi = chunk > 0 && er.x == vc;                               // first row began in an earlier chunk
        const bool tail_multi = vl >= 0 && (chunk + 1) * CHUNK < E && er.y == vl;     // last row runs on
        // slots of a row that spans chunks (only the chunk's first and last row can)
        const bool x0 = (head_multi && v0 == vc) || (tail_multi && v0 == vl);
        const bool x3 = (head_multi && v3 == vc) || (tail_multi && v3 == vl);

        Words<WT> c0, c1, c2, c3;
#pragma unroll
        for (int i = 0; i < WT; ++i) c0.w[i] = c1.w[i] = c2.w[i] = c3.w[i] = 0;
        const bool work = __any(g0 || g1 || g2 || g3);                 // else: no live neighbour behind these 256 slots
        if (!work && LIVE && tiles == 1) continue;                     // nothing to gather, nothing to store (all-zero rows are not written), nothing to mark
        if (g0) c0 = gather_words<WT>(front + (size_t)u0 * Wp + woff);
        if (g1) c1 = gather_words<WT>(front + (size_t)u1 * Wp + woff);
        if (g2) c2 = gather_words<WT>(front + (size_t)u2 * Wp + woff);
        if (g3) c3 = gather_words<WT>(front + (size_t)u3 * Wp + woff);
        // mask of row v: what reached it before this level = seen[v] | front[v].  front[v] (level - 1's gain) is committed to
        // seen by the housekeeping blocks of THIS launch: either order gives the same mask.  Rows whose live bit is clear
        // have an all-zero (possibly never written) frontier row: not loaded.
        // Round 4: the first and the last row's loads (reachability + frontier, predicated, no use in between) go out together with
        // the gathers; as a chain of calls each frontier load sat inside an `if (live)` whose merge point waited for it, and the ISA
        // showed up to four serial round trips behind the gathers.  (All four rows' loads at once: 142 registers, three waves per
        // SIMD instead of four, every level 3-6 us SLOWER -- profiles/r04_level_times_batched_masks.txt.)
        // (Round 4, after the finalise kernel's lesson: this chain compiles to up to four serial load - wait rounds behind the gathers.
        //  Requesting the first and last row's masks with the gathers and the interior rows' in a second batch was built and
        //  A/B-ed as separate library builds, tools/ab_lib.py: BFS 205-212 us against 193-197 us for this chain; all four rows at
        //  once needs 142 registers, three waves per SIMD, every level 3-6 us slower.  Requesting the rows of the slots next to
        //  the chunk with the index loads made no measurable difference either.  profiles/r04_level_ab_libs.txt)
        {
            auto row_mask = [&](int v) {
                Words<WT> m = (variant & 4) ? load_words_nt<WT>(seen + (size_t)v * Wp + woff) : load_words<WT>(seen + (size_t)v * Wp + woff);
                if (is_live(v)) {
                    const Words<WT> f = load_words<WT>(front + (size_t)v * Wp + woff);
#pragma unroll
                    for (int i = 0; i < WT; ++i) m.w[i] |= f.w[i];
                }
                return m;
            };
            Words<WT> s0, s1, s2, s3;
#pragma unroll
            for (int i = 0; i < WT; ++i) s0.w[i] = s1.w[i] = s2.w[i] = s3.w[i] = 0;
            if (work && v0 >= 0) s0 = row_mask(v0);
            if (work && v3 >= 0) s3 = v3 == v0 ? s0 : row_mask(v3);
            // an interior row (neither the lane's first nor last row)
            if (work && v1 >= 0) s1 = v1 == v0 ? s0 : (v1 == v3 ? s3 : row_mask(v1));
            if (work && v2 >= 0) s2 = v2 == v1 ? s1 : (v2 == v3 ? s3 : row_mask(v2));
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                c0.w[i] &= ~s0.w[i];
                c1.w[i] &= ~s1.w[i];
                c2.w[i] &= ~s2.w[i];
                c3.w[i] &= ~s3.w[i];
            }
        }
        const u64 any = any_bits<WT>(c0) | any_bits<WT>(c1) | any_bits<WT>(c2) | any_bits<WT>(c3);
        STAMP(2);
        if (__any(any != 0)) {                                         // else: nothing new through these 256 edges
            // inclusive OR along the lane's own slots, restarting where the row changes
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                if (v1 == v0) c1.w[i] |= c0.w[i];
                if (v2 == v1) c2.w[i] |= c1.w[i];
                if (v3 == v2) c3.w[i] |= c2.w[i];
            }
            // across lanes: segmented scan over each lane's LAST run (row v3); a lane starts a new segment unless all
            // its slots share one row and that row is also the previous lane's last row
            const int pv3 = dpp_mov<DPP_WAVE_SHR1>(v3);
            const bool connects = lane > 0 && pv3 == v0 && v0 >= 0;
            Words<WT> t = c3;
            bool head = !(connects && v0 == v3);
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
        STAMP(3);
        // Emit every run that ends in this lane (the slot after it belongs to another row, or the chunk ends).
        const int nv0 = dpp_mov<DPP_WAVE_SHL1>(v0);
        const int after3 = lane == 63 ? -3 : nv0;
        const size_t i0 = (size_t)v0 * Wp + woff, i1 = (size_t)v1 * Wp + woff, i2 = (size_t)v2 * Wp + woff,
                     i3 = (size_t)v3 * Wp + woff;
        const bool e0 = v0 >= 0 && v0 != v1, e1 = v1 >= 0 && v1 != v2, e2 = v2 >= 0 && v2 != v3, e3 = v3 >= 0 && v3 != after3;
        const bool x1 = (head_multi && v1 == vc) || (tail_multi && v1 == vl);
        const bool x2 = (head_multi && v2 == vc) || (tail_multi && v2 == vl);
        const bool n0 = e0 && any_bits<WT>(c0) != 0, n1 = e1 && any_bits<WT>(c1) != 0, n2 = e2 && any_bits<WT>(c2) != 0,
                   n3 = e3 && any_bits<WT>(c3) != 0;
        // With the live table an all-zero row need not be written: nobody gathers a row whose live bit is clear.
        // (Several tiles share one live bit per node: then zeros are written too, so a live row is exact in every tile.)
        const bool dense = !LIVE || tiles > 1;
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
        STAMP(4);
        found |= n0 || n1 || n2 || n3;
        if constexpr (LIVE) {
            // Mark the rows that received something.  The chunk's rows are a short ascending run of node ids: build each
            // 32-bit table word with a wave-wide OR and let one lane publish it (per-row atomics -- ~30 to every word
            // from a few waves -- cost 14 us per dense level).
            if (__any(n0 || n1 || n2 || n3)) {
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
                    if (n0) atomicOr(&wave_words[(v0 >> 5) - wfirst], 1u << (v0 & 31));
                    if (n1) atomicOr(&wave_words[(v1 >> 5) - wfirst], 1u << (v1 & 31));
                    if (n2) atomicOr(&wave_words[(v2 >> 5) - wfirst], 1u << (v2 & 31));
                    if (n3) atomicOr(&wave_words[(v3 >> 5) - wfirst], 1u << (v3 & 31));
                    __builtin_amdgcn_wave_barrier();
                    if (lane <= kmax) {
                        const unsigned m = wave_words[lane];
                        if (m) atomicOr(&live_acc[wfirst + lane], m);
                    }
                } else {                                       // a run with wide gaps (isolated nodes in between)
                    if (n0) atomicOr(&live_acc[v0 >> 5], 1u << (v0 & 31));
                    if (n1) atomicOr(&live_acc[v1 >> 5], 1u << (v1 & 31));
                    if (n2) atomicOr(&live_acc[v2 >> 5], 1u << (v2 & 31));
                    if (n3) atomicOr(&live_acc[v3 >> 5], 1u << (v3 & 31));
                }
            }
        }
        STAMP(5);
    }
    STAMP(6);
    return found;
}

template <int WT, int LIVE>      // LIVE: 0 no table, 1 table staged in LDS, 2 table read from global memory (big graphs)
__global__ __launch_bounds__(256) void k_bfs_level(const int *__restrict__ erow, const int *__restrict__ col,
                                                   int E, int N, int Wp, const u64 *__restrict__ front,
                                                   u64 *__restrict__ seen, u64 *__restrict__ acc,
                                                   u64 *__restrict__ idle, u64 *__restrict__ hop_planes,
                                                   size_t plane_elems, int level, BfsCtl *ctl, const int *aux,
                                                   int expand_blocks, const unsigned *__restrict__ live,
                                                   unsigned *__restrict__ live_acc, unsigned *__restrict__ live_idle,
                                                   int live_words, int variant, const LevelCopy cp) {
    if ((int)blockIdx.x >= cp.first_block) {                          // copy role: runs whether or not the BFS is over
        if (blockIdx.y == 0) level_copy_role(cp);
        return;
    }
    if (bfs_over(ctl, aux, level)) return;
    const int lane = threadIdx.x & 63;
    int woff = blockIdx.y * WT;                                    // housekeeping: tile = blockIdx.y
    if ((int)blockIdx.x >= expand_blocks) {
        // Housekeeping blocks (beside the expand waves, not on their critical path):
        //  (1) clear, two levels ahead: the live table and the accumulator words of the rows that span chunks;
        //  (2) COMMIT level - 1 for every node: a node whose frontier row is non-zero gained those anchors at level - 1
        //      -> reachability plane and hop-bit planes.  The expand waves never commit: they mask with seen[v] | front[v],
        //      which is the same whether this commit has landed or not (OR is idempotent), and their chain ends at the store
        //      of the next frontier instead of a plane read-modify-write behind it.
        const int hb = cp.first_block - expand_blocks;
        level_housekeeping<WT, LIVE>(E, N, Wp, front, seen, idle, hop_planes, plane_elems, level, aux, live, live_idle, live_words,
                                     ((int)blockIdx.x - expand_blocks) * blockDim.x + threadIdx.x, hb * blockDim.x, woff, blockIdx.y == 0);
        return;
    }
    // Expand waves: with more than one word tile per node (K > 256) the tiles of ONE chunk go to adjacent waves of the
    // same block, so the 128-byte frontier line that all of them gather from is fetched from L2 once and served to the
    // others by the CU's L1 (one tile per launch row of blocks fetched it once per tile, from different CUs).
    const int tiles = gridDim.y;
    // POPE_KNOB_LEVEL_VARIANT bit 1: blocks are dealt round-robin over the 8 XCDs (block b and b + 8 share one); give every
    // XCD a CONTIGUOUS range of chunks, so that the seen / accumulator rows it touches are one eighth of those arrays.
    int bx = blockIdx.x;
    if (variant & 2) {
        const int q = expand_blocks >> 3, r = expand_blocks & 7, x = bx & 7;
        bx = x * q + min(x, r) + (bx >> 3);
    }
    const int wid = ((blockIdx.y * expand_blocks + bx) * blockDim.x + threadIdx.x) >> 6;
    const int wave = wid / tiles;                                  // which stream of chunks this wave walks
    woff = (wid - wave * tiles) * WT;
    const int nwaves = (expand_blocks * blockDim.x) >> 6;
    const int nchunks = (E + CHUNK - 1) >> CHUNK_SHIFT;
    // The first chunk's slot loads are issued before the live table is staged: they fly while LDS fills.
    int4 vr = make_int4(-1, -1, -1, -1), ur = make_int4(0, 0, 0, 0);
    // bit 0: the index streams are read once per launch: non-temporal, so that they do not evict the frontier from L2
    auto load_idx = [&](const int *p) {
        if (variant & 1) {
            const i32x4v t = __builtin_nontemporal_load(reinterpret_cast<const i32x4v *>(p));
            return make_int4(t.x, t.y, t.z, t.w);
        }
        return *reinterpret_cast<const int4 *>(p);
    };
    if (wave < nchunks && wave * CHUNK + lane * SLOTS < E) {
        vr = load_idx(erow + wave * CHUNK + lane * SLOTS);
        ur = load_idx(col + wave * CHUNK + lane * SLOTS);
    }
    extern __shared__ uint4 live_lds4[];
    const unsigned *live_lds = reinterpret_cast<const unsigned *>(live_lds4);
    if constexpr (LIVE == 1) {
        stage_live_table(live, live_words, live_lds4);
        __syncthreads();
    }
    __shared__ unsigned wave_live_words[4][8];                     // per wave: the live-table words its chunk's rows fall into
    const bool found = level_expand<WT, LIVE>(erow, col, E, Wp, front, seen, acc, live, live_acc, live_lds, variant, level, lane, wave, nwaves,
                                              nchunks, woff, tiles, vr, ur, wave_live_words[threadIdx.x >> 6]);
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
                                                  const int *__restrict__ aux, int *report, int ticket, int x_row_begin) {
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
        if (x && v >= x_row_begin) {                               // rows below: copied by the level launches (LevelCopy)
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
// MODE: 0 plain stores (default), 1 non-temporal stores (kept for A/B, tools/ab_finalize.py).
// n_shards > 1 (multi-GPU): `planes` holds the all-gathered shards back to back (shard_elems words apart, K anchors
// each); a row's columns of ALL shards are written in one pass, so the [N, F + shards*K] matrix is streamed once.
template <int MODE>
__global__ __launch_bounds__(256) void k_finalize_fast(const u64 *__restrict__ planes, size_t plane_elems,
                                                       int n_hop_bits, const int *__restrict__ max_hop_dev, int N, int K,
                                                       int Wp, const float *__restrict__ x, int F,
                                                       float *__restrict__ out, long long out_cols, int c0,
                                                       int n_shards, size_t shard_elems, const int *__restrict__ aux,
                                                       int *report, int ticket, int x_row_begin) {
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
        if (x && v >= x_row_begin) {                           // rows below were copied by the level launches (LevelCopy)
            const f32x4 *xs = reinterpret_cast<const f32x4 *>(x + (size_t)v * F);
            for (int q = lane; q < F4; q += 64) {
                const f32x4 t = __builtin_nontemporal_load(xs + q);
                if (MODE == 1) __builtin_nontemporal_store(t, orow + q);
                else orow[q] = t;
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
            if (MODE == 1) __builtin_nontemporal_store(r, erow + q);
            else erow[q] = r;
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
                                                       int ticket, int x_row_begin, int contiguous) {
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
    // (contiguous != 0 -- A/B, POPE_KNOB_FINALIZE_VARIANT 5 -- a block of consecutive items per wave, as k_finalize_fast deals rows).
    const int F4 = F >> 2, K4 = K >> 2, n_emb = K4 * n_shards;
    const int n_seg = (n_emb + 64 * EP - 1) / (64 * EP);
    const int items = N * n_seg;                                         // < 2^31: checked on the host
    const int per = (items + nwaves - 1) / nwaves;
    const int i_begin = contiguous ? wave * per : wave, i_end = contiguous ? min(items, i_begin + per) : items, i_step = contiguous ? 1 : nwaves;
    if (i_begin >= i_end) return;
    auto row_of = [&](int i, int &seg) { const int v = (int)((unsigned)i / (unsigned)n_seg); seg = i - v * n_seg; return v; };
    int seg = 0, v = row_of(i_begin, seg);
    FinRow<XP, EP> cur = fin_load<XP, EP>(planes, plane_elems, n_hop_bits, Wp, x, F4, v, lane, x && seg == 0 && v >= x_row_begin, K4, n_emb, shard_elems, seg * 64 * EP);
    for (int i = i_begin; i < i_end; i += i_step) {
        FinRow<XP, EP> nxt = cur;
        int seg_n = 0, v_n = 0;
        if (i + i_step < i_end) {
            v_n = row_of(i + i_step, seg_n);
            nxt = fin_load<XP, EP>(planes, plane_elems, n_hop_bits, Wp, x, F4, v_n, lane, x && seg_n == 0 && v_n >= x_row_begin, K4, n_emb, shard_elems, seg_n * 64 * EP);
        }
        f32x4 *orow = reinterpret_cast<f32x4 *>(out + (size_t)v * out_cols);
        if (XP > 0 && x && seg == 0 && v >= x_row_begin) {
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
                                                       int ticket, int x_row_begin) {
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
        if (XP > 0 && x && seg == 0 && v >= x_row_begin) {
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
        if (XP > 0 && x && seg == 0 && v >= x_row_begin) {
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

// ------------------------------------------------------------------------------------------------
// The sparse LAST levels of the BFS and the finalise kernel in ONE launch (round 4)
// ------------------------------------------------------------------------------------------------
// On the Flickr-shaped graph the last four level launches touch 3 174, 64, 1 and 0 nodes and still cost a kernel boundary
// (~3.3 us) plus a pass over the index streams each (27 us of the step), with the whole chip waiting; the finalise kernel
// behind them is a 100 us stream whose feature copy depends on nothing the BFS computes.  k_tail_finalize runs both at once:
//   * blocks 0 .. bfs_blocks-1 (dispatched first, so they are resident from the start) run levels first_level, first_level+1,
//     ... with the SAME per-level code as k_bfs_level (level_housekeeping + level_expand over their share of the chunks) and a
//     barrier among themselves between levels -- two stages of agent-scope counters in the control block, the arrival a
//     release, the departure an acquire -- until a level finds nothing (the launch after the last productive level commits
//     that level, as in k_bfs_level) or level_stop is reached (hop planes are only cleared for levels < 16: a deeper graph
//     answers "not done" and the host continues with level launches);
//   * block 0 then publishes the verdict (pinned report + ticket for the host, ctl->tail_done for the device);
//   * every other block copies its share of x's rows meanwhile (16 pieces of 16 bytes in flight per lane), waits for
//     tail_done, and expands its share of the planes into the K embedding columns (k_finalize_fast's arithmetic).
// Every wait is bounded by the 100 MHz real-time counter (TAIL_WAIT_TICKS): a wait that runs out sets ctl->tail_failed and
// the role moves on, so the grid always drains; the host reports the flag as an error.
constexpr unsigned long long TAIL_WAIT_TICKS = 200000000ull;        // 2 s

struct TailArgs {
    const int *erow, *col;
    int E, N, Wp;
    u64 *front[3];
    unsigned *live[3];
    int live_words;
    u64 *seen, *hop_planes;
    size_t plane_elems;
    BfsCtl *ctl;
    const int *aux;
    int first_level, level_stop, bfs_blocks, variant;
    int K, F;
    const float *x;
    float *out;
    long long out_cols;
    int x_row_begin;
    int *report;
    int ticket;
};

__device__ __forceinline__ bool tail_barrier(BfsCtl *ctl, int B, unsigned episode) {
    __shared__ int ok_s;
    __syncthreads();                                   // every store of this block has been issued and waited for
    if (threadIdx.x == 0) {
        const int G = B < TAIL_GROUPS ? B : TAIL_GROUPS, g = (int)blockIdx.x % G;
        const unsigned members = (unsigned)((B - g + G - 1) / G);
        // ONE release fence before the arrival and ONE acquire fence after the departure; the counters themselves are relaxed
        // agent-scope atomics (they bypass the non-coherent caches without maintaining them): an acquire on every poll is a
        // cache invalidation per iteration and made a level inside this kernel cost ~100 us.
        __threadfence();
        const unsigned old = __hip_atomic_fetch_add(&ctl->tail_group[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == members * episode) __hip_atomic_fetch_add(&ctl->tail_top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        while (__hip_atomic_load(&ctl->tail_top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G * episode) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > TAIL_WAIT_TICKS) { ok = 0; break; }
        }
        __threadfence();
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

template <int WT, int LIVE>
__global__ __launch_bounds__(256) void k_tail_finalize(const TailArgs a) {
    const int lane = threadIdx.x & 63;
    const int B = a.bfs_blocks;
    extern __shared__ uint4 live_lds4[];
    __shared__ unsigned wave_live_words[4][8];
    if ((int)blockIdx.x < B) {
        // ---- BFS role ----
        BfsCtl *ctl = a.ctl;
        int failed = 0;
        if (!bfs_over(ctl, a.aux, a.first_level)) {
            const int wave = ((int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x) >> 6, nwaves = (B * (int)blockDim.x) >> 6;
            const int nchunks = (a.E + CHUNK - 1) >> CHUNK_SHIFT;
            unsigned episode = 0;
            for (int level = a.first_level;; ++level) {
                const u64 *front = a.front[(level - 1) % 3];
                u64 *acc = a.front[level % 3], *idle = a.front[(level + 1) % 3];
                const unsigned *live = a.live[(level - 1) % 3];
                unsigned *live_acc = a.live[level % 3], *live_idle = a.live[(level + 1) % 3];
                int4 vr = make_int4(-1, -1, -1, -1), ur = make_int4(0, 0, 0, 0);
                if (wave < nchunks && wave * CHUNK + lane * SLOTS < a.E) {
                    vr = *reinterpret_cast<const int4 *>(a.erow + wave * CHUNK + lane * SLOTS);
                    ur = *reinterpret_cast<const int4 *>(a.col + wave * CHUNK + lane * SLOTS);
                }
                if constexpr (LIVE == 1) {
                    stage_live_table(live, a.live_words, live_lds4);
                    __syncthreads();
                }
                level_housekeeping<WT, LIVE>(a.E, a.N, a.Wp, front, a.seen, idle, a.hop_planes, a.plane_elems, level, a.aux, live, live_idle,
                                             a.live_words, (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x, B * (int)blockDim.x, 0, true);
                const bool found = level_expand<WT, LIVE>(a.erow, a.col, a.E, a.Wp, front, a.seen, acc, live, live_acc,
                                                          reinterpret_cast<const unsigned *>(live_lds4), a.variant, level, lane, wave, nwaves, nchunks,
                                                          0, 1, vr, ur, wave_live_words[threadIdx.x >> 6]);
                if (__any(found) && lane == 0) raise_level(ctl, level);
                if (!tail_barrier(ctl, B, ++episode)) { failed = 1; break; }
                // every block reads the same word here: all raises of this level came before the barrier
                const int la = __hip_atomic_load(&ctl->last_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (la < level || level + 1 >= a.level_stop) break;
            }
        }
        if (failed && threadIdx.x == 0) __hip_atomic_store(&ctl->tail_failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            // The other BFS blocks are past their last barrier too (or this block ran out of patience): publish.
            const int la = __hip_atomic_load(&ctl->last_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int bad = failed | __hip_atomic_load(&ctl->tail_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.report) {
                a.report[0] = la;
                a.report[1] = csr_flags(ctl, a.aux) | (bad ? BFS_FLAG_TAIL_FAILED : 0);
                __hip_atomic_store(&a.report[2], a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(&ctl->tail_done, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // ---- copy + expansion role ----
    const int nb = (int)gridDim.x - B, cb = (int)blockIdx.x - B;
    const int w = threadIdx.x >> 6;
    const unsigned F4 = (unsigned)a.F >> 2, opitch4 = (unsigned)(a.out_cols >> 2);
    if (a.x && F4) {
        const unsigned piece_begin = (unsigned)a.x_row_begin * F4, piece_end = (unsigned)a.N * F4;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.x);
        f32x4 *dst = reinterpret_cast<f32x4 *>(a.out);
        for (unsigned batch = (unsigned)cb;; batch += (unsigned)nb) {
            const unsigned long long base64 = (unsigned long long)piece_begin + ((unsigned long long)batch * 4u + w) * (64u * LEVEL_COPY_PIECES) + lane;
            if ((unsigned long long)piece_begin + (unsigned long long)batch * LEVEL_COPY_BLOCK_PIECES >= piece_end) break;
            if (base64 >= piece_end) continue;
            const unsigned base = (unsigned)base64;
            f32x4 v[LEVEL_COPY_PIECES];
#pragma unroll
            for (int j = 0; j < LEVEL_COPY_PIECES; ++j) {
                const unsigned long long p = (unsigned long long)base + 64u * j;
                if (p < piece_end) v[j] = __builtin_nontemporal_load(src + p);
            }
            unsigned row = base / F4, q = base - row * F4;
#pragma unroll
            for (int j = 0; j < LEVEL_COPY_PIECES; ++j) {
                const unsigned long long p = (unsigned long long)base + 64u * j;
                if (p < piece_end) {
                    if (a.variant & 8) __builtin_nontemporal_store(v[j], dst + (size_t)row * opitch4 + q);
                    else dst[(size_t)row * opitch4 + q] = v[j];
                }
                q += 64u;
                while (q >= F4) { q -= F4; ++row; }
            }
        }
    }
    // wait for the verdict of the BFS blocks
    __shared__ int go_s;
    __shared__ float inv[16];
    if (threadIdx.x < 16) inv[threadIdx.x] = 1.0f / (float)(threadIdx.x + 1);
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int go = 1;
        while (__hip_atomic_load(&a.ctl->tail_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.ticket) {
            __builtin_amdgcn_s_sleep(32);
            if (__builtin_amdgcn_s_memrealtime() - t0 > TAIL_WAIT_TICKS) { go = 0; break; }
        }
        __threadfence();
        if (!go) __hip_atomic_store(&a.ctl->tail_failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        go_s = go;
    }
    __syncthreads();
    if (!go_s) return;
    const int m = __hip_atomic_load(&a.ctl->last_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n_hop_bits = m > 0 ? 32 - __clz(m) : 0;
    const int wave = cb * 4 + w, nwaves = nb * 4;
    const int per = (a.N + nwaves - 1) / nwaves;
    const int v_begin = wave * per, v_end = min(a.N, v_begin + per);
    const int K4 = a.K >> 2;
    const u64 *planes = a.seen;                                  // plane 0 = reachability, planes 1.. = hop bits (contiguous)
    for (int v = v_begin; v < v_end; ++v) {
        f32x4 *erow = reinterpret_cast<f32x4 *>(a.out + (size_t)v * a.out_cols + a.F);
        const size_t wbase = (size_t)v * a.Wp;
        for (int q = lane; q < K4; q += 64) {
            const int j = q * 4;
            const size_t widx = wbase + (j >> 6);
            const int bit = j & 63;
            const unsigned reach = (unsigned)(planes[widx] >> bit) & 15u;
            unsigned t = 0;
            for (int b = 0; b < n_hop_bits; ++b)
                t |= ((unsigned)(planes[(size_t)(b + 1) * a.plane_elems + widx] >> bit) & 15u) << (4 * b);
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

extern "C" size_t pope_bfs_scratch_bytes(int64_t N, int64_t E, int32_t K) {
    if (N < 0 || E < 0 || K <= 0) return 0;
    (void)E;
    // control block | anchors[K] | three rotating frontier planes | their three live-bit tables
    return CTL_BYTES + align_up((size_t)K * sizeof(long long), 256) + 3 * align_up(pope_plane_bytes(N, K), 256) + 3 * live_bytes(N);
}

constexpr int LIVE_MAX_NODES = 256 * 1024;   // live table of 32 KB per block in LDS (4 blocks per CU); beyond: read from global
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
static int g_live_mode = -1;            // -1: by graph size (LDS table up to LIVE_MAX_NODES, global table beyond)
static int g_finalize_variant = 1;      // 0: generic kernel, 1: pipelined fast path (default), 2: round 1-3 fast path + non-temporal stores, 3 / 4: two launches, 5: pipelined with contiguous rows, 7: round 1-3 fast path
static int g_finalize_blocks = 256 * 8;
static bool g_finalize_blocks_set = false;   // POPE_KNOB_FINALIZE_BLOCKS given: it also sizes the pipelined kernel (default 4 096 blocks)
static int g_level_blocks = 0;           // cap on the expand blocks of a level launch (0: one wave per chunk up to 2048 blocks)
static int g_tail_level = 0;             // POPE_KNOB_TAIL_LEVEL: first level that runs inside k_tail_finalize (0: no tail kernel)
static int g_tail_blocks = 256;          // POPE_KNOB_TAIL_BLOCKS: BFS blocks of k_tail_finalize
static int g_prepare_merge = 1;          // POPE_KNOB_PREPARE_MERGE: 1 (default) = pope_geodesic_run clears, seeds and builds the CSR in ONE launch (k_prepare); 0 = two launches
static int g_level_variant = 0;          // POPE_KNOB_LEVEL_VARIANT bits: 1 nt index streams, 2 XCD-contiguous chunks, 4 nt reachability loads
// POPE_KNOB_LEVEL_COPY: per mille of x's rows that level launch l of pope_geodesic_run copies in its copy role (LevelCopy).
// Index 0 is unused.  Launches the speculative window does not reach leave their share to the finalise kernel.
constexpr int LEVEL_COPY_SLOTS = 16;
static int g_level_copy_permille[LEVEL_COPY_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
namespace pope { int g_gemm_force_tile = 0, g_pairwise_kernel = 0, g_fail_host_register = 0, g_host_result_mode = 0, g_gemm_split_bf16 = 0, g_gather_lds_pad_kb = 0, g_sage_forward_overlap = 1, g_gemm_small_tile16 = 1, g_gemm_tile16_buffers = 4; extern int g_sage_lanes; }

extern "C" int pope_debug_set(int32_t knob, int32_t value) {
    clear_error();
    switch (knob) {
    case POPE_KNOB_LIVE_MODE:        g_live_mode = value; break;
    case POPE_KNOB_FINALIZE_VARIANT: g_finalize_variant = value; break;
    case POPE_KNOB_FINALIZE_BLOCKS:  g_finalize_blocks = value > 0 ? value : 256 * 8; g_finalize_blocks_set = value > 0; break;
    case POPE_KNOB_GEMM_TILE:        pope::g_gemm_force_tile = value; break;
    case POPE_KNOB_PAIRWISE_KERNEL:  pope::g_pairwise_kernel = value; break;
    case POPE_KNOB_COPY_BATCHES:     pope::g_copy_batches_per_wave = value; break;
    case POPE_KNOB_LEVEL_BLOCKS:     g_level_blocks = value; break;
    case POPE_KNOB_FAIL_HOST_REGISTER: pope::g_fail_host_register = value; break;
    case POPE_KNOB_HOST_RESULT_MODE: pope::g_host_result_mode = value; break;
    case POPE_KNOB_GEMM_SPLIT_BF16:  pope::g_gemm_split_bf16 = value; break;
    case POPE_KNOB_GATHER_LDS_PAD_KB: pope::g_gather_lds_pad_kb = value; break;
    case POPE_KNOB_SAGE_FORWARD_OVERLAP: pope::g_sage_forward_overlap = value; break;
    case POPE_KNOB_GEMM_TILE16_BUFFERS: pope::g_gemm_tile16_buffers = value == 4 ? 4 : 3; break;
    case POPE_KNOB_GEMM_SMALL_TILE16: pope::g_gemm_small_tile16 = value; break;
    case POPE_KNOB_SAGE_LANES:       pope::g_sage_lanes = value; break;
    case POPE_KNOB_LEVEL_VARIANT:    g_level_variant = value; break;
    case POPE_KNOB_TAIL_LEVEL:       g_tail_level = value; break;
    case POPE_KNOB_TAIL_BLOCKS:      g_tail_blocks = value > 0 ? value : 256; break;
    case POPE_KNOB_PREPARE_MERGE:    g_prepare_merge = value; break;
    case POPE_KNOB_LEVEL_COPY: {                                     // value = level << 16 | per mille; level 0: every launch
        const int lv = (value >> 16) & 0xff, pm = value & 0xffff;
        if (lv >= LEVEL_COPY_SLOTS || pm > 1000) { set_error("pope_debug_set: level copy %d / %d", lv, pm); return POPE_ERR_INVALID; }
        if (lv == 0) for (int i = 1; i < LEVEL_COPY_SLOTS; ++i) g_level_copy_permille[i] = pm;
        else g_level_copy_permille[lv] = pm;
        break;
    }
    default: set_error("pope_debug_set: unknown knob %d", knob); return POPE_ERR_INVALID;
    }
    return POPE_OK;
}

template <int WT>
static void launch_level(int E, int N, int Wp, const int *col, const int *erow, const int *aux, const u64 *front, u64 *seen,
                         u64 *acc, u64 *idle, u64 *hop_planes, size_t plane_elems, int level, BfsCtl *ctl,
                         const unsigned *live, unsigned *live_acc, unsigned *live_idle, int live_words, LevelCopy cp, hipStream_t stream) {
    const int nchunks = (E + CHUNK - 1) >> CHUNK_SHIFT;
    int expand_blocks = (nchunks + 3) / 4;                           // one wave per chunk ...
    if (expand_blocks > 256 * 8) expand_blocks = 256 * 8;            // ... up to 8 blocks per CU, then waves loop
    if (g_level_blocks > 0 && expand_blocks > g_level_blocks) expand_blocks = g_level_blocks;   // A/B: POPE_KNOB_LEVEL_BLOCKS
    int house_blocks = (N + 255) / 256;                              // the commit of the previous level: one thread per node
    if (house_blocks > 1024) house_blocks = 1024;                    // (+ the clears: rows that span chunks, the live table)
    cp.first_block = expand_blocks + house_blocks;                   // the copy role's blocks come last: dispatched behind the BFS roles
    cp.blocks = (int)((cp.piece_end - cp.piece_begin + LEVEL_COPY_BLOCK_PIECES - 1) / LEVEL_COPY_BLOCK_PIECES);
    const int gx = cp.first_block + cp.blocks;
    profile_mark(stream, level, 0);
    const int mode = g_live_mode >= 0 ? g_live_mode : (live_words <= LIVE_MAX_NODES / 32 ? 1 : 2);
    if (mode == 1 && live_words <= LIVE_MAX_NODES / 32)
        hipLaunchKernelGGL((k_bfs_level<WT, 1>), dim3(gx, Wp / WT), dim3(256),
                           align_up((size_t)live_words * sizeof(unsigned), 16), stream, erow, col, E, N, Wp, front, seen, acc, idle, hop_planes,
                           plane_elems, level, ctl, aux, expand_blocks, live, live_acc, live_idle, live_words, g_level_variant, cp);
    else if (mode == 2)
        hipLaunchKernelGGL((k_bfs_level<WT, 2>), dim3(gx, Wp / WT), dim3(256), 0, stream, erow,
                           col, E, N, Wp, front, seen, acc, idle, hop_planes, plane_elems, level, ctl, aux, expand_blocks, live,
                           live_acc, live_idle, live_words, g_level_variant, cp);
    else
        hipLaunchKernelGGL((k_bfs_level<WT, 0>), dim3(gx, Wp / WT), dim3(256), 0, stream, erow,
                           col, E, N, Wp, front, seen, acc, idle, hop_planes, plane_elems, level, ctl, aux, expand_blocks, live,
                           live_acc, live_idle, live_words, g_level_variant, cp);
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
    b.level_limit = 1ll << plane_capacity;
    return slot_acquire(&b.slot, (size_t)K);
}

// One launch clears the control block, the three frontier buffers, the reachability plane, the first hop planes and
// (pope_geodesic_run) the CSR status header.
static void bfs_enqueue_clear(const Bfs &b, int *aux_header, hipStream_t stream) {
    const int eager = b.capacity < EAGER_PLANES ? b.capacity : EAGER_PLANES;
    hipLaunchKernelGGL(k_zero, dim3(2048), dim3(256), 0, stream, (uint4 *)b.base,
                       (b.front_off + 3 * align_up(b.plane_bytes, 256) + 3 * live_bytes(b.N)) / 16, (uint4 *)b.seen,
                       (size_t)(1 + eager) * b.plane_bytes / 16, (uint4 *)aux_header,
                       aux_header ? (size_t)AUX_HEADER * sizeof(int) / 16 : (size_t)0);
}

// Anchors go through pinned, device-mapped host memory and the seed kernel reads them in place: no copy kernel.
static int bfs_enqueue_seed(const Bfs &b, const int64_t *anchors_host, hipStream_t stream) {
    memcpy(b.slot->anchors, anchors_host, (size_t)b.K * sizeof(long long));
    hipLaunchKernelGGL(k_bfs_seed, dim3((b.K + 255) / 256), dim3(256), 0, stream, b.slot->anchors_dev, b.K, b.Wp, b.seen, b.front[0], b.live[0]);
    return POPE_OK;
}

static int bfs_enqueue_init(const Bfs &b, const int64_t *anchors_host, hipStream_t stream) {
    bfs_enqueue_clear(b, nullptr, stream);
    return bfs_enqueue_seed(b, anchors_host, stream);
}

// Which rows of x the level launches of one pope_geodesic_run copy (LevelCopy): launch l takes rows [cut[l - 1], cut[l]).
struct CopyPlan {
    const float *x = nullptr;
    float *out = nullptr;
    unsigned F4 = 0, opitch4 = 0;
    int cut[LEVEL_COPY_SLOTS] = {};
    int levels = 0;                     // launches 1 .. levels carry a slice
    int rows() const { return cut[levels]; }
};

// out[:, :F] = x in 16-byte pieces with 32-bit piece indices, and a finalise kernel that can start at a row of its choice.
static bool level_copy_eligible(const float *x, int32_t F, const float *out, int64_t out_cols, int64_t N, int32_t K) {
    return x && out && F > 0 && (F & 3) == 0 && (K & 3) == 0 && (out_cols & 3) == 0 && aligned16(x) && aligned16(out) &&
           (uint64_t)N * (uint64_t)(F / 4) < (1ull << 32);
}

static CopyPlan make_copy_plan(const float *x, int32_t F, float *out, int64_t out_cols, int64_t N, int launches) {
    CopyPlan p;
    p.x = x; p.out = out; p.F4 = (unsigned)F / 4; p.opitch4 = (unsigned)(out_cols / 4);
    p.levels = launches < LEVEL_COPY_SLOTS - 1 ? launches : LEVEL_COPY_SLOTS - 1;
    int64_t acc = 0;                                                 // per mille so far
    for (int l = 1; l <= p.levels; ++l) {
        acc += g_level_copy_permille[l];
        if (acc > 1000) acc = 1000;
        p.cut[l] = (int)(N * acc / 1000);
    }
    return p;
}

// Enqueue levels [level, stop) (clipped to what the hop-bit capacity can represent); returns the next level.
static int bfs_enqueue_levels(const Bfs &b, int level, int stop, hipStream_t stream, const CopyPlan *plan = nullptr) {
    const int first = level;
    profile_mark(stream, 0, 0, true);                                // span mode: one event pair around the whole run
    for (; level < stop; ++level) {
        if (level >= b.level_limit || b.E == 0) break;
        if ((level & (level - 1)) == 0 && level >= (1 << EAGER_PLANES)) {   // first level with this hop bit
            int bit = 0;
            while ((1 << bit) < level) ++bit;
            hipLaunchKernelGGL(k_zero, dim3(1024), dim3(256), 0, stream, (uint4 *)(b.hop_planes + (size_t)bit * b.plane_elems),
                               b.plane_bytes / 16, (uint4 *)nullptr, (size_t)0, (uint4 *)nullptr, (size_t)0);
        }
        const u64 *prev = b.front[(level - 1) % 3];           // frontier of level - 1
        u64 *next = b.front[level % 3];                          // receives the frontier of this level
        u64 *idle = b.front[(level + 1) % 3];                    // next level's accumulator: rows spanning chunks cleared now
        const unsigned *lp = b.live[(level - 1) % 3];
        unsigned *ln = b.live[level % 3], *li = b.live[(level + 1) % 3];
        LevelCopy cp;
        if (plan && level <= plan->levels && plan->cut[level] > plan->cut[level - 1]) {
            cp.x = plan->x; cp.out = plan->out; cp.F4 = plan->F4; cp.opitch4 = plan->opitch4;
            cp.piece_begin = (unsigned)plan->cut[level - 1] * plan->F4;
            cp.piece_end = (unsigned)plan->cut[level] * plan->F4;
        }
        if (b.Wp == 1)      launch_level<1>(b.E, b.N, b.Wp, b.col, b.erow, b.aux, prev, b.seen, next, idle, b.hop_planes, b.plane_elems, level, b.ctl, lp, ln, li, b.live_words, cp, stream);
        else if (b.Wp == 2) launch_level<2>(b.E, b.N, b.Wp, b.col, b.erow, b.aux, prev, b.seen, next, idle, b.hop_planes, b.plane_elems, level, b.ctl, lp, ln, li, b.live_words, cp, stream);
        else                launch_level<4>(b.E, b.N, b.Wp, b.col, b.erow, b.aux, prev, b.seen, next, idle, b.hop_planes, b.plane_elems, level, b.ctl, lp, ln, li, b.live_words, cp, stream);
    }
    profile_mark(stream, 0, 1, true);
    if (g_profile.enabled && g_profile.span_only && !g_profile.level.empty())
        g_profile.level.back() = -(level - first);                  // span entries carry minus the number of launches
    return level;
}

// The tail kernel (k_tail_finalize): levels [first_level, level_stop) until one finds nothing, the verdict, and -- with `out` --
// the rest of the feature copy and the embedding columns.  The BFS blocks must all be resident at once (they wait for one
// another): their number is capped at half of what the device holds of this kernel, the copy blocks fill the rest.
template <int WT, int LIVE>
static int launch_tail_t(const TailArgs &a0, size_t lds, hipStream_t stream) {
    TailArgs a = a0;
    static std::atomic<int> resident_cache{0};
    int resident = resident_cache.load(std::memory_order_relaxed);
    if (!resident) {
        int per_cu = 0, dev = 0, cus = 0;
        POPE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tail_finalize<WT, LIVE>, 256, LIVE == 1 ? 32 * 1024 : 0));
        POPE_HIP(hipGetDevice(&dev));
        POPE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        resident = per_cu * cus;
        POPE_REQUIRE(resident >= 2, "tail kernel: the device holds %d blocks", resident);
        resident_cache.store(resident, std::memory_order_relaxed);
    }
    if (a.bfs_blocks > resident / 2) a.bfs_blocks = resident / 2;
    const int copy_blocks = a.out ? resident - a.bfs_blocks : 0;
    hipLaunchKernelGGL((k_tail_finalize<WT, LIVE>), dim3(a.bfs_blocks + copy_blocks), dim3(256), lds, stream, a);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

static int launch_tail(const TailArgs &a, hipStream_t stream) {
    const int mode = g_live_mode >= 0 ? g_live_mode : (a.live_words <= LIVE_MAX_NODES / 32 ? 1 : 2);
    const int live = mode == 1 && a.live_words <= LIVE_MAX_NODES / 32 ? 1 : (mode == 2 ? 2 : 0);
    const size_t lds = live == 1 ? align_up((size_t)a.live_words * sizeof(unsigned), 16) : 0;
#define POPE_TAIL(WT)                                                      \
    (live == 1 ? launch_tail_t<WT, 1>(a, lds, stream) : live == 2 ? launch_tail_t<WT, 2>(a, lds, stream) : launch_tail_t<WT, 0>(a, lds, stream))
    if (a.Wp == 1) return POPE_TAIL(1);
    if (a.Wp == 2) return POPE_TAIL(2);
    return POPE_TAIL(4);
#undef POPE_TAIL
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
    if (flags & BFS_FLAG_TAIL_FAILED) {
        set_error("geodesic bfs: a bounded wait inside the tail kernel ran out (its BFS blocks did not meet)");
        return POPE_ERR_HIP;
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

static int finalize_enqueue(const u64 *planes, int n_hop_bits, const int *max_hop_dev, int64_t N, int32_t K,
                            const float *x, int32_t F, float *out, int64_t out_cols, int32_t c0, hipStream_t stream,
                            int n_shards = 1, size_t shard_elems = 0, const int *aux = nullptr, int *report = nullptr,
                            int ticket = 0, int x_row_begin = 0) {
    const int Wp = words_for(K);
    const size_t plane_elems = (size_t)N * Wp;
    const bool vec = F % 4 == 0 && K % 4 == 0 && c0 % 4 == 0 && out_cols % 4 == 0 && aligned16(out) && (!x || aligned16(x));
    dim3 grid(capped_grid((size_t)N * 64, 256)), block(256);
    // The device-side depth (max_hop_dev) is only used by pope_geodesic_run, whose speculative window stops at
    // LEVEL_BATCH = 12 levels: at most 4 hop bits.  With a host-side count the fast path needs n_hop_bits <= 4.
    if (n_shards > 1 && !(vec && n_hop_bits <= 4)) {          // generic kernel: one launch per shard
        for (int g = 0; g < n_shards; ++g) {
            int rc = finalize_enqueue(planes + (size_t)g * shard_elems, n_hop_bits, max_hop_dev, N, K, g == 0 ? x : nullptr, F, out,
                                      out_cols, c0 + g * K, stream);
            if (rc) return rc;
        }
        return POPE_OK;
    }
    if (vec && (g_finalize_variant > 0 || n_shards > 1) && (max_hop_dev || n_hop_bits <= 4)) {
        dim3 fgrid(g_finalize_blocks);                          // 8 blocks per CU, contiguous row blocks per wave
        // POPE_KNOB_FINALIZE_VARIANT 3 / 4 (round-4 A/B): the embedding columns and the feature copy as TWO launches on the same
        // stream -- 3: columns first (that kernel carries the verdict, so the host still hears it when the BFS ends), then
        // side_copy.hip's copy kernel (5.9 TB/s alone); 4: the copy first.
        if ((g_finalize_variant == 3 || g_finalize_variant == 4) && x && n_shards == 1 && x_row_begin < N &&
            SideCopy::eligible(x, F, out, out_cols, N)) {
            const float *xs = x + (size_t)x_row_begin * F;
            float *os = out + (size_t)x_row_begin * out_cols;
            if (g_finalize_variant == 4) { int rc = enqueue_copy_features(xs, F, os, out_cols, N - x_row_begin, stream); if (rc) return rc; }
            hipLaunchKernelGGL(k_finalize_fast<0>, fgrid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, (long long)out_cols, c0, n_shards, shard_elems, aux, report, ticket, (int)N);
            if (g_finalize_variant == 3) { int rc = enqueue_copy_features(xs, F, os, out_cols, N - x_row_begin, stream); if (rc) return rc; }
            POPE_HIP(hipGetLastError());
            return POPE_OK;
        }
        // the pipelined kernel (default, variant 1; 5: with contiguous row blocks; 7: the round 1-3 kernel): shapes it has instances for
        {
            const int xp = !x ? 0 : (F <= 256 ? 1 : F <= 512 ? 2 : F <= 1024 ? 4 : -1);
            const int64_t ne = (int64_t)(K / 4) * n_shards;
            const int ep = ne <= 64 ? 1 : ne <= 128 ? 2 : 4;             // wider rows: segments of 256 pieces, one work item each
            const int64_t items = N * ((ne + 64 * ep - 1) / (64 * ep));
            const int64_t witems = N * ((ne / 16 + 15) / 16);
            if (g_finalize_variant == 1 && xp >= 0 && ne > 64 && (K & 63) == 0 && witems + 32768 * 4 < INT32_MAX) {      // wide rows: one load per plane half-word, shuffles to the lanes
                dim3 wgrid(g_finalize_blocks_set ? g_finalize_blocks : (unsigned)std::min<int64_t>(std::max<int64_t>((witems + 3) / 4, 256), 32768));
#define POPE_FIN_WIDE(XP)                                                                                                                 \
    hipLaunchKernelGGL((k_finalize_wide<XP>), wgrid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, \
                       (long long)out_cols, c0, n_shards, shard_elems, aux, report, ticket, x_row_begin)
                if (xp == 0) POPE_FIN_WIDE(0); else if (xp == 1) POPE_FIN_WIDE(1); else if (xp == 2) POPE_FIN_WIDE(2); else POPE_FIN_WIDE(4);
#undef POPE_FIN_WIDE
                POPE_HIP(hipGetLastError());
                return POPE_OK;
            }
            if ((g_finalize_variant == 1 || g_finalize_variant == 5) && xp >= 0 && items + 32768 * 4 < INT32_MAX) {
                const int contiguous = g_finalize_variant == 5;
                // one row per wave by default (grid sweep, profiles/r04_finalize_pipe*.txt: 2 048 blocks 0.2479 ms, 4 096 0.2456, 8 192
                // 0.2416, 16 384 0.2394, one row per wave 0.2395, 32 768 0.2400): short-lived waves in row order
                const int64_t one_row_per_wave = std::min<int64_t>(std::max<int64_t>((items + 3) / 4, 256), 32768);
                dim3 pgrid(g_finalize_blocks_set ? g_finalize_blocks : (unsigned)one_row_per_wave);
#define POPE_FIN_PIPE(XP, EP)                                                                                                             \
    hipLaunchKernelGGL((k_finalize_pipe<XP, EP>), pgrid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, \
                       (long long)out_cols, c0, n_shards, shard_elems, aux, report, ticket, x_row_begin, contiguous)
                if (xp == 0)      { if (ep == 1) POPE_FIN_PIPE(0, 1); else if (ep == 2) POPE_FIN_PIPE(0, 2); else POPE_FIN_PIPE(0, 4); }
                else if (xp == 1) { if (ep == 1) POPE_FIN_PIPE(1, 1); else if (ep == 2) POPE_FIN_PIPE(1, 2); else POPE_FIN_PIPE(1, 4); }
                else if (xp == 2) { if (ep == 1) POPE_FIN_PIPE(2, 1); else if (ep == 2) POPE_FIN_PIPE(2, 2); else POPE_FIN_PIPE(2, 4); }
                else              { if (ep == 1) POPE_FIN_PIPE(4, 1); else if (ep == 2) POPE_FIN_PIPE(4, 2); else POPE_FIN_PIPE(4, 4); }
#undef POPE_FIN_PIPE
                POPE_HIP(hipGetLastError());
                return POPE_OK;
            }
        }
        if (g_finalize_variant == 2)
            hipLaunchKernelGGL(k_finalize_fast<1>, fgrid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, (long long)out_cols, c0, n_shards, shard_elems, aux, report, ticket, x_row_begin);
        else
            hipLaunchKernelGGL(k_finalize_fast<0>, fgrid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, (long long)out_cols, c0, n_shards, shard_elems, aux, report, ticket, x_row_begin);
        POPE_HIP(hipGetLastError());
        return POPE_OK;
    }
    if (vec)
        hipLaunchKernelGGL(k_finalize<true>, grid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, (long long)out_cols, c0, aux, report, ticket, x_row_begin);
    else
        hipLaunchKernelGGL(k_finalize<false>, grid, block, 0, stream, planes, plane_elems, n_hop_bits, max_hop_dev, (int)N, K, Wp, x, F, out, (long long)out_cols, c0, aux, report, ticket, x_row_begin);
    POPE_HIP(hipGetLastError());
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
    // The level launches copy part of out[:, :F] = x in their copy role (LevelCopy); the finalise kernel copies the rest.
    CopyPlan plan;
    if (out && level_copy_eligible(x, F, out, out_cols, N, K)) plan = make_copy_plan(x, F, out, out_cols, N, window);
    memcpy(b.slot->anchors, anchors_host, (size_t)K * sizeof(long long));     // this call's pinned, device-mapped slot: read in place
    if (g_prepare_merge && K <= PREP_MAX_ANCHORS && E > 0) {
        // one launch: clear + seed role beside the speculative CSR role (k_prepare)
        static std::atomic<unsigned> epochs{0};
        unsigned epoch = ++epochs & 0x1fffffffu;
        if (epoch == 0) epoch = ++epochs & 0x1fffffffu;
        PrepSeeds seeds;
        for (int j = 0; j < K; ++j) seeds.a[j] = (int)anchors_host[j];          // (validated by bfs_setup)
        const int eager = b.capacity < EAGER_PLANES ? b.capacity : EAGER_PLANES;
        const size_t na = (b.front_off + 3 * align_up(b.plane_bytes, 256) + 3 * live_bytes(b.N)) / 16, nb = (size_t)(1 + eager) * b.plane_bytes / 16;
        const long long *src = (const long long *)edge_index, *dst = src + E;
        // (knob values above 1, for A/B: low 16 bits = the clear role's block count, high 16 bits = a cap on the CSR role's)
        const int zero_blocks = (g_prepare_merge & 0xffff) > 1 ? (g_prepare_merge & 0xffff) : 1024;
        const bool pairs = (E & 1) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(col) | reinterpret_cast<uintptr_t>(erow)) & 15u) == 0;
        unsigned csr_blocks = pairs ? capped_grid(E / 2, 256) : capped_grid(E, 256);
        if ((g_prepare_merge >> 16) > 0) csr_blocks = std::min<unsigned>(csr_blocks, (unsigned)(g_prepare_merge >> 16));
        if (pairs)
            hipLaunchKernelGGL(k_prepare<true>, dim3(zero_blocks + csr_blocks), dim3(256), 0, stream, src, dst, (int)E, (int)N, rowptr, col, erow, aux,
                               (uint4 *)b.base, na, (uint4 *)b.seen, nb, zero_blocks, epoch, seeds, K, b.Wp, b.seen, b.front[0], b.live[0]);
        else
            hipLaunchKernelGGL(k_prepare<false>, dim3(zero_blocks + csr_blocks), dim3(256), 0, stream, src, dst, (int)E, (int)N, rowptr, col, erow, aux,
                               (uint4 *)b.base, na, (uint4 *)b.seen, nb, zero_blocks, epoch, seeds, K, b.Wp, b.seen, b.front[0], b.live[0]);
        POPE_HIP(hipGetLastError());
    } else {
        bfs_enqueue_clear(b, aux, stream);                    // BFS state and the CSR status header in one launch
        SeedArgs seed;
        seed.anchors = b.slot->anchors_dev; seed.K = K; seed.Wp = b.Wp; seed.seen = b.seen; seed.front = b.front[0]; seed.live = b.live[0];
        rc = csr_build(edge_index, E, N, rowptr, col, erow, aux, ws + L.csr_scratch, L.planes - L.csr_scratch, 2, seed, stream);
        if (rc) return rc;
    }
    // Round 4: the sparse last levels run inside the finalise kernel's launch (k_tail_finalize) when POPE_KNOB_TAIL_LEVEL names
    // the first of them: one word tile per node, levels below 16, and -- with an output -- the fast expansion's shapes.
    const int level_stop = (int)std::min<long long>(b.level_limit, 1 << EAGER_PLANES);
    const bool tail = g_tail_level >= 2 && g_tail_level <= window && g_tail_level < level_stop && b.Wp <= 4 && E > 0 &&
                      (!out || ((F & 3) == 0 && (K & 3) == 0 && (out_cols & 3) == 0 && aligned16(out) && (!x || aligned16(x)) &&
                                (uint64_t)N * (uint64_t)(F / 4 + 1) < (1ull << 32)));
    int level = bfs_enqueue_levels(b, 1, tail ? g_tail_level : 1 + window, stream, plan.levels ? &plan : nullptr);
    const int x_row_begin = plan.levels ? plan.cut[std::min(level - 1, plan.levels)] : 0;      // rows the launches really took
    // The finalise kernel writes the verdict into the pinned report when it starts: no report launch, and the host
    // returns as soon as the BFS is known to be complete -- `out` is finished in stream order.
    int ticket = 0;
    if (tail && level == g_tail_level) {
        ticket = b.slot->ticket = b.slot->ticket == INT32_MAX ? 1 : b.slot->ticket + 1;
        TailArgs a;
        a.erow = b.erow; a.col = b.col; a.E = b.E; a.N = b.N; a.Wp = b.Wp;
        for (int i = 0; i < 3; ++i) { a.front[i] = b.front[i]; a.live[i] = b.live[i]; }
        a.live_words = b.live_words; a.seen = b.seen; a.hop_planes = b.hop_planes; a.plane_elems = b.plane_elems;
        a.ctl = b.ctl; a.aux = b.aux; a.first_level = level; a.level_stop = level_stop; a.bfs_blocks = g_tail_blocks; a.variant = g_level_variant;
        a.K = K; a.F = out ? F : 0; a.x = out ? x : nullptr; a.out = out; a.out_cols = out_cols; a.x_row_begin = x_row_begin;
        a.report = b.slot->report_dev; a.ticket = ticket;
        if ((rc = launch_tail(a, stream))) return rc;
        level = level_stop;                               // what the tail kernel may have run
    } else if (out) {
        ticket = b.slot->ticket = b.slot->ticket == INT32_MAX ? 1 : b.slot->ticket + 1;
        if ((rc = finalize_enqueue(planes, 0, &b.ctl->last_active, N, K, x, F, out, out_cols, 0, stream, 1, 0, aux,
                                   b.slot->report_dev, ticket, x_row_begin))) return rc;
    }
    int last_active = 0;
    bool done = false;
    rc = bfs_poll(b, level, &last_active, &done, stream, ticket);
    if (rc == POPE_ERR_UNSORTED) {                        // general path: counting sort, then start over
        clear_error();
        if ((rc = csr_fallback((const long long *)edge_index, (const long long *)edge_index + E, (int)E, (int)N, rowptr, col,
                               erow, aux, ws + L.csr_scratch, stream))) return rc;
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
    if (out && (rc = finalize_enqueue(planes, hop_bits(last_active), nullptr, N, K, x, F, out, out_cols, 0, stream, 1, 0, nullptr, nullptr, 0,
                                      x_row_begin))) return rc;
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
