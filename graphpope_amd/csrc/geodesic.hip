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
#ifndef POPE_HBM_BLOCKS
#define POPE_HBM_BLOCKS 0             // A/B: cap of the expand blocks of the 8-word kernel on HBM-resident graphs (0: 2 048)
#endif
#ifndef POPE_L2_LOOP
#define POPE_L2_LOOP (256 * 2)        // cap of the expand blocks of the 8-word, wave-per-tile kernel (graphs with the LDS live table): what is resident at once; 0 = no cap
#endif
#ifndef POPE_WT8_L2_WAVES
#define POPE_WT8_L2_WAVES 1           // waves per SIMD that kernel must fit (1: the compiler's own 178 registers = 2 waves, no scratch)
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
    // The seeds: the anchors are read in place from the call's pinned host slot -- a PCIe round trip per load -- so the last
    // ceil(K / 256) blocks take 256 anchors each, one load per thread (one block looping over 1 024 anchors made four serial
    // round trips: 18.6 us for this launch at configs[3]'s shape where the CSR pass itself needs 8).
    const int seed_blocks = (K + (int)blockDim.x - 1) / (int)blockDim.x;
    if (K > 0 && (int)gridDim.x >= seed_blocks) {
        const int sb = (int)blockIdx.x - ((int)gridDim.x - seed_blocks);
        if (sb >= 0) {
            const int j = sb * (int)blockDim.x + (int)threadIdx.x;
            if (j < K) seed_anchor(anchors[j], j, Wp, seen, front, live);
        }
    } else if (K > 0 && blockIdx.x == gridDim.x - 1) {
        for (int j = threadIdx.x; j < K; j += blockDim.x) seed_anchor(anchors[j], j, Wp, seen, front, live);
    }
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

#include "geodesic_level.h"        // Words, DPP moves, live tables, k_live_summary, k_bfs_level

#include "geodesic_expand.h"       // write_report, k_finalize*, k_hops, k_hop_codes, k_column_stats_*, k_concat

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
    if (POPE_HBM_BLOCKS != 0 && WT == 8 && TILES != 2 && expand_blocks > POPE_HBM_BLOCKS) expand_blocks = POPE_HBM_BLOCKS;   // (A/B: only what is resident)
    if (TILES == 2) expand_blocks *= tiles;                          // ... and per tile (4 waves per block: the tiles of a chunk share a block for 1, 2 or 4 tiles)
    // The 8-word kernel holds two blocks per CU (178 registers; three when held to 168): more blocks than that queue behind them and
    // their waves START late -- the 1 024-anchor levels of the Flickr-shaped graph were 2.3 such rounds.  Only what is resident is
    // launched; its waves walk their chunks in a loop, the next chunk's indices requested a chunk ahead (as the HBM-resident graphs'
    // waves do).  BFS at 512 / 1 024 anchors: every block launched 0.276 / 0.425 ms; 768 blocks at 3 waves per SIMD 0.257 / 0.413 (512
    // and 1 024 blocks: 0.272 / 0.419, 0.273 / 0.412); 512 blocks at the compiler's own 2 waves, no scratch: 0.250 / 0.388 (640:
    // 0.252 / 0.398) -- profiles/r05_ab_flickr_loop.txt.
    if (POPE_L2_LOOP != 0 && WT == 8 && TILES == 2 && expand_blocks > POPE_L2_LOOP) expand_blocks = POPE_L2_LOOP / tiles * tiles;
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
        // tests/test_soak_gpu.py hit it on its fourth call; random and 0xFF fills, which round 4 soaked, could not.)
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
