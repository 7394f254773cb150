// Geodesic GraphPOPE embedding on MI355X (gfx950): CSR build, bit-parallel multi-source BFS, finalise.
//
// Replaces /root/reference/utils.py:64-135 (one NetworkX bidirectional BFS per (node, anchor) pair on a
// multiprocessing pool, then 1/len(path), tensor conversion and torch.cat).  Design:
//
//  * anchors are packed 64 per uint64 word; every node carries W words, so one pass over the CSR advances
//    the BFS of all K anchors by one level ("MS-BFS");
//  * the level kernel is BOTTOM-UP (pull): node v ORs the frontier words of its out-neighbours, because
//    hop(v -> anchor) = 1 + min over edges v -> u of hop(u -> anchor).  Only v's owner writes v's state, so
//    there are no atomics and the result is independent of scheduling;
//  * hop counts are stored bit-sliced: plane b gets `new` OR-ed in when bit b of the level is set.  State is
//    a few N*W*8-byte planes that live in L2 / Infinity Cache; the 4*N*K-byte float matrix is written once,
//    coalesced, by the finalise kernel straight into the [N, F+K] output (no transpose, no torch.cat).
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace pope {

// ------------------------------------------------------------------------------------------------
// CSR build
// ------------------------------------------------------------------------------------------------
enum { CSR_FLAG_BAD_INDEX = 1, CSR_FLAG_UNSORTED = 2 };

struct CsrCtl {        // device control block, first 16 bytes of the scratch
    int flags;
    int max_degree;
    int pad[2];
};

__global__ __launch_bounds__(256) void k_csr_count(const long long *__restrict__ src,
                                                   const long long *__restrict__ dst, int E, int N,
                                                   int *__restrict__ cnt, CsrCtl *ctl) {
    int flags = 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        long long s = src[e], d = dst[e];
        if (s < 0 || s >= N || d < 0 || d >= N) {
            flags |= CSR_FLAG_BAD_INDEX;
        } else {
            atomicAdd(&cnt[s], 1);
            if (e > 0 && src[e - 1] > s) flags |= CSR_FLAG_UNSORTED;
        }
    }
    if (flags) atomicOr(&ctl->flags, flags);
}

__global__ __launch_bounds__(256) void k_csr_maxdeg(const int *__restrict__ rowptr, int N, CsrCtl *ctl) {
    int m = 0;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x)
        m = max(m, rowptr[v + 1] - rowptr[v]);
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&ctl->max_degree, m);
}

// SORTED: edge_index is already grouped by source, position e is its CSR slot.
template <bool SORTED>
__global__ __launch_bounds__(256) void k_csr_fill(const long long *__restrict__ src,
                                                  const long long *__restrict__ dst, int E,
                                                  const int *__restrict__ rowptr, int *__restrict__ cursor,
                                                  int *__restrict__ col) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        if (SORTED) {
            col[e] = (int)dst[e];
        } else {
            int s = (int)src[e];
            col[rowptr[s] + atomicAdd(&cursor[s], 1)] = (int)dst[e];
        }
    }
}

static size_t scan_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (int *)nullptr, (int *)nullptr, 0, n, rocprim::plus<int>());
    return bytes;
}

// ------------------------------------------------------------------------------------------------
// BFS
// ------------------------------------------------------------------------------------------------
struct BfsCtl {          // device control block at the start of the BFS scratch
    int last_active;     // highest level at which some (node, anchor) pair was newly reached
    int pad[3];
};

__device__ __forceinline__ u64 valid_mask(int K, int word) {
    int bits = K - 64 * word;
    return bits >= 64 ? ~0ull : (bits <= 0 ? 0ull : ((1ull << bits) - 1ull));
}

__global__ void k_bfs_seed(const long long *__restrict__ anchors, int K, int Wp, u64 *seen, u64 *front) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= K) return;
    size_t idx = (size_t)anchors[j] * Wp + (j >> 6);
    u64 bit = 1ull << (j & 63);
    atomicOr(&seen[idx], bit);       // duplicate anchors share a node: distinct bits of the same words
    atomicOr(&front[idx], bit);
}

// Commit the words a node gained at `level`: frontier for the next level, reachability, hop-bit planes.
__device__ __forceinline__ void commit(u64 fresh, size_t idx, u64 *front_next, u64 *seen, u64 *hop_planes,
                                       size_t plane_elems, int level) {
    front_next[idx] = fresh;
    if (fresh) {
        seen[idx] |= fresh;
        for (int b = 0, l = level; l; ++b, l >>= 1)
            if (l & 1) hop_planes[(size_t)b * plane_elems + idx] |= fresh;
    }
}

// One BFS level, bottom-up.  WT = words per tile (1, 2 or 4); blockIdx.y selects the tile of a node's words.
// A "group" of GROUP lanes owns one node: lane = slot * WT + word, S = GROUP / WT edge slots.
// Rows longer than BIG_DEG are deferred and then swept by the whole 256-thread block.
template <int WT, int GROUP>
__global__ __launch_bounds__(256) void k_bfs_pull(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                  int N, int K, int Wp,
                                                  const u64 *__restrict__ front_prev, u64 *__restrict__ front_next,
                                                  u64 *__restrict__ seen, u64 *__restrict__ hop_planes,
                                                  size_t plane_elems, int level, BfsCtl *ctl) {
    constexpr int S = GROUP / WT;            // edge slots per group
    constexpr int GROUPS = 256 / GROUP;      // groups per block
    constexpr int NPB = 64;                  // nodes per block
    constexpr int BIG_DEG = 32 * S;          // longer rows go to the block sweep
    constexpr int BS = 256 / WT;             // edge slots in the block sweep
    static_assert(NPB % GROUPS == 0 && GROUP <= 64 && GROUP % WT == 0, "shape");

    // The previous level reached nothing new: the BFS is over, every later launch is a no-op.
    if (__hip_atomic_load(&ctl->last_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < level - 1) return;

    __shared__ int big_rows[NPB];
    __shared__ int n_big;
    __shared__ u64 red[4 * WT];
    if (threadIdx.x == 0) n_big = 0;
    __syncthreads();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int gl = tid % GROUP;              // lane inside the group
    const int w = gl % WT;                   // word inside the tile
    const int slot = gl / WT;
    const int group = tid / GROUP;
    const int word = blockIdx.y * WT + w;    // word inside the node
    const u64 vmask = valid_mask(K, word);
    const int gshift = lane / GROUP * GROUP; // first lane of this group inside the wave
    const u64 gbits = (GROUP == 64) ? ~0ull : (((1ull << GROUP) - 1ull) << gshift);
    bool found = false;

    for (int i = 0; i < NPB / GROUPS; ++i) {
        const int v = blockIdx.x * NPB + i * GROUPS + group;
        const bool in_range = v < N;
        int beg = 0, end = 0;
        u64 unseen = 0;
        size_t idx = 0;
        if (in_range) {
            idx = (size_t)v * Wp + word;
            unseen = ~seen[idx] & vmask;
            beg = rowptr[v];
            end = rowptr[v + 1];
        }
        // Nodes every anchor of this tile has already reached never look at their edges again.
        const bool open = (__ballot(unseen != 0) & gbits) != 0;
        const bool big = open && (end - beg) > BIG_DEG;
        if (big && gl == 0) big_rows[atomicAdd(&n_big, 1)] = v;

        u64 acc = 0;
        if (open && !big) {
            int e = beg + slot;
            for (; e + 3 * S < end; e += 4 * S) {          // four independent gathers in flight
                int u0 = col[e], u1 = col[e + S], u2 = col[e + 2 * S], u3 = col[e + 3 * S];
                u64 f0 = front_prev[(size_t)u0 * Wp + word];
                u64 f1 = front_prev[(size_t)u1 * Wp + word];
                u64 f2 = front_prev[(size_t)u2 * Wp + word];
                u64 f3 = front_prev[(size_t)u3 * Wp + word];
                acc |= (f0 | f1) | (f2 | f3);
            }
            for (; e < end; e += S) acc |= front_prev[(size_t)col[e] * Wp + word];
        }
#pragma unroll
        for (int off = WT; off < GROUP; off <<= 1) acc |= __shfl_xor(acc, off);
        if (in_range && slot == 0 && !big) {
            u64 fresh = acc & unseen;
            commit(fresh, idx, front_next, seen, hop_planes, plane_elems, level);
            found |= fresh != 0;
        }
    }
    __syncthreads();

    // Block sweep of the deferred long rows: 256 / WT edge slots, four gathers in flight per lane.
    const int bw = tid % WT, bslot = tid / WT, bword = blockIdx.y * WT + bw;
    const u64 bmask = valid_mask(K, bword);
    const int nb = n_big;
    for (int r = 0; r < nb; ++r) {
        const int v = big_rows[r];
        const size_t idx = (size_t)v * Wp + bword;
        const u64 unseen = ~seen[idx] & bmask;
        const int beg = rowptr[v], end = rowptr[v + 1];
        u64 acc = 0;
        int e = beg + bslot;
        for (; e + 3 * BS < end; e += 4 * BS) {
            int u0 = col[e], u1 = col[e + BS], u2 = col[e + 2 * BS], u3 = col[e + 3 * BS];
            u64 f0 = front_prev[(size_t)u0 * Wp + bword];
            u64 f1 = front_prev[(size_t)u1 * Wp + bword];
            u64 f2 = front_prev[(size_t)u2 * Wp + bword];
            u64 f3 = front_prev[(size_t)u3 * Wp + bword];
            acc |= (f0 | f1) | (f2 | f3);
        }
        for (; e < end; e += BS) acc |= front_prev[(size_t)col[e] * Wp + bword];
#pragma unroll
        for (int off = WT; off < 64; off <<= 1) acc |= __shfl_xor(acc, off);
        if (lane < WT) red[(tid >> 6) * WT + lane] = acc;
        __syncthreads();
        if (tid < WT) {
            u64 fresh = (red[tid] | red[WT + tid] | red[2 * WT + tid] | red[3 * WT + tid]) & unseen;
            commit(fresh, idx, front_next, seen, hop_planes, plane_elems, level);
            found |= fresh != 0;
        }
        __syncthreads();
    }

    if (__any(found) && lane == 0)
        __hip_atomic_store(&ctl->last_active, level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                                                  int n_hop_bits, int N, int K, int Wp,
                                                  const float *__restrict__ x, int F, float *__restrict__ out,
                                                  long long out_cols, int c0) {
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

__global__ __launch_bounds__(256) void k_concat(const float *__restrict__ x, int N, int F, float *__restrict__ out,
                                                long long out_cols, bool vec) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int v = wave; v < N; v += nwaves) {
        const float *xrow = x + (size_t)v * F;
        float *orow = out + (size_t)v * out_cols;
        if (vec) {
            for (int q = lane; q < F / 4; q += 64)
                reinterpret_cast<float4 *>(orow)[q] = reinterpret_cast<const float4 *>(xrow)[q];
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
extern "C" size_t pope_csr_scratch_bytes(int64_t N, int64_t E) {
    (void)E;
    if (N < 0) return 0;
    // control block | cnt[N + 1] | rocPRIM scan temp
    return 256 + align_up((size_t)(N + 1) * sizeof(int), 256) + align_up(scan_temp_bytes((size_t)N + 1), 256);
}

extern "C" int pope_csr_build(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr, int32_t *col,
                              void *scratch, size_t scratch_bytes, int32_t *max_degree_host, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(N >= 0 && N < INT32_MAX && E >= 0 && E < INT32_MAX, "pope_csr_build: need 0 <= N, E < 2^31 (N=%lld E=%lld)",
                 (long long)N, (long long)E);
    POPE_REQUIRE(rowptr && col && scratch && (edge_index || E == 0), "pope_csr_build: null pointer");
    if (scratch_bytes < pope_csr_scratch_bytes(N, E)) {
        set_error("pope_csr_build: scratch %zu < %zu bytes", scratch_bytes, pope_csr_scratch_bytes(N, E));
        return POPE_ERR_WORKSPACE;
    }
    char *base = (char *)scratch;
    CsrCtl *ctl = (CsrCtl *)base;
    int *cnt = (int *)(base + 256);
    void *scan_tmp = base + 256 + align_up((size_t)(N + 1) * sizeof(int), 256);
    size_t scan_bytes = scan_temp_bytes((size_t)N + 1);

    POPE_HIP(hipMemsetAsync(base, 0, 256 + (size_t)(N + 1) * sizeof(int), stream));
    const long long *src = (const long long *)edge_index, *dst = src + E;
    if (E > 0)
        hipLaunchKernelGGL(k_csr_count, dim3(capped_grid(E, 256)), dim3(256), 0, stream, src, dst, (int)E, (int)N, cnt, ctl);
    POPE_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, cnt, rowptr, 0, (size_t)N + 1, rocprim::plus<int>(), stream));
    if (N > 0)
        hipLaunchKernelGGL(k_csr_maxdeg, dim3(capped_grid(N, 256)), dim3(256), 0, stream, rowptr, (int)N, ctl);
    CsrCtl h;
    POPE_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, stream));
    POPE_HIP(hipStreamSynchronize(stream));
    if (h.flags & CSR_FLAG_BAD_INDEX) {
        set_error("pope_csr_build: edge_index holds a node id outside [0, %lld)", (long long)N);
        return POPE_ERR_INDEX;
    }
    if (E > 0) {
        if (h.flags & CSR_FLAG_UNSORTED) {
            POPE_HIP(hipMemsetAsync(cnt, 0, (size_t)(N + 1) * sizeof(int), stream));
            hipLaunchKernelGGL(k_csr_fill<false>, dim3(capped_grid(E, 256)), dim3(256), 0, stream, src, dst, (int)E, rowptr, cnt, col);
        } else {
            hipLaunchKernelGGL(k_csr_fill<true>, dim3(capped_grid(E, 256)), dim3(256), 0, stream, src, dst, (int)E, rowptr, cnt, col);
        }
    }
    POPE_HIP(hipGetLastError());
    if (max_degree_host) *max_degree_host = h.max_degree;
    return POPE_OK;
}

extern "C" int32_t pope_words(int32_t K) { return K <= 0 ? 0 : words_for(K); }

extern "C" size_t pope_plane_bytes(int64_t N, int32_t K) {
    if (N < 0 || K <= 0) return 0;
    return (size_t)N * words_for(K) * sizeof(u64);
}

extern "C" size_t pope_bfs_scratch_bytes(int64_t N, int32_t K) {
    if (N < 0 || K <= 0) return 0;
    // control block | anchors[K] | two frontier planes
    return 256 + align_up((size_t)K * sizeof(long long), 256) + 2 * align_up(pope_plane_bytes(N, K), 256);
}

template <int WT, int GROUP>
static void launch_pull(int N, int K, int Wp, const int *rowptr, const int *col, const u64 *fp, u64 *fn, u64 *seen,
                        u64 *hop_planes, size_t plane_elems, int level, BfsCtl *ctl, hipStream_t stream) {
    dim3 grid((N + 63) / 64, Wp / WT);
    hipLaunchKernelGGL((k_bfs_pull<WT, GROUP>), grid, dim3(256), 0, stream, rowptr, col, N, K, Wp, fp, fn, seen,
                       hop_planes, plane_elems, level, ctl);
}

extern "C" int pope_geodesic_bfs(const int32_t *rowptr, const int32_t *col, int64_t N, int64_t E,
                                 const int64_t *anchors_host, int32_t K, uint64_t *planes_, int32_t plane_capacity,
                                 void *scratch, size_t scratch_bytes, int32_t *max_hop_host, int32_t *n_hop_bits_host,
                                 void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(N > 0 && N < INT32_MAX && E >= 0 && E < INT32_MAX, "pope_geodesic_bfs: need 0 < N < 2^31, 0 <= E < 2^31");
    POPE_REQUIRE(K > 0 && plane_capacity >= 1 && plane_capacity <= 31, "pope_geodesic_bfs: need K > 0 and 1 <= plane_capacity <= 31");
    POPE_REQUIRE(rowptr && col && anchors_host && planes_ && scratch, "pope_geodesic_bfs: null pointer");
    if (scratch_bytes < pope_bfs_scratch_bytes(N, K)) {
        set_error("pope_geodesic_bfs: scratch %zu < %zu bytes", scratch_bytes, pope_bfs_scratch_bytes(N, K));
        return POPE_ERR_WORKSPACE;
    }
    for (int j = 0; j < K; ++j)
        if (anchors_host[j] < 0 || anchors_host[j] >= N) {
            set_error("pope_geodesic_bfs: anchor %d = %lld outside [0, %lld)", j, (long long)anchors_host[j], (long long)N);
            return POPE_ERR_INDEX;
        }
    const int Wp = words_for(K);
    const size_t plane_elems = (size_t)N * Wp;
    const size_t plane_bytes = plane_elems * sizeof(u64);
    u64 *planes = (u64 *)planes_;
    u64 *seen = planes;
    u64 *hop_planes = planes + plane_elems;
    char *base = (char *)scratch;
    BfsCtl *ctl = (BfsCtl *)base;
    long long *anchors_dev = (long long *)(base + 256);
    u64 *front[2];
    front[0] = (u64 *)(base + 256 + align_up((size_t)K * sizeof(long long), 256));
    front[1] = (u64 *)((char *)front[0] + align_up(plane_bytes, 256));

    POPE_HIP(hipMemsetAsync(ctl, 0, 256, stream));
    POPE_HIP(hipMemcpyAsync(anchors_dev, anchors_host, (size_t)K * sizeof(long long), hipMemcpyHostToDevice, stream));
    POPE_HIP(hipMemsetAsync(seen, 0, plane_bytes, stream));
    POPE_HIP(hipMemsetAsync(front[0], 0, plane_bytes, stream));
    hipLaunchKernelGGL(k_bfs_seed, dim3((K + 255) / 256), dim3(256), 0, stream, anchors_dev, K, Wp, seen, front[0]);

    const long long level_limit = 1ll << plane_capacity;      // levels 1 .. limit-1 fit plane_capacity bits
    int level = 1, batch = 8, last_active = 0;
    for (;;) {
        const int stop = level + batch;                         // enqueue levels [level, stop)
        for (; level < stop; ++level) {
            if (level >= level_limit) break;
            if ((level & (level - 1)) == 0) {                    // first level with this hop bit: clear its plane
                int b = 0;
                while ((1 << b) < level) ++b;
                POPE_HIP(hipMemsetAsync(hop_planes + (size_t)b * plane_elems, 0, plane_bytes, stream));
            }
            const u64 *fp = front[(level - 1) & 1];
            u64 *fn = front[level & 1];
            if (Wp == 1)      launch_pull<1, 8>((int)N, K, Wp, rowptr, col, fp, fn, seen, hop_planes, plane_elems, level, ctl, stream);
            else if (Wp == 2) launch_pull<2, 16>((int)N, K, Wp, rowptr, col, fp, fn, seen, hop_planes, plane_elems, level, ctl, stream);
            else              launch_pull<4, 16>((int)N, K, Wp, rowptr, col, fp, fn, seen, hop_planes, plane_elems, level, ctl, stream);
        }
        POPE_HIP(hipMemcpyAsync(&last_active, &ctl->last_active, sizeof(int), hipMemcpyDeviceToHost, stream));
        POPE_HIP(hipStreamSynchronize(stream));
        if (last_active < level - 1) break;                      // some enqueued level found nothing: finished
        if (level >= level_limit) {
            // the last representable level still discovered nodes: deeper levels may exist
            set_error("pope_geodesic_bfs: hop count needs more than %d bits", plane_capacity);
            return POPE_ERR_HOP_OVERFLOW;
        }
        if (batch < 1024) batch *= 2;
    }
    POPE_HIP(hipGetLastError());
    int bits = 0;
    while ((1 << bits) <= last_active) ++bits;
    if (max_hop_host) *max_hop_host = last_active;
    if (n_hop_bits_host) *n_hop_bits_host = bits;
    return POPE_OK;
}

extern "C" int pope_geodesic_finalize(const uint64_t *planes, int32_t n_hop_bits, int64_t N, int32_t K,
                                      const float *x, int32_t F, float *out, int64_t out_cols, int32_t c0,
                                      void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(planes && out, "pope_geodesic_finalize: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && K > 0 && F >= 0 && c0 >= 0 && n_hop_bits >= 0 && n_hop_bits <= 31,
                 "pope_geodesic_finalize: bad size");
    POPE_REQUIRE(out_cols >= (int64_t)F + c0 + K, "pope_geodesic_finalize: out_cols %lld < F + c0 + K = %lld",
                 (long long)out_cols, (long long)F + c0 + K);
    const int Wp = words_for(K);
    const size_t plane_elems = (size_t)N * Wp;
    const bool vec = F % 4 == 0 && K % 4 == 0 && c0 % 4 == 0 && out_cols % 4 == 0 && aligned16(out) && (!x || aligned16(x));
    dim3 grid(capped_grid((size_t)N * 64, 256)), block(256);
    if (vec)
        hipLaunchKernelGGL(k_finalize<true>, grid, block, 0, stream, (const u64 *)planes, plane_elems, n_hop_bits, (int)N, K, Wp, x, F, out, (long long)out_cols, c0);
    else
        hipLaunchKernelGGL(k_finalize<false>, grid, block, 0, stream, (const u64 *)planes, plane_elems, n_hop_bits, (int)N, K, Wp, x, F, out, (long long)out_cols, c0);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
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

extern "C" int pope_concat(const float *x, int64_t N, int32_t F, float *out, int64_t out_cols, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(x && out, "pope_concat: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && F > 0 && out_cols >= F, "pope_concat: bad size");
    const bool vec = F % 4 == 0 && out_cols % 4 == 0 && aligned16(out) && aligned16(x);
    hipLaunchKernelGGL(k_concat, dim3(capped_grid((size_t)N * 64, 256)), dim3(256), 0, stream, x, (int)N, F, out,
                       (long long)out_cols, vec);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}
