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

struct CsrCtl {        // device control block, first bytes of the scratch
    int flags;
    int pad[3];
};

// Fast path, speculative: PyG stores edge_index grouped by source (coalesced), so slot e of the CSR is edge e
// and rowptr is where the source changes.  One streaming pass, no atomics, no scan.  If a pair is out of
// order the flag is raised and the counting path below redoes the build.
__global__ __launch_bounds__(256) void k_csr_sorted(const long long *__restrict__ src,
                                                    const long long *__restrict__ dst, int E, int N,
                                                    int *__restrict__ rowptr, int *__restrict__ col,
                                                    int *__restrict__ erow, CsrCtl *ctl) {
    int flags = 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const long long s = src[e], d = dst[e];
        if (s < 0 || s >= N || d < 0 || d >= N) {
            flags |= CSR_FLAG_BAD_INDEX;
            continue;
        }
        long long prev = e > 0 ? src[e - 1] : -1;
        if (prev > s) {
            flags |= CSR_FLAG_UNSORTED;
        } else if (prev >= -1 && prev < s) {
            for (long long r = prev + 1; r <= s; ++r) rowptr[r] = e;      // rows prev+1 .. s start here
        }
        if (e == E - 1)
            for (long long r = s + 1; r <= N; ++r) rowptr[r] = E;
        col[e] = (int)d;
        erow[e] = (int)s;
    }
    if (flags) atomicOr(&ctl->flags, flags);
}

__global__ __launch_bounds__(256) void k_fill_int(int *p, int n, int value) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = value;
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

template <int WT> struct Words { u64 w[WT]; };

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

template <int WT>
__device__ __forceinline__ void store_words(u64 *__restrict__ p, const Words<WT> &r) {
    if constexpr (WT == 1) {
        p[0] = r.w[0];
    } else {
#pragma unroll
        for (int i = 0; i < WT; i += 2) *reinterpret_cast<ulonglong2 *>(p + i) = make_ulonglong2(r.w[i], r.w[i + 1]);
    }
}

// One BFS level, phase 1 ("expand"), bottom-up and EDGE-parallel: lane = one CSR slot e = (v -> u).
//   cand = front[u] & ~seen[v]           anchors that reach v through u and had not reached v before
// Slots are sorted by v, so a row is a run of consecutive lanes: a segmented OR-scan over the wave
// combines each run.  Work per wave is 64 edges whatever the degree distribution (no long rows, no
// dependent pointer chase: erow/col are coalesced streams), and there are NO atomics:
//   * the run that contains a row's FIRST slot is the row's "owner piece" and is stored to acc[v];
//   * a run that continues a row begun in an earlier chunk is a "continuation piece": there is at most
//     one per 64-slot chunk (its first run) and it is stored to cont[chunk]; k_bfs_update ORs the
//     continuation pieces of the few rows that span chunks.
// WT = words per tile (1, 2 or 4); blockIdx.y selects the tile of a node's W words.
template <int WT>
__global__ __launch_bounds__(256) void k_bfs_expand(const int *__restrict__ erow, const int *__restrict__ col,
                                                    int E, int Wp, const u64 *__restrict__ front,
                                                    const u64 *__restrict__ seen, u64 *__restrict__ acc,
                                                    u64 *__restrict__ cont, int level, BfsCtl *ctl) {
    // The previous level reached nothing new: the BFS is over, every later launch is a no-op.
    if (__hip_atomic_load(&ctl->last_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < level - 1) return;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nchunks = (E + 63) >> 6;
    const int woff = blockIdx.y * WT;
    bool found = false;
    for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
        const int e = chunk * 64 + lane;
        const bool valid = e < E;
        int v = -1, u = 0;
        if (valid) {
            v = erow[e];
            u = col[e];
        }
        const int v0 = __shfl(v, 0);                                       // row of the chunk's first slot
        const bool head_continues = chunk > 0 && erow[chunk * 64 - 1] == v0;
        Words<WT> c;
#pragma unroll
        for (int i = 0; i < WT; ++i) c.w[i] = 0;
        if (valid) {
            const Words<WT> f = load_words<WT>(front + (size_t)u * Wp + woff);
            const Words<WT> s = load_words<WT>(seen + (size_t)v * Wp + woff);
#pragma unroll
            for (int i = 0; i < WT; ++i) c.w[i] = f.w[i] & ~s.w[i];
        }
        u64 any = 0;
#pragma unroll
        for (int i = 0; i < WT; ++i) any |= c.w[i];
        u64 *cont_c = cont + (size_t)chunk * Wp + woff;
        Words<WT> zero;
#pragma unroll
        for (int i = 0; i < WT; ++i) zero.w[i] = 0;
        if (!__any(any != 0)) {                                            // nothing new through these 64 edges
            if (lane == 0) store_words<WT>(cont_c, zero);
            continue;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {                                 // segmented inclusive OR-scan keyed by v
            const int pv = __shfl_up(v, d);
            const bool take = lane >= d && pv == v;
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                const u64 pc = __shfl_up(c.w[i], d);
                if (take) c.w[i] |= pc;
            }
        }
        const int nv = __shfl_down(v, 1);
        const bool tail = valid && (lane == 63 || nv != v);                // last lane of its run
        if (tail) {
            u64 nz = 0;
#pragma unroll
            for (int i = 0; i < WT; ++i) nz |= c.w[i];
            if (head_continues && v == v0) {
                store_words<WT>(cont_c, c);                                // continuation piece of a row begun earlier
            } else {
                if (nz) store_words<WT>(acc + (size_t)v * Wp + woff, c);   // owner piece; acc was zero: plain store
                if (v == v0) store_words<WT>(cont_c, zero);                // this chunk continues nothing
            }
            found |= nz != 0;
        }
    }
    // Raise the "this level reached something" flag.  Same-address device-scope stores serialise at the memory
    // side (tens of ns each), so a wave stores only while the flag still shows an older level.
    if (__any(found) && lane == 0 &&
        __hip_atomic_load(&ctl->last_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != level)
        __hip_atomic_store(&ctl->last_active, level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Phase 2 ("update"): one thread per (node, word).  acc holds the owner piece; rows that span several
// 64-slot chunks add their continuation pieces.  The result is exactly the set of newly reached anchors
// (expand masked with ~seen, which nobody modified meanwhile), i.e. the next frontier: commit it to the
// reachability plane and the hop-bit planes, and clear the old frontier so it can be the next accumulator.
__global__ __launch_bounds__(256) void k_bfs_update(const int *__restrict__ rowptr, int Wp, size_t plane_elems,
                                                    u64 *__restrict__ fresh_front, u64 *__restrict__ old_front,
                                                    const u64 *__restrict__ cont, u64 *__restrict__ seen,
                                                    u64 *__restrict__ hop_planes, int level, BfsCtl *ctl) {
    if (__hip_atomic_load(&ctl->last_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < level) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane_elems; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i / Wp), w = (int)(i % Wp);
        old_front[i] = 0;
        u64 fresh = fresh_front[i];
        const int p0 = rowptr[v], p1 = rowptr[v + 1];
        if (p1 > p0) {
            const int c0 = p0 >> 6, c1 = (p1 - 1) >> 6;
            if (c1 > c0) {
                u64 extra = 0;
                for (int c = c0 + 1; c <= c1; ++c) extra |= cont[(size_t)c * Wp + w];
                if (extra & ~fresh) {
                    fresh |= extra;
                    fresh_front[i] = fresh;
                }
            }
        }
        if (fresh) {
            seen[i] |= fresh;
            for (int b = 0, l = level; l; ++b, l >>= 1)
                if (l & 1) hop_planes[(size_t)b * plane_elems + i] |= fresh;
        }
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
    // control block | cnt[N + 1] | rocPRIM scan temp      (the last two only used for unsorted edge lists)
    return 256 + align_up((size_t)(N + 1) * sizeof(int), 256) + align_up(scan_temp_bytes((size_t)N + 1), 256);
}

extern "C" int pope_csr_build(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr, int32_t *col,
                              int32_t *erow, void *scratch, size_t scratch_bytes, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(N >= 0 && N < INT32_MAX && E >= 0 && E < INT32_MAX, "pope_csr_build: need 0 <= N, E < 2^31 (N=%lld E=%lld)",
                 (long long)N, (long long)E);
    POPE_REQUIRE(rowptr && scratch && ((edge_index && col && erow) || E == 0), "pope_csr_build: null pointer");
    if (scratch_bytes < pope_csr_scratch_bytes(N, E)) {
        set_error("pope_csr_build: scratch %zu < %zu bytes", scratch_bytes, pope_csr_scratch_bytes(N, E));
        return POPE_ERR_WORKSPACE;
    }
    char *base = (char *)scratch;
    CsrCtl *ctl = (CsrCtl *)base;
    int *cnt = (int *)(base + 256);
    void *scan_tmp = base + 256 + align_up((size_t)(N + 1) * sizeof(int), 256);
    const long long *src = (const long long *)edge_index, *dst = src + E;

    if (E == 0) {
        POPE_HIP(hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * sizeof(int), stream));
        return POPE_OK;
    }
    POPE_HIP(hipMemsetAsync(ctl, 0, 256, stream));
    hipLaunchKernelGGL(k_csr_sorted, dim3(capped_grid(E, 256)), dim3(256), 0, stream, src, dst, (int)E, (int)N, rowptr, col, erow, ctl);
    CsrCtl h;
    POPE_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, stream));
    POPE_HIP(hipStreamSynchronize(stream));
    if (h.flags & CSR_FLAG_BAD_INDEX) {
        set_error("pope_csr_build: edge_index holds a node id outside [0, %lld)", (long long)N);
        return POPE_ERR_INDEX;
    }
    if (h.flags & CSR_FLAG_UNSORTED) {
        size_t scan_bytes = scan_temp_bytes((size_t)N + 1);
        POPE_HIP(hipMemsetAsync(cnt, 0, (size_t)(N + 1) * sizeof(int), stream));
        hipLaunchKernelGGL(k_csr_count, dim3(capped_grid(E, 256)), dim3(256), 0, stream, src, (int)E, cnt);
        POPE_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, cnt, rowptr, 0, (size_t)N + 1, rocprim::plus<int>(), stream));
        POPE_HIP(hipMemsetAsync(cnt, 0, (size_t)(N + 1) * sizeof(int), stream));
        hipLaunchKernelGGL(k_csr_scatter, dim3(capped_grid(E, 256)), dim3(256), 0, stream, src, dst, (int)E, rowptr, cnt, col, erow);
    }
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int32_t pope_words(int32_t K) { return K <= 0 ? 0 : words_for(K); }

extern "C" size_t pope_plane_bytes(int64_t N, int32_t K) {
    if (N < 0 || K <= 0) return 0;
    return (size_t)N * words_for(K) * sizeof(u64);
}

static size_t cont_bytes(int64_t E, int32_t K) { return align_up((size_t)((E + 63) / 64 + 1) * words_for(K) * sizeof(u64), 256); }

extern "C" size_t pope_bfs_scratch_bytes(int64_t N, int64_t E, int32_t K) {
    if (N < 0 || E < 0 || K <= 0) return 0;
    // control block | anchors[K] | two frontier planes | continuation pieces (one per 64 CSR slots)
    return 256 + align_up((size_t)K * sizeof(long long), 256) + 2 * align_up(pope_plane_bytes(N, K), 256) + cont_bytes(E, K);
}

template <int WT>
static void launch_expand(int E, int Wp, const int *erow, const int *col, const u64 *front, const u64 *seen, u64 *acc,
                          u64 *cont, int level, BfsCtl *ctl, hipStream_t stream) {
    dim3 grid(capped_grid((size_t)E, 256, 256u * 32u), Wp / WT);
    hipLaunchKernelGGL((k_bfs_expand<WT>), grid, dim3(256), 0, stream, erow, col, E, Wp, front, seen, acc, cont, level, ctl);
}

extern "C" int pope_geodesic_bfs(const int32_t *rowptr, const int32_t *col, const int32_t *erow, int64_t N, int64_t E,
                                 const int64_t *anchors_host, int32_t K, uint64_t *planes_, int32_t plane_capacity,
                                 void *scratch, size_t scratch_bytes, int32_t *max_hop_host, int32_t *n_hop_bits_host,
                                 void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(N > 0 && N < INT32_MAX && E >= 0 && E < INT32_MAX, "pope_geodesic_bfs: need 0 < N < 2^31, 0 <= E < 2^31");
    POPE_REQUIRE(K > 0 && plane_capacity >= 1 && plane_capacity <= 31, "pope_geodesic_bfs: need K > 0 and 1 <= plane_capacity <= 31");
    POPE_REQUIRE(rowptr && ((erow && col) || E == 0) && anchors_host && planes_ && scratch, "pope_geodesic_bfs: null pointer");
    if (scratch_bytes < pope_bfs_scratch_bytes(N, E, K)) {
        set_error("pope_geodesic_bfs: scratch %zu < %zu bytes", scratch_bytes, pope_bfs_scratch_bytes(N, E, K));
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
    u64 *cont = (u64 *)((char *)front[1] + align_up(plane_bytes, 256));

    POPE_HIP(hipMemsetAsync(ctl, 0, 256, stream));
    POPE_HIP(hipMemcpyAsync(anchors_dev, anchors_host, (size_t)K * sizeof(long long), hipMemcpyHostToDevice, stream));
    POPE_HIP(hipMemsetAsync(seen, 0, plane_bytes, stream));
    POPE_HIP(hipMemsetAsync(front[0], 0, 2 * align_up(plane_bytes, 256), stream));      // both frontier buffers
    hipLaunchKernelGGL(k_bfs_seed, dim3((K + 255) / 256), dim3(256), 0, stream, anchors_dev, K, Wp, seen, front[0]);

    const long long level_limit = 1ll << plane_capacity;      // levels 1 .. limit-1 fit plane_capacity bits
    int level = 1, batch = 8, last_active = 0;
    for (;;) {
        const int stop = level + batch;                         // enqueue levels [level, stop)
        for (; level < stop; ++level) {
            if (level >= level_limit || E == 0) break;
            if ((level & (level - 1)) == 0) {                    // first level with this hop bit: clear its plane
                int b = 0;
                while ((1 << b) < level) ++b;
                POPE_HIP(hipMemsetAsync(hop_planes + (size_t)b * plane_elems, 0, plane_bytes, stream));
            }
            u64 *prev = front[(level - 1) & 1];               // frontier of level - 1
            u64 *next = front[level & 1];                        // all zero: accumulates, becomes the new frontier
            if (Wp == 1)      launch_expand<1>((int)E, Wp, erow, col, prev, seen, next, cont, level, ctl, stream);
            else if (Wp == 2) launch_expand<2>((int)E, Wp, erow, col, prev, seen, next, cont, level, ctl, stream);
            else              launch_expand<4>((int)E, Wp, erow, col, prev, seen, next, cont, level, ctl, stream);
            hipLaunchKernelGGL(k_bfs_update, dim3(capped_grid(plane_elems, 256)), dim3(256), 0, stream, rowptr, Wp,
                               plane_elems, next, prev, cont, seen, hop_planes, level, ctl);
        }
        POPE_HIP(hipMemcpyAsync(&last_active, &ctl->last_active, sizeof(int), hipMemcpyDeviceToHost, stream));
        POPE_HIP(hipStreamSynchronize(stream));
        if (last_active < level - 1 || E == 0) break;           // some enqueued level found nothing: finished
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
