// GPU fan-out neighbour sampling for the GraphSAGE mini-batches.
//
// Replaces the host side of /root/reference/main.py:100-116: PyG NeighborSampler(adj_t, sizes=[25, 10]) ->
// torch_sparse.sample_adj in DataLoader worker processes, followed by the host gather data.x[n_id] (main.py:118-123).
// One hop = for every target node keep all its neighbours if it has at most `fanout`, else `fanout` DISTINCT ones,
// then relabel: the new node list is the targets (same order) followed by the newly met nodes in order of first
// appearance, and the block's CSR uses those local ids -- the layout sage_conv_forward consumes.
//
// Sampling without replacement needs no per-row state here: slot j of a row with d > fanout neighbours takes
// neighbour perm(j), where perm is a keyed pseudo-random PERMUTATION of [0, d) (4-round Feistel network on the next
// even power of two, cycle-walked into range); the first `fanout` outputs of a permutation are distinct by
// construction.  The key mixes the caller's seed, the hop and the node id, so the sample is a pure function of its
// arguments: the CPU checker used by the tests restates it and the comparison is bit for bit.  (The reference's own random stream --
// torch_sparse's C++ RNG inside DataLoader workers -- is not reproducible outside that stack: SURVEY.md §7 trap 9.)
#include <cstring>

#include "common.h"

namespace pope {

__host__ __device__ __forceinline__ unsigned mix32(unsigned h) {
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h ^= h >> 12; h *= 0x297A2D39u;
    h ^= h >> 15;
    return h;
}

__host__ __device__ __forceinline__ unsigned row_key(unsigned long long seed, int hop, int node) {
    return mix32((unsigned)seed ^ mix32((unsigned)(seed >> 32) + 0x9E3779B1u * (unsigned)(hop + 1)) ^ mix32((unsigned)node * 0x85EBCA6Bu + 0x165667B1u));
}

// Keyed permutation of [0, d): Feistel on 2 * hb bits (the smallest even width covering d), cycle-walked.
__host__ __device__ __forceinline__ int feistel_perm(int i, int d, unsigned key) {
    int bits = 2;
    while ((1 << bits) < d) ++bits;
    const int hb = (bits + 1) >> 1;
    const unsigned mask = (1u << hb) - 1u;
    unsigned x = (unsigned)i;
    do {
        unsigned l = x >> hb, r = x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const unsigned f = mix32(r * 0x9E3779B1u + key + (unsigned)round * 0x85EBCA6Bu) & mask;
            const unsigned nl = r;
            r = l ^ f;
            l = nl;
        }
        x = (l << hb) | r;
    } while (x >= (unsigned)d);
    return (int)x;
}

// t_dev != nullptr (device-extent mode, sage_sample_batch_device): the launch covers the CAPACITY T and the true number
// of targets is read on the device; slots past it count zero neighbours, so the scan over the capacity still ends in nnz.
__device__ __forceinline__ int true_count(const int *t_dev, int cap) {
    if (!t_dev) return cap;
    const int v = *t_dev;
    return v < cap ? v : cap;
}

// map[g] = smallest "position key" at which node g occurs: targets occupy keys [0, T) (k_sample_count_scan), sampled slot p key T + p.
__global__ __launch_bounds__(256) void k_sample_pick(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                     const long long *__restrict__ targets, int T_cap, int fanout,
                                                     unsigned long long seed, int hop, const int *__restrict__ out_rowptr,
                                                     int *__restrict__ picked, int *__restrict__ map, const int *__restrict__ t_dev,
                                                     const unsigned long long *__restrict__ seed_dev, int *__restrict__ rank) {
    const int T = true_count(t_dev, T_cap);
    if (seed_dev) seed += *seed_dev;                               // replayed launches (HIP graph): the seed lives on the device
    const int stride = fanout < 0 ? 1 : fanout;
    const long long total = (long long)T * stride;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(q / stride), j = (int)(q % stride);
        const int beg_out = out_rowptr[i], c = out_rowptr[i + 1] - beg_out;
        if (j >= c) continue;
        const int g = (int)targets[i];
        const int beg = rowptr[g], d = rowptr[g + 1] - beg;
        const int pick = d <= c ? j : feistel_perm(j, d, row_key(seed, hop, g));
        const int u = col[beg + pick];
        const int p = beg_out + j;
        picked[p] = u;
        rank[p] = -1;                      // "not ranked yet": k_sample_rank_relabel's readers wait on it
        atomicMin(&map[u], T + p);
    }
}

// (Measured and rejected, round 3: the count and flag passes as rocprim transform iterators feeding rocprim's scans instead
//  of launches of their own -- two launches less per hop, but the scans went from ~5 us to 19-21 us each: their per-item
//  functor loads (rowptr[targets[i]], map[picked[p]]) run at ITEMS_PER_THREAD-fold lower parallelism than a flat kernel.
//  The fused kernels further down do their own chained scan instead.)
// fanout < 0 ("all neighbours"): rows are copied whole, one thread per output slot.
__global__ __launch_bounds__(256) void k_sample_all(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                    const long long *__restrict__ targets, int T,
                                                    const int *__restrict__ out_rowptr, int cap, int *__restrict__ picked,
                                                    int *__restrict__ map, int *__restrict__ rank) {
    const int i = blockIdx.x;                                      // one block per target row
    if (i >= T) return;
    const int g = (int)targets[i];
    const int beg = rowptr[g], beg_out = out_rowptr[i], c = out_rowptr[i + 1] - beg_out;
    for (int j = threadIdx.x; j < c && beg_out + j < cap; j += blockDim.x) {     // cap too small: reported by the host below
        const int u = col[beg + j];
        picked[beg_out + j] = u;
        rank[beg_out + j] = -1;
        atomicMin(&map[u], T + beg_out + j);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused passes (round 3): count + scan and flag + scan each in ONE launch -- nine launches per hop became three, and in a
// captured step a launch of a 2 us kernel still costs ~4.5 us.  The scans are chained inside the launch: a block draws a
// ticket (so a block only ever waits for blocks that have started), publishes the sum of its 1 024 items in a status word
// and adds up the status words of ALL lower tickets -- they depend on nothing but their blocks' own items, so there is no
// serial chain (at most ~400 words per block).  Status words and tickets are zeroed one launch ahead (k_sample_begin at
// the start of a batch, then hop h's first kernel for hop h + 1: two regions alternate, like the two position-key maps).
// ------------------------------------------------------------------------------------------------------------------
constexpr int CHAIN_ITEMS = 1024;               // items per block: 256 threads x 4

struct Chain {
    int *ticket;                                // one counter per kernel
    unsigned long long *status;                 // (1 << 32) | block sum, 0 = not there yet
};

__device__ __forceinline__ int chain_ticket(const Chain &c) {
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(c.ticket, 1);
    __syncthreads();
    return s_ticket;
}

// Exclusive scan of `v` over the 256 threads of the block; *total = the block's sum.
__device__ __forceinline__ int block_exclusive_scan(int v, int *total) {
    __shared__ int s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int t = s_wave[w];
        if (w < wave) before += t;
        all += t;
    }
    __syncthreads();                            // s_wave is reused by the caller's next scan
    *total = all;
    return before + inc - v;
}

// Publishes this block's sum under its ticket and returns the sum of all lower tickets' (every thread gets it).
__device__ __forceinline__ int chain_prefix(const Chain &c, int ticket, int block_sum) {
    __shared__ int s_part[4];
    if (threadIdx.x == 0)
        __hip_atomic_store(&c.status[ticket], (1ull << 32) | (unsigned)block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int sum = 0;
    for (int j = threadIdx.x; j < ticket; j += blockDim.x) {
        unsigned long long w;
        do {
            w = __hip_atomic_load(&c.status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((w >> 32) == 0);               // ticket j was drawn before ours: that block is running or done
        sum += (int)(unsigned)w;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    const int r = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    return r;
}

// map[:] = 0x7F7F7F7F (larger than any position key) and words[:] = 0; grid-stride, any grid.
__device__ __forceinline__ void clear_region(int *__restrict__ map, int N, unsigned long long *__restrict__ words, int n_words) {
    const int n4 = N >> 2;
    const int4 v = make_int4(0x7F7F7F7F, 0x7F7F7F7F, 0x7F7F7F7F, 0x7F7F7F7F);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) reinterpret_cast<int4 *>(map)[i] = v;
    if (blockIdx.x == 0 && (int)threadIdx.x < (N & 3)) map[(n4 << 2) + threadIdx.x] = 0x7F7F7F7F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) words[i] = 0ull;
}

// Start of a batch: the first hop's map and chain state.
__global__ __launch_bounds__(256) void k_sample_begin(int *__restrict__ map, int N, unsigned long long *__restrict__ words, int n_words) {
    clear_region(map, N, words, n_words);
}

// The per-target counts and their scan: out_rowptr[i] = number of sampled edges of the targets before i, i in [0, T_cap]
// (targets past the true count contribute nothing, so every entry from T on holds nnz).  Also prepares the NEXT hop's map
// and chain state (next_map / next_words: buffers this hop does not touch).
__global__ __launch_bounds__(256) void k_sample_count_scan(const int *__restrict__ rowptr, const long long *__restrict__ targets, int T_cap,
                                                           int fanout, int *__restrict__ out_rowptr, int *__restrict__ map,
                                                           long long *__restrict__ out_n_id, const int *__restrict__ t_dev,
                                                           int *__restrict__ dims, Chain chain, int *__restrict__ next_map, int N,
                                                           unsigned long long *__restrict__ next_words, int n_next_words,
                                                           const long long *__restrict__ first_dev, const long long *__restrict__ labels,
                                                           long long *__restrict__ y_out) {
    const int T = true_count(t_dev, T_cap);
    if (first_dev) targets += *first_dev;          // epoch mode: this batch's targets start at a device cursor into the epoch's order
    const int t = chain_ticket(chain);
    if (t == 0 && threadIdx.x == 0 && dims) dims[0] = T;           // n_dst of this block
    const int base = t * CHAIN_ITEMS + (int)threadIdx.x * 4;
    int c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + k;
        c[k] = 0;
        if (i < T) {
            const long long g = targets[i];
            const int d = rowptr[g + 1] - rowptr[g];
            c[k] = (fanout < 0 || d <= fanout) ? d : fanout;
            map[g] = i;                    // position key of a target = its index (target lists hold distinct nodes)
            out_n_id[i] = g;               // n_id starts with the targets, in order
            if (labels) y_out[i] = labels[g];   // main.py:122 y = data.y[n_id[:batch_size]]
        }
    }
    int block_sum;
    const int ex = block_exclusive_scan(c[0] + c[1] + c[2] + c[3], &block_sum);
    int run = chain_prefix(chain, t, block_sum) + ex;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k <= T_cap) out_rowptr[base + k] = run;
        run += c[k];
    }
    if (next_map) clear_region(next_map, N, next_words, n_next_words);
}

// The first-occurrence flags, their scan AND the local ids, one launch.  A node's position key tells everything: a target
// keeps its index; a new node first seen at slot p0 = key - T gets T + rank[p0] (rank = exclusive scan of the flags), and the
// thread that owns p0 also writes it into n_id.  rank[p0] may belong to another block: p0 <= p, so that block drew a lower
// ticket and is running or done, and the reader waits on the word itself (the pick kernel left -1 there).  Blocks past nnz
// have nothing to do -- and nobody waits for them.  The thread that owns slot nnz reports the two counts.
__global__ __launch_bounds__(256) void k_sample_rank_relabel(const int *__restrict__ picked, const int *__restrict__ out_rowptr, int T_cap,
                                                             int cap, const int *__restrict__ map, int *__restrict__ rank,
                                                             int *__restrict__ out_col, long long *__restrict__ n_id,
                                                             long long *__restrict__ report, const int *__restrict__ t_dev,
                                                             int *__restrict__ dims, Chain chain) {
    const int T = true_count(t_dev, T_cap);
    const int nnz_true = out_rowptr[T];
    const int nnz = min(nnz_true, cap);                            // (more only in "all neighbours" mode with an undersized buffer: reported below)
    const int t = chain_ticket(chain);
    if (t * CHAIN_ITEMS > nnz) return;
    const int base = t * CHAIN_ITEMS + (int)threadIdx.x * 4;
    int f[4], u[4], key[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = base + k;
        u[k] = p < nnz ? picked[p] : 0;
        key[k] = p < nnz ? map[u[k]] : 0;
        f[k] = (p < nnz && key[k] == T + p) ? 1 : 0;
    }
    int block_sum;
    const int ex = block_exclusive_scan(f[0] + f[1] + f[2] + f[3], &block_sum);
    int run = chain_prefix(chain, t, block_sum) + ex;
    int mine[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = base + k;
        mine[k] = run;
        if (p < nnz) __hip_atomic_store(&rank[p], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == nnz) {                                            // run = number of new nodes
            if (dims) {                                            // device-extent mode: the sizes stay on the device
                dims[1] = T + run;                                 // n_src: targets + distinct new nodes
                dims[2] = nnz_true;
                dims[3] = 0;
            }
            if (report) {
                report[0] = nnz_true;
                report[1] = T + run;
                __threadfence_system();                            // `report` is pinned host memory: no copy kernel
            }
        }
        run += f[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = base + k;
        if (p >= nnz) continue;
        if (key[k] < T) {
            out_col[p] = key[k];
        } else {
            const int p0 = key[k] - T;
            int r = mine[k];
            if (p0 != p) {
                do {
                    r = __hip_atomic_load(&rank[p0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while (r < 0);
            }
            out_col[p] = T + r;
            if (p0 == p) n_id[T + r] = u[k];
        }
    }
}

// Per (device, host thread) pinned report slot, created on first use (a worker thread may sample beside the main thread).
static int pinned_report(long long **host, long long **dev) {
    static thread_local long long *h[64] = {nullptr}, *d[64] = {nullptr};
    int id = 0;
    POPE_HIP(hipGetDevice(&id));
    POPE_REQUIRE(id >= 0 && id < 64, "device index %d out of range", id);
    if (!h[id]) {
        POPE_HIP(hipHostMalloc((void **)&h[id], 64, hipHostMallocMapped));
        POPE_HIP(hipHostGetDevicePointer((void **)&d[id], h[id], 0));
    }
    *host = h[id];
    *dev = d[id];
    return POPE_OK;
}

struct SampleLayout {
    size_t picked, rank, map, newid, report, state, total;   // map / newid: the position-key maps of even / odd hops
    int cap, nb_count, nb_flag, state_words;     // chained-scan state: two regions of state_words 64-bit words each
};

static SampleLayout sample_layout(int64_t N, int64_t T, int64_t cap) {
    SampleLayout L;
    L.cap = (int)cap;
    size_t o = 0;
    L.picked = o; o += align_up((size_t)(cap + 1) * 4, 256);
    L.rank = o;   o += align_up((size_t)(cap + 1) * 4, 256);
    L.map = o;    o += align_up((size_t)N * 4, 256);
    L.newid = o;  o += align_up((size_t)N * 4, 256);
    L.report = o; o += 256;
    L.nb_count = (int)((T + 1 + CHAIN_ITEMS - 1) / CHAIN_ITEMS);
    L.nb_flag = (int)((cap + 1 + CHAIN_ITEMS - 1) / CHAIN_ITEMS);
    L.state_words = 2 + L.nb_count + L.nb_flag;                  // [ticket of count_scan, ticket of flag_rank, status words ...]
    L.state = o;  o += 2 * align_up((size_t)L.state_words * 8, 256);
    L.total = o;
    return L;
}

}  // namespace pope

using namespace pope;

extern "C" size_t sage_sample_scratch_bytes(int64_t N, int64_t n_targets, int64_t nnz_capacity) {
    if (N <= 0 || n_targets <= 0 || nnz_capacity < 0) return 0;
    return sample_layout(N, n_targets, nnz_capacity).total;
}

// One hop, enqueued without waiting.  t_dev / dims / seed_dev: device-extent mode (see sage_sample_batch_device); report:
// pinned host words for the host-sized mode.
// `L` is the layout of the scratch buffer (sized for the largest hop of the call); `first` = this is the first hop enqueued by
// the call (its map and chain state are cleared by a launch of their own), `prepare_next` = another hop follows (this hop's
// first kernel clears that hop's map and state, which live in the other of the two regions).
static int enqueue_hop(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *targets, int64_t n_targets, int32_t fanout,
                       uint64_t seed, int32_t hop, int32_t *out_rowptr, int32_t *out_col, int64_t nnz_capacity, int64_t *out_n_id,
                       void *scratch, size_t scratch_bytes, const SampleLayout &L, bool first, bool prepare_next, const int *t_dev, int *dims,
                       const unsigned long long *seed_dev, long long *report_dev, hipStream_t stream, const char *who,
                       const long long *first_dev = nullptr, const long long *labels = nullptr, long long *y_out = nullptr) {
    if (scratch_bytes < L.total) {
        set_error("%s: scratch %zu < %zu bytes", who, scratch_bytes, L.total);
        return POPE_ERR_WORKSPACE;
    }
    char *base = (char *)scratch;
    int *picked = (int *)(base + L.picked), *rank = (int *)(base + L.rank);
    const size_t region_bytes = align_up((size_t)L.state_words * 8, 256);
    const int r = hop & 1;
    int *map = (int *)(base + (r ? L.newid : L.map)), *other_map = (int *)(base + (r ? L.map : L.newid));
    unsigned long long *words = (unsigned long long *)(base + L.state + r * region_bytes);
    unsigned long long *other_words = (unsigned long long *)(base + L.state + (1 - r) * region_bytes);
    const Chain count_chain{(int *)words, words + 2}, flag_chain{(int *)(words + 1), words + 2 + L.nb_count};
    const int T = (int)n_targets, cap = (int)nnz_capacity;
    if ((T + 1 + CHAIN_ITEMS - 1) / CHAIN_ITEMS > L.nb_count || (cap + 1 + CHAIN_ITEMS - 1) / CHAIN_ITEMS > L.nb_flag) {
        set_error("%s: hop %d is larger than the scratch layout", who, hop);
        return POPE_ERR_WORKSPACE;
    }

    if (first)
        hipLaunchKernelGGL(k_sample_begin, dim3(capped_grid((size_t)N / 4 + 1, 256)), dim3(256), 0, stream, map, (int)N, words, L.state_words);
    hipLaunchKernelGGL(k_sample_count_scan, dim3((T + 1 + CHAIN_ITEMS - 1) / CHAIN_ITEMS), dim3(256), 0, stream, rowptr, (const long long *)targets, T,
                       fanout, out_rowptr, map, (long long *)out_n_id, t_dev, dims, count_chain, prepare_next ? other_map : nullptr, (int)N,
                       other_words, L.state_words, first_dev, labels, y_out);
    targets = out_n_id;                          // the later kernels read the targets where the first one wrote them: n_id[0 .. T)
    if (fanout < 0) {
        // all neighbours: the total is only known on the device; the caller sized nnz_capacity for it
        hipLaunchKernelGGL(k_sample_all, dim3(T), dim3(256), 0, stream, rowptr, col, (const long long *)targets, T, out_rowptr, cap, picked, map, rank);
    } else {
        hipLaunchKernelGGL(k_sample_pick, dim3(capped_grid((size_t)T * fanout, 256)), dim3(256), 0, stream, rowptr, col,
                           (const long long *)targets, T, fanout, (unsigned long long)seed, hop, out_rowptr, picked, map, t_dev, seed_dev, rank);
    }
    hipLaunchKernelGGL(k_sample_rank_relabel, dim3((cap + 1 + CHAIN_ITEMS - 1) / CHAIN_ITEMS), dim3(256), 0, stream, picked, out_rowptr, T, cap, map, rank,
                       out_col, (long long *)out_n_id, report_dev, t_dev, dims, flag_chain);
    POPE_HIP(hipGetLastError());
    return POPE_OK;
}

extern "C" int sage_sample_hop(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *targets, int64_t n_targets,
                               int32_t fanout, uint64_t seed, int32_t hop, int32_t *out_rowptr, int32_t *out_col,
                               int64_t nnz_capacity, int64_t *out_n_id, int64_t *nnz_host, int64_t *n_src_host, void *scratch,
                               size_t scratch_bytes, void *stream_) {
    clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    POPE_REQUIRE(rowptr && col && targets && out_rowptr && out_col && out_n_id && scratch && nnz_host && n_src_host,
                 "sage_sample_hop: null pointer");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && n_targets > 0 && n_targets < INT32_MAX && nnz_capacity >= 0 && nnz_capacity < INT32_MAX,
                 "sage_sample_hop: bad size");
    POPE_REQUIRE(fanout != 0, "sage_sample_hop: fanout must be > 0, or < 0 for all neighbours");
    POPE_REQUIRE(fanout < 0 || nnz_capacity >= n_targets * (int64_t)fanout, "sage_sample_hop: nnz_capacity %lld < n_targets * fanout",
                 (long long)nnz_capacity);
    long long *rep = nullptr, *rep_dev = nullptr;                 // the two counts come back through pinned, device-mapped memory
    int rc = pinned_report(&rep, &rep_dev);
    if (rc) return rc;
    rc = enqueue_hop(rowptr, col, N, targets, n_targets, fanout, seed, hop, out_rowptr, out_col, nnz_capacity, out_n_id, scratch,
                     scratch_bytes, sample_layout(N, n_targets, nnz_capacity), true, false, nullptr, nullptr, nullptr, rep_dev, stream,
                     "sage_sample_hop");
    if (rc) return rc;
    POPE_HIP(hipStreamSynchronize(stream));
    POPE_HIP(hipGetLastError());
    if (rep[0] > nnz_capacity) {
        set_error("sage_sample_hop: %lld sampled edges exceed nnz_capacity %lld", rep[0], (long long)nnz_capacity);
        return POPE_ERR_WORKSPACE;
    }
    *nnz_host = rep[0];
    *n_src_host = rep[1];
    return POPE_OK;
}

// Device-extent form of sage_sample_batch: NO host synchronisation.  Every hop runs over its capacity (hop h may meet up to
// t_cap[h] targets) and reads the true target count of hop h - 1 on the device; dims[h] = {n_dst, n_src, nnz, 0} (int32,
// device) is what the SAGE entry points take as their `dims` argument.  Capturable into a HIP graph: with seed_dev the
// draw follows a device word (sage_advance_counters bumps it between replays).
static int sample_batch_device_impl(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *seeds, int64_t n_seeds,
                                    const int32_t *fanouts_host, int32_t n_hops, uint64_t seed, const uint64_t *seed_dev,
                                    int32_t *const *out_rowptr, int32_t *const *out_col, int64_t *const *out_n_id, int32_t *dims,
                                    void *scratch, size_t scratch_bytes, hipStream_t stream, const char *who, const long long *first_dev,
                                    const long long *labels, long long *y_out) {
    const int64_t *targets = seeds;
    int64_t t_cap = n_seeds, t_last = n_seeds, cap_last = 0;
    for (int h = 0; h < n_hops; ++h) {                              // the scratch layout of the largest (= last) hop serves every hop
        POPE_REQUIRE(fanouts_host[h] > 0, "%s: fan-outs must be positive", who);
        t_last = t_cap;
        cap_last = t_cap * (int64_t)fanouts_host[h];
        POPE_REQUIRE(t_cap + cap_last < INT32_MAX, "%s: capacity of hop %d exceeds 31 bits", who, h);
        t_cap += cap_last;
    }
    const SampleLayout L = sample_layout(N, t_last, cap_last);
    t_cap = n_seeds;
    for (int h = 0; h < n_hops; ++h) {
        const int64_t cap = t_cap * (int64_t)fanouts_host[h];
        const int rc = enqueue_hop(rowptr, col, N, targets, t_cap, fanouts_host[h], seed, h, out_rowptr[h], out_col[h], cap, out_n_id[h],
                                   scratch, scratch_bytes, L, h == 0, h + 1 < n_hops, h == 0 ? nullptr : dims + 4 * (h - 1) + 1, dims + 4 * h,
                                   (const unsigned long long *)seed_dev, nullptr, stream, who, h == 0 ? first_dev : nullptr,
                                   h == 0 ? labels : nullptr, h == 0 ? y_out : nullptr);
        if (rc) return rc;
        targets = out_n_id[h];
        t_cap = t_cap + cap;
    }
    return POPE_OK;
}

extern "C" int sage_sample_batch_device(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *seeds, int64_t n_seeds,
                                        const int32_t *fanouts_host, int32_t n_hops, uint64_t seed, const uint64_t *seed_dev,
                                        int32_t *const *out_rowptr, int32_t *const *out_col, int64_t *const *out_n_id, int32_t *dims,
                                        void *scratch, size_t scratch_bytes, void *stream_) {
    clear_error();
    POPE_REQUIRE(rowptr && col && seeds && fanouts_host && out_rowptr && out_col && out_n_id && dims && scratch && n_hops > 0 && n_hops <= 16,
                 "sage_sample_batch_device: null pointer or bad hop count");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && n_seeds > 0, "sage_sample_batch_device: bad size");
    return sample_batch_device_impl(rowptr, col, N, seeds, n_seeds, fanouts_host, n_hops, seed, seed_dev, out_rowptr, out_col, out_n_id, dims, scratch,
                                    scratch_bytes, (hipStream_t)stream_, "sage_sample_batch_device", nullptr, nullptr, nullptr);
}

// The batch an epoch's loader would hand out next (main.py:100-123: NeighborSampler(node_idx, batch_size, shuffle) + y =
// data.y[n_id[:batch_size]]) without the loader: the seeds are order[*first_dev .. + n_seeds) -- `order` = the epoch's
// (shuffled) node list on the device, `first_dev` a device word the caller advances by n_seeds per step
// (sage_advance_counters) -- and, with `labels`, y_out[i] = labels[seed i].  The caller keeps *first_dev + n_seeds within
// `order`.  Otherwise sage_sample_batch_device: no synchronisation, capturable, replayable.
extern "C" int sage_sample_epoch_batch_device(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *order, const int64_t *first_dev,
                                              int64_t n_seeds, const int64_t *labels, int64_t *y_out, const int32_t *fanouts_host, int32_t n_hops,
                                              uint64_t seed, const uint64_t *seed_dev, int32_t *const *out_rowptr, int32_t *const *out_col,
                                              int64_t *const *out_n_id, int32_t *dims, void *scratch, size_t scratch_bytes, void *stream_) {
    clear_error();
    POPE_REQUIRE(rowptr && col && order && first_dev && fanouts_host && out_rowptr && out_col && out_n_id && dims && scratch && n_hops > 0 && n_hops <= 16,
                 "sage_sample_epoch_batch_device: null pointer or bad hop count");
    POPE_REQUIRE((labels == nullptr) == (y_out == nullptr), "sage_sample_epoch_batch_device: labels and y_out go together");
    POPE_REQUIRE(N > 0 && N < INT32_MAX && n_seeds > 0, "sage_sample_epoch_batch_device: bad size");
    return sample_batch_device_impl(rowptr, col, N, order, n_seeds, fanouts_host, n_hops, seed, seed_dev, out_rowptr, out_col, out_n_id, dims, scratch,
                                    scratch_bytes, (hipStream_t)stream_, "sage_sample_epoch_batch_device", (const long long *)first_dev,
                                    (const long long *)labels, (long long *)y_out);
}

// All hops of a mini-batch in one call (main.py:100-116 NeighborSampler(sizes=[25, 10]) draws every hop of a batch at once):
// hop h samples around hop h-1's node list, so the per-hop synchronisations stay, but the host-side glue between the hops
// (allocation, slicing, argument marshalling in Python: ~25 us per hop in a launch-bound training step) is gone.
// Buffers are sized by CAPACITY: hop h may meet up to t_capacity[h] targets (t_capacity[0] = n_seeds,
// t_capacity[h] = t_capacity[h-1] * (1 + fanout[h-1])) and keep up to t_capacity[h] * fanout[h] edges.
extern "C" int sage_sample_batch(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *seeds, int64_t n_seeds,
                                 const int32_t *fanouts_host, int32_t n_hops, uint64_t seed, int32_t *const *out_rowptr,
                                 int32_t *const *out_col, int64_t *const *out_n_id, int64_t *nnz_host, int64_t *n_src_host,
                                 void *scratch, size_t scratch_bytes, void *stream_) {
    clear_error();
    POPE_REQUIRE(fanouts_host && out_rowptr && out_col && out_n_id && nnz_host && n_src_host && n_hops > 0 && n_hops <= 16,
                 "sage_sample_batch: null pointer or bad hop count");
    const int64_t *targets = seeds;
    int64_t T = n_seeds;
    for (int h = 0; h < n_hops; ++h) {
        POPE_REQUIRE(fanouts_host[h] > 0, "sage_sample_batch: fan-outs must be positive (use sage_sample_hop for 'all neighbours')");
        const int rc = sage_sample_hop(rowptr, col, N, targets, T, fanouts_host[h], seed, h, out_rowptr[h], out_col[h],
                                       T * (int64_t)fanouts_host[h], out_n_id[h], &nnz_host[h], &n_src_host[h], scratch, scratch_bytes,
                                       stream_);
        if (rc) return rc;
        targets = out_n_id[h];
        T = n_src_host[h];
    }
    return POPE_OK;
}
