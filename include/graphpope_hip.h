/*
 * graphpope_hip.h -- C ABI of libgraphpope_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for GraphPOPE's one hot path: what a binding of the
 * reference (/root/reference, 100 % Python) would load with ctypes.  INTEGRATION.md shows that stub.
 * Plain pointers and sizes only; every buffer is allocated by the caller.  Unless a parameter name ends in
 * `_host`, pointers are DEVICE pointers valid on the device that is current on the calling thread
 * (hipSetDevice / torch.cuda.set_device), and work is enqueued on `stream` (a hipStream_t passed as
 * void*, NULL = the default stream).  The library owns no device memory.  What it keeps between calls, all of it
 * host-side: the per-thread error string; per device, a mutex-guarded ring of small pinned host slots (a call's
 * anchor ids and its BFS verdict travel through a slot of its own, reused only after the event behind its last
 * device-side user has completed -- calls from several host threads or on several streams do not share staging);
 * and two process-global diagnostic facilities that are off by default and NOT thread-safe: the level-timing hook
 * (pope_profile_levels) and the A/B knobs (pope_debug_set).
 *
 * Every function returns 0 on success or a negative POPE_ERR_* code; pope_last_error() then holds
 * a message for the calling thread.
 *
 * Which reference interface each entry point replaces (file:line into /root/reference):
 *
 *   pope_csr_build           utils.py:121        G = to_networkx(data)           (graph build from edge_index)
 *   pope_geodesic_bfs        utils.py:64-81,     the per-(node, anchor) nx.shortest_path loop and its
 *                            utils.py:92-114     multiprocessing fan-out, as one batched multi-source BFS
 *   pope_geodesic_finalize   utils.py:73,125,    1/len(path), tensor conversion and torch.cat((x, emb), 1)
 *                            utils.py:129-135
 *   pope_geodesic_run        utils.py:144-145    get_geodesic_distance_vector + concat_into_features in one call
 *   pope_geodesic_column_stats  utils.py:50-54   the BFS sums behind nx.closeness_centrality (biased anchor selection)
 *   pope_pagerank_weights /  utils.py:26-30      nx.pagerank_scipy power iteration as SpMV over the device CSR
 *   pope_pagerank_step                           (biased anchor selection, README's best row)
 *   pope_geodesic_hops       (no counterpart)    the integer hop matrix the floats are made of; parity tests
 *   pope_kmeans_plusplus /   utils.py:168-170    KMeans(n_clusters=K).fit(X).cluster_centers_ (k-means++ seeding, Lloyd
 *   pope_kmeans_lloyd_step                       iterations with the MFMA tile as the assignment step)
 *   pope_pairwise_minmax     utils.py:158-176    sklearn cosine/euclidean pairwise + MinMaxScaler
 *   pope_pairwise_features   utils.py:158-177    the same + concat_into_features inside the tile kernel
 *   pope_concat              utils.py:129-135    torch.cat for the node2vec branch (device-resident callers)
 *   pope_host_copy_2d        utils.py:129-135    torch.cat's feature half on the host cores (host -> host callers)
 *   sage_conv_forward /      main.py:206 and     PyG SAGEConv((x_src, x_dst), adj_t): mean aggregation over the
 *   sage_conv_backward       PyG SAGEConv [3p]   sampled CSR + lin_l + lin_r, and its gradients
 *   sage_bn_relu_dropout_forward / _backward   main.py:207-209
 *                                BatchNorm1d + relu_ + F.dropout of the hidden layers, forward and backward
 *   sage_cross_entropy_*     main.py:216         F.cross_entropy(y_hat, y): loss and gradient in two launches
 *   sage_adam_step           main.py:244         torch.optim.Adam step, all parameter tensors in one launch
 *   sage_sample_hop          main.py:100-116     NeighborSampler -> torch_sparse.sample_adj (one hop), relabelled block
 */
#ifndef GRAPHPOPE_HIP_H
#define GRAPHPOPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POPE_OK                 0
#define POPE_ERR_INVALID       -1   /* bad argument (null pointer, negative size, K <= 0, ...)            */
#define POPE_ERR_HIP           -2   /* a HIP runtime call failed; message has hipGetErrorString           */
#define POPE_ERR_INDEX         -3   /* edge_index or anchor id outside [0, N)                             */
#define POPE_ERR_HOP_OVERFLOW  -4   /* a hop count does not fit the plane capacity / hop dtype given      */
#define POPE_ERR_WORKSPACE     -5   /* caller workspace too small                                         */
#define POPE_ERR_NO_DEVICE     -6   /* no gfx950 device visible                                           */
#define POPE_ERR_UNSORTED      -7   /* CSR was built with defer_check and edge_index is not sorted by source */

/* Message for the last error on the calling thread ("" if none).  Never NULL; valid until the next call. */
const char *pope_last_error(void);

/* Library / build identification, e.g. "graphpope_hip 0.2 gfx950". */
const char *pope_version(void);

/*
 * POPE_OK if a gfx950 device is visible to the calling process (and, if cu_count is not NULL, its number of compute
 * units); POPE_ERR_NO_DEVICE otherwise -- the product path has no CPU fallback (utils.Graphpope raises).
 */
int pope_require_device(int32_t *cu_count_host);

/*
 * Diagnostic knobs for A/B runs (tools/ab_*.py and the kernel-variant tests): process-global, not thread-safe, never
 * needed by a caller.  value < 0 restores the automatic choice where one exists.
 */
#define POPE_KNOB_LIVE_MODE         0   /* level kernel: -1 auto (by graph size), 1 live-bit table staged in LDS, 2 table read from global memory, 3 global table behind a summary in LDS */
#define POPE_KNOB_FINALIZE_VARIANT  1   /* 1 (default) the pipelined finalise kernels (every load of a row in flight, next row requested before this one is stored, rows dealt round-robin; wide rows through LDS tables); 7 the round 1-3 kernel (serial loops); 0 the generic kernel -- kept so that tests can compare their bits; 8 / 9 / 10: as 1, with wide rows on the table kernel k_finalize_lut only without features (8, the default), always (9: features by the copy kernel in front of it) or never (10: the shuffle kernel k_finalize_wide); 11 / 12: k_finalize_lut over several short-rowed shards takes a batch per (shard, block of rows) (11, the default) or the flat (row, half-word) order (12) */
#define POPE_KNOB_FINALIZE_BLOCKS   2   /* grid of the finalise kernel (default: one work item per wave for the pipelined kernels, 2048 blocks for the round 1-3 one) */
#define POPE_KNOB_GEMM_TILE         3   /* SAGE GEMM: 0 auto, 1 64x64, 2 64x128, 3 128x256 tiles, 4 / 5 stream-K without loader waves, 6 stream-K with stages of 64 in two 80 KB buffers, 7 stream-K instead of the chip-fitted whole tiles (gemm_tile16.h) */
#define POPE_KNOB_PAIRWISE_KERNEL   4   /* node2vec embedding: 0 auto (anchor-resident persistent kernel for depths <= 128), 1 one tile per block, 2 / 3 persistent kernel with one / two consumer sets */
#define POPE_KNOB_COPY_BATCHES      5   /* node2vec embedding: 16-piece batches per wave of the feature-copy kernel beside the tile kernel (default 1) */
#define POPE_KNOB_FAIL_HOST_REGISTER 7  /* host -> host boundary, tests of the fallbacks, bit mask: 1 = every hipHostRegister is refused (edge_index then goes up through pinned staging), 2 = behave as if the pinned ring's hipHostMalloc had been refused (float columns through the bounce buffer), 4 = behave as if the 4 MB bounce buffer had been refused too (blocking copies by the runtime) */
#define POPE_KNOB_SAGE_FORWARD_OVERLAP 14 /* sage_conv_forward(_indexed): 1 (default) the gather runs beside the x_dst half of the projection in one launch, the agg half follows; 0 gather, then the whole projection */
#define POPE_KNOB_GEMM_SMALL_TILE16 15  /* SAGE forward products too small for stream-K: 1 (default) = whole tiles of 16 or 32 rows (gemm_tile16.h), 0 = the 64 x 64 tile kernel */
#define POPE_KNOB_GEMM_TILE16_BUFFERS 18 /* whole-tile forward GEMM: 4 (default) or 3 LDS stage buffers (two or one stage times to hide a request; same bits) */
#define POPE_KNOB_PREPARE_MERGE     19  /* pope_geodesic_run: 1 (default) = the clear + seed of the BFS state and the speculative CSR build as two roles of ONE launch (k_prepare; at most 256 anchors per call); 0 = two launches */
#define POPE_KNOB_STREAMK_XCD       20  /* SAGE weight-gradient GEMM (stream-K): 0 units dealt to the blocks in block order; 1 (default) = XCD-aware: the blocks that work on the same depth range of the tiles sharing their B rows are neighbours in one XCD (csrc/gemm_streamk_tn.h) -- the partial sums of a tile are cut at other depths, so the last bits may differ; deterministic either way */
/* (Knob numbers 6, 8-13, 16 and 17 belonged to experiments that were measured slower and removed in round 5 -- a copy role and block caps
 * in the level launches, the last levels inside the finalise launch, side streams in the SAGE backward pass, split-bf16 products, the
 * one-launch SAGE layer, the registered result mode: DESIGN.md keeps their figures.) */
int pope_debug_set(int32_t knob, int32_t value);

/* ------------------------------------------------------------------------------------------------
 * Graph: forward CSR (row = source node, columns = targets) in int32, built on the device into
 * caller-allocated arrays.  Distance is measured node -> anchor along edge direction (utils.py:73
 * nx.shortest_path(G, source=node, target=anchor) on a DiGraph), so the bottom-up BFS pulls over a
 * node's OUT-edges: this is the only adjacency the path needs.
 * ------------------------------------------------------------------------------------------------ */
size_t pope_csr_scratch_bytes(int64_t N, int64_t E);
size_t pope_csr_aux_elems(int64_t E);          /* int32 elements of `aux` below */

/*
 * edge_index: int64 [2, E] row-major on the device, PyG convention: edge e is
 * edge_index[e] -> edge_index[E + e].  Self-loops and repeated edges are allowed and kept.
 * Written (all caller-allocated, int32):
 *   rowptr [N + 1], col [E rounded up to a multiple of 4, >= 4], erow [same]: CSR slot p holds the edge erow[p] -> col[p], slots sorted
 *       by erow (erow is the row id of every slot: the edge-parallel BFS kernel streams erow/col instead of
 *       chasing rowptr).  Column order inside a row: the edge list's own order if edge_index is already sorted by source
 *       (PyG's coalesced order: preserved, no atomics); otherwise ascending by target (a row-wise sort follows the
 *       counting scatter, so the CSR is a pure function of the edge set: repeated builds are identical).
 *   aux [pope_csr_aux_elems(E)]: header (counts, status flags) + the rows that span several 256-slot chunks.
 * defer_check = 0: synchronises `stream` once; returns POPE_ERR_INDEX for an id outside [0, N) and falls back
 *   to a counting sort when edge_index is not sorted by source.
 * defer_check = 1: fully asynchronous; only the sorted fast path is attempted and its verdict stays in aux:
 *   pope_geodesic_bfs then returns POPE_ERR_INDEX / POPE_ERR_UNSORTED (rebuild with defer_check = 0).
 * Requires 0 <= N < 2^31 and 0 <= E < 2^31.
 */
int pope_csr_build(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr, int32_t *col, int32_t *erow,
                   int32_t *aux, void *scratch, size_t scratch_bytes, int32_t defer_check, void *stream);

/*
 * The same CSR in canonical form whatever the order of edge_index: rows by source, every row's targets ascending
 * (repeated edges adjacent).  Synchronises `stream` once (index check).  Used by the rankings below.
 */
int pope_csr_build_canonical(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr, int32_t *col, int32_t *erow,
                             int32_t *aux, void *scratch, size_t scratch_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Biased anchor selection: PageRank scores (utils.py:26-30 nx.pagerank_scipy(to_networkx(data)); NetworkX 3: nx.pagerank),
 * float64, reproduced bit for bit so that the ascending stable sort picks the reference's anchors.
 *   pope_pagerank_weights  w[j] = 1 / (distinct targets of j), 0 for dangling nodes, from the canonical CSR by source.
 *   pope_pagerank_step     x_out[i] = alpha * (sum over the in-neighbours j of i, ascending, of w[j] * x[j]
 *                                               + dangling_sum / N) + (1 - alpha) / N
 *                          over the canonical CSR of the REVERSED edge list (rows = targets, entries = sources).  One
 *                          power iteration; the host binding owns the loop (dangling_sum = sum of x over the dangling
 *                          nodes in index order, convergence on sum |x - xlast| < N * tol, as SciPy / NumPy evaluate them).
 * Asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------------ */
int pope_pagerank_weights(const int32_t *rowptr, const int32_t *col, int64_t N, double *w, void *stream);
int pope_pagerank_step(const int32_t *rowptr_by_target, const int32_t *sources, int64_t N, const double *x, const double *w,
                       double dangling_sum, double alpha, double *x_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K-means anchors of the node2vec branch (utils.py:168-170  KMeans(n_clusters=K).fit(X).cluster_centers_, scikit-learn
 * defaults).  The host binding owns the control flow and the random numbers (the reference's KMeans draws from the global
 * NumPy stream: one choice() for the first seed, 2 + int(log K) uniforms per further seed); the device does the distances,
 * the reductions and the centre updates.  X is float32 [N, D], already centred by its column means (KMeans.fit does that).
 *   pope_column_moments     sum[c], sumsq[c] of every column in float64 (column means; the tolerance 1e-4 x mean variance)
 *   pope_shift_columns      out = X + sign * shift[c]
 *   pope_kmeans_plusplus    k-means++ seeding: chosen[K] (device int64) = the seeded rows, in order.  uniforms_host:
 *                           (K - 1) x n_trials doubles in [0, 1), step-major.  No host synchronisation after the upload.
 *   pope_kmeans_lloyd_step  one Lloyd iteration: labels (N x K products on the f32 MFMA tile + row argmin, first minimum
 *                           wins), new centres (mean of the members, float64 sums in index order; an empty cluster keeps
 *                           its centre), *changed (device int) = some label differs from labels_prev, *shift_total (device
 *                           double) = sum of squared centre shifts.  Deterministic.  Asynchronous on `stream`.
 * scratch: pope_kmeans_scratch_bytes(N, D, K) for the two kmeans calls; 4096 * D * 8 bytes for pope_column_moments.
 * ------------------------------------------------------------------------------------------------ */
size_t pope_kmeans_scratch_bytes(int64_t N, int32_t D, int32_t K);
int pope_column_moments(const float *X, int64_t N, int32_t D, double *sum, double *sumsq, void *scratch, size_t scratch_bytes,
                        void *stream);
int pope_shift_columns(const float *X, const float *shift, int64_t N, int32_t D, float sign, float *out, void *stream);
int pope_kmeans_plusplus(const float *X, int64_t N, int32_t D, int32_t K, int64_t first_id, const double *uniforms_host,
                         int32_t n_trials, int64_t *chosen, void *scratch, size_t scratch_bytes, void *stream);
int pope_kmeans_lloyd_step(const float *X, int64_t N, int32_t D, const float *centers, int32_t K, float *centers_new,
                           int32_t *labels, const int32_t *labels_prev, int32_t *changed, double *shift_total,
                           void *scratch, size_t scratch_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Geodesic embedding.
 *
 * Hop counts are kept BIT-SLICED ("hop planes"): anchors are packed 64 per uint64 word, W = pope_words(K)
 * words per node, and a plane is a uint64 [N, W] array.  planes[0] is the reachability plane (bit j of
 * node v set <=> a directed path v -> anchors[j] exists); planes[1 + b] holds bit b of the hop count.
 * This is also the shard exchange format for multi-GPU runs: a rank that owns anchors
 * [k0, k0 + K_local) produces planes for K_local and the shards are concatenated by an all-gather.
 * ------------------------------------------------------------------------------------------------ */

/* Words per node for K anchors: ceil(K / 64) rounded up to 1, 2 or a multiple of 4. */
int32_t pope_words(int32_t K);

/* Bytes of one plane, and of the scratch pope_geodesic_bfs needs (a control block, the anchors, three rotating frontier planes,
 * their three one-bit-per-node live tables and one summary of a live table). */
size_t pope_plane_bytes(int64_t N, int32_t K);
size_t pope_bfs_scratch_bytes(int64_t N, int64_t E, int32_t K);

/*
 * Multi-source BFS from all K anchors at once over the CSR of pope_csr_build.
 *   anchors_host   int64 [K] on the HOST (the reference samples them on the host: utils.py:22-24); duplicates kept.
 *   planes         uint64 [plane_capacity + 1, N, W]; need not be initialised.  On return planes
 *                  [0, 1 + *n_hop_bits) are valid, the rest untouched.
 *   scratch        pope_bfs_scratch_bytes(N, E, K) bytes.
 *   max_hop_host   (out, host) largest finite hop count found.
 *   n_hop_bits_host(out, host) number of hop-bit planes written = bits needed for *max_hop.
 * Returns POPE_ERR_HOP_OVERFLOW if a hop count would need more than plane_capacity bits.
 * Synchronises `stream` (the level loop polls a device flag every few levels).
 */
int pope_geodesic_bfs(const int32_t *rowptr, const int32_t *col, const int32_t *erow, const int32_t *aux,
                      int64_t N, int64_t E,
                      const int64_t *anchors_host, int32_t K, uint64_t *planes, int32_t plane_capacity, void *scratch, size_t scratch_bytes,
                      int32_t *max_hop_host, int32_t *n_hop_bits_host, void *stream);

/*
 * The same BFS in two halves (identical arguments): _begin enqueues the clears, the seed and the first 12 levels and
 * returns without waiting; _finish synchronises `stream`, reads the verdict and keeps going if the graph is deeper.
 * Work enqueued on `stream` in between (an all-gather of planes[0 .. 5), the finalise kernel) runs speculatively: it is
 * valid iff _finish reports *n_hop_bits <= 4 (planes 1..4 are cleared up front, so unused hop-bit planes read as zero).
 * The pinned anchor staging buffer is per device: do not begin a second BFS on the device before finishing the first.
 */
int pope_geodesic_bfs_begin(const int32_t *rowptr, const int32_t *col, const int32_t *erow, const int32_t *aux,
                            int64_t N, int64_t E, const int64_t *anchors_host, int32_t K, uint64_t *planes,
                            int32_t plane_capacity, void *scratch, size_t scratch_bytes, void *stream);
int pope_geodesic_bfs_finish(const int32_t *rowptr, const int32_t *col, const int32_t *erow, const int32_t *aux,
                             int64_t N, int64_t E, const int64_t *anchors_host, int32_t K, uint64_t *planes,
                             int32_t plane_capacity, void *scratch, size_t scratch_bytes, int32_t *max_hop_host,
                             int32_t *n_hop_bits_host, void *stream);

/*
 * out[v, 0:F] = x[v, :],  out[v, F + c0 + j] = 1.0f / (hops(v, anchor j) + 1), 0.0f if unreachable,
 * for the K_shard anchors of one shard.  out is float32 [N, out_cols] row-major with out_cols >= F + c0 + K_shard;
 * x may be NULL (then only the embedding columns are written: used for shards after the first).
 * Asynchronous on `stream`.
 */
int pope_geodesic_finalize(const uint64_t *planes, int32_t n_hop_bits, int64_t N, int32_t K_shard,
                           const float *x, int32_t F, float *out, int64_t out_cols, int32_t c0, void *stream);

/*
 * Multi-GPU form: `planes` holds n_shards all-gathered shards back to back, shard g at planes + g * shard_stride_words,
 * each [1 + n_hop_bits, N, W(K_shard)] for K_shard anchors; writes x and the columns of ALL shards
 * (out[v, F + g * K_shard + j]) in one pass over the output.  Asynchronous on `stream`.
 */
int pope_geodesic_finalize_shards(const uint64_t *planes, int32_t n_shards, int64_t shard_stride_words, int32_t n_hop_bits,
                                  int64_t N, int32_t K_shard, const float *x, int32_t F, float *out, int64_t out_cols,
                                  void *stream);

/* The name of the finalise kernel the three calls above and pope_geodesic_run launch for a shape, as a kernel trace shows it
 * ("k_finalize_pipe<2, 1>", "k_finalize_wide<0>", ...), written into name[0 .. cap): lets a measurement label itself with the
 * kernel that really ran (utils.py:129-135 is one torch.cat whatever the shape).  Aligned bases and row pitches assumed. */
int pope_finalize_kernel_name(int64_t N, int32_t K_shard, int32_t F, int32_t has_x, int32_t n_shards, char *name, size_t cap);

/* The same for the level kernel of a BFS over N nodes from K anchors ("k_bfs_level<4, 1, 0>": words per tile, how the live-bit table is
 * read, how a node's tiles are walked -- csrc/geodesic.hip: level_choice). */
int pope_level_kernel_name(int64_t N, int32_t K, char *name, size_t cap);

/*
 * The whole geodesic hot path in ONE call (what utils.py:137-147 does after sampling the anchors):
 * edge_index -> CSR -> multi-source BFS -> out[v, 0:F] = x[v, :], out[v, F + j] = 1 / (hops(v, anchor j) + 1).
 * Everything is enqueued speculatively (sorted-CSR fast path, 12 BFS levels, the finalise kernel reading the
 * depth from device memory).  The host then waits ONCE, for the BFS verdict only: the finalise kernel publishes it
 * in pinned host memory when it starts, and the call returns while `out` is still being written -- `out`, like the
 * result of any asynchronous call, is complete in `stream` order (synchronise `stream` before reading it from the
 * host or from another stream).  Unsorted edge lists and graphs deeper than 10 hops transparently take the general
 * path (stream synchronisations in between).
 * workspace: pope_geodesic_run_workspace_bytes(N, E, K, plane_capacity) bytes, device memory, no initialisation
 * needed; afterwards pope_geodesic_run_planes() locates the hop planes inside it (planes [0, 1 + *n_hop_bits) valid).
 * out may be NULL (BFS only).  Returns POPE_ERR_HOP_OVERFLOW if plane_capacity bits cannot hold the depth.
 */
size_t pope_geodesic_run_workspace_bytes(int64_t N, int64_t E, int32_t K, int32_t plane_capacity);
uint64_t *pope_geodesic_run_planes(void *workspace, int64_t N, int64_t E, int32_t K, int32_t plane_capacity);
int pope_geodesic_run(const int64_t *edge_index, int64_t E, int64_t N, const int64_t *anchors_host, int32_t K,
                      const float *x, int32_t F, float *out, int64_t out_cols, int32_t plane_capacity,
                      void *workspace, size_t workspace_bytes, int32_t *max_hop_host, int32_t *n_hop_bits_host,
                      void *stream);

/*
 * Measurement hook (bench.py): enable = 1: every BFS level launch is bracketed by HIP events on the launch stream and
 * pope_profile_read returns, per level launched since enabling, the level number and the elapsed milliseconds of its
 * kernel (waits for the events).  enable = 2: ONE event pair around each enqueued run of level launches (no events
 * between the kernels, so the pipeline is undisturbed); pope_profile_read then returns one entry per run with
 * levels[i] = -(number of launches in the run) and the elapsed milliseconds of the whole run.  0 disables.
 * Not thread-safe; off by default.
 */
void pope_profile_levels(int32_t enable);
int32_t pope_profile_read(int32_t *levels_host, float *level_ms_host, int32_t capacity);

/* Integer hop matrix: hops int32 [N, K] node-major, -1 = unreachable.  Asynchronous on `stream`. */
int pope_geodesic_hops(const uint64_t *planes, int32_t n_hop_bits, int64_t N, int32_t K,
                       int32_t *hops, void *stream);

/*
 * Column statistics of the hop matrix, straight from the planes: reach[j] = number of nodes with a path to anchor j
 * (the anchor included), hop_sum[j] = sum of their hop counts; both int64 [K] on the device.  What closeness
 * centrality needs (utils.py:50-54: nx.closeness_centrality uses inward distances on a DiGraph).  Asynchronous.
 */
size_t pope_column_stats_scratch_bytes(int32_t K);
int pope_geodesic_column_stats(const uint64_t *planes, int32_t n_hop_bits, int64_t N, int32_t K, int64_t *hop_sum,
                               int64_t *reach, void *scratch, size_t scratch_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * node2vec-space embedding: pairwise distance to the anchor rows + per-column min-max scaling.
 * ------------------------------------------------------------------------------------------------ */
#define POPE_METRIC_COSINE_DISTANCE   0   /* 'distance'   -> sklearn cosine_distances   */
#define POPE_METRIC_COSINE_SIMILARITY 1   /* 'similarity' -> sklearn cosine_similarity  */
#define POPE_METRIC_EUCLIDEAN         2   /* 'euclidean'  -> sklearn euclidean_distances */

size_t pope_pairwise_scratch_bytes(int64_t N, int32_t K, int32_t D);

/*
 * X float32 [N, D]; A float32 [K, D] (the anchor rows); writes the min-max scaled [N, K] block into
 * out[v, c0 + j] of a float32 [N, out_cols] matrix.  Scaling is per column over all N rows:
 * y = e * (1 / range) + (0 - min * (1 / range)), range < 10 * FLT_EPSILON treated as 1.
 * Asynchronous on `stream`.
 */
int pope_pairwise_minmax(const float *X, int64_t N, int32_t D, const float *A, int32_t K, int32_t metric,
                         float *out, int64_t out_cols, int32_t c0, void *scratch, size_t scratch_bytes,
                         void *stream);
/* utils.py:165-167 as written there -- embedding[anchor_nodes] and the distances in one call: anchor j is row anchor_ids[j]
 * (DEVICE int64, each in [0, N)) of X, read through the ids by the tile kernel (no gathered copy of the anchor rows, no gather
 * launch in front of the call).  x = NULL / F = 0: the embedding columns alone.  Scratch: pope_pairwise_by_id_scratch_bytes. */
size_t pope_pairwise_by_id_scratch_bytes(int64_t N, int32_t K, int32_t D);
int pope_pairwise_features_by_id(const float *x, int32_t F, const float *X, int64_t N, int32_t D, const int64_t *anchor_ids, int32_t K,
                                 int32_t metric, float *out, int64_t out_cols, int32_t c0, void *scratch, size_t scratch_bytes,
                                 void *stream);

/*
 * The same with the feature half of utils.py:177 concat_into_features fused in: out[v, 0:F] = x[v, :] (x float32 [N, F],
 * may be NULL) is streamed by the blocks of the tile kernel between their product and their epilogue, next to the embedding
 * columns out[v, c0 + j], c0 >= F: the copy's HBM time hides under the MFMA work.  Asynchronous on `stream`.
 */
int pope_pairwise_features(const float *x, int32_t F, const float *X, int64_t N, int32_t D, const float *A, int32_t K,
                           int32_t metric, float *out, int64_t out_cols, int32_t c0, void *scratch, size_t scratch_bytes,
                           void *stream);

/* out[v, 0:F] = x[v, :] for a float32 [N, out_cols] matrix (the feature half of torch.cat). Asynchronous. */
int pope_concat(const float *x, int64_t N, int32_t F, float *out, int64_t out_cols, void *stream);

/*
 * HOST memory on both sides: copy `rows` rows of `row_bytes` bytes from src (row pitch src_pitch_bytes) to dst (row
 * pitch dst_pitch_bytes) with `threads` host threads (<= 0: all hardware threads) and streaming stores; returns when
 * the copy is complete.  The host half of torch.cat((data.x, embedding), 1) (utils.py:129-135) for the host -> host
 * Graphpope call: data.x never crosses PCIe -- it goes straight into the result tensor's first F columns on the
 * host cores while the GPU computes and ships only the K embedding columns.  No HIP call is made.
 */
int pope_host_copy_2d(const void *src_host, int64_t src_pitch_bytes, void *dst_host, int64_t dst_pitch_bytes,
                      int64_t row_bytes, int64_t rows, int32_t threads);

/*
 * The embedding half of the same torch.cat: `rows` rows of `row_bytes` bytes from DEVICE memory (pitch src_pitch_bytes)
 * into the last K columns of the host result (dst_host, row pitch dst_pitch_bytes; pinned memory makes it a true
 * asynchronous DMA on `stream`).  Only these K columns ever cross PCIe.  Asynchronous on `stream`.
 */
int pope_copy_2d_to_host(const void *src, int64_t src_pitch_bytes, void *dst_host, int64_t dst_pitch_bytes,
                         int64_t row_bytes, int64_t rows, void *stream);

/*
 * The whole torch.cat((data.x, embedding), 1) of the host -> host call (utils.py:129-135) into an ORDINARY PAGEABLE
 * result, as the reference returns one: out[:, :x_row_bytes] = x_host rows (host threads, streaming stores) and
 * out[:, x_row_bytes : x_row_bytes + emb_row_bytes] = emb rows (DEVICE memory, DMA on `stream`, behind whatever produced
 * emb there).  The feature columns are filled by `threads` host threads over `chunks` row chunks (<= 0: 8; first touch:
 * MADV_HUGEPAGE is applied first); the embedding rows come over in 8 MB chunks through three pinned slots that are allocated
 * once per process, and the worker threads copy each landed chunk out while the next one is on the bus -- no page of the
 * result is ever registered with the runtime.  If the pinned ring is refused (or a row is wider than a slot) the rows are staged
 * through a 4 MB pinned bounce buffer of the library's own and copied out by the calling thread (slower, same bytes; only if
 * that buffer is refused as well does a blocking copy by the runtime write the pageable rows).  `stream` is synchronised
 * before the call returns.  x_row_bytes or emb_row_bytes may be 0.
 */
int pope_assemble_host_result(const void *x_host, int64_t x_pitch_bytes, int64_t x_row_bytes, const void *emb,
                              int64_t emb_pitch_bytes, int64_t emb_row_bytes, void *out_host, int64_t out_pitch_bytes,
                              int64_t rows, int32_t threads, int32_t chunks, void *stream);

/*
 * The same in two halves, so that the page faults and the feature copy run UNDERNEATH the upload of edge_index and the GPU
 * work: pope_assemble_begin starts the host threads on out[:, :x_row_bytes] = x_host and returns a handle (NULL + an error
 * message on bad arguments); pope_assemble_finish brings the embedding columns down as pope_assemble_host_result does,
 * waits for everything and frees the handle (also when it fails).  pope_assemble_abort frees a handle that will not be finished.
 */
void *pope_assemble_begin(const void *x_host, int64_t x_pitch_bytes, int64_t x_row_bytes, void *out_host, int64_t out_pitch_bytes,
                          int64_t rows, int32_t threads, int32_t chunks);
int pope_assemble_finish(void *handle, const void *emb, int64_t emb_pitch_bytes, int64_t emb_row_bytes, void *stream);
void pope_assemble_abort(void *handle);
/* Optional, returns at once: the process's pinned ring (24 MB, 2 ms of hipHostMalloc) is allocated by a helper thread on
 * `device` (< 0: the thread's default), beside the caller's GPU work, instead of inside the first pope_assemble_finish. */
void pope_assemble_prepare(int32_t device);
/* 1 if pope_assemble_finish_codes can be used now (the pinned ring exists or could be allocated by this call), else 0 -- the
 * caller then brings float columns down with pope_assemble_finish, which stages them through the bounce buffer.  Waits for an
 * allocation that pope_assemble_prepare started. */
int32_t pope_assemble_ring_ready(void);

/*
 * The geodesic embedding in its transport form (utils.py:73 1 / len(path), one byte per element instead of four):
 * pope_geodesic_hop_codes writes codes[v * pitch + j] = 0 if node v has no path to anchor j, hops + 1 otherwise, and the
 * 256 floats the bytes stand for (lut[0] = 0, lut[c] = 1 / c, the finalise kernel's own arithmetic) -- hop counts above 254
 * are refused (POPE_ERR_INVALID; use pope_geodesic_finalize).  pope_assemble_finish_codes is pope_assemble_finish for that
 * form (it needs the pinned ring): K code bytes per row cross PCIe -- a quarter of the float columns -- and the worker threads write
 * out[:, x_row_bytes + 4 j] = lut[code] while they copy out of the ring; the host looks floats up, it computes none.
 */
int pope_geodesic_hop_codes(const uint64_t *planes, int32_t n_hop_bits, int32_t max_hop, int64_t N, int32_t K, uint8_t *codes,
                            int64_t codes_pitch_bytes, float *lut, void *stream);
int pope_assemble_finish_codes(void *handle, const uint8_t *codes, int64_t codes_pitch_bytes, int32_t K, const float *lut, void *stream);

/*
 * Caller-owned pageable HOST memory as a DMA endpoint for the length of one call: pope_host_pin registers the WHOLE PAGES
 * inside [host, host + bytes) with the HIP runtime -- never the partial pages at the two ends, which a heap allocation shares
 * with other objects (POPE_ERR_HIP if refused, or if fewer than 1 MB of whole pages lie inside: the caller then stages through
 * pinned memory instead); pope_copy_to_device enqueues the asynchronous host -> device copy on `stream` (for a pinned buffer:
 * the body by DMA from the registered pages, the two end fragments as plain copies); pope_host_unpin releases the pages (after
 * the copy has completed; POPE_ERR_INVALID for a pointer pope_host_pin did not pin).  Used for edge_index (utils.py:121):
 * 14.4 MB go up straight from the caller's tensor.
 */
int pope_host_pin(const void *host, size_t bytes);
int pope_host_unpin(const void *host);
int pope_copy_to_device(const void *src_host, void *dst, size_t bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * SAGEConv over a sampled bipartite block (CSR by destination; destinations are the first n_dst sources).
 *   out = lin_l(mean_{j in N(i)} x_src[j]) + lin_r(x_src[i]),  lin_l with bias, lin_r without.
 * ------------------------------------------------------------------------------------------------ */
/*
 * Device extents.  A mini-batch sampled on the device (sage_sample_batch_device) has data-dependent sizes; reading them
 * back costs a host synchronisation per step.  Every SAGE entry point below therefore takes `dims`: NULL (the int64 size
 * arguments ARE the sizes), or a DEVICE pointer to int32 [4] = {n_dst, n_src, nnz, 0} as the sampler wrote it.  The size
 * arguments are then CAPACITIES (buffer extents, grid sizing; true value <= capacity) and every kernel reads the true
 * sizes on the device: nothing is read back, and the same launches can be captured into a HIP graph and replayed on new
 * batches.  Rows beyond the true extents of an output are left untouched.
 */
size_t sage_conv_scratch_bytes(int64_t n_src, int64_t n_dst, int64_t nnz, int32_t c_in, int32_t c_out);   /* backward */
size_t sage_conv_forward_scratch_bytes(int64_t n_dst, int32_t c_in, int32_t c_out);                         /* forward  */

/*
 * rowptr int32 [n_dst + 1], col int32 [nnz] (indices into x_src), x_src f32 [n_src, c_in],
 * w_l / w_r f32 [c_out, c_in], b_l f32 [c_out] or NULL, out f32 [n_dst, c_out].
 * agg (f32 [n_dst, c_in]) receives the mean-aggregated features (kept for backward).
 * scratch: sage_conv_forward_scratch_bytes(n_dst, c_in, c_out) bytes (0 for small layers: scratch may be NULL then) --
 * the partial-tile slabs of the stream-K projection; with less, the projection falls back to whole-tile kernels.
 * Asynchronous.
 */
int sage_conv_forward(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz,
                      const float *x_src, int32_t c_in, const float *w_l, const float *b_l, const float *w_r,
                      int32_t c_out, float *agg, float *out, void *scratch, size_t scratch_bytes, const int32_t *dims,
                      void *stream);

/* The aggregation half on its own: agg[i, :] = mean_{p in row i} x_src[col[p], :] (zero for empty rows).  Asynchronous. */
int sage_gather_mean(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz,
                     const float *x_src, int32_t c_in, float *agg, void *stream);

/*
 * Gradients.  grad_out f32 [n_dst, c_out].  grad_x f32 [n_src, c_in] (may be NULL), grad_w_l, grad_w_r
 * f32 [c_out, c_in], grad_b_l f32 [c_out] (may be NULL) are OVERWRITTEN.  Asynchronous.
 */
/*
 * sage_conv_forward on rows of a resident feature matrix: block-local source j is feats[n_id[j]] (feats: [n_rows, c_in],
 * n_id: int64 [n_src] on the device, its first n_dst entries are the destinations).  Replaces convert_batch's
 * x = data.x[n_id] (main.py:118-123) followed by the layer: the [n_src, c_in] gather is never materialised.
 * x_dst (out): [n_dst, c_in] = feats[n_id[:n_dst]], to be passed to sage_conv_backward as x_src (with n_src = n_dst and
 * grad_x = NULL: features have no gradient).
 */
int sage_conv_forward_indexed(const int32_t *rowptr, const int32_t *col, const int64_t *n_id, int64_t n_src, int64_t n_dst,
                              int64_t nnz, const float *feats, int64_t n_rows, int32_t c_in, const float *w_l, const float *b_l,
                              const float *w_r, int32_t c_out, float *agg, float *x_dst, float *out, void *scratch,
                              size_t scratch_bytes, const int32_t *dims, void *stream);
/*
 * The same two forward calls for a layer whose output goes into BatchNorm (main.py:206-207: x = convs[i](...); x = bns[i](x)): the
 * projection's epilogue also leaves the column sums of `out` and of its squares, per row tile, in bn_pa / bn_pb ([bn_parts_cap, c_out]
 * float64 each; bn_parts_cap >= ceil(n_dst / 16) always suffices) -- the first stage of the BatchNorm statistics, which otherwise is a
 * launch of its own that reads `out` back.  bn_info (HOST, 2 ints, out): [0] = row tiles written (0: this shape's kernels produce no
 * statistics -- call sage_bn_relu_dropout_forward as usual), [1] = rows per tile; pass both to sage_bn_relu_dropout_forward_stats.
 */
int sage_conv_forward_stats(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz, const float *x_src,
                            int32_t c_in, const float *w_l, const float *b_l, const float *w_r, int32_t c_out, float *agg, float *out,
                            void *scratch, size_t scratch_bytes, const int32_t *dims, double *bn_pa, double *bn_pb, int32_t bn_parts_cap,
                            int32_t *bn_info, void *stream);
int sage_conv_forward_indexed_stats(const int32_t *rowptr, const int32_t *col, const int64_t *n_id, int64_t n_src, int64_t n_dst,
                                    int64_t nnz, const float *feats, int64_t n_rows, int32_t c_in, const float *w_l, const float *b_l,
                                    const float *w_r, int32_t c_out, float *agg, float *x_dst, float *out, void *scratch,
                                    size_t scratch_bytes, const int32_t *dims, double *bn_pa, double *bn_pb, int32_t bn_parts_cap,
                                    int32_t *bn_info, void *stream);
int sage_conv_backward(const int32_t *rowptr, const int32_t *col, int64_t n_src, int64_t n_dst, int64_t nnz,
                       const float *x_src, const float *agg, int32_t c_in, const float *w_l, const float *w_r,
                       int32_t c_out, const float *grad_out, float *grad_x, float *grad_w_l, float *grad_b_l,
                       float *grad_w_r, void *scratch, size_t scratch_bytes, const int32_t *dims, void *stream);
/* The indexed pair without a matrix of the destination rows.  sage_conv_forward_indexed accepts x_dst = NULL (scratch:
 * sage_conv_forward_indexed_scratch_bytes), and sage_conv_backward_indexed computes the same gradients as sage_conv_backward
 * reading x_dst[i] = feats[n_id[i]] through n_id (the weight-gradient kernel's loader follows the index; kernel paths that
 * cannot build the rows in the scratch tail).  The input features get no gradient.  main.py:118-123, 206. */
size_t sage_conv_forward_indexed_scratch_bytes(int64_t n_dst, int32_t c_in, int32_t c_out);
size_t sage_conv_backward_indexed_scratch_bytes(int64_t n_src, int64_t n_dst, int64_t nnz, int32_t c_in, int32_t c_out);
int sage_conv_backward_indexed(const int32_t *rowptr, const int32_t *col, const int64_t *n_id, int64_t n_src, int64_t n_dst,
                               int64_t nnz, const float *feats, int64_t n_rows, const float *agg, int32_t c_in, const float *w_l,
                               const float *w_r, int32_t c_out, const float *grad_out, float *grad_w_l, float *grad_b_l,
                               float *grad_w_r, void *scratch, size_t scratch_bytes, const int32_t *dims, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fan-out neighbour sampling on the device (one hop of PyG NeighborSampler / torch_sparse.sample_adj, main.py:100-116).
 *   rowptr / col    int32 CSR of the whole graph by TARGET node (row v lists the nodes whose features v aggregates;
 *                   adj_t in the reference).  For the symmetric Flickr / PubMed graphs this is pope_csr_build's CSR.
 *   targets         int64 [n_targets] distinct node ids on the device.
 *   fanout          > 0: rows with more neighbours keep `fanout` distinct ones, chosen by a keyed pseudo-random
 *                   permutation that is a pure function of (seed, hop, node); < 0: keep all neighbours.
 *   out_rowptr      int32 [n_targets + 1], out_col int32 [nnz_capacity]: the sampled block, LOCAL ids.
 *   out_n_id        int64 [n_targets + nnz_capacity]: targets (same order) followed by the newly met nodes in order
 *                   of first appearance; out_col indexes into it.
 *   nnz_host / n_src_host   (out, host) edges kept and length of out_n_id.
 * nnz_capacity >= n_targets * fanout (fanout > 0) or the sum of the targets' degrees (fanout < 0).
 * Synchronises `stream` once (the two counts come back to the host).
 * ------------------------------------------------------------------------------------------------ */
size_t sage_sample_scratch_bytes(int64_t N, int64_t n_targets, int64_t nnz_capacity);
int sage_sample_hop(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *targets, int64_t n_targets,
                    int32_t fanout, uint64_t seed, int32_t hop, int32_t *out_rowptr, int32_t *out_col, int64_t nnz_capacity,
                    int64_t *out_n_id, int64_t *nnz_host, int64_t *n_src_host, void *scratch, size_t scratch_bytes,
                    void *stream);

/*
 * Every hop of a mini-batch in one call: hop h samples `fanouts_host[h]` neighbours around hop h-1's node list (hop 0:
 * `seeds`), exactly as n_hops successive sage_sample_hop calls with hop = h would.  out_rowptr / out_col / out_n_id:
 * HOST arrays of n_hops device pointers sized by capacity -- with t_cap[0] = n_seeds and
 * t_cap[h] = t_cap[h-1] * (1 + fanouts[h-1]):  out_rowptr[h] t_cap[h] + 1 ints, out_col[h] t_cap[h] * fanouts[h] ints,
 * out_n_id[h] t_cap[h] * (1 + fanouts[h]) int64.  nnz_host / n_src_host: (out, host) [n_hops].
 * scratch: sage_sample_scratch_bytes(N, t_cap[last], t_cap[last] * fanouts[last]).  Synchronises `stream` once per hop.
 */
int sage_sample_batch(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *seeds, int64_t n_seeds,
                      const int32_t *fanouts_host, int32_t n_hops, uint64_t seed, int32_t *const *out_rowptr,
                      int32_t *const *out_col, int64_t *const *out_n_id, int64_t *nnz_host, int64_t *n_src_host, void *scratch,
                      size_t scratch_bytes, void *stream);

/*
 * sage_sample_batch WITHOUT any host synchronisation (the device-extent form): every hop runs over its capacity and reads
 * the true target count of the hop before it on the device.  dims (out, device): int32 [n_hops][4], dims[h] =
 * {n_dst, n_src, nnz, 0} of hop h -- what the SAGE entry points take as `dims`; out_rowptr[h] rows past n_dst are empty,
 * out_col[h] / out_n_id[h] entries past nnz / n_src are unspecified.  seed_dev (device uint64, or NULL) is ADDED to `seed`
 * on the device: a captured launch follows that word (sage_advance_counters).  Buffers and scratch as sage_sample_batch.
 */
int sage_sample_batch_device(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *seeds, int64_t n_seeds,
                             const int32_t *fanouts_host, int32_t n_hops, uint64_t seed, const uint64_t *seed_dev,
                             int32_t *const *out_rowptr, int32_t *const *out_col, int64_t *const *out_n_id, int32_t *dims,
                             void *scratch, size_t scratch_bytes, void *stream);
/* The next batch of an epoch without a loader (main.py:100-123): seeds = order[*first_dev .. + n_seeds) with `order` the epoch's
 * (shuffled) node list and `first_dev` a device word the caller advances by n_seeds per step; with `labels` (and y_out)
 * y_out[i] = labels[seed i] (main.py:122).  Everything else as sage_sample_batch_device. */
int sage_sample_epoch_batch_device(const int32_t *rowptr, const int32_t *col, int64_t N, const int64_t *order, const int64_t *first_dev,
                                   int64_t n_seeds, const int64_t *labels, int64_t *y_out, const int32_t *fanouts_host, int32_t n_hops,
                                   uint64_t seed, const uint64_t *seed_dev, int32_t *const *out_rowptr, int32_t *const *out_col,
                                   int64_t *const *out_n_id, int32_t *dims, void *scratch, size_t scratch_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Hidden-layer epilogue: BatchNorm1d + ReLU + dropout as one op  (main.py:207-209)
 *
 * Replaces   x = self.bns[i](x); x = x.relu_(); x = F.dropout(x, p=self.dropout, training=self.training)
 * and its autograd.  x, y and the grad_x / grad_y matrices are [M, C] float32 row-major on the device; gamma, beta, the running and
 * the saved statistics and grad_gamma / grad_beta are [C].
 *   training != 0: batch statistics (biased variance), running_mean / running_var updated in place with `momentum`
 *                  (unbiased variance), as torch.nn.BatchNorm1d; both running pointers may be NULL (track_running_stats=False).
 *                  Dropout keeps an element with probability 1 - p and scales it by 1 / (1 - p); the mask is a pure
 *                  function of (seed, element index), recomputed in backward: no mask tensor.
 *   training == 0: running statistics, no dropout.
 * save_mean / save_rstd (out) feed the backward call.  grad_gamma / grad_beta may be NULL.  Asynchronous.
 * rows_dev (device int32, or NULL): the true row count, M then being the capacity ("Device extents" above).
 * seed_dev (device uint64, or NULL): added to `seed` on the device (a replayed HIP graph draws a new mask per replay).
 * scratch: sage_bn_scratch_bytes(C) bytes.
 * ------------------------------------------------------------------------------------------------ */
size_t sage_bn_scratch_bytes(int32_t C);
int sage_bn_relu_dropout_forward(const float *x, int64_t M, int32_t C, const float *gamma, const float *beta,
                                 float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum, float eps, int32_t training,
                                 float p, uint64_t seed, float *y, float *save_mean, float *save_rstd, void *scratch,
                                 size_t scratch_bytes, const int32_t *rows_dev, const uint64_t *seed_dev, void *stream);
/* The forward pass with the first stage of the statistics taken from the kernel that wrote x (sage_conv_forward_stats: pa / pb
 * [parts, C] float64, one part per rows_per_part rows; with rows_dev only the parts in front of the true row count are read): two
 * launches instead of three.  Same arithmetic otherwise (float64 sums; the order of the additions differs from the three-launch form). */
int sage_bn_relu_dropout_forward_stats(const float *x, int64_t M, int32_t C, const float *gamma, const float *beta,
                                       float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum, float eps,
                                       int32_t training, float p, uint64_t seed, float *y, float *save_mean, float *save_rstd, void *scratch,
                                       size_t scratch_bytes, const int32_t *rows_dev, const uint64_t *seed_dev, const double *pa,
                                       const double *pb, int32_t parts, int32_t rows_per_part, void *stream);
int sage_bn_relu_dropout_backward(const float *x, const float *grad_y, int64_t M, int32_t C, const float *gamma,
                                  const float *beta, const float *save_mean, const float *save_rstd, int32_t training,
                                  float p, uint64_t seed, float *grad_x, float *grad_gamma, float *grad_beta, void *scratch,
                                  size_t scratch_bytes, const int32_t *rows_dev, const uint64_t *seed_dev, void *stream);
/* The backward pass, and out of its float64 sums the column sums of grad_x (grad_x_colsum, float32 [C]): the bias gradient of the layer
 * that produced x (main.py:206-207), which sage_conv_backward otherwise computes by reading grad_x back (pass it grad_b_l = NULL). */
int sage_bn_relu_dropout_backward_bias(const float *x, const float *grad_y, int64_t M, int32_t C, const float *gamma,
                                       const float *beta, const float *save_mean, const float *save_rstd, int32_t training,
                                       float p, uint64_t seed, float *grad_x, float *grad_gamma, float *grad_beta, void *scratch,
                                       size_t scratch_bytes, const int32_t *rows_dev, const uint64_t *seed_dev, float *grad_x_colsum,
                                       void *stream);

/*
 * One Adam step over every parameter tensor in a single launch  (main.py:244: torch.optim.Adam(self.parameters(), lr)).
 * params / grads / exp_avg / exp_avg_sq: HOST arrays of n_tensors device pointers (float32, contiguous, numel[t] elements).
 * Update rule of torch.optim.Adam (amsgrad off): decoupled nothing, weight_decay added to the gradient, bias
 * corrections from `step` (1-based, the step being taken).  Hyper-parameters are doubles: 1 - beta, lr / (1 - beta1^t)
 * and sqrt(1 - beta2^t) are formed in double and rounded to float once, as torch does.  Asynchronous.
 * step_dev (device int64, or NULL): the step count read on the device instead of `step` (replayed HIP graphs).
 */
int sage_adam_step(int32_t n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                   float *const *exp_avg_sq, const int64_t *numel, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step, const int64_t *step_dev, void *stream);
/* The same step with the LAST STAGE OF THIS STEP'S CROSS-ENTROPY as one more block of the same launch (main.py:216 + 244: the loss is a
 * number the host logs after the step, it need not hold up the backward pass): xent_rows = the row_scratch that
 * sage_cross_entropy_forward(fused = 2) filled, xent_n its row count; loss_out[0] = the mean loss, loss_out[1] = 1 / count. */
int sage_adam_step_loss(int32_t n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                        float *const *exp_avg_sq, const int64_t *numel, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int64_t step, const int64_t *step_dev, const float *xent_rows, int64_t xent_n,
                        float *loss_out, void *stream);

/*
 * Plumbing of a training step replayed as a HIP graph (main.py:213-222 training_step + Lightning's backward / optimizer
 * step, with no host in the loop).  sage_advance_counters: counters[i] += increments_host[i] for i < n <= 8, one launch
 * (the dropout / sampling seeds and Adam's step count live in device words).  sage_copy_segments: n device-to-device
 * copies in ONE launch (a batch of a pre-sampled pool into the step's fixed buffers); dst / src / bytes are HOST arrays.
 */
int sage_advance_counters(int64_t *counters, const int64_t *increments_host, int32_t n, void *stream);
int sage_copy_segments(int32_t n, void *const *dst, const void *const *src, const int64_t *bytes, void *stream);

/*
 * Cross-entropy with integer labels, mean over the rows whose label is not ignore_index  (main.py:216, 224, 233:
 * F.cross_entropy(y_hat, y)).  Forward writes the scalar loss, 1 / (number of counted rows) and the UNSCALED gradient
 * softmax(logits) - onehot(target) [N, C] (zero rows for ignored labels); backward multiplies it by the upstream scalar
 * gradient and by 1 / count, both read on the device.  row_scratch: N floats.  *bad_label (device int, zeroed by the
 * caller) is set when a label lies outside [0, C) and is not ignore_index.  Asynchronous.
 * fused != 0: ONE launch for the whole forward pass, and grad_unscaled then holds the gradient ALREADY SCALED by 1 / count,
 * i.e. the gradient of the mean loss for an upstream gradient of 1 -- a training step that seeds its backward pass with 1
 * (loss.backward()) needs no backward launch at all; sage_cross_entropy_backward must not be applied to it.
 * fused == 2: as fused == 1, but the scalar loss and 1 / count are NOT written: the caller hands row_scratch to sage_adam_step_loss,
 * whose launch finishes them (one launch less per training step).
 */
int sage_cross_entropy_forward(const float *logits, const int64_t *target, int64_t N, int32_t C, int64_t ignore_index,
                               float *loss, float *grad_unscaled, float *inv_count, float *row_scratch, int32_t *bad_label,
                               int32_t fused, void *stream);
int sage_cross_entropy_backward(const float *grad_unscaled, int64_t N, int32_t C, const float *grad_loss, const float *inv_count,
                                float *grad_logits, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHPOPE_HIP_H */
