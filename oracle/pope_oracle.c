/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the GraphPOPE geodesic embedding, used only as the parity checker
 * (tests/, __graft_entry__.smoke()) and as bench.py's `cpu_baseline` leg.  Nothing under
 * graphpope_amd/ may import, link or call this file.
 *
 * Parity status: PINNED.  tests/golden/geodesic_*.npz were produced by running the reference's own
 * utils.Graphpope / get_geodesic_distance_vector in the build container (tests/golden/make_goldens.py)
 * and tests/test_oracle.py checks this file against every one of them bit-for-bit.
 *
 * What it restates (file:line into /root/reference):
 *   utils.py:64-81   shortest_path_length      value(node, anchor) = 1/len(shortest_path(node -> anchor)),
 *                                               NetworkXNoPath -> 0.   len() counts NODES, so value = 1/(hops+1).
 *   utils.py:92-107  all_pairs_..._parallel     row i of the result is node i (slices merged in node order).
 *   utils.py:116-126 get_geodesic_distance_vector  result tensor is float32 [N, K], column j <-> anchors[j].
 *   utils.py:129-135 concat_into_features       out = cat(x, emb) along dim 1 -> [N, F+K] float32.
 *
 * The reference runs one bidirectional BFS per (node, anchor) pair.  Hop counts are unique, so one BFS
 * per anchor over the REVERSED edges (distance is measured node -> anchor along edge direction) gives
 * the identical matrix; that is what this file does.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Reverse CSR: for every node u, the list of v with an edge v -> u. */
static int build_reverse_csr(const int64_t *edge_index, int64_t E, int64_t N,
                             int64_t **rowptr_out, int32_t **col_out)
{
    const int64_t *src = edge_index, *dst = edge_index + E;
    int64_t *rowptr = (int64_t *)calloc((size_t)N + 1, sizeof(int64_t));
    int32_t *col = (int32_t *)malloc((size_t)(E > 0 ? E : 1) * sizeof(int32_t));
    int64_t *fill;
    if (!rowptr || !col) { free(rowptr); free(col); return -1; }
    for (int64_t e = 0; e < E; ++e) {
        if (src[e] < 0 || src[e] >= N || dst[e] < 0 || dst[e] >= N) { free(rowptr); free(col); return -2; }
        rowptr[dst[e] + 1]++;
    }
    for (int64_t i = 0; i < N; ++i) rowptr[i + 1] += rowptr[i];
    fill = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    if (!fill) { free(rowptr); free(col); return -1; }
    memcpy(fill, rowptr, (size_t)N * sizeof(int64_t));
    for (int64_t e = 0; e < E; ++e) col[fill[dst[e]]++] = (int32_t)src[e];
    free(fill);
    *rowptr_out = rowptr;
    *col_out = col;
    return 0;
}

/*
 * hops[v*K + j] = number of edges on a shortest directed path v -> anchors[j], or -1 if none.
 * Duplicate anchors give duplicate columns; an anchor's own entry is 0.
 * Returns 0, -1 (out of memory) or -2 (an index outside [0, N)).
 */
int oracle_geodesic_hops(const int64_t *edge_index, int64_t E, int64_t N,
                         const int64_t *anchors, int32_t K, int32_t *hops)
{
    int64_t *rowptr = NULL;
    int32_t *col = NULL, *queue = NULL, *dist = NULL;
    int rc = build_reverse_csr(edge_index, E, N, &rowptr, &col);
    if (rc) return rc;
    queue = (int32_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int32_t));
    dist = (int32_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int32_t));
    if (!queue || !dist) { rc = -1; goto done; }
    for (int32_t j = 0; j < K; ++j) {
        int64_t a = anchors[j], head = 0, tail = 0;
        if (a < 0 || a >= N) { rc = -2; goto done; }
        for (int64_t v = 0; v < N; ++v) dist[v] = -1;
        dist[a] = 0;
        queue[tail++] = (int32_t)a;
        while (head < tail) {
            int32_t u = queue[head++];
            for (int64_t p = rowptr[u]; p < rowptr[u + 1]; ++p) {
                int32_t v = col[p];
                if (dist[v] < 0) { dist[v] = dist[u] + 1; queue[tail++] = v; }
            }
        }
        for (int64_t v = 0; v < N; ++v) hops[v * K + j] = dist[v];
    }
done:
    free(rowptr); free(col); free(queue); free(dist);
    return rc;
}

/*
 * utils.py:73,125: Python evaluates 1/len(path) in float64, the list of Python floats is then cast
 * to float32 by torch.as_tensor.  Unreachable is the integer 0 -> 0.0f.
 */
void oracle_hops_to_embedding(const int32_t *hops, int64_t n_entries, float *emb)
{
    for (int64_t i = 0; i < n_entries; ++i)
        emb[i] = hops[i] < 0 ? 0.0f : (float)(1.0 / (double)(hops[i] + 1));
}

/* utils.py:134: torch.cat((data.x, embedding), 1). */
void oracle_concat(const float *x, int64_t N, int64_t F, const float *emb, int64_t K, float *out)
{
    for (int64_t v = 0; v < N; ++v) {
        memcpy(out + v * (F + K), x + v * F, (size_t)F * sizeof(float));
        memcpy(out + v * (F + K) + F, emb + v * K, (size_t)K * sizeof(float));
    }
}
