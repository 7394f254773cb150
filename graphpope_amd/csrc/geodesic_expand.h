// Hop planes -> the embedding: the finalise kernels (generic, rounds 1-3, pipelined, shuffle, LDS tables), the hop matrix, the byte
// codes of the host boundary, the column statistics and the feature copy.  Not a header in its own right: csrc/geodesic.hip includes
// it inside namespace pope, behind the control block and the CSR status helpers (BfsCtl, csr_flags).
// Replaces utils.py:73, 116-135 (1 / (hops + 1) next to the features): DESIGN.md section 3, lessons 18 and 20.
#pragma once

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
