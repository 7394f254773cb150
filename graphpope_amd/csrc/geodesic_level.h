// The level kernel of the multi-source BFS (k_bfs_level<WT, LIVE, TILES>) and its helpers: word tiles, DPP moves, the live-bit
// tables, housekeeping and expand roles.  Not a header in its own right: csrc/geodesic.hip includes it inside namespace pope, behind
// the control block and the CSR status helpers it uses (BfsCtl, bfs_over, raise_level, AUX_*, CHUNK, SLOTS, the POPE_* macros).
// Replaces utils.py:64-81 (nx.shortest_path per node and anchor): DESIGN.md section 3.
#pragma once

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
    constexpr bool AHEAD = (LIVE >= 2 || (WT == 8 && POPE_L2_LOOP != 0)) && POPE_AHEAD != 0;
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
__global__ __launch_bounds__(256, WT == 8 ? (TILES == 1 ? POPE_WT8_LOOP_WAVES : (LIVE == 1 ? POPE_WT8_L2_WAVES : POPE_WT8_WAVES)) : 1) void k_bfs_level(const int *__restrict__ erow, const int *__restrict__ col,
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
