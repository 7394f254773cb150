// Host-side half of the drop-in boundary (no device code): assembling the reference's [N, F+K] result tensor on the
// host cores while the GPU computes / ships only the K embedding columns.
//
// Replaces /root/reference/utils.py:129-135 concat_into_features -> torch.cat((data.x, embedding), 1) for the
// host -> host call: the features never cross PCIe (they do not change); they are copied once, host to host, by a
// few threads with streaming stores, underneath the GPU work.
//
// pope_assemble_host_result (round 3) does that INTO AN ORDINARY PAGEABLE TENSOR, like the one the reference returns.
// The reference calls Graphpope once per process (utils.py:195-208 memoises), so what counts is the first call: a
// 270 MB hipHostMalloc'd result took 34 ms to allocate and stayed page-locked for the life of the process.  Here worker
// threads copy data.x into the result (first touch: with MADV_HUGEPAGE the faults are 2 MB each) while the K embedding
// columns arrive, chunk by chunk, in a small pinned ring of the library's own, and the same threads copy each landed chunk
// out.  The HIP runtime is never handed a page of the result.  (Round 3's other transport -- registering the result's own pages
// chunk by chunk for pitched DMA -- measured slower on first and repeated calls, was where every host-side fault of rounds 3-4
// lived, and was removed in round 5 together with the pre-fault helper threads: DESIGN.md section 1 keeps the figures.)
#include <emmintrin.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "graphpope_hip.h"

namespace pope {
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();
extern int g_fail_host_register;         // pope_debug_set(POPE_KNOB_FAIL_HOST_REGISTER): tests of the fallback path
}  // namespace pope

namespace {

// One row segment of `bytes` bytes; dst 16-byte aligned segments go out as non-temporal stores (the destination is
// written once and not read back by these threads: no read-for-ownership traffic, the caches keep the sources).
inline void copy_segment(const char *src, char *dst, size_t bytes) {
    size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
    if (head > bytes) head = bytes;
    if (head) {
        memcpy(dst, src, head);
        src += head; dst += head; bytes -= head;
    }
    const size_t vec = bytes / 64 * 64;
    for (size_t o = 0; o < vec; o += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + o));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + o + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + o + 32));
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + o + 48));
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + o), a);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + o + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + o + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + o + 48), d);
    }
    if (bytes > vec) memcpy(dst + vec, src + vec, bytes - vec);
}

void copy_rows(const char *src, size_t src_pitch, char *dst, size_t dst_pitch, size_t row_bytes, int64_t r0, int64_t r1) {
    if (src_pitch == row_bytes && dst_pitch == row_bytes) {             // contiguous on both sides: one long segment
        copy_segment(src + (size_t)r0 * row_bytes, dst + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes);
    } else {
        for (int64_t r = r0; r < r1; ++r) copy_segment(src + (size_t)r * src_pitch, dst + (size_t)r * dst_pitch, row_bytes);
    }
    _mm_sfence();
}

void copy_bytes(const char *src, char *dst, size_t b0, size_t b1) {
    copy_segment(src + b0, dst + b0, b1 - b0);
    _mm_sfence();
}

int clamp_threads(int32_t threads) {
    int t = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (t < 1) t = 1;
    if (t > 64) t = 64;
    return t;
}

}  // namespace

extern "C" int pope_host_copy_2d(const void *src_host, int64_t src_pitch_bytes, void *dst_host, int64_t dst_pitch_bytes,
                                 int64_t row_bytes, int64_t rows, int32_t threads) {
    if (!src_host || !dst_host || row_bytes < 0 || rows < 0 || src_pitch_bytes < row_bytes || dst_pitch_bytes < row_bytes)
        return POPE_ERR_INVALID;
    if (rows == 0 || row_bytes == 0) return POPE_OK;
    const size_t total = (size_t)rows * (size_t)row_bytes;
    int t = clamp_threads(threads);
    if (total < ((size_t)1 << 20)) t = 1;                               // not worth a thread below 1 MiB
    const char *s = static_cast<const char *>(src_host);
    char *d = static_cast<char *>(dst_host);
    const bool flat = src_pitch_bytes == row_bytes && dst_pitch_bytes == row_bytes;
    if (flat) {
        // one contiguous run (a flat buffer arrives as ONE row): split by byte ranges, 4 KiB aligned, not by rows
        if ((size_t)t > total / 4096) t = (int)(total / 4096 > 0 ? total / 4096 : 1);
        if (t == 1) {
            copy_bytes(s, d, 0, total);
            return POPE_OK;
        }
        const size_t per = (total / (size_t)t + 4095) / 4096 * 4096;
        std::vector<std::thread> pool;
        pool.reserve((size_t)t - 1);
        for (int i = 1; i < t; ++i) {
            const size_t b0 = (size_t)i * per, b1 = b0 + per < total ? b0 + per : total;
            if (b0 >= total) break;
            pool.emplace_back(copy_bytes, s, d, b0, b1);
        }
        copy_bytes(s, d, 0, per < total ? per : total);
        for (auto &th : pool) th.join();
        return POPE_OK;
    }
    if ((int64_t)t > rows) t = (int)rows;
    if (t == 1) {
        copy_rows(s, (size_t)src_pitch_bytes, d, (size_t)dst_pitch_bytes, (size_t)row_bytes, 0, rows);
        return POPE_OK;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)t - 1);
    const int64_t per = (rows + t - 1) / t;
    for (int i = 1; i < t; ++i) {
        const int64_t r0 = i * per, r1 = r0 + per < rows ? r0 + per : rows;
        if (r0 >= rows) break;
        pool.emplace_back(copy_rows, s, (size_t)src_pitch_bytes, d, (size_t)dst_pitch_bytes, (size_t)row_bytes, r0, r1);
    }
    copy_rows(s, (size_t)src_pitch_bytes, d, (size_t)dst_pitch_bytes, (size_t)row_bytes, 0, per < rows ? per : rows);
    for (auto &th : pool) th.join();
    return POPE_OK;
}

// ------------------------------------------------------------------------------------------------
// Caller-owned pageable memory as a DMA endpoint for the length of one call
// ------------------------------------------------------------------------------------------------
// Round 4: only the WHOLE pages inside [host, host + bytes) are registered -- the partial pages at the two ends of a heap
// allocation also hold other objects (the allocator's headers, neighbouring tensors), and registering those registers memory
// this library knows nothing about.  pope_copy_to_device sends the two fragments (under a page each) as plain pageable copies,
// which the runtime stages through its own buffers, and the body by DMA from the registered pages.
namespace {
struct PinnedRange { uintptr_t lo, hi; };
std::mutex g_pin_mu;
std::vector<std::pair<const void *, PinnedRange>> g_pins;          // live registrations made by pope_host_pin, by the caller's pointer
}  // namespace

extern "C" int pope_host_pin(const void *host, size_t bytes) {
    pope::clear_error();
    if (!host || bytes == 0) {
        pope::set_error("pope_host_pin: null pointer or zero size");
        return POPE_ERR_INVALID;
    }
    if (pope::g_fail_host_register & 1) {
        pope::set_error("pope_host_pin: refused (POPE_KNOB_FAIL_HOST_REGISTER)");
        return POPE_ERR_HIP;
    }
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE), b = reinterpret_cast<uintptr_t>(host);
    const uintptr_t lo = (b + page - 1) & ~(page - 1), hi = (b + bytes) & ~(page - 1);
    if (hi <= lo || hi - lo < ((uintptr_t)1 << 20)) {
        pope::set_error("pope_host_pin: fewer than 1 MB of whole pages inside the buffer: stage it instead");
        return POPE_ERR_HIP;
    }
    const hipError_t e = hipHostRegister(reinterpret_cast<void *>(lo), hi - lo, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pope::set_error("hipHostRegister(%zu bytes) failed: %s", (size_t)(hi - lo), hipGetErrorString(e));
        return POPE_ERR_HIP;
    }
    std::lock_guard<std::mutex> lock(g_pin_mu);
    g_pins.emplace_back(host, PinnedRange{lo, hi});
    return POPE_OK;
}

extern "C" int pope_host_unpin(const void *host) {
    pope::clear_error();
    if (!host) return POPE_ERR_INVALID;
    PinnedRange r{0, 0};
    {
        std::lock_guard<std::mutex> lock(g_pin_mu);
        for (size_t i = 0; i < g_pins.size(); ++i)
            if (g_pins[i].first == host) {
                r = g_pins[i].second;
                g_pins.erase(g_pins.begin() + (long)i);
                break;
            }
    }
    if (!r.hi) {
        pope::set_error("pope_host_unpin: %p was not pinned by pope_host_pin", host);
        return POPE_ERR_INVALID;
    }
    const hipError_t e = hipHostUnregister(reinterpret_cast<void *>(r.lo));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pope::set_error("hipHostUnregister failed: %s", hipGetErrorString(e));
        return POPE_ERR_HIP;
    }
    return POPE_OK;
}

extern "C" int pope_copy_to_device(const void *src_host, void *dst, size_t bytes, void *stream) {
    pope::clear_error();
    if (!src_host || !dst) {
        pope::set_error("pope_copy_to_device: null pointer");
        return POPE_ERR_INVALID;
    }
    if (bytes == 0) return POPE_OK;
    PinnedRange r{0, 0};
    {
        std::lock_guard<std::mutex> lock(g_pin_mu);
        for (auto &p : g_pins)
            if (p.first == src_host) r = p.second;
    }
    const uintptr_t b = reinterpret_cast<uintptr_t>(src_host), e_ = b + bytes;
    auto piece = [&](uintptr_t lo, uintptr_t hi) -> int {
        if (hi <= lo) return POPE_OK;
        const hipError_t e = hipMemcpyAsync(static_cast<char *>(dst) + (lo - b), reinterpret_cast<const void *>(lo), hi - lo, hipMemcpyHostToDevice, (hipStream_t)stream);
        if (e != hipSuccess) {
            pope::set_error("hipMemcpyAsync(H2D, %zu bytes) failed: %s", (size_t)(hi - lo), hipGetErrorString(e));
            return POPE_ERR_HIP;
        }
        return POPE_OK;
    };
    if (r.hi && r.lo >= b && r.hi <= e_) {             // pinned by pope_host_pin: the body from the registered pages, the two ends apart
        int rc = piece(b, r.lo);
        if (!rc) rc = piece(r.lo, r.hi);
        if (!rc) rc = piece(r.hi, e_);
        return rc;
    }
    return piece(b, e_);
}

// ------------------------------------------------------------------------------------------------
// out[:, :F] = x (host cores) and out[:, F:] = emb (DMA), into a pageable result, chunk by chunk
// ------------------------------------------------------------------------------------------------
// Phase times of the last pope_assemble_host_result call of the process, in ms (diagnostic: tools/boundary_breakdown.py):
// 0 madvise, 1 waiting for the chunks' host copies, 2 the ring's allocation (first call), 3 enqueueing DMAs, 4 joining the workers,
// 5 waiting for the stream, 6 unused, 7 total.
static double g_assemble_trace[8];
static std::mutex g_assemble_trace_mu;       // two assemblies may finish at the same time (found by tools/host_race_harness under -fsanitize=thread)
extern "C" void pope_debug_boundary_trace(double *host8) {
    std::lock_guard<std::mutex> lock(g_assemble_trace_mu);
    for (int i = 0; i < 8; ++i) host8[i] = g_assemble_trace[i];
}
static void publish_trace(const double *tr) {
    std::lock_guard<std::mutex> lock(g_assemble_trace_mu);
    for (int i = 0; i < 8; ++i) g_assemble_trace[i] = tr[i];
}

namespace {

inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// dst[0 .. n) (floats, 4-byte aligned) = the values the code bytes stand for.  `pair` maps TWO code bytes (low byte first) to
// their two floats, so one 8-byte load serves two elements; only the entries of codes that occur are ever touched (hop
// counts are small), so the lookups stay in L1.  Streaming stores once dst is 16-byte aligned, like copy_segment.
inline void expand_codes(const unsigned char *src, char *dst_, size_t n, const uint64_t *pair) {
    float *dst = reinterpret_cast<float *>(dst_);
    const float *single = reinterpret_cast<const float *>(pair);                  // pair[c] for c < 256 is (lut[c], lut[0]): element 2c is lut[c]
    size_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 15u)) { dst[i] = single[2 * (size_t)src[i]]; ++i; }
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, src + i, 8);
        const __m128i lo = _mm_set_epi64x((long long)pair[(w >> 16) & 0xFFFFu], (long long)pair[w & 0xFFFFu]);
        const __m128i hi = _mm_set_epi64x((long long)pair[(w >> 48) & 0xFFFFu], (long long)pair[(w >> 32) & 0xFFFFu]);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i), lo);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 4), hi);
    }
    for (; i < n; ++i) dst[i] = single[2 * (size_t)src[i]];
}

struct Assembly {
    const char *x = nullptr;
    size_t x_pitch = 0, x_row = 0;
    char *out = nullptr;
    size_t out_pitch = 0;
    int64_t rows = 0;
    int chunks = 0, slices = 0;                  // every chunk is cut into `slices` row slices, one work item each
    std::vector<int64_t> chunk_lo;               // chunks + 1 row boundaries
    std::atomic<int> next{0};
    std::vector<std::atomic<int>> done;          // per chunk: slices finished
    // ring mode (pope_assemble_finish): the embedding columns arrive in a small pinned ring, chunk by chunk, and the same
    // threads copy them out.  Items (ring chunk e / slices, slice e % slices) become takable once the chunk is published.
    std::atomic<int> emb_total{-1};              // number of embedding items, -1 until finish() knows it
    std::atomic<int> emb_next{0}, emb_published{0};
    std::atomic<bool> aborted{false};
    std::vector<std::atomic<int>> emb_done;      // per ring chunk: slices copied out
    const char *ring[4] = {nullptr, nullptr, nullptr, nullptr};
    int ring_slots = 0;
    int64_t emb_chunk_rows = 0;
    size_t emb_row = 0;                          // bytes of one row in the ring
    const uint64_t *pair_lut = nullptr;          // hop-code transport: the ring holds one byte per element, this table the floats
    size_t emb_elems = 0;                        //   ... elements (bytes) per row
    // Threads that run out of feature items before finish() has said how many embedding items there will be (an assembly begun
    // BEFORE the upload and the GPU work) sleep here instead of spinning through the whole GPU phase; begin_embedding() /
    // abort() wake them.  Once the embedding phase has begun the waits are short (a chunk is on the bus) and they yield-spin.
    std::mutex idle_m;
    std::condition_variable idle_cv;

    explicit Assembly(int nchunks) : done((size_t)nchunks), emb_done(0) {
        for (auto &d : done) d.store(0, std::memory_order_relaxed);
    }

    void x_item(int it) {
        const int c = it / slices, s = it % slices;
        const int64_t lo = chunk_lo[(size_t)c], n = chunk_lo[(size_t)c + 1] - lo;
        const int64_t r0 = lo + n * s / slices, r1 = lo + n * (s + 1) / slices;
        if (r1 > r0) {
            if (x_row) {
                for (int64_t r = r0; r < r1; ++r) copy_segment(x + (size_t)r * x_pitch, out + (size_t)r * out_pitch, x_row);
                _mm_sfence();
            }
        }
        done[(size_t)c].fetch_add(1, std::memory_order_release);
    }

    void emb_item(int e) {
        const int c = e / slices, s = e % slices;
        const int64_t lo = (int64_t)c * emb_chunk_rows, hi = lo + emb_chunk_rows < rows ? lo + emb_chunk_rows : rows, n = hi - lo;
        const int64_t r0 = lo + n * s / slices, r1 = lo + n * (s + 1) / slices;
        const char *src = ring[c % ring_slots];
        if (pair_lut) {
            for (int64_t r = r0; r < r1; ++r)
                expand_codes(reinterpret_cast<const unsigned char *>(src) + (size_t)(r - lo) * emb_row, out + (size_t)r * out_pitch + x_row, emb_elems, pair_lut);
        } else {
            for (int64_t r = r0; r < r1; ++r) copy_segment(src + (size_t)(r - lo) * emb_row, out + (size_t)r * out_pitch + x_row, emb_row);
        }
        _mm_sfence();
        emb_done[(size_t)c].fetch_add(1, std::memory_order_release);
    }

    void begin_embedding(int total_items) {
        {
            std::lock_guard<std::mutex> lock(idle_m);
            emb_total.store(total_items, std::memory_order_release);
        }
        idle_cv.notify_all();
    }

    void abort() {
        {
            std::lock_guard<std::mutex> lock(idle_m);
            aborted.store(true, std::memory_order_release);
        }
        idle_cv.notify_all();
    }

    void work() {
        const int items = chunks * slices;
        for (;;) {
            const int tot = emb_total.load(std::memory_order_acquire);
            if (tot >= 0) {                      // a landed ring chunk first: its slot is what the next DMA waits for
                int e = emb_next.load(std::memory_order_relaxed);
                if (e < tot && e / slices < emb_published.load(std::memory_order_acquire)) {
                    if (emb_next.compare_exchange_weak(e, e + 1, std::memory_order_relaxed)) emb_item(e);
                    continue;
                }
            }
            if (next.load(std::memory_order_relaxed) < items) {
                const int it = next.fetch_add(1, std::memory_order_relaxed);
                if (it < items) x_item(it);
                continue;
            }
            if (aborted.load(std::memory_order_acquire)) return;
            if (tot >= 0 && emb_next.load(std::memory_order_relaxed) >= tot) return;
            if (tot < 0) {                       // nothing to do until finish() (or abort()): sleep, do not spin
                std::unique_lock<std::mutex> lock(idle_m);
                idle_cv.wait(lock, [&] { return emb_total.load(std::memory_order_acquire) >= 0 || aborted.load(std::memory_order_acquire); });
                continue;
            }
            std::this_thread::yield();
        }
    }
};

// The pinned ring the embedding columns travel through: allocated once per process (24 MB), guarded by a mutex that a
// finishing assembly holds (concurrent host -> host calls of one process take turns in this last phase).
constexpr int RING_SLOTS = 3;
constexpr size_t RING_SLOT_BYTES = (size_t)8 << 20;
struct PinnedRing {
    std::mutex mu;
    char *slot[RING_SLOTS] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[RING_SLOTS] = {nullptr, nullptr, nullptr};
    bool ready = false, failed = false, preparing = false;
    float *lut_pinned = nullptr;                 // hop-code transport: where the device's 256 floats land ...
    float lut_seen[256];                         // ... the table pair_lut was built from ...
    std::vector<uint64_t> pair_lut;              // ... and the 65 536-entry two-byte table (512 KB, built once: the floats never change)
};
PinnedRing g_ring;

}  // namespace

namespace {

// The threads of an assembly.  One assembly at a time borrows the process's parked workers (created on first use, woken
// through a condition variable: ~0.1 ms for sixteen, where creating them costs 0.3-0.4 ms of a 2 ms call); a second
// assembly running at the same time starts threads of its own, as every assembly did before.  The parked workers are never
// destroyed (detached, their bookkeeping leaked on purpose: no destructor order to get wrong at exit).
struct Parked {
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    Assembly *job = nullptr;
    unsigned long long generation = 0;
    int threads = 0, want = 0, running = 0;
    bool busy = false;
};
Parked *g_parked = new Parked;

void parked_worker(int index, unsigned long long seen) {
    Parked &p = *g_parked;
    std::unique_lock<std::mutex> lock(p.m);
    for (;;) {
        p.cv_job.wait(lock, [&] { return p.generation != seen; });
        seen = p.generation;
        if (index >= p.want) continue;
        Assembly *job = p.job;
        lock.unlock();
        job->work();
        lock.lock();
        if (--p.running == 0) p.cv_done.notify_all();
    }
}

struct Workers {
    std::vector<std::thread> own;
    bool parked = false;

    void start(Assembly *a, int t) {
        Parked &p = *g_parked;
        {
            std::unique_lock<std::mutex> lock(p.m);
            if (!p.busy) {
                p.busy = true;
                while (p.threads < t) {
                    std::thread(parked_worker, p.threads, p.generation).detach();
                    ++p.threads;
                }
                p.job = a; p.want = t; p.running = t;
                ++p.generation;
                parked = true;
            }
        }
        if (parked) {
            p.cv_job.notify_all();
            return;
        }
        own.reserve((size_t)t);
        for (int i = 0; i < t; ++i) own.emplace_back([a] { a->work(); });
    }

    void join() {
        if (parked) {
            Parked &p = *g_parked;
            std::unique_lock<std::mutex> lock(p.m);
            p.cv_done.wait(lock, [&] { return p.running == 0; });
            p.busy = false;
            parked = false;
        }
        for (auto &th : own) th.join();
        own.clear();
    }
};

struct HostAssembly {
    Assembly a;
    Workers pool;
    size_t total = 0;
    double t_begin = 0, t_madvise = 0;
    explicit HostAssembly(int nch) : a(nch) {}
};

}  // namespace

// The work that does not need the embedding -- page faults and the feature copy -- starts here, on `threads` host threads,
// and runs underneath whatever the caller does next (upload edge_index, enqueue and wait for the GPU).
extern "C" void *pope_assemble_begin(const void *x_host, int64_t x_pitch_bytes, int64_t x_row_bytes, void *out_host,
                                     int64_t out_pitch_bytes, int64_t rows, int32_t threads, int32_t chunks) {
    pope::clear_error();
    if (!out_host || rows <= 0 || x_row_bytes < 0 || (x_row_bytes > 0 && (!x_host || x_pitch_bytes < x_row_bytes)) || out_pitch_bytes < x_row_bytes ||
        out_pitch_bytes <= 0) {
        pope::set_error("pope_assemble_begin: null pointer or bad size");
        return nullptr;
    }
    const double t_begin = now_ms();
    char *out = static_cast<char *>(out_host);
    const size_t total = (size_t)rows * (size_t)out_pitch_bytes;
    // 2 MB faults instead of 4 KB ones where the kernel allows it (THP "madvise" or "always"); a refusal changes nothing
    if (!getenv("GRAPHPOPE_NO_HUGEPAGE")) {
        const uintptr_t huge = (uintptr_t)1 << 21, b = reinterpret_cast<uintptr_t>(out);
        const uintptr_t lo = (b + huge - 1) & ~(huge - 1), hi = (b + total) & ~(huge - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    }
    int t = clamp_threads(threads);
    int nch = chunks > 0 ? chunks : 8;
    if (total < ((size_t)8 << 20)) nch = 1;
    if ((int64_t)nch > rows) nch = (int)rows;
    HostAssembly *h = new HostAssembly(nch);
    h->total = total;
    h->t_begin = t_begin;
    h->t_madvise = now_ms() - t_begin;
    Assembly &a = h->a;
    a.x = static_cast<const char *>(x_host); a.x_pitch = (size_t)x_pitch_bytes; a.x_row = (size_t)x_row_bytes;
    a.out = out; a.out_pitch = (size_t)out_pitch_bytes; a.rows = rows; a.chunks = nch;
    a.slices = t;
    a.chunk_lo.resize((size_t)nch + 1);
    // the first chunk is small so that the first DMA starts early; the rest are equal
    for (int c = 0; c <= nch; ++c) a.chunk_lo[(size_t)c] = rows * c / nch;
    if (nch >= 4) a.chunk_lo[1] = rows / (2 * nch);
    h->pool.start(&h->a, t);
    return h;
}

// Give up an assembly that will not be finished (an error between begin and finish): waits for the host threads.
extern "C" void pope_assemble_abort(void *handle) {
    HostAssembly *h = static_cast<HostAssembly *>(handle);
    if (!h) return;
    h->a.abort();
    h->pool.join();
    delete h;
}

// The three pinned slots, the code table and their events; g_ring.mu held.  Allocated once per process (2 ms of hipHostMalloc)
// and never released.  false: no pinned ring -- the runtime refused the 24 MB (not retried), or POPE_KNOB_FAIL_HOST_REGISTER bit
// 1 says to behave as if it had -- and the caller sends float columns through the bounce buffer instead.
// (A ring of ORDINARY memory as the fallback was built and withdrawn: asynchronous copies into heap memory that is freed and
// reused afterwards left the runtime with stale mappings, and a later, unrelated transfer died with a GPU memory fault at a
// host heap address.)
static bool ring_usable_locked() {
    if (pope::g_fail_host_register & 2) return false;
    if (g_ring.ready) return true;
    if (g_ring.failed) return false;
    bool ok = true;
    for (int i = 0; i < RING_SLOTS && ok; ++i)
        ok = hipHostMalloc(reinterpret_cast<void **>(&g_ring.slot[i]), RING_SLOT_BYTES, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&g_ring.ev[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&g_ring.lut_pinned), 256 * sizeof(float), hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (int i = 0; i < RING_SLOTS; ++i) {
            if (g_ring.slot[i]) (void)hipHostFree(g_ring.slot[i]);
            if (g_ring.ev[i]) (void)hipEventDestroy(g_ring.ev[i]);
            g_ring.slot[i] = nullptr; g_ring.ev[i] = nullptr;
        }
        if (g_ring.lut_pinned) (void)hipHostFree(g_ring.lut_pinned);
        g_ring.lut_pinned = nullptr;
        g_ring.failed = true;
    }
    g_ring.ready = ok;
    return ok;
}

// The first host -> host call of a process can have the ring allocated beside its GPU work: this returns at once and a
// helper thread does the allocation on `device`; pope_assemble_finish takes the same mutex and so waits for it if need be.
extern "C" void pope_assemble_prepare(int32_t device) {
    {
        std::unique_lock<std::mutex> lock(g_ring.mu);
        if (g_ring.ready || g_ring.failed || g_ring.preparing) return;
        g_ring.preparing = true;
    }
    std::thread([device] {
        std::unique_lock<std::mutex> lock(g_ring.mu);
        if (device >= 0) (void)hipSetDevice(device);
        (void)ring_usable_locked();
    }).detach();
}

// 1 if the byte transport can be used now (the pinned ring exists, or could be allocated here), else 0; waits for an
// allocation that pope_assemble_prepare started.  The caller decides between pope_assemble_finish_codes and float columns.
extern "C" int32_t pope_assemble_ring_ready(void) {
    std::unique_lock<std::mutex> lock(g_ring.mu);
    return ring_usable_locked() ? 1 : 0;
}

// Rows [r0, r1) of a DEVICE matrix (row bytes `eb`, pitch `epitch`) into pageable host rows at dst0 + r * dpitch WITHOUT giving
// the runtime the pageable pointer: 4 MB pinned bounce buffer (allocated once per process, guarded by its own mutex),
// one blocking pitched copy per buffer-full, memcpy out.  Slow (nothing overlaps) and only used when the pinned ring is not
// available (refused, or rows wider than a ring slot).  Rows wider than the buffer travel in column pieces (round 5: they used to
// fall through to the runtime).  Only if even the 4 MB are refused does a BLOCKING hipMemcpy2D write the rows directly (the
// runtime's own staging; nothing is in flight when it returns, and no page of the result is registered anywhere).
constexpr size_t BOUNCE_BYTES = (size_t)4 << 20;
static std::mutex g_bounce_mu;
static char *g_bounce = nullptr;
static bool g_bounce_failed = false;

static int bounce_rows(const char *embp, size_t epitch, size_t eb, char *dst0, size_t dpitch, int64_t r0, int64_t r1, hipStream_t stream) {
    if (r1 <= r0 || eb == 0) return POPE_OK;
    std::lock_guard<std::mutex> lock(g_bounce_mu);
    if (!g_bounce && !g_bounce_failed && !(pope::g_fail_host_register & 4)) {
        if (hipHostMalloc(reinterpret_cast<void **>(&g_bounce), BOUNCE_BYTES, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            g_bounce = nullptr;
            g_bounce_failed = true;
        }
    }
    const bool use_bounce = g_bounce && !(pope::g_fail_host_register & 4);
    if (!use_bounce) {
        hipError_t e = hipStreamSynchronize(stream);
        if (e == hipSuccess)
            e = hipMemcpy2D(dst0 + (size_t)r0 * dpitch, dpitch, embp + (size_t)r0 * epitch, epitch, eb, (size_t)(r1 - r0), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            pope::set_error("hipMemcpy2D(D2H, rows %lld..%lld) failed: %s", (long long)r0, (long long)r1, hipGetErrorString(e));
            return POPE_ERR_HIP;
        }
        return POPE_OK;
    }
    const size_t piece = eb < BOUNCE_BYTES ? eb : BOUNCE_BYTES;                // bytes of a row per pass (a whole row where it fits)
    const int64_t per = (int64_t)(BOUNCE_BYTES / piece);
    for (size_t c0 = 0; c0 < eb; c0 += piece) {
        const size_t cb = c0 + piece < eb ? piece : eb - c0;
        for (int64_t r = r0; r < r1; r += per) {
            const int64_t n = r + per < r1 ? per : r1 - r;
            hipError_t e = (epitch == eb && cb == eb) ? hipMemcpyAsync(g_bounce, embp + (size_t)r * epitch, (size_t)n * eb, hipMemcpyDeviceToHost, stream)
                                                      : hipMemcpy2DAsync(g_bounce, cb, embp + (size_t)r * epitch + c0, epitch, cb, (size_t)n, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {
                pope::set_error("D2H through the bounce buffer (rows %lld..%lld) failed: %s", (long long)r, (long long)(r + n), hipGetErrorString(e));
                return POPE_ERR_HIP;
            }
            for (int64_t i = 0; i < n; ++i) memcpy(dst0 + (size_t)(r + i) * dpitch + c0, g_bounce + (size_t)i * cb, cb);
        }
    }
    return POPE_OK;
}

// `lut_dev` == nullptr: `emb` holds the float columns themselves.  Otherwise `emb` holds one code byte per element
// (pope_geodesic_hop_codes) and `lut_dev` the 256 floats the codes stand for: a quarter of the bytes cross PCIe and the host
// threads look the floats up while they copy out of the ring (ring mode only).
static int assemble_finish_impl(void *handle, const void *emb, int64_t emb_pitch_bytes, int64_t emb_row_bytes, const float *lut_dev, void *stream_) {
    pope::clear_error();
    hipStream_t stream = (hipStream_t)stream_;
    HostAssembly *h = static_cast<HostAssembly *>(handle);
    if (!h) {
        pope::set_error("pope_assemble_finish: null handle");
        return POPE_ERR_INVALID;
    }
    Assembly &a = h->a;
    Workers &pool = h->pool;
    const bool coded = lut_dev != nullptr;
    const size_t out_row = (size_t)(emb_row_bytes < 0 ? 0 : emb_row_bytes) * (coded ? sizeof(float) : 1);
    if (emb_row_bytes < 0 || (emb_row_bytes > 0 && (!emb || emb_pitch_bytes < emb_row_bytes)) || a.out_pitch < a.x_row + out_row ||
        (coded && emb_row_bytes == 0)) {
        pope_assemble_abort(h);
        pope::set_error("pope_assemble_finish: null pointer or bad size");
        return POPE_ERR_INVALID;
    }
    double tr[8] = {h->t_madvise, 0, 0, 0, 0, 0, 0, 0};
    const double t_begin = h->t_begin;
    char *out = a.out;
    const int64_t rows = a.rows, x_row_bytes = (int64_t)a.x_row;
    const int nch = a.chunks;
    int rc = POPE_OK;
    const uintptr_t base = reinterpret_cast<uintptr_t>(out);
    const size_t xb = (size_t)x_row_bytes, eb = (size_t)emb_row_bytes, pitch = a.out_pitch, epitch = (size_t)emb_pitch_bytes;
    const char *embp = static_cast<const char *>(emb);
    auto S = [&](int64_t r) { return base + (uintptr_t)r * pitch + xb; };          // where row r's embedding columns start
    // The embedding columns land in three 8 MB pinned slots allocated once per process, and the threads that copied the features
    // copy each landed chunk out into the pageable result while the next one is on the bus.  No page of the result is ever
    // registered, so nothing about the call depends on how the kernel driver handles pinning and unpinning a fresh
    // quarter-gigabyte range (which made repeated calls erratic, 4 .. 18 ms, when round 3 tried it).
    bool ring_done = false;
    if (eb > 0 && eb <= RING_SLOT_BYTES) {
        std::unique_lock<std::mutex> lock(g_ring.mu);
        double t0 = now_ms();
        const bool usable = ring_usable_locked();
        tr[2] = now_ms() - t0;
        if (usable) {
            const int64_t crow = (int64_t)(RING_SLOT_BYTES / eb);
            const int ne = (int)((rows + crow - 1) / crow);
            a.emb_row = eb; a.emb_elems = eb; a.emb_chunk_rows = crow; a.ring_slots = RING_SLOTS;
            if (coded && hipMemcpyAsync(g_ring.lut_pinned, lut_dev, 256 * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess) {
                (void)hipGetLastError();
                pope::set_error("D2H of the code table failed");
                rc = POPE_ERR_HIP;
            }
            for (int i = 0; i < RING_SLOTS; ++i) a.ring[i] = g_ring.slot[i];
            a.emb_done = std::vector<std::atomic<int>>((size_t)ne);
            for (auto &d : a.emb_done) d.store(0, std::memory_order_relaxed);
            a.begin_embedding(ne * a.slices);
            auto publish = [&](int c) {                                            // chunk c has landed: hand it to the threads
                const double t1 = now_ms();
                const hipError_t e = hipEventSynchronize(g_ring.ev[c % RING_SLOTS]);
                tr[5] += now_ms() - t1;
                if (e != hipSuccess && rc == POPE_OK) {
                    pope::set_error("hipEventSynchronize(ring chunk %d) failed: %s", c, hipGetErrorString(e));
                    rc = POPE_ERR_HIP;
                }
                if (rc == POPE_OK && coded && c == 0) {                           // the table travelled ahead of chunk 0
                    if (g_ring.pair_lut.empty() || memcmp(g_ring.lut_seen, g_ring.lut_pinned, sizeof(g_ring.lut_seen)) != 0) {
                        memcpy(g_ring.lut_seen, g_ring.lut_pinned, sizeof(g_ring.lut_seen));
                        g_ring.pair_lut.resize(65536);
                        uint32_t bits[256];
                        memcpy(bits, g_ring.lut_seen, sizeof(bits));
                        for (uint32_t p = 0; p < 65536; ++p) g_ring.pair_lut[p] = (uint64_t)bits[p & 255u] | ((uint64_t)bits[p >> 8] << 32);
                    }
                    a.pair_lut = g_ring.pair_lut.data();
                }
                if (rc == POPE_OK) a.emb_published.store(c + 1, std::memory_order_release);
            };
            for (int c = 0; c < ne && rc == POPE_OK; ++c) {
                if (c >= RING_SLOTS) {                                             // the slot's previous chunk must have been copied out
                    t0 = now_ms();
                    while (a.emb_done[(size_t)(c - RING_SLOTS)].load(std::memory_order_acquire) < a.slices) std::this_thread::yield();
                    tr[1] += now_ms() - t0;
                }
                const int64_t lo = (int64_t)c * crow, n = (lo + crow < rows ? lo + crow : rows) - lo;
                t0 = now_ms();
                hipError_t e = epitch == eb ? hipMemcpyAsync(g_ring.slot[c % RING_SLOTS], embp + (size_t)lo * epitch, (size_t)n * eb, hipMemcpyDeviceToHost, stream)
                                            : hipMemcpy2DAsync(g_ring.slot[c % RING_SLOTS], eb, embp + (size_t)lo * epitch, epitch, eb, (size_t)n,
                                                               hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipEventRecord(g_ring.ev[c % RING_SLOTS], stream);
                tr[3] += now_ms() - t0;
                if (e != hipSuccess) {
                    pope::set_error("D2H into the pinned ring (chunk %d) failed: %s", c, hipGetErrorString(e));
                    rc = POPE_ERR_HIP;
                    break;
                }
                if (c >= 1) publish(c - 1);
            }
            if (rc == POPE_OK) publish(ne - 1);
            if (rc != POPE_OK) {                                                   // let the threads go; drain what is in flight
                a.abort();
                a.begin_embedding(0);
                (void)hipStreamSynchronize(stream);
            }
            t0 = now_ms();
            pool.join();
            tr[4] = now_ms() - t0;
            ring_done = true;
        } else {
            a.begin_embedding(0);                                                  // no ring: the bounce buffer below
        }
    } else {
        a.begin_embedding(0);
    }
    if (coded && !ring_done) {
        pope_assemble_abort(h);
        pope::set_error("pope_assemble_finish_codes: no pinned ring (hipHostMalloc refused or rows wider than a ring slot)");
        return POPE_ERR_HIP;
    }
    if (ring_done) {
        tr[7] = now_ms() - t_begin;
        publish_trace(tr);
        delete h;
        return rc;
    }
    // No pinned ring (refused, or rows wider than a slot): wait for the feature copy, then stage the columns through the library's
    // 4 MB pinned bounce buffer and copy them out in this thread.  The runtime never sees the result's pages.
    for (int c = 0; c < nch; ++c) {
        const double t1 = now_ms();
        while (a.done[(size_t)c].load(std::memory_order_acquire) < a.slices) std::this_thread::yield();
        tr[1] += now_ms() - t1;
    }
    if (eb > 0) rc = bounce_rows(embp, epitch, eb, reinterpret_cast<char *>(S(0)), pitch, 0, rows, stream);
    double t0 = now_ms();
    pool.join();
    tr[4] = now_ms() - t0;
    t0 = now_ms();
    const hipError_t es = hipStreamSynchronize(stream);
    tr[5] = now_ms() - t0;
    if (es != hipSuccess && rc == POPE_OK) {
        pope::set_error("hipStreamSynchronize failed: %s", hipGetErrorString(es));
        rc = POPE_ERR_HIP;
    }
    tr[7] = now_ms() - t_begin;
    publish_trace(tr);
    delete h;
    return rc;
}

extern "C" int pope_assemble_finish(void *handle, const void *emb, int64_t emb_pitch_bytes, int64_t emb_row_bytes, void *stream_) {
    return assemble_finish_impl(handle, emb, emb_pitch_bytes, emb_row_bytes, nullptr, stream_);
}

extern "C" int pope_assemble_finish_codes(void *handle, const uint8_t *codes, int64_t codes_pitch_bytes, int32_t K, const float *lut, void *stream_) {
    if (!lut) {
        pope_assemble_abort(handle);
        pope::clear_error();
        pope::set_error("pope_assemble_finish_codes: null code table");
        return POPE_ERR_INVALID;
    }
    return assemble_finish_impl(handle, codes, codes_pitch_bytes, K, lut, stream_);
}

extern "C" int pope_assemble_host_result(const void *x_host, int64_t x_pitch_bytes, int64_t x_row_bytes, const void *emb,
                                         int64_t emb_pitch_bytes, int64_t emb_row_bytes, void *out_host, int64_t out_pitch_bytes,
                                         int64_t rows, int32_t threads, int32_t chunks, void *stream_) {
    pope::clear_error();
    if (!out_host || rows < 0 || x_row_bytes < 0 || emb_row_bytes < 0 || (x_row_bytes > 0 && (!x_host || x_pitch_bytes < x_row_bytes)) ||
        (emb_row_bytes > 0 && (!emb || emb_pitch_bytes < emb_row_bytes)) || out_pitch_bytes < x_row_bytes + emb_row_bytes) {
        pope::set_error("pope_assemble_host_result: null pointer or bad size");
        return POPE_ERR_INVALID;
    }
    if (rows == 0) return POPE_OK;
    void *h = pope_assemble_begin(x_host, x_pitch_bytes, x_row_bytes, out_host, out_pitch_bytes, rows, threads, chunks);
    if (!h) return POPE_ERR_INVALID;
    return pope_assemble_finish(h, emb, emb_pitch_bytes, emb_row_bytes, stream_);
}
