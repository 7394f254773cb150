// A fake HIP runtime for graphpope_amd/csrc/host.cc (CPU only, no GPU, no libamdhip64): one "copy engine" thread per
// stream executes the enqueued copies and event records in order, asynchronously to the caller, so that host.cc's threads
// (parked worker pool, ring hand-off, registration bookkeeping) run against something that behaves like a device queue and
// can be put under -fsanitize=thread / -fsanitize=address.  "Device" memory is ordinary host memory.
// Checks of its own: a DMA destination that is neither inside a hipHostMalloc'd block nor inside ONE registered range is
// counted (fake_unpinned_async_targets()), a registration that overlaps a live one or an unregistration of an unknown base
// fails like the real runtime's.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

namespace {

struct Event {
    std::mutex m;
    std::condition_variable cv;
    unsigned long long recorded = 0, completed = 0;
};

struct Queue {
    std::mutex m;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> ops;
    bool running = false;
    std::thread worker;
    Queue() {
        worker = std::thread([this] {
            std::unique_lock<std::mutex> lock(m);
            for (;;) {
                cv.wait(lock, [&] { return !ops.empty(); });
                auto op = std::move(ops.front());
                ops.pop_front();
                running = true;
                lock.unlock();
                std::this_thread::sleep_for(std::chrono::microseconds(20));      // a transfer takes a while
                op();
                lock.lock();
                running = false;
                if (ops.empty()) idle.notify_all();
            }
        });
        worker.detach();
    }
    void push(std::function<void()> op) {
        {
            std::lock_guard<std::mutex> lock(m);
            ops.push_back(std::move(op));
        }
        cv.notify_one();
    }
    void drain() {
        std::unique_lock<std::mutex> lock(m);
        idle.wait(lock, [&] { return ops.empty() && !running; });
    }
};

std::mutex g_m;
std::map<void *, Queue *> g_queues;
std::map<char *, size_t> g_pinned, g_registered;
std::atomic<long> g_unpinned_targets{0};

Queue &queue_of(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_m);
    Queue *&q = g_queues[(void *)s];
    if (!q) q = new Queue;
    return *q;
}

bool inside(const std::map<char *, size_t> &ranges, const char *p, size_t n) {
    auto it = ranges.upper_bound(const_cast<char *>(p));
    if (it == ranges.begin()) return false;
    --it;
    return p >= it->first && p + n <= it->first + it->second;
}

void note_host_target(const void *dst, size_t n) {
    std::lock_guard<std::mutex> lock(g_m);
    const char *p = static_cast<const char *>(dst);
    if (!inside(g_pinned, p, n) && !inside(g_registered, p, n)) ++g_unpinned_targets;
}

}  // namespace

extern "C" long fake_unpinned_async_targets() { return g_unpinned_targets.load(); }
extern "C" void fake_stream_synchronize(void *stream) { queue_of((hipStream_t)stream).drain(); }
extern "C" long fake_live_registrations() {
    std::lock_guard<std::mutex> lock(g_m);
    return (long)g_registered.size();
}

extern "C" {

hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    *p = aligned_alloc(4096, (n + 4095) / 4096 * 4096);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lock(g_m);
    g_pinned[static_cast<char *>(*p)] = n;
    return hipSuccess;
}

hipError_t hipHostFree(void *p) {
    {
        std::lock_guard<std::mutex> lock(g_m);
        g_pinned.erase(static_cast<char *>(p));
    }
    free(p);
    return hipSuccess;
}

hipError_t hipHostRegister(void *p, size_t n, unsigned) {
    std::lock_guard<std::mutex> lock(g_m);
    char *b = static_cast<char *>(p);
    for (auto &r : g_registered)
        if (b < r.first + r.second && r.first < b + n) return hipErrorHostMemoryAlreadyRegistered;
    g_registered[b] = n;
    return hipSuccess;
}

hipError_t hipHostUnregister(void *p) {
    std::lock_guard<std::mutex> lock(g_m);
    return g_registered.erase(static_cast<char *>(p)) ? hipSuccess : hipErrorHostMemoryNotRegistered;
}

hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
    *e = reinterpret_cast<hipEvent_t>(new Event);
    return hipSuccess;
}

hipError_t hipEventDestroy(hipEvent_t e) {
    delete reinterpret_cast<Event *>(e);
    return hipSuccess;
}

hipError_t hipEventRecord(hipEvent_t e_, hipStream_t s) {
    Event *e = reinterpret_cast<Event *>(e_);
    unsigned long long ticket;
    {
        std::lock_guard<std::mutex> lock(e->m);
        ticket = ++e->recorded;
    }
    queue_of(s).push([e, ticket] {
        {
            std::lock_guard<std::mutex> lock(e->m);
            if (e->completed < ticket) e->completed = ticket;
        }
        e->cv.notify_all();
    });
    return hipSuccess;
}

hipError_t hipEventSynchronize(hipEvent_t e_) {
    Event *e = reinterpret_cast<Event *>(e_);
    std::unique_lock<std::mutex> lock(e->m);
    const unsigned long long want = e->recorded;
    e->cv.wait(lock, [&] { return e->completed >= want; });
    return hipSuccess;
}

hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s) {
    if (kind == hipMemcpyDeviceToHost) note_host_target(dst, n);
    queue_of(s).push([dst, src, n] { memcpy(dst, src, n); });
    return hipSuccess;
}

hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind,
                            hipStream_t s) {
    if (kind == hipMemcpyDeviceToHost && height) note_host_target(dst, (height - 1) * dpitch + width);
    queue_of(s).push([=] {
        for (size_t r = 0; r < height; ++r) memcpy(static_cast<char *>(dst) + r * dpitch, static_cast<const char *>(src) + r * spitch, width);
    });
    return hipSuccess;
}

hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    for (size_t r = 0; r < height; ++r) memcpy(static_cast<char *>(dst) + r * dpitch, static_cast<const char *>(src) + r * spitch, width);
    return hipSuccess;
}

hipError_t hipStreamSynchronize(hipStream_t s) {
    queue_of(s).drain();
    return hipSuccess;
}

hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "fake hip error"; }

}  // extern "C"
