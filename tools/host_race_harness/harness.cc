// Drives graphpope_amd/csrc/host.cc (the host half of the host -> host boundary, utils.py:129-147) against fake_hip.cc:
// the parked worker pool, the pinned-ring hand-off, the bounce transport (whole rows and column pieces of rows wider than the
// buffer), the runtime's blocking copy behind it, aborts, and two assemblies at once -- every result compared with a plain reference;
// no page of a result may ever be registered, and no asynchronous copy may target unpinned memory.  Built by run.sh under -fsanitize=thread and -fsanitize=address.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "graphpope_hip.h"

namespace pope {            // what abi.cpp / geodesic.hip provide in the real library
int g_fail_host_register = 0;
static thread_local char g_error[512];
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
void clear_error() { g_error[0] = 0; }
}  // namespace pope

extern "C" long fake_unpinned_async_targets();
extern "C" long fake_live_registrations();
extern "C" void fake_stream_synchronize(void *stream);

static int g_failures = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            ++g_failures;                                  \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                  \
            fprintf(stderr, " [%s]\n", pope::g_error);     \
        }                                                  \
    } while (0)

struct Case {
    int64_t rows;
    int f, k;
    std::vector<float> x, emb, out;
    std::vector<uint8_t> codes;
    float lut[256];
    Case(int64_t rows_, int f_, int k_, unsigned seed) : rows(rows_), f(f_), k(k_), x((size_t)rows_ * f_), emb((size_t)rows_ * k_), out((size_t)rows_ * (f_ + k_), -1.f),
                                                          codes((size_t)rows_ * k_) {
        srand(seed);
        for (auto &v : x) v = (float)rand() / RAND_MAX;
        lut[0] = 0.f;
        for (int c = 1; c < 256; ++c) lut[c] = 1.0f / (float)c;
        for (size_t i = 0; i < codes.size(); ++i) {
            codes[i] = (uint8_t)(rand() % 12);
            emb[i] = lut[codes[i]];
        }
    }
    bool ok() const {
        for (int64_t r = 0; r < rows; ++r) {
            if (f && memcmp(&out[(size_t)r * (f + k)], &x[(size_t)r * f], (size_t)f * 4)) return false;
            if (memcmp(&out[(size_t)r * (f + k) + f], &emb[(size_t)r * k], (size_t)k * 4)) return false;
        }
        return true;
    }
};

enum How { FLOATS, CODES, ABORT };

static void run_case(Case &c, How how, int threads, void *stream, int delay_us = 0) {
    if (getenv("HARNESS_VERBOSE")) fprintf(stderr, "case rows %lld f %d k %d refuse %d how %d threads %d stream %p\n", (long long)c.rows, c.f, c.k, pope::g_fail_host_register, (int)how, threads, stream);
    void *h = pope_assemble_begin(c.f ? c.x.data() : nullptr, (int64_t)c.f * 4, (int64_t)c.f * 4, c.out.data(), (int64_t)(c.f + c.k) * 4, c.rows, threads, 0);
    CHECK(h != nullptr, "begin");
    if (!h) return;
    if (delay_us) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));    // the "GPU phase": the workers must sleep, not spin
    if (how == ABORT) {
        pope_assemble_abort(h);
        return;
    }
    int rc = how == CODES ? pope_assemble_finish_codes(h, c.codes.data(), c.k, c.k, c.lut, stream)
                          : pope_assemble_finish(h, c.emb.data(), (int64_t)c.k * 4, (int64_t)c.k * 4, stream);
    CHECK(rc == POPE_OK, "finish rc %d", rc);
    CHECK(c.ok(), "result differs (rows %lld f %d k %d refuse %d how %d)", (long long)c.rows, c.f, c.k, pope::g_fail_host_register, (int)how);
}

int main() {
    // 1. the ring transport, floats and codes, repeated (the parked pool is reused), with and without feature columns
    for (int rep = 0; rep < 4; ++rep)
        for (int f : {500, 0, 3}) {
            Case a(30000 + 17 * rep, f, 256, 1 + rep), b(30000 + 17 * rep, f, 256, 7 + rep);
            run_case(a, FLOATS, 16, nullptr, rep ? 300 : 0);
            run_case(b, CODES, 16, nullptr, rep ? 300 : 0);
        }
    CHECK(fake_live_registrations() == 0, "%ld pages of a result registered", fake_live_registrations());
    // 2. the ring refused (knob bit 2): the bounce buffer -- whole rows, and rows wider than the 4 MB buffer in column pieces (also
    //    wider than an 8 MB ring slot, so they take this path with the ring available too); the runtime must never be given a
    //    pageable destination for an asynchronous copy, and nothing is registered
    {
        const long before = fake_unpinned_async_targets();
        pope::g_fail_host_register = 2;
        Case a(40000, 500, 64, 12), b(1000, 8, 1100, 13), wide(3, 2, (9 << 20) / 4 + 3, 17);
        run_case(a, FLOATS, 8, nullptr);
        run_case(b, FLOATS, 3, nullptr);
        run_case(wide, FLOATS, 2, nullptr);
        pope::g_fail_host_register = 0;
        Case wide2(2, 0, (9 << 20) / 4 + 1, 18);
        run_case(wide2, FLOATS, 2, nullptr);
        CHECK(fake_unpinned_async_targets() == before, "%ld asynchronous copies into unpinned, unregistered host memory", fake_unpinned_async_targets() - before);
        CHECK(fake_live_registrations() == 0, "%ld registrations left behind", fake_live_registrations());
    }
    // 3. ring and bounce buffer refused (bits 2 | 4): blocking copies by the runtime
    {
        pope::g_fail_host_register = 6;
        Case c(5000, 20, 40, 14);
        run_case(c, FLOATS, 4, nullptr);
        pope::g_fail_host_register = 0;
    }
    // 4. codes without a ring are refused with an error, not a hang
    {
        pope::g_fail_host_register = 2;
        Case b(20000, 100, 128, 16);
        void *h = pope_assemble_begin(b.x.data(), 400, 400, b.out.data(), (100 + 128) * 4, b.rows, 8, 0);
        CHECK(h && pope_assemble_finish_codes(h, b.codes.data(), 128, 128, b.lut, nullptr) == POPE_ERR_HIP, "codes without a ring must fail cleanly");
        pope::g_fail_host_register = 0;
    }
    // 5. aborts: before any work, and with the workers asleep in the "GPU phase"
    for (int rep = 0; rep < 3; ++rep) {
        Case a(25000, 500, 256, 20 + rep);
        run_case(a, ABORT, 16, nullptr, rep * 200);
    }
    // 6. two assemblies at once on two streams (one borrows the parked pool, the other starts threads of its own), repeatedly
    for (int rep = 0; rep < 6; ++rep) {
        Case a(30000, 200, 256, 30 + rep), b(28000, 64, 128, 40 + rep);
        std::thread t1([&] { run_case(a, rep & 1 ? CODES : FLOATS, 8, (void *)0x10, 100); });
        std::thread t2([&] { run_case(b, FLOATS, 6, (void *)0x20, 50); });
        t1.join();
        t2.join();
    }
    // 6b. uploads: only whole pages inside the caller's buffer are registered, the two ends travel as plain copies
    {
        std::vector<char> src((3 << 20) + 777), dst(src.size(), 0);
        for (size_t i = 0; i < src.size(); ++i) src[i] = (char)(i * 131 + 7);
        const char *p = src.data() + 13;                       // not page aligned
        const size_t n = src.size() - 13 - 5;
        CHECK(pope_host_pin(p, n) == POPE_OK, "pin");
        CHECK(fake_live_registrations() == 1, "one registration expected");
        CHECK(pope_copy_to_device(p, dst.data(), n, (void *)0x30) == POPE_OK, "copy");
        fake_stream_synchronize((void *)0x30);
        CHECK(memcmp(dst.data(), p, n) == 0, "uploaded bytes differ");
        CHECK(pope_host_unpin(p) == POPE_OK && fake_live_registrations() == 0, "unpin");
        CHECK(pope_host_unpin(p) == POPE_ERR_INVALID, "a second unpin must be refused");
        CHECK(pope_host_pin(src.data() + 1, 8192) == POPE_ERR_HIP, "a buffer without 1 MB of whole pages must be refused");
    }
    // 7. the whole-call entry and the 2-D host copy
    {
        Case a(12345, 77, 33, 50);
        CHECK(pope_assemble_host_result(a.x.data(), 77 * 4, 77 * 4, a.emb.data(), 33 * 4, 33 * 4, a.out.data(), (77 + 33) * 4, a.rows, 5, 3, nullptr) == POPE_OK, "host_result");
        CHECK(a.ok(), "host_result differs");
        std::vector<float> dst(a.x.size());
        CHECK(pope_host_copy_2d(a.x.data(), 77 * 4, dst.data(), 77 * 4, 77 * 4, a.rows, 7) == POPE_OK && dst == a.x, "host_copy_2d");
    }
    printf(g_failures ? "host harness: %d FAILURES\n" : "host harness: all cases passed\n", g_failures);
    return g_failures ? 1 : 0;
}
