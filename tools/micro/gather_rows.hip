// What does the chip deliver for the access pattern of the BFS level kernel -- a coalesced stream of 4-byte indices, each naming a
// random ROW of a table that is gathered by one lane -- as a function of the row size (16 / 32 / 64 / 128 bytes = 128 / 256 / 512 /
// 1 024 anchors) and of the table size (inside the 256 MB Infinity Cache or beyond it)?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_rows tools/micro/gather_rows.hip && /tmp/gather_rows
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -- /tmp/gather_rows        (calibrates FETCH_SIZE for this pattern)
// A lane owns 4 consecutive indices (one 16-byte load), gathers its 4 rows with every load requested before the first use, folds them
// with OR and stores ONE 8-byte word per lane -- the shape of k_bfs_level's expand waves without the scan.  `slots` indices per
// launch; the number of distinct rows gathered is what a uniform draw gives.  Prints the time, the gather rate, the useful bytes per
// second and (for the beyond-cache sizes) what a fetch granularity of 64 or 128 bytes would make of it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;

template <int WORDS>      // 8-byte words per row: 2, 4, 8, 16
__global__ __launch_bounds__(256) void k_gather(const int *__restrict__ idx, size_t slots, const u64 *__restrict__ table, u64 *__restrict__ out) {
    const size_t lane4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (lane4 >= slots) return;
    const int4 u = *reinterpret_cast<const int4 *>(idx + lane4);
    ulonglong2 v[4][WORDS / 2];
    const int rows[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int w = 0; w < WORDS / 2; ++w) v[s][w] = reinterpret_cast<const ulonglong2 *>(table + (size_t)rows[s] * WORDS)[w];
    u64 acc = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int w = 0; w < WORDS / 2; ++w) acc |= v[s][w].x | v[s][w].y;
    out[lane4 / 4] = acc;
}

template <int WORDS>
static float run(const int *idx, size_t slots, const u64 *table, u64 *out, int reps) {
    const unsigned blocks = (unsigned)((slots / 4 + 255) / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_gather<WORDS>, dim3(blocks), dim3(256), 0, 0, idx, slots, table, out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gather<WORDS>, dim3(blocks), dim3(256), 0, 0, idx, slots, table, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const size_t slots = (size_t)64 << 20;                     // 67 M indices (R-MAT scale 22 has 65 M CSR slots)
    const size_t max_rows = (size_t)4 << 20;                   // 4.2 M rows, as R-MAT scale 22 has nodes
    const size_t max_bytes = max_rows * 128;
    std::vector<int> h(slots);
    int *idx;
    u64 *table, *out;
    CK(hipMalloc(&idx, slots * 4));
    CK(hipMalloc(&table, max_bytes));
    CK(hipMalloc(&out, slots / 4 * 8));
    CK(hipMemset(table, 1, max_bytes));
    printf("%-10s %-10s %-9s %9s %12s %12s %14s\n", "row bytes", "rows", "table MB", "ms", "G gathers/s", "useful TB/s", "TB/s at 128 B");
    for (size_t rows : {(size_t)64 << 10, (size_t)512 << 10, (size_t)4 << 20}) {
        std::mt19937_64 rng(7 + rows);
        for (size_t i = 0; i < slots; ++i) h[i] = (int)(rng() % rows);
        CK(hipMemcpy(idx, h.data(), slots * 4, hipMemcpyHostToDevice));
        for (int bytes : {16, 32, 64, 128}) {
            float ms = 0;
            switch (bytes) {
            case 16:  ms = run<2>(idx, slots, table, out, 5); break;
            case 32:  ms = run<4>(idx, slots, table, out, 5); break;
            case 64:  ms = run<8>(idx, slots, table, out, 5); break;
            default:  ms = run<16>(idx, slots, table, out, 5); break;
            }
            const double g = slots / (ms * 1e-3);
            printf("%-10d %-10zu %-9.1f %9.3f %12.1f %12.2f %14.2f\n", bytes, rows, rows * bytes / 1e6, ms, g / 1e9, g * bytes / 1e12,
                   g * std::max(bytes, 128) / 1e12);
        }
    }
    return 0;
}
