// What does the chip take for WRITING the embedding columns of the result -- [N, out_cols] float32, columns [col0, col0 + width) -- with no
// loads and no arithmetic at all?  The roof of the finalise kernels' store side.  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/column_fill tools/micro/column_fill.hip && /tmp/column_fill
// A wave writes 8 KB runs (8 store instructions of 1 KB, k_finalize_lut's shape) of the flat (row, 128-byte unit) sequence.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_fill(float *out, long long out_cols, int col0, int hpr_shift, long long total) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const unsigned mask = (1u << hpr_shift) - 1u;
    const float4 val = make_float4(1.f, 2.f, 3.f, 4.f);
    for (long long batch = wave; batch * 64 < total; batch += nwaves) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long g = batch * 64 + 8 * e + (lane >> 3);
            if (g < total) {
                const unsigned v = (unsigned)(g >> hpr_shift), hw = (unsigned)g & mask;
                *reinterpret_cast<float4 *>(out + (size_t)v * out_cols + col0 + hw * 32 + (lane & 7) * 4) = val;
            }
        }
    }
}

static void run(const char *what, long long N, long long out_cols, int col0, int width, int blocks) {
    float *out;
    CK(hipMalloc(&out, (size_t)N * out_cols * 4));
    int hpr_shift = 0;
    while ((32 << hpr_shift) < width) ++hpr_shift;
    const long long total = N << hpr_shift;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, out, out_cols, col0, hpr_shift, total);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, out, out_cols, col0, hpr_shift, total);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)N * width * 4;
    printf("%-44s %5d blocks  %8.1f us  %6.2f TB/s\n", what, blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    CK(hipFree(out));
}

int main() {
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        run("flickr 8 x 256: [89250, 2548] cols 500..2548", 89250, 2548, 500, 2048, blocks);
        run("flickr 8 x 128: [89250, 1524] cols 500..1524", 89250, 1524, 500, 1024, blocks);
        run("rmat22 512: [4194304, 512] all", 4194304, 512, 0, 512, blocks);
    }
    return 0;
}
