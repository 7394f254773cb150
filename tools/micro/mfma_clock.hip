// What is the shader clock while every SIMD of the chip issues f32 MFMAs back to back, and what rate is that?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_clock tools/micro/mfma_clock.hip && /tmp/mfma_clock
// s_memtime counts shader clocks, s_memrealtime a constant 100 MHz: their ratio over a wave's loop is the clock the wave ran at.
// The f32 MFMA peak of MI355X_MICROARCH.md (157.3 TFLOP/s) is 256 CUs x 4 SIMDs x 64 FLOP/clock x 2.4 GHz.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: ten independent v_mfma_f32_16x16x4_f32 accumulators per wave, back to back; MODE 1: no MFMA, s_sleep loop of about the same length
// MODE 2: as the consumer waves of k_gemm_tile16: per 40 MFMAs seven ds_read_b128 whose results are the operands of the NEXT 40
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters, float seed) {
    __shared__ float4 lds[2048];                           // 32 KB
    if (MODE == 2) {
        for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_float4(seed, seed * 0.5f, seed * 0.25f, 1.0f);
        __syncthreads();
    }
    f32x4 acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    const float a = seed + (threadIdx.x & 7), b = seed * 0.5f + (threadIdx.x & 3);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 2) {
            static float4 fa[2][5], fb[2][2];              // (registers: the loop is unrolled by two below)
            (void)fa; (void)fb;
        } else {
            __builtin_amdgcn_s_sleep(20);                  // 40 MFMAs x 32 clocks = 1280 clocks = 20 x 64
        }
    }
    if (MODE == 3) {                                       // the operand pattern of k_gemm_tile16's pass: 5 x 2 accumulators, A from five float4, B from two
        float4 fa[5], fb[2];
#pragma unroll
        for (int i = 0; i < 5; ++i) fa[i] = make_float4(a + i, a - i, a * i, a + 2 * i);
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[t] = make_float4(b + t, b - t, b * (t + 2), b + 3 * t);
#pragma unroll
        for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(fa[i].x), "+v"(fa[i].y), "+v"(fa[i].z), "+v"(fa[i].w));
#pragma unroll
        for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(fb[t].x), "+v"(fb[t].y), "+v"(fb[t].z), "+v"(fb[t].w));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[i * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb[t].x, acc[i * 2 + t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[i * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb[t].y, acc[i * 2 + t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[i * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb[t].z, acc[i * 2 + t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[i * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb[t].w, acc[i * 2 + t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (MODE == 4 || MODE == 5) {                          // the same pattern with the accumulators forced into arch VGPRs (4) or AccVGPRs (5)
        float4 fa[5], fb[2];
#pragma unroll
        for (int i = 0; i < 5; ++i) fa[i] = make_float4(a + i, a - i, a * i, a + 2 * i);
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[t] = make_float4(b + t, b - t, b * (t + 2), b + 3 * t);
#pragma unroll
        for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(fa[i].x), "+v"(fa[i].y), "+v"(fa[i].z), "+v"(fa[i].w));
#pragma unroll
        for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(fb[t].x), "+v"(fb[t].y), "+v"(fb[t].z), "+v"(fb[t].w));
#define MF(A, B, C) do { if (MODE == 4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B)); \
                         else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(C) : "v"(A), "v"(B)); } while (0)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) MF(fa[i].x, fb[t].x, acc[i * 2 + t]);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) MF(fa[i].y, fb[t].y, acc[i * 2 + t]);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) MF(fa[i].z, fb[t].z, acc[i * 2 + t]);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) MF(fa[i].w, fb[t].w, acc[i * 2 + t]);
        }
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    }
    if (MODE == 2) {
        float4 fa[2][5], fb[2][2];
        const int base = (threadIdx.x & 63) + 64 * (threadIdx.x >> 6) * 7;
        auto rd = [&](int set, int it) {
#pragma unroll
            for (int i = 0; i < 5; ++i) fa[set][i] = lds[(base + 64 * i + it) & 2047];
#pragma unroll
            for (int t = 0; t < 2; ++t) fb[set][t] = lds[(base + 64 * (5 + t) + it) & 2047];
        };
        rd(0, 0);
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                rd(p ^ 1, it + p + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[i * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p][i].x, fb[p][t].x, acc[i * 2 + t], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[i * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p][i].y, fb[p][t].y, acc[i * 2 + t], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[i * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p][i].z, fb[p][t].z, acc[i * 2 + t], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[i * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[p][i].w, fb[p][t].w, acc[i * 2 + t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[wave * 2] = c1 - c0;
        out[wave * 2 + 1] = r1 - r0;
    }
    if (s == 12345.678f) out[0] = 0;                       // keep the accumulators alive
}

template <int MODE>
static void run(const char *what, int blocks, int iters, unsigned long long *d) {
    const int waves = blocks * 4;
    std::vector<unsigned long long> h(waves * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> mhz(waves), us(waves);
        for (int w = 0; w < waves; ++w) {
            us[w] = h[2 * w + 1] / 100.0;
            mhz[w] = h[2 * w] / us[w];
        }
        std::sort(mhz.begin(), mhz.end());
        std::sort(us.begin(), us.end());
        const double flop = MODE != 1 ? (double)waves * iters * 40 * 2048.0 : 0.0;
        printf("%-34s blocks %4d  launch %8.1f us  wave loop median %8.1f us  shader clock min %6.0f median %6.0f max %6.0f MHz",
               what, blocks, ms * 1e3, us[waves / 2], mhz[0], mhz[waves / 2], mhz[waves - 1]);
        if (MODE != 1) printf("  %6.1f TFLOP/s by the median wave loop (%.3f of 157.3)", flop / (us[waves / 2] * 1e-6) / 1e12, flop / (us[waves / 2] * 1e-6) / 157.3e12);
        printf("\n");
    }
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 4096 * 2 * 8);
    run<1>("no MFMA (s_sleep), 1 wave / SIMD", 256, 200, d);
    run<0>("MFMA back to back, 1 wave / SIMD", 256, 200, d);        // ~ 110 us at 2.4 GHz
    run<0>("MFMA back to back, 1 wave / SIMD", 256, 2000, d);       // ~ 1.1 ms
    run<0>("MFMA back to back, 2 waves / SIMD", 512, 1000, d);
    run<3>("MFMA, operands as in k_gemm_tile16", 256, 2000, d);
    run<4>("same, accumulators in arch VGPRs", 256, 2000, d);
    run<5>("same, accumulators in AccVGPRs", 256, 2000, d);
    run<2>("MFMA + 7 ds_read_b128 per 40, 1 wave/SIMD", 256, 2000, d);
    run<2>("MFMA + 7 ds_read_b128 per 40, 2 waves/SIMD", 512, 1000, d);
    run<0>("MFMA on 64 CUs only", 64, 2000, d);
    run<1>("no MFMA (s_sleep), 1 wave / SIMD", 256, 200, d);
    return 0;
}
