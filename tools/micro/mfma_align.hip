// Does the register alignment of the accumulator tuple change the rate of v_mfma_f32_32x32x2_f32?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_align tools/micro/mfma_align.hip && /tmp/mfma_align
#include <hip/hip_runtime.h>
#include <cstdio>

#define MFMA8(C0, C1)                                              \
    "v_mfma_f32_32x32x2_f32 " C0 ", v0, v1, " C0 "\n"              \
    "v_mfma_f32_32x32x2_f32 " C1 ", v0, v1, " C1 "\n"              \
    "v_mfma_f32_32x32x2_f32 " C0 ", v0, v1, " C0 "\n"              \
    "v_mfma_f32_32x32x2_f32 " C1 ", v0, v1, " C1 "\n"              \
    "v_mfma_f32_32x32x2_f32 " C0 ", v0, v1, " C0 "\n"              \
    "v_mfma_f32_32x32x2_f32 " C1 ", v0, v1, " C1 "\n"              \
    "v_mfma_f32_32x32x2_f32 " C0 ", v0, v1, " C0 "\n"              \
    "v_mfma_f32_32x32x2_f32 " C1 ", v0, v1, " C1 "\n"

#define CLOBBERS "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
    "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",   \
    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",   \
    "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"

template <int VARIANT>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (VARIANT == 0) asm volatile(MFMA8("v[4:19]", "v[20:35]") ::: CLOBBERS);      // bases 4-aligned
        if (VARIANT == 1) asm volatile(MFMA8("v[2:17]", "v[18:33]") ::: CLOBBERS);      // bases = 2 mod 4
        if (VARIANT == 2) asm volatile(MFMA8("v[8:23]", "v[24:39]") ::: CLOBBERS);      // bases 8-aligned
        if (VARIANT == 3) asm volatile(MFMA8("v[50:65]", "v[34:49]") ::: CLOBBERS);     // as the compiler chose in k_pairwise_persistent
        if (VARIANT == 4) asm volatile(MFMA8("v[16:31]", "v[32:47]") ::: CLOBBERS);     // 16-aligned
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[VARIANT] = t1 - t0;
}

int main() {
    unsigned long long *d, h[8] = {0};
    hipMalloc(&d, sizeof(h));
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, iters);
        hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, iters);
        hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, d, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"acc base % 4 == 0 (v4, v20)", "acc base % 4 == 2 (v2, v18)", "acc base % 8 == 0 (v8, v24)", "v50 / v34", "acc base % 16 == 0"};
    for (int v = 0; v < 5; ++v) printf("%-32s %.1f ticks per MFMA\n", names[v], (double)h[v] / (iters * 8.0));
    return 0;
}
