// Does the issue rate of back-to-back v_mfma_f32_16x16x4_f32 depend on WHICH registers hold A and B?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_banks tools/micro/mfma_banks.hip && /tmp/mfma_banks
// Forty MFMAs per trip on ten accumulators (a[0:39]), the pass of k_gemm_tile16: slot c of A block i against slot c of B block t.
// Registers are fixed by hand (inline asm): A block i = v[40+4i .. 43+4i], B block t = v[60+4t .. 63+4t].
//   V0  one A and one B register for all forty                      (the constant-operand loop of mfma_clock.hip)
//   V1  the kernel's pattern: A = v[40+4i+c], B = v[60+4t+c]       (A and B of an MFMA in the same bank, c)
//   V2  B rotated by one: A = v[40+4i+c], B = v[60+4t+(c+1)%4]     (different banks)
//   V3  A fixed per slot (v[40+c]), B as V1                         (only B changes between neighbours)
//   V4  as V1 but the ten MFMAs of a slot ordered t-major (B changes every fifth instead of every other)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

static std::string body(int variant) {
    std::string s;
    char buf[128];
    for (int c = 0; c < 4; ++c) {
        for (int k = 0; k < 10; ++k) {
            int i, t;
            if (variant == 4) { t = k / 5; i = k % 5; } else { i = k / 2; t = k % 2; }
            int a = 40 + 4 * i + c, b = 60 + 4 * t + c;
            if (variant == 0) { a = 40; b = 60; }
            if (variant == 2) b = 60 + 4 * t + (c + 1) % 4;
            if (variant == 3) a = 40 + c;
            const int acc = 4 * (i * 2 + t);
            snprintf(buf, sizeof buf, "v_mfma_f32_16x16x4_f32 a[%d:%d], v%d, v%d, a[%d:%d]\n", acc, acc + 3, a, b, acc, acc + 3);
            s += buf;
        }
    }
    return s;
}


__global__ __launch_bounds__(256) void k0(unsigned long long *out, int iters, float seed) {
    // operands: small per-lane values; accumulators start at zero
    asm volatile(
        "v_cvt_f32_u32 v40, %0\n"
        "v_add_f32 v41, 1.25, v40\n"
        "v_add_f32 v42, 1.50, v40\n"
        "v_add_f32 v43, 1.75, v40\n"
        "v_add_f32 v44, 2.00, v40\n"
        "v_add_f32 v45, 2.25, v40\n"
        "v_add_f32 v46, 2.50, v40\n"
        "v_add_f32 v47, 2.75, v40\n"
        "v_add_f32 v48, 3.00, v40\n"
        "v_add_f32 v49, 3.25, v40\n"
        "v_add_f32 v50, 3.50, v40\n"
        "v_add_f32 v51, 3.75, v40\n"
        "v_add_f32 v52, 4.00, v40\n"
        "v_add_f32 v53, 4.25, v40\n"
        "v_add_f32 v54, 4.50, v40\n"
        "v_add_f32 v55, 4.75, v40\n"
        "v_add_f32 v56, 5.00, v40\n"
        "v_add_f32 v57, 5.25, v40\n"
        "v_add_f32 v58, 5.50, v40\n"
        "v_add_f32 v59, 5.75, v40\n"
        "v_add_f32 v60, 6.00, v40\n"
        "v_add_f32 v61, 6.25, v40\n"
        "v_add_f32 v62, 6.50, v40\n"
        "v_add_f32 v63, 6.75, v40\n"
        "v_add_f32 v64, 7.00, v40\n"
        "v_add_f32 v65, 7.25, v40\n"
        "v_add_f32 v66, 7.50, v40\n"
        "v_add_f32 v67, 7.75, v40\n"
        "v_accvgpr_write_b32 a0, 0\n"
        "v_accvgpr_write_b32 a1, 0\n"
        "v_accvgpr_write_b32 a2, 0\n"
        "v_accvgpr_write_b32 a3, 0\n"
        "v_accvgpr_write_b32 a4, 0\n"
        "v_accvgpr_write_b32 a5, 0\n"
        "v_accvgpr_write_b32 a6, 0\n"
        "v_accvgpr_write_b32 a7, 0\n"
        "v_accvgpr_write_b32 a8, 0\n"
        "v_accvgpr_write_b32 a9, 0\n"
        "v_accvgpr_write_b32 a10, 0\n"
        "v_accvgpr_write_b32 a11, 0\n"
        "v_accvgpr_write_b32 a12, 0\n"
        "v_accvgpr_write_b32 a13, 0\n"
        "v_accvgpr_write_b32 a14, 0\n"
        "v_accvgpr_write_b32 a15, 0\n"
        "v_accvgpr_write_b32 a16, 0\n"
        "v_accvgpr_write_b32 a17, 0\n"
        "v_accvgpr_write_b32 a18, 0\n"
        "v_accvgpr_write_b32 a19, 0\n"
        "v_accvgpr_write_b32 a20, 0\n"
        "v_accvgpr_write_b32 a21, 0\n"
        "v_accvgpr_write_b32 a22, 0\n"
        "v_accvgpr_write_b32 a23, 0\n"
        "v_accvgpr_write_b32 a24, 0\n"
        "v_accvgpr_write_b32 a25, 0\n"
        "v_accvgpr_write_b32 a26, 0\n"
        "v_accvgpr_write_b32 a27, 0\n"
        "v_accvgpr_write_b32 a28, 0\n"
        "v_accvgpr_write_b32 a29, 0\n"
        "v_accvgpr_write_b32 a30, 0\n"
        "v_accvgpr_write_b32 a31, 0\n"
        "v_accvgpr_write_b32 a32, 0\n"
        "v_accvgpr_write_b32 a33, 0\n"
        "v_accvgpr_write_b32 a34, 0\n"
        "v_accvgpr_write_b32 a35, 0\n"
        "v_accvgpr_write_b32 a36, 0\n"
        "v_accvgpr_write_b32 a37, 0\n"
        "v_accvgpr_write_b32 a38, 0\n"
        "v_accvgpr_write_b32 a39, 0\n"
        :: "v"(threadIdx.x & 7) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
        "v_mfma_f32_16x16x4_f32 a[0:3], v40, v60, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v40, v60, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v40, v60, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v40, v60, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v40, v60, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v40, v60, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v40, v60, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v40, v60, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v40, v60, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v40, v60, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v40, v60, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v40, v60, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v40, v60, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v40, v60, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v40, v60, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v40, v60, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v40, v60, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v40, v60, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v40, v60, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v40, v60, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v40, v60, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v40, v60, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v40, v60, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v40, v60, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v40, v60, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v40, v60, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v40, v60, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v40, v60, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v40, v60, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v40, v60, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v40, v60, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v40, v60, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v40, v60, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v40, v60, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v40, v60, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v40, v60, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v40, v60, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v40, v60, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v40, v60, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v40, v60, a[36:39]\n"
        ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(s) :: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[wave * 2] = c1 - c0; out[wave * 2 + 1] = r1 - r0; }
    if (s == 12345.678f) out[0] = 0;
}

__global__ __launch_bounds__(256) void k1(unsigned long long *out, int iters, float seed) {
    // operands: small per-lane values; accumulators start at zero
    asm volatile(
        "v_cvt_f32_u32 v40, %0\n"
        "v_add_f32 v41, 1.25, v40\n"
        "v_add_f32 v42, 1.50, v40\n"
        "v_add_f32 v43, 1.75, v40\n"
        "v_add_f32 v44, 2.00, v40\n"
        "v_add_f32 v45, 2.25, v40\n"
        "v_add_f32 v46, 2.50, v40\n"
        "v_add_f32 v47, 2.75, v40\n"
        "v_add_f32 v48, 3.00, v40\n"
        "v_add_f32 v49, 3.25, v40\n"
        "v_add_f32 v50, 3.50, v40\n"
        "v_add_f32 v51, 3.75, v40\n"
        "v_add_f32 v52, 4.00, v40\n"
        "v_add_f32 v53, 4.25, v40\n"
        "v_add_f32 v54, 4.50, v40\n"
        "v_add_f32 v55, 4.75, v40\n"
        "v_add_f32 v56, 5.00, v40\n"
        "v_add_f32 v57, 5.25, v40\n"
        "v_add_f32 v58, 5.50, v40\n"
        "v_add_f32 v59, 5.75, v40\n"
        "v_add_f32 v60, 6.00, v40\n"
        "v_add_f32 v61, 6.25, v40\n"
        "v_add_f32 v62, 6.50, v40\n"
        "v_add_f32 v63, 6.75, v40\n"
        "v_add_f32 v64, 7.00, v40\n"
        "v_add_f32 v65, 7.25, v40\n"
        "v_add_f32 v66, 7.50, v40\n"
        "v_add_f32 v67, 7.75, v40\n"
        "v_accvgpr_write_b32 a0, 0\n"
        "v_accvgpr_write_b32 a1, 0\n"
        "v_accvgpr_write_b32 a2, 0\n"
        "v_accvgpr_write_b32 a3, 0\n"
        "v_accvgpr_write_b32 a4, 0\n"
        "v_accvgpr_write_b32 a5, 0\n"
        "v_accvgpr_write_b32 a6, 0\n"
        "v_accvgpr_write_b32 a7, 0\n"
        "v_accvgpr_write_b32 a8, 0\n"
        "v_accvgpr_write_b32 a9, 0\n"
        "v_accvgpr_write_b32 a10, 0\n"
        "v_accvgpr_write_b32 a11, 0\n"
        "v_accvgpr_write_b32 a12, 0\n"
        "v_accvgpr_write_b32 a13, 0\n"
        "v_accvgpr_write_b32 a14, 0\n"
        "v_accvgpr_write_b32 a15, 0\n"
        "v_accvgpr_write_b32 a16, 0\n"
        "v_accvgpr_write_b32 a17, 0\n"
        "v_accvgpr_write_b32 a18, 0\n"
        "v_accvgpr_write_b32 a19, 0\n"
        "v_accvgpr_write_b32 a20, 0\n"
        "v_accvgpr_write_b32 a21, 0\n"
        "v_accvgpr_write_b32 a22, 0\n"
        "v_accvgpr_write_b32 a23, 0\n"
        "v_accvgpr_write_b32 a24, 0\n"
        "v_accvgpr_write_b32 a25, 0\n"
        "v_accvgpr_write_b32 a26, 0\n"
        "v_accvgpr_write_b32 a27, 0\n"
        "v_accvgpr_write_b32 a28, 0\n"
        "v_accvgpr_write_b32 a29, 0\n"
        "v_accvgpr_write_b32 a30, 0\n"
        "v_accvgpr_write_b32 a31, 0\n"
        "v_accvgpr_write_b32 a32, 0\n"
        "v_accvgpr_write_b32 a33, 0\n"
        "v_accvgpr_write_b32 a34, 0\n"
        "v_accvgpr_write_b32 a35, 0\n"
        "v_accvgpr_write_b32 a36, 0\n"
        "v_accvgpr_write_b32 a37, 0\n"
        "v_accvgpr_write_b32 a38, 0\n"
        "v_accvgpr_write_b32 a39, 0\n"
        :: "v"(threadIdx.x & 7) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
        "v_mfma_f32_16x16x4_f32 a[0:3], v40, v60, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v40, v64, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v44, v60, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v44, v64, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v48, v60, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v48, v64, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v52, v60, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v52, v64, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v56, v60, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v56, v64, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v41, v61, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v41, v65, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v45, v61, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v45, v65, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v49, v61, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v49, v65, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v53, v61, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v53, v65, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v57, v61, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v57, v65, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v42, v62, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v42, v66, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v46, v62, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v46, v66, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v50, v62, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v50, v66, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v54, v62, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v54, v66, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v58, v62, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v58, v66, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v43, v63, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v43, v67, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v47, v63, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v47, v67, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v51, v63, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v51, v67, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v55, v63, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v55, v67, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v59, v63, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v59, v67, a[36:39]\n"
        ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(s) :: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[wave * 2] = c1 - c0; out[wave * 2 + 1] = r1 - r0; }
    if (s == 12345.678f) out[0] = 0;
}

__global__ __launch_bounds__(256) void k2(unsigned long long *out, int iters, float seed) {
    // operands: small per-lane values; accumulators start at zero
    asm volatile(
        "v_cvt_f32_u32 v40, %0\n"
        "v_add_f32 v41, 1.25, v40\n"
        "v_add_f32 v42, 1.50, v40\n"
        "v_add_f32 v43, 1.75, v40\n"
        "v_add_f32 v44, 2.00, v40\n"
        "v_add_f32 v45, 2.25, v40\n"
        "v_add_f32 v46, 2.50, v40\n"
        "v_add_f32 v47, 2.75, v40\n"
        "v_add_f32 v48, 3.00, v40\n"
        "v_add_f32 v49, 3.25, v40\n"
        "v_add_f32 v50, 3.50, v40\n"
        "v_add_f32 v51, 3.75, v40\n"
        "v_add_f32 v52, 4.00, v40\n"
        "v_add_f32 v53, 4.25, v40\n"
        "v_add_f32 v54, 4.50, v40\n"
        "v_add_f32 v55, 4.75, v40\n"
        "v_add_f32 v56, 5.00, v40\n"
        "v_add_f32 v57, 5.25, v40\n"
        "v_add_f32 v58, 5.50, v40\n"
        "v_add_f32 v59, 5.75, v40\n"
        "v_add_f32 v60, 6.00, v40\n"
        "v_add_f32 v61, 6.25, v40\n"
        "v_add_f32 v62, 6.50, v40\n"
        "v_add_f32 v63, 6.75, v40\n"
        "v_add_f32 v64, 7.00, v40\n"
        "v_add_f32 v65, 7.25, v40\n"
        "v_add_f32 v66, 7.50, v40\n"
        "v_add_f32 v67, 7.75, v40\n"
        "v_accvgpr_write_b32 a0, 0\n"
        "v_accvgpr_write_b32 a1, 0\n"
        "v_accvgpr_write_b32 a2, 0\n"
        "v_accvgpr_write_b32 a3, 0\n"
        "v_accvgpr_write_b32 a4, 0\n"
        "v_accvgpr_write_b32 a5, 0\n"
        "v_accvgpr_write_b32 a6, 0\n"
        "v_accvgpr_write_b32 a7, 0\n"
        "v_accvgpr_write_b32 a8, 0\n"
        "v_accvgpr_write_b32 a9, 0\n"
        "v_accvgpr_write_b32 a10, 0\n"
        "v_accvgpr_write_b32 a11, 0\n"
        "v_accvgpr_write_b32 a12, 0\n"
        "v_accvgpr_write_b32 a13, 0\n"
        "v_accvgpr_write_b32 a14, 0\n"
        "v_accvgpr_write_b32 a15, 0\n"
        "v_accvgpr_write_b32 a16, 0\n"
        "v_accvgpr_write_b32 a17, 0\n"
        "v_accvgpr_write_b32 a18, 0\n"
        "v_accvgpr_write_b32 a19, 0\n"
        "v_accvgpr_write_b32 a20, 0\n"
        "v_accvgpr_write_b32 a21, 0\n"
        "v_accvgpr_write_b32 a22, 0\n"
        "v_accvgpr_write_b32 a23, 0\n"
        "v_accvgpr_write_b32 a24, 0\n"
        "v_accvgpr_write_b32 a25, 0\n"
        "v_accvgpr_write_b32 a26, 0\n"
        "v_accvgpr_write_b32 a27, 0\n"
        "v_accvgpr_write_b32 a28, 0\n"
        "v_accvgpr_write_b32 a29, 0\n"
        "v_accvgpr_write_b32 a30, 0\n"
        "v_accvgpr_write_b32 a31, 0\n"
        "v_accvgpr_write_b32 a32, 0\n"
        "v_accvgpr_write_b32 a33, 0\n"
        "v_accvgpr_write_b32 a34, 0\n"
        "v_accvgpr_write_b32 a35, 0\n"
        "v_accvgpr_write_b32 a36, 0\n"
        "v_accvgpr_write_b32 a37, 0\n"
        "v_accvgpr_write_b32 a38, 0\n"
        "v_accvgpr_write_b32 a39, 0\n"
        :: "v"(threadIdx.x & 7) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
        "v_mfma_f32_16x16x4_f32 a[0:3], v40, v61, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v40, v65, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v44, v61, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v44, v65, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v48, v61, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v48, v65, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v52, v61, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v52, v65, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v56, v61, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v56, v65, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v41, v62, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v41, v66, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v45, v62, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v45, v66, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v49, v62, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v49, v66, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v53, v62, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v53, v66, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v57, v62, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v57, v66, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v42, v63, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v42, v67, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v46, v63, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v46, v67, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v50, v63, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v50, v67, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v54, v63, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v54, v67, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v58, v63, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v58, v67, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v43, v60, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v43, v64, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v47, v60, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v47, v64, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v51, v60, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v51, v64, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v55, v60, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v55, v64, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v59, v60, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v59, v64, a[36:39]\n"
        ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(s) :: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[wave * 2] = c1 - c0; out[wave * 2 + 1] = r1 - r0; }
    if (s == 12345.678f) out[0] = 0;
}

__global__ __launch_bounds__(256) void k3(unsigned long long *out, int iters, float seed) {
    // operands: small per-lane values; accumulators start at zero
    asm volatile(
        "v_cvt_f32_u32 v40, %0\n"
        "v_add_f32 v41, 1.25, v40\n"
        "v_add_f32 v42, 1.50, v40\n"
        "v_add_f32 v43, 1.75, v40\n"
        "v_add_f32 v44, 2.00, v40\n"
        "v_add_f32 v45, 2.25, v40\n"
        "v_add_f32 v46, 2.50, v40\n"
        "v_add_f32 v47, 2.75, v40\n"
        "v_add_f32 v48, 3.00, v40\n"
        "v_add_f32 v49, 3.25, v40\n"
        "v_add_f32 v50, 3.50, v40\n"
        "v_add_f32 v51, 3.75, v40\n"
        "v_add_f32 v52, 4.00, v40\n"
        "v_add_f32 v53, 4.25, v40\n"
        "v_add_f32 v54, 4.50, v40\n"
        "v_add_f32 v55, 4.75, v40\n"
        "v_add_f32 v56, 5.00, v40\n"
        "v_add_f32 v57, 5.25, v40\n"
        "v_add_f32 v58, 5.50, v40\n"
        "v_add_f32 v59, 5.75, v40\n"
        "v_add_f32 v60, 6.00, v40\n"
        "v_add_f32 v61, 6.25, v40\n"
        "v_add_f32 v62, 6.50, v40\n"
        "v_add_f32 v63, 6.75, v40\n"
        "v_add_f32 v64, 7.00, v40\n"
        "v_add_f32 v65, 7.25, v40\n"
        "v_add_f32 v66, 7.50, v40\n"
        "v_add_f32 v67, 7.75, v40\n"
        "v_accvgpr_write_b32 a0, 0\n"
        "v_accvgpr_write_b32 a1, 0\n"
        "v_accvgpr_write_b32 a2, 0\n"
        "v_accvgpr_write_b32 a3, 0\n"
        "v_accvgpr_write_b32 a4, 0\n"
        "v_accvgpr_write_b32 a5, 0\n"
        "v_accvgpr_write_b32 a6, 0\n"
        "v_accvgpr_write_b32 a7, 0\n"
        "v_accvgpr_write_b32 a8, 0\n"
        "v_accvgpr_write_b32 a9, 0\n"
        "v_accvgpr_write_b32 a10, 0\n"
        "v_accvgpr_write_b32 a11, 0\n"
        "v_accvgpr_write_b32 a12, 0\n"
        "v_accvgpr_write_b32 a13, 0\n"
        "v_accvgpr_write_b32 a14, 0\n"
        "v_accvgpr_write_b32 a15, 0\n"
        "v_accvgpr_write_b32 a16, 0\n"
        "v_accvgpr_write_b32 a17, 0\n"
        "v_accvgpr_write_b32 a18, 0\n"
        "v_accvgpr_write_b32 a19, 0\n"
        "v_accvgpr_write_b32 a20, 0\n"
        "v_accvgpr_write_b32 a21, 0\n"
        "v_accvgpr_write_b32 a22, 0\n"
        "v_accvgpr_write_b32 a23, 0\n"
        "v_accvgpr_write_b32 a24, 0\n"
        "v_accvgpr_write_b32 a25, 0\n"
        "v_accvgpr_write_b32 a26, 0\n"
        "v_accvgpr_write_b32 a27, 0\n"
        "v_accvgpr_write_b32 a28, 0\n"
        "v_accvgpr_write_b32 a29, 0\n"
        "v_accvgpr_write_b32 a30, 0\n"
        "v_accvgpr_write_b32 a31, 0\n"
        "v_accvgpr_write_b32 a32, 0\n"
        "v_accvgpr_write_b32 a33, 0\n"
        "v_accvgpr_write_b32 a34, 0\n"
        "v_accvgpr_write_b32 a35, 0\n"
        "v_accvgpr_write_b32 a36, 0\n"
        "v_accvgpr_write_b32 a37, 0\n"
        "v_accvgpr_write_b32 a38, 0\n"
        "v_accvgpr_write_b32 a39, 0\n"
        :: "v"(threadIdx.x & 7) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
        "v_mfma_f32_16x16x4_f32 a[0:3], v40, v60, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v40, v64, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v40, v60, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v40, v64, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v40, v60, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v40, v64, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v40, v60, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v40, v64, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v40, v60, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v40, v64, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v41, v61, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v41, v65, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v41, v61, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v41, v65, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v41, v61, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v41, v65, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v41, v61, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v41, v65, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v41, v61, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v41, v65, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v42, v62, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v42, v66, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v42, v62, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v42, v66, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v42, v62, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v42, v66, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v42, v62, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v42, v66, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v42, v62, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v42, v66, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v43, v63, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v43, v67, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v43, v63, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v43, v67, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v43, v63, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v43, v67, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v43, v63, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v43, v67, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v43, v63, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v43, v67, a[36:39]\n"
        ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(s) :: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[wave * 2] = c1 - c0; out[wave * 2 + 1] = r1 - r0; }
    if (s == 12345.678f) out[0] = 0;
}

__global__ __launch_bounds__(256) void k4(unsigned long long *out, int iters, float seed) {
    // operands: small per-lane values; accumulators start at zero
    asm volatile(
        "v_cvt_f32_u32 v40, %0\n"
        "v_add_f32 v41, 1.25, v40\n"
        "v_add_f32 v42, 1.50, v40\n"
        "v_add_f32 v43, 1.75, v40\n"
        "v_add_f32 v44, 2.00, v40\n"
        "v_add_f32 v45, 2.25, v40\n"
        "v_add_f32 v46, 2.50, v40\n"
        "v_add_f32 v47, 2.75, v40\n"
        "v_add_f32 v48, 3.00, v40\n"
        "v_add_f32 v49, 3.25, v40\n"
        "v_add_f32 v50, 3.50, v40\n"
        "v_add_f32 v51, 3.75, v40\n"
        "v_add_f32 v52, 4.00, v40\n"
        "v_add_f32 v53, 4.25, v40\n"
        "v_add_f32 v54, 4.50, v40\n"
        "v_add_f32 v55, 4.75, v40\n"
        "v_add_f32 v56, 5.00, v40\n"
        "v_add_f32 v57, 5.25, v40\n"
        "v_add_f32 v58, 5.50, v40\n"
        "v_add_f32 v59, 5.75, v40\n"
        "v_add_f32 v60, 6.00, v40\n"
        "v_add_f32 v61, 6.25, v40\n"
        "v_add_f32 v62, 6.50, v40\n"
        "v_add_f32 v63, 6.75, v40\n"
        "v_add_f32 v64, 7.00, v40\n"
        "v_add_f32 v65, 7.25, v40\n"
        "v_add_f32 v66, 7.50, v40\n"
        "v_add_f32 v67, 7.75, v40\n"
        "v_accvgpr_write_b32 a0, 0\n"
        "v_accvgpr_write_b32 a1, 0\n"
        "v_accvgpr_write_b32 a2, 0\n"
        "v_accvgpr_write_b32 a3, 0\n"
        "v_accvgpr_write_b32 a4, 0\n"
        "v_accvgpr_write_b32 a5, 0\n"
        "v_accvgpr_write_b32 a6, 0\n"
        "v_accvgpr_write_b32 a7, 0\n"
        "v_accvgpr_write_b32 a8, 0\n"
        "v_accvgpr_write_b32 a9, 0\n"
        "v_accvgpr_write_b32 a10, 0\n"
        "v_accvgpr_write_b32 a11, 0\n"
        "v_accvgpr_write_b32 a12, 0\n"
        "v_accvgpr_write_b32 a13, 0\n"
        "v_accvgpr_write_b32 a14, 0\n"
        "v_accvgpr_write_b32 a15, 0\n"
        "v_accvgpr_write_b32 a16, 0\n"
        "v_accvgpr_write_b32 a17, 0\n"
        "v_accvgpr_write_b32 a18, 0\n"
        "v_accvgpr_write_b32 a19, 0\n"
        "v_accvgpr_write_b32 a20, 0\n"
        "v_accvgpr_write_b32 a21, 0\n"
        "v_accvgpr_write_b32 a22, 0\n"
        "v_accvgpr_write_b32 a23, 0\n"
        "v_accvgpr_write_b32 a24, 0\n"
        "v_accvgpr_write_b32 a25, 0\n"
        "v_accvgpr_write_b32 a26, 0\n"
        "v_accvgpr_write_b32 a27, 0\n"
        "v_accvgpr_write_b32 a28, 0\n"
        "v_accvgpr_write_b32 a29, 0\n"
        "v_accvgpr_write_b32 a30, 0\n"
        "v_accvgpr_write_b32 a31, 0\n"
        "v_accvgpr_write_b32 a32, 0\n"
        "v_accvgpr_write_b32 a33, 0\n"
        "v_accvgpr_write_b32 a34, 0\n"
        "v_accvgpr_write_b32 a35, 0\n"
        "v_accvgpr_write_b32 a36, 0\n"
        "v_accvgpr_write_b32 a37, 0\n"
        "v_accvgpr_write_b32 a38, 0\n"
        "v_accvgpr_write_b32 a39, 0\n"
        :: "v"(threadIdx.x & 7) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
        "v_mfma_f32_16x16x4_f32 a[0:3], v40, v60, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v44, v60, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v48, v60, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v52, v60, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v56, v60, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v40, v64, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v44, v64, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v48, v64, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v52, v64, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v56, v64, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v41, v61, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v45, v61, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v49, v61, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v53, v61, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v57, v61, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v41, v65, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v45, v65, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v49, v65, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v53, v65, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v57, v65, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v42, v62, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v46, v62, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v50, v62, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v54, v62, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v58, v62, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v42, v66, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v46, v66, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v50, v66, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v54, v66, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v58, v66, a[36:39]\n"
        "v_mfma_f32_16x16x4_f32 a[0:3], v43, v63, a[0:3]\n"
        "v_mfma_f32_16x16x4_f32 a[8:11], v47, v63, a[8:11]\n"
        "v_mfma_f32_16x16x4_f32 a[16:19], v51, v63, a[16:19]\n"
        "v_mfma_f32_16x16x4_f32 a[24:27], v55, v63, a[24:27]\n"
        "v_mfma_f32_16x16x4_f32 a[32:35], v59, v63, a[32:35]\n"
        "v_mfma_f32_16x16x4_f32 a[4:7], v43, v67, a[4:7]\n"
        "v_mfma_f32_16x16x4_f32 a[12:15], v47, v67, a[12:15]\n"
        "v_mfma_f32_16x16x4_f32 a[20:23], v51, v67, a[20:23]\n"
        "v_mfma_f32_16x16x4_f32 a[28:31], v55, v67, a[28:31]\n"
        "v_mfma_f32_16x16x4_f32 a[36:39], v59, v67, a[36:39]\n"
        ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(s) :: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[wave * 2] = c1 - c0; out[wave * 2 + 1] = r1 - r0; }
    if (s == 12345.678f) out[0] = 0;
}

template <typename K>
static void run(const char *what, K kern, int blocks, int iters, unsigned long long *d) {
    const int waves = blocks * 4;
    std::vector<unsigned long long> h(waves * 2);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> clk(waves), us(waves);
        for (int w = 0; w < waves; ++w) { us[w] = h[2 * w + 1] / 100.0; clk[w] = (double)h[2 * w] / ((double)iters * 40); }
        std::sort(clk.begin(), clk.end());
        std::sort(us.begin(), us.end());
        printf("%-62s %d waves/SIMD  median %6.2f shader clocks per MFMA  (loop %8.1f us, %6.1f TFLOP/s)\n", what, blocks / 256, clk[waves / 2],
               us[waves / 2], (double)waves * iters * 40 * 2048.0 / (us[waves / 2] * 1e-6) / 1e12);
    }
}

void run_all() {
    unsigned long long *d;
    hipMalloc(&d, 4096 * 2 * 8);
    const char *names[5] = {"V0 one A, one B register", "V1 kernel pattern: A, B of an MFMA in the same bank", "V2 B rotated: different banks",
                            "V3 A fixed per slot, B as V1", "V4 V1 ordered t-major (B changes every fifth)"};
    for (int w = 1; w <= 2; ++w) {
        run(names[0], k0, 256 * w, 2000 / w, d);
        run(names[1], k1, 256 * w, 2000 / w, d);
        run(names[2], k2, 256 * w, 2000 / w, d);
        run(names[3], k3, 256 * w, 2000 / w, d);
        run(names[4], k4, 256 * w, 2000 / w, d);
    }
}

int main() {
    // generate the kernels' source, compile it at run time with hiprtc?  Simpler: print the asm bodies so that the five kernels below
    // can be checked against them -- the bodies are pasted in by the macro generator at the bottom of this file.
    for (int v = 0; v < 5; ++v) {
        if (getenv("MFMA_BANKS_PRINT")) printf("---- V%d\n%s", v, body(v).c_str());
    }
    run_all();
    return 0;
}
