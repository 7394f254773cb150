// LDS-DMA (global_load_lds_dwordx4) wave-instructions shared by the stream-K GEMMs and the persistent pairwise kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace pope {

static __device__ __attribute__((aligned(16))) float g_sk_zero[4];   // never written: the source of depth padding

// One LDS-DMA wave-instruction: lane l's 16 bytes at `src` land at LDS byte address lds_wave_base + 16 * l.
// Inline asm on purpose: beside a __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) waits vmcnt(0) in front of EVERY
// ds_read -- it cannot tell the buffer being filled from the buffer being read -- which serialises the prefetch with the
// MFMAs of the current stage.  The asm form is invisible to that bookkeeping; the stage loop waits for it itself
// (s_waitcnt vmcnt(0) + barrier, one stage later).  M0 holds the LDS base and is compiler-reserved: saved, written and
// restored inside the one statement (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void sk_glds16(const float *src, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_wave_base) : "memory");
}

// The same with the source as (wave-uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset): no vector arithmetic at all
// in front of the DMA -- f32 MFMAs run on the vector ALUs' issue slots, so every VALU instruction of a wave that shares a
// SIMD with an MFMA wave waits for a gap in its stream (stamps: 500 cycles per DMA with 64-bit per-lane address arithmetic).
__device__ __forceinline__ void sk_glds16_saddr(const float *base_uniform, unsigned lane_byte_off, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(base_uniform), "s"(lds_wave_base) : "memory");
}

}  // namespace pope
