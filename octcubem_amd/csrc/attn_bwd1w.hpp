// Pieces shared by the one-wave-per-SIMD attention backward kernels (attn_bwd1w.hip: head_dim 32, attn_bwd1w64.hip: head_dim 64).
#pragma once
#include "attn_tile.hpp"

namespace octmae {
namespace bwd1w_util {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return mfma16x16(a, b, c); }

template <int CNT>
__device__ __forceinline__ void wait_vm() {
  static_assert(CNT >= 0 && CNT < 64, "vmcnt is 6 bits");
  __builtin_amdgcn_s_waitcnt((CNT & 15) | ((CNT >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void lds_dma16(unsigned m0v, unsigned voff, i32x4_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory", "m0");
}
// the workspace traffic (fp32 dQ partial sums, read-modify-written once per key block) with its own cache policy switches:
// BWD1W_WS_ST_AUX = aux bits of the store (0 default, 2 = nt), -DBWD1W_WS_LD_NT = nt on the LDS-DMA loads of the old values
#ifndef BWD1W_WS_ST_AUX
#define BWD1W_WS_ST_AUX 0
#endif
// timing-only experiment (-DBWD1W_ST_SMALL): every write-out store lands in the first 64 KB of the (batch, head)'s workspace
#ifdef BWD1W_ST_SMALL
#define BWD1W_RVO(x) ((x) & 0xFFFFu)
#else
#define BWD1W_RVO(x) (x)
#endif
__device__ __forceinline__ void lds_dma16_ws(unsigned m0v, unsigned voff, i32x4_t rsrc) {
#ifdef BWD1W_WS_LD_NT
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory", "m0");
#else
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory", "m0");
#endif
}
__device__ __forceinline__ void lds_dma4(unsigned m0v, unsigned voff, i32x4_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ i32x4_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4_t r = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, 0x00020000};
  r[0] = __builtin_amdgcn_readfirstlane(r[0]);
  r[1] = __builtin_amdgcn_readfirstlane(r[1]);
  r[2] = __builtin_amdgcn_readfirstlane(r[2]);
  r[3] = __builtin_amdgcn_readfirstlane(r[3]);
  return r;
}
// 8-byte chunk c8 (4 bf16) of row `row` of a [rows][64 B] image: ds_write_b64 by 16 consecutive rows and the transposed reads of
// 4 consecutive rows (one aligned 256-byte line) are both conflict-free for any within-row permutation that separates the 8
// even (odd) rows of a 16-row run
__device__ __forceinline__ int img_off(int row, int c8) { return row * 64 + ((c8 ^ ((row >> 1) & 7)) << 3); }

// dV^T / dK^T accumulate in the ACCUMULATOR half of the register file (128 of this wave's 512 registers), where only MFMAs touch
// them; the MFMAs whose results the vector ALU consumes (S, dP, dQ^T) are the compiler's builtins with VGPR destinations
// (-mllvm -amdgpu-mfma-vgpr-form, see the Makefile).  One function cannot have both forms from builtins, hence the asm.  Hazards
// (the compiler pads nothing inside asm): the A / B operands are written by v_cvt_pk at least two instructions earlier (the
// generator pins their last producer), the accumulate chain needs no wait states, and the read-out after the loop sits behind
// explicit s_nops.
#define MFMA_ACC(acc, a, b) asm volatile(OCTMAE_MFMA32_ASM " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
// The lane id, computed where it is asked for (volatile: not merged with other copies, not hoisted): code behind the tile loop
// that needs per-lane addresses derives them from this instead of keeping them -- or the lane id -- in registers across the loop.
__device__ __forceinline__ int lane_id_fresh() {
  int x;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
  return x;
}
// end of a sub-step: this wave's dS image rows are written (LDS operations complete in order: at most the N reads issued after
// the last image write may still be pending), then the workgroup barrier
#ifdef ABL_NO_BARRIER
#define SUBSTEP_END(N) __builtin_amdgcn_s_waitcnt(0xC07F | 0)
#else
#define SUBSTEP_END(N)                                            \
  do {                                                            \
    __builtin_amdgcn_s_waitcnt(0xC07F | ((N) << 8));              \
    __builtin_amdgcn_s_barrier();                                 \
  } while (0)
#endif

}  // namespace bwd1w_util
}  // namespace octmae
