// Shared pieces of the attention kernels (attn.hip: forward and the two-kernel backward; attn_bwd.hip: the fused backward):
// the swizzled 64-row LDS tile, its LDS-DMA ring, accumulator -> operand conversion.
#pragma once
#include "common.hpp"

namespace octmae {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int HD>
struct Tile {  // 64 rows x HD bf16, row-major, XOR-swizzled 16-byte chunks
  static constexpr int ROWB = HD * 2;
  static constexpr int BYTES = 64 * ROWB;
  static constexpr int CHUNKS = HD / 8;
  __device__ static __forceinline__ int sw(int row) {
    if (HD == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return (row >> 2) & 3;
  }
  __device__ static __forceinline__ int off(int row, int c) { return row * ROWB + ((c ^ sw(row)) << 4); }

  // A-operand fragment, rows on the MFMA row index: element j = T[rb + r][16 s + 8 h + j]
  __device__ static __forceinline__ bf16x8 row_frag(const char* t, int rb, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(t + off(rb + (lane & 31), 2 * s + (lane >> 5)));
  }
  // A-operand fragment of T^T matching an accumulator-derived B operand (k order of a 32x32 accumulator):
  // element j = T[kb + 16 s + 8 (j>>2) + 4 h + (j&3)][db + r]
  __device__ static __forceinline__ bf16x8 tr_frag(const char* t, int kb, int s, int db, int lane) {
    const int h = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const int c = (db >> 3) + 2 * gi + (p >> 1);
    const int r0 = kb + 16 * s + 4 * h + q;
    const bf16x4 lo = lds_tr_read(t + off(r0, c) + (p & 1) * 8);
    const bf16x4 hi = lds_tr_read(t + off(r0 + 8, c) + (p & 1) * 8);
    return cat4(lo, hi);
  }
};

// ---- hand-placed LDS addressing (attn_bwd.hip `units`, attn.hip forward): per-lane byte offsets computed once, kept opaque so
// that the compiler does not re-derive them from the lane id inside the loops, and dereferenced as address-space-3 pointers
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}
// LDS accesses by byte OFFSET (address space 3 stated explicitly: pointer arithmetic through integers would otherwise come
// back as generic pointers and flat_load / flat_store)
#define OCTMAE_LDS_PTR(T, off) (reinterpret_cast<__attribute__((address_space(3))) T*>((__attribute__((address_space(3))) char*)(size_t)(off)))
template <class T>
__device__ __forceinline__ T lds_ld(unsigned off) { return *OCTMAE_LDS_PTR(const T, off); }
template <class T>
__device__ __forceinline__ void lds_st(unsigned off, T v) { *OCTMAE_LDS_PTR(T, off) = v; }
__device__ __forceinline__ bf16x4 lds_tr_ld(unsigned off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(OCTMAE_LDS_PTR(bf16x4, off));
}

// (query/key block, head, batch) of this workgroup.  The hardware hands consecutive linear block ids to the 8 XCDs round
// robin, so with the plain (x, y, z) mapping the ~N/128 workgroups that share one (batch, head)'s K/V land on all 8 XCDs and
// every XCD's L2 fetches those K/V separately (measured: 5.5x the algorithmic HBM bytes).  xcd_remap gives each XCD a
// contiguous range of logical ids instead, i.e. whole (batch, head) groups.
struct BlockCoord { int x, head, b; };
__device__ __forceinline__ BlockCoord attn_block_coord() {
  const int gx = gridDim.x, gy = gridDim.y;
  const int n = gx * gy * (int)gridDim.z;
#ifdef ATT_NO_XCD_REMAP
  const int lin = (int)(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z));
#else
  const int lin = xcd_remap((int)(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z)), n);
#endif
  return BlockCoord{lin % gx, (lin / gx) % gy, lin / (gx * gy)};
}

// ---- K/V tile ring filled by LDS-DMA -------------------------------------------------------------------------------
// Tiles go global -> LDS directly (buffer_load_dwordx4 ... lds: no staging registers, no ds_write, no VALU) into an
// NB-deep ring, NB-1 tiles ahead of their use.  The instruction is issued from inline asm on purpose: the compiler drains
// vmcnt to 0 before every LDS read it cannot prove disjoint from a tracked LDS-DMA (all ds_read_tr), which would serialise
// the ring; untracked, the only vmcnt waits in the loop are the counted ones in dma_wait_barrier().  Rows beyond the
// descriptor's range (keys >= N, and whole tiles past the last one, which are still fetched so that every step issues the
// same number of loads) read as zero.
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

template <int HD, int NWAVES>
struct TileDma {
  using T = Tile<HD>;
  static constexpr int PIECES = T::BYTES / 1024;            // 1-KiB pieces (64 lanes x 16 B) per tile
  static constexpr int PER_WAVE = PIECES / NWAVES;           // DMA instructions per wave per tile
  static_assert(PIECES % NWAVES == 0 && PER_WAVE >= 1, "tile pieces must divide over the waves");
  i32x4_t rsrc;
  unsigned voff[PER_WAVE];      // per-lane byte offset of this wave's chunk within tile 0
  unsigned tile_stride;
  int wid;

  __device__ __forceinline__ void init(const bf16_t* base, size_t row_stride, int nrows, int wid_, int lane) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned nrec = (unsigned)(((size_t)(nrows - 1) * row_stride + HD) * 2);
    rsrc = i32x4_t{(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)nrec, 0x00020000};
    wid = __builtin_amdgcn_readfirstlane(wid_);
    tile_stride = (unsigned)(64 * row_stride * 2);
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int q = (wid * PER_WAVE + i) * 64 + lane;        // LDS chunk q  <-  global chunk (row, c ^ swizzle(row))
      const int row = q / T::CHUNKS, c = (q % T::CHUNKS) ^ T::sw(row);
      voff[i] = (unsigned)(((size_t)row * row_stride + c * 8) * 2);
    }
  }
  // tile index `t` -> LDS byte address `lds_tile` (wave-uniform)
  __device__ __forceinline__ void load(int t, unsigned lds_tile) const {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_tile + (unsigned)((wid * PER_WAVE + i) * 1024)));
      const unsigned off = voff[i] + (unsigned)t * tile_stride;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is "reserved"; nothing else in these kernels lives in it
      asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(off), "s"(rsrc) : "memory", "m0");
#pragma clang diagnostic pop
    }
  }
};

// s_waitcnt vmcnt(CNT) lgkmcnt(0); s_barrier  -- the CNT youngest LDS-DMA loads of this wave stay in flight across the barrier
template <int CNT>
__device__ __forceinline__ void dma_wait_barrier() {
  static_assert(CNT >= 0 && CNT < 64, "vmcnt is 6 bits");
  // gfx9 s_waitcnt immediate: vmcnt = {[15:14],[3:0]}, expcnt [6:4] = 7 (no wait), lgkmcnt [11:8] = 0
  __builtin_amdgcn_s_waitcnt((CNT & 15) | ((CNT >> 4) << 14) | (7 << 4) | (0 << 8));
#ifndef ATT_EXPERIMENT_NO_BARRIER
  __builtin_amdgcn_s_barrier();
#endif
}

// accumulator registers 8s..8s+7 -> bf16 B-operand fragment of k-step s
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int s) {
  u32x4 w;
  w[0] = pack2bf(a[8 * s + 0], a[8 * s + 1]);
  w[1] = pack2bf(a[8 * s + 2], a[8 * s + 3]);
  w[2] = pack2bf(a[8 * s + 4], a[8 * s + 5]);
  w[3] = pack2bf(a[8 * s + 6], a[8 * s + 7]);
  return __builtin_bit_cast(bf16x8, w);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// The attention kernels are bound by VALU issue (softmax), not by the MFMA pipe: rocprofv3 shows per-wave
// SQ_ACTIVE_INST_VALU x resident waves ~ 100 % of the SIMD, and tools/ubench/valu_rate.hip prices the opcodes at
// 2 cycles (v_fma/v_add/v_max), 4 (v_max3, v_cvt_pk_bf16_f32, every v_pk_*_f32) and 8 (v_exp_f32) per wave-instruction.
// So the per-score work is cut to  exp + cvt (+ max):
//   * Q (forward, dQ) / K (dK/dV) fragments are pre-multiplied by scale*log2(e) once per workgroup and the S accumulator
//     starts at the row constant (-running max, or -LSE*log2e in backward): the MFMA result IS the exp2 argument;
//   * (the softmax row sums can come from one extra MFMA against an all-ones operand instead of 32 v_add per tile --
//     ONES_SUM in the forward; off since the kernel time became ~ MFMA + VALU cycles, see there);
//   * the running max moves only when a row exceeds it by more than RESCALE_SLACK (log2 units), P <= 2^8.
constexpr float RESCALE_SLACK = 8.0f;

__device__ __forceinline__ bf16x8 scale_frag(u32x4 v, float s) {
  u32x4 w;
#pragma unroll
  for (int e = 0; e < 4; ++e) w[e] = pack2bf(bflo(v[e]) * s, bfhi(v[e]) * s);
  return __builtin_bit_cast(bf16x8, w);
}

}  // namespace octmae
