// Fused (single-pass) attention backward for gfx950, head_dim 32 and 64, arbitrary sequence length.
//
// Backward of softmax((q k^T) scale) v  (Pre-training/custom_util/video_vit.py:130-134; flash path: flash-attn MHA built at
// models_mae_joint_res_flash_attn.py:131-149) in ONE sweep over the scores: S, dP and exp are computed once per score and
// feed all three gradients -- 5 matrix products and 1 exp per score, against 7 and 2 in the two-kernel form of attn.hip
// (dQ kernel: S, dP, dQ; dK/dV kernel: S, dP, dV, dK).
//
// Work split (the guide's "key on the lane" structure, Appendix B 'Attention backward'):
//   * one workgroup = 8 waves (two per SIMD) per (batch, head); it walks the sequence's KEY BLOCKS of KB = 8 waves x KW keys
//     (512 keys at head_dim 32, 256 at 64) one after the other;
//   * inside a key block each wave owns KW keys: its K and V fragments stay in registers, dK^T and dV^T of those keys
//     accumulate in registers while the workgroup sweeps all 64-row query tiles (Q, dO and the two per-query constants arrive
//     by LDS-DMA in a 3-deep ring);
//   * S = Q K^T and dP = dO V^T are computed with the key on the MFMA lane (32x32x16), so P and dS are already the B operands
//     of dV^T += dO^T P and dK^T += Q^T dS; only dS crosses LDS, once, as a [key][query] image of the whole key block;
//   * dQ^T[d][q] += K^T[d][key] dS^T[key][q] is then taken over ALL keys of the block by 16x16x32 MFMAs -- every wave owns
//     16x16 output tiles of the 64-query tile, its K^T fragments are loop invariants -- so no fp32 partial sums cross waves
//     (head_dim 32 splits the keys in two halves: one 1-KiB exchange between two waves per tile);
//   * dQ is summed over key blocks in an fp32 workspace by the SAME lane of the SAME workgroup (read-modify-write in program
//     order): no atomics, bit-reproducible;
//   * the keys past the last full block (N = 5121 and 1281 leave ONE: the cls token) go to a second, small kernel that splits
//     the QUERY tiles over the waves instead, adds its share to the workspace, scales and writes dQ as bf16.
// Numerics as in attn.hip: bf16 operands (P, dS rounded to bf16 for the MFMAs), fp32 accumulation, S in the exp2 domain
// (K pre-multiplied by scale*log2e), row constants -lse*log2e and -delta as the initial accumulators.
#include <type_traits>

#include "attn_tile.hpp"
#include "attn_bwd_tail1.hpp"
#include "../../include/octmae.h"

namespace octmae {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ f32x4_t mfma16(bf16x8 a, bf16x8 b, f32x4_t c) {
  return mfma16x16(a, b, c);
}

#ifndef BWD_DMA_MID
#define BWD_DMA_MID(HD) ((HD) == 64)
#endif
#ifdef BWD_STAMP
// diagnostic build only: per-wave cycle sums of the tile loop's segments (never read by the kernel itself)
__device__ unsigned long long g_bwd_stamp[512 * 8 * 8];
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(i)                                   \
  do {                                             \
    const unsigned long long now__ = stamp();      \
    seg[i] += now__ - tlast;                       \
    tlast = now__;                                 \
  } while (0)
#else
#define STAMP(i)
#endif

template <int HD>
struct BwdCfg {
  static constexpr int NW = 8;                          // waves per workgroup
  static constexpr int KW = (HD == 32) ? 64 : 32;       // keys per wave
  static constexpr int NG = KW / 32;                    // 32-key groups per wave
  static constexpr int KB = NW * KW;                    // keys per block
  static constexpr int NB = 3;                          // Q / dO / constants ring depth (64-row tiles)
  static constexpr int KS = HD / 16, DB = HD / 32;
  static constexpr bool PAIR = (HD == 32);              // dQ keys split over a pair of waves
  static constexpr int KH = PAIR ? KB / 2 : KB;         // keys one wave covers in the dQ product
  static constexpr int QSTEPS = KH / 32;                // 16x16x32 k-steps per output tile
  static constexpr int NT = PAIR ? 1 : 2;               // dQ output tiles a wave read-modify-writes per query tile
  static constexpr int NIMG = 1;                        // dS images
  using T = Tile<HD>;
  static constexpr int PD = 2 * T::BYTES / 1024 / NW;   // LDS-DMA data pieces per wave per tile (Q and dO together)
  static_assert(PD >= 1 && PD * NW * 1024 == 2 * T::BYTES, "tile pieces must divide over the waves");
  // LDS map (the rings come first: their fragment reads reach everything from one base register through the offset field)
  static constexpr int QR = 0;                          // Q ring
  static constexpr int OR_ = QR + NB * T::BYTES;        // dO ring
  static constexpr int CR = OR_ + NB * T::BYTES;        // constants ring: [NB][2][64] f32
  // pair exchange [NW][64 lanes] f32x4
  static constexpr int XCH = CR + NB * 512;
  static constexpr int XCH_BYTES = PAIR ? NW * 1024 : 0;
  // workspace values of the tile in flight, brought in by LDS-DMA: [NW][NT][64 lanes] f32x4
  static constexpr int OLD = XCH + XCH_BYTES;
  static constexpr int OLD_BYTES = NW * NT * 1024;
  static constexpr int IMG = OLD + OLD_BYTES;           // dS image(s) [KB keys][64 queries] bf16; image 0 also stages the
  static constexpr int IMG_BYTES = KB * 128;            //   block's K rows [KB][HD] at the start of a key block
  static constexpr int LDS = IMG + NIMG * IMG_BYTES;
  static_assert(KB * HD * 2 <= IMG_BYTES, "K staging must fit one image");
  static_assert(LDS <= 160 * 1024, "LDS budget");
  // 8-byte chunk c of staged K row `row`: conflict-free transposed reads of rows {0..3, 8..11} / {4..7, 12..15}
  __device__ static __forceinline__ int kst_off(int row, int c) {
    if (HD == 32) return row * 64 + ((c ^ (((row >> 3) & 1) << 2)) << 3);
    const int sw = (row & 1) | (((row >> 2) & 1) << 1) | (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 3);
    return row * 128 + ((c ^ sw) << 3);
  }
};

// bit permutation of the low 4 row bits: conflict-free ds_write_b64 by 16 consecutive rows AND conflict-free transposed reads
// of rows {0..3, 8..11} / {4..7, 12..15} (attention backward; SQ_LDS_BANK_CONFLICT reads 0.1 % of the LDS cycles)
__device__ __forceinline__ int img_sw(int row) {
  return (row & 1) | (((row >> 2) & 1) << 1) | (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 3);
}
// byte offset of the 8-byte chunk c8 (4 queries) of key row `row` in a dS image (128-byte rows = 64 queries)
__device__ __forceinline__ int img_off(int row, int c8) { return row * 128 + ((c8 ^ img_sw(row)) << 3); }

// s_waitcnt vmcnt(CNT) -- the CNT youngest vector-memory operations of this wave stay in flight
template <int CNT>
__device__ __forceinline__ void wait_vm() {
  static_assert(CNT >= 0 && CNT < 64, "vmcnt is 6 bits");
  __builtin_amdgcn_s_waitcnt((CNT & 15) | ((CNT >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

__device__ __forceinline__ void lds_dma16(unsigned m0v, unsigned voff, i32x4_t rsrc) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory", "m0");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void lds_dma4(unsigned m0v, unsigned voff, i32x4_t rsrc) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory", "m0");
#pragma clang diagnostic pop
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

// 8 bf16 A / B operand of a 32x32x16 or 16x16x32 MFMA from a PLAIN row-major [rows][rowb bytes] LDS array through two
// transposed reads: lane 4q+p of each 16-lane group supplies row (row0 + q), columns col0 + 4p .. +3; lane i of the group
// receives column col0 + i of those 4 rows; the second read takes the rows `dr` further down.
__device__ __forceinline__ bf16x8 tr_pair(const char* base, int rowb, int row0, int col0, int dr, int lane) {
  const int q = (lane >> 2) & 3, p = lane & 3;
  const char* a = base + (row0 + q) * rowb + (col0 + 4 * p) * 2;
  return cat4(lds_tr_read(a), lds_tr_read(a + dr * rowb));
}

// =====================================================================================================
// per-query constants, padded: rowc[0][bh][q] = -lse*log2e, rowc[1][bh][q] = -rowsum(dO * O); q in [N, NPAD): (-1e30, 0)
// (a padded query row then gives P = exp2(-1e30) = 0 and dS = 0 whatever the zero-filled Q / dO rows produce)
// =====================================================================================================
template <int HD>
__global__ __launch_bounds__(256) void attn_rowconst_pad_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                                const float* __restrict__ lse, float* __restrict__ rowc, int B,
                                                                int N, int NPAD, int H) {
  // One workgroup per 64 consecutive (padded) query rows of a sample: each wave reduces 16 rows (a row of O / dO is one or two
  // fully coalesced 1 KiB loads), the per-(row, head) results cross LDS and leave as 64 contiguous floats per head -- written
  // straight from the reducing lanes they were 4-byte stores into H different planes (0.46 ms per decoder layer, now 0.25).
  __shared__ float sh[16][64 + 1];              // delta: [head (<= 16 per pass)][row]
  const size_t plane = (size_t)B * H * NPAD;
  constexpr int LPH = HD / 8;                   // lanes per head
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int D = H * HD, nchunk = D / 8;
  const int tiles_per_b = NPAD / 64;
  for (int blk = blockIdx.x; blk < B * tiles_per_b; blk += gridDim.x) {
    const int b = blk / tiles_per_b, q0 = (blk % tiles_per_b) * 64;
    for (int h0 = 0; h0 < H; h0 += 16) {        // 16 heads (64 x LPH lanes = one or two passes over the chunks) at a time
      const int hn = (H - h0) < 16 ? (H - h0) : 16;
      for (int c0 = h0 * LPH; c0 < (h0 + hn) * LPH; c0 += 64) {
        const int c = c0 + lane;
        const bool cok = c < (h0 + hn) * LPH && c < nchunk;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
          const int rl = wid * 16 + i, q = q0 + rl;
          float sum = 0.f;
          if (cok && q < N) {
            const size_t src = ((size_t)b * N + q) * D + 8 * c;
            const u32x4 a = *reinterpret_cast<const u32x4*>(o + src);
            const u32x4 d = *reinterpret_cast<const u32x4*>(dout + src);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += bflo(a[e]) * bflo(d[e]) + bfhi(a[e]) * bfhi(d[e]);
          }
#pragma unroll
          for (int m = 1; m < LPH; m <<= 1) sum += __shfl_xor(sum, m, 64);
          if (cok && (lane % LPH) == 0) sh[c / LPH - h0][rl] = -sum;
        }
      }
      __syncthreads();
      // 256 threads x 4 rows: head = tid / 16, rows 4 (tid % 16) .. + 3
      const int hh = tid >> 4, r4 = (tid & 15) * 4;
      if (hh < hn) {
        const size_t i0 = ((size_t)b * H + h0 + hh) * NPAD + q0 + r4;
        f32x4 vl, vd;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = q0 + r4 + e;
          vl[e] = (q < N) ? -lse[((size_t)b * H + h0 + hh) * N + q] * LOG2E : -1.0e30f;
          vd[e] = (q < N) ? sh[hh][r4 + e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(rowc + i0) = vl;
        *reinterpret_cast<f32x4*>(rowc + plane + i0) = vd;
      }
      __syncthreads();
    }
  }
}

// The same padded planes from a delta that already exists: delta[B * N][H] (fp32, -rowsum(dO * O) per token row and head), written by
// the epilogue of the proj dgrad GEMM that produced dO (octmae_linear_dgrad_delta, csrc/gemm.hip) -- no second pass over O and dO.
// One workgroup per 64 (padded) query rows of a sample: the [64][H] block of delta is contiguous; it crosses LDS and leaves as 64
// contiguous floats per head, beside -lse * log2e.
__global__ __launch_bounds__(256) void attn_rowconst_from_delta_kernel(const float* __restrict__ delta, const float* __restrict__ lse,
                                                                       float* __restrict__ rowc, int B, int N, int NPAD, int H) {
  __shared__ float sh[64][64 + 1];              // [row][head]   (H <= 64)
  const size_t plane = (size_t)B * H * NPAD;
  const int tid = threadIdx.x;
  const int tiles_per_b = NPAD / 64;
  for (int blk = blockIdx.x; blk < B * tiles_per_b; blk += gridDim.x) {
    const int b = blk / tiles_per_b, q0 = (blk % tiles_per_b) * 64;
    for (int i = tid; i < 64 * H; i += 256) {
      const int rl = i / H, hh = i - rl * H;
      sh[rl][hh] = (q0 + rl < N) ? delta[((size_t)b * N + q0 + rl) * H + hh] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 16 * H; i += 256) {   // head i / 16, rows 4 (i % 16) .. + 3
      const int hh = i >> 4, r4 = (i & 15) * 4;
      const size_t i0 = ((size_t)b * H + hh) * NPAD + q0 + r4;
      f32x4 vl, vd;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = q0 + r4 + e;
        vl[e] = (q < N) ? -lse[((size_t)b * H + hh) * N + q] * LOG2E : -1.0e30f;
        vd[e] = sh[r4 + e][hh];
      }
      *reinterpret_cast<f32x4*>(rowc + i0) = vl;
      *reinterpret_cast<f32x4*>(rowc + plane + i0) = vd;
    }
    __syncthreads();
  }
}

// =====================================================================================================
// main kernel: the full key blocks
// =====================================================================================================
template <int HD>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ rowc, float* __restrict__ dq_ws,
                                                               bf16_t* __restrict__ dqkv, int N, int NPAD, int H, int nkb,
                                                               float scale) {
  using C = BwdCfg<HD>;
  using T = typename C::T;
  constexpr int KS = C::KS, DB = C::DB, NG = C::NG, NB = C::NB, PD = C::PD, NT = C::NT;
  extern __shared__ __attribute__((aligned(128))) char smem[];    // 128: the XOR chunk selectors below act on address bits 0..6
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // (batch, head) pair.  Consecutive block ids go to the 8 XCDs round robin; xcd_remap hands every XCD a CONTIGUOUS range of
  // pairs instead, i.e. all 16 heads of a sample: their Q / dO rows are adjacent 64- or 128-byte pieces of the same lines.
  // (Measured: no effect on the kernel time -- it is not memory bound -- but half the L2 misses.)
  const int bh = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int b = bh / H, head = bh % H;
  const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)head * HD;
  const bf16_t* kb_ = qb + (size_t)H * HD;
  const bf16_t* vb_ = qb + (size_t)2 * H * HD;
  const bf16_t* dob = dout + (size_t)b * N * os + (size_t)head * HD;
  const float sc2 = scale * LOG2E;
  const int ntiles = (N + 63) / 64;

  // ---- LDS-DMA plan of this wave: PD data pieces (Q pieces first, then dO pieces) + one constants row per tile
  const i32x4_t rsQ = make_rsrc(qb, (unsigned)(((size_t)(N - 1) * rs + HD) * 2));
  const i32x4_t rsO = make_rsrc(dob, (unsigned)(((size_t)(N - 1) * os + HD) * 2));
  constexpr int PIECES = T::BYTES / 1024;
  unsigned dvoff[PD], dlds[PD], dstride[PD];
  bool disq[PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) {
    const int gp = wid * PD + i;
    const bool isq = gp < PIECES;                // wave-uniform
    const int piece = isq ? gp : gp - PIECES;
    const int q = piece * 64 + lane;
    const int row = q / T::CHUNKS, c = (q % T::CHUNKS) ^ T::sw(row);
    const size_t stride = isq ? rs : os;
    dvoff[i] = (unsigned)(((size_t)row * stride + c * 8) * 2);
    dstride[i] = (unsigned)(64 * stride * 2);
    dlds[i] = (unsigned)((isq ? C::QR : C::OR_) + piece * 1024);
    disq[i] = isq;
  }
  const i32x4_t rsC = make_rsrc(rowc + (size_t)(wid & 1) * gridDim.x * NPAD + (size_t)bh * NPAD, (unsigned)((size_t)NPAD * 4));
  auto issue = [&](int tt) {                     // Q / dO / constants of tile tt -> ring slot tt % NB   (PD + 1 operations)
    const int slot = tt % NB;
#pragma unroll
    for (int i = 0; i < PD; ++i) {
      const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + dlds[i] + (unsigned)(slot * T::BYTES)));
      lds_dma16(m0v, dvoff[i] + (unsigned)tt * dstride[i], disq[i] ? rsQ : rsO);
    }
    const unsigned m0c = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + C::CR + slot * 512 + (wid & 1) * 256));
    lds_dma4(m0c, (unsigned)((tt * 64 + lane) * 4), rsC);
  };

  // ---- dQ workspace of this (batch, head): fp32 [N][HD]; rows >= N fall outside the descriptor (loads 0, stores dropped)
  float* wsb = dq_ws + (size_t)bh * N * HD;
  const i32x4_t rsW = make_rsrc(wsb, (unsigned)((size_t)N * HD * 4));            // for the LDS-DMA loads (inline asm)
  const __amdgpu_buffer_rsrc_t rsWs = __builtin_amdgcn_make_buffer_rsrc(           // the same range for the builtin stores
      reinterpret_cast<void*>(((unsigned long long)(unsigned)rsW[1] << 32) | (unsigned)rsW[0]), 0, rsW[2], 0x00020000);
  // this wave's dQ^T output tiles: rows d = 16 dt + 4 (lane >> 4) + e, columns q = 16 qt + (lane & 15)
  const int dt = C::PAIR ? ((wid >> 1) & 1) : (wid & 3);
  const int qpair = wid >> 2;
  const int khalf = C::PAIR ? (wid & 1) : 0;
  const int own = C::PAIR ? khalf : 0;            // PAIR: the tile (of its two) this wave keeps and read-modify-writes
  const int c16 = lane & 15, kg = lane >> 4;
  unsigned wsoff[2];                              // byte offset within tile 0's rows of output tile qt = 2 qpair + i
#pragma unroll
  for (int i = 0; i < 2; ++i) wsoff[i] = (unsigned)(((16 * (2 * qpair + i) + c16) * HD + 16 * dt + 4 * kg) * 4);
  const unsigned wsown = own ? wsoff[1] : wsoff[0];
  const unsigned ws_tile = (unsigned)(64 * HD * 4);
  // dQ product operands (transposed reads): dS image rows 32 ks + 8 kg + q4 (+4), chunk (4 qt + p) ^ img_sw(row); the second
  // output tile (qt + 1) is chunk bit 2, i.e. address ^ 32
  const int q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int p2row = khalf * C::KH + 8 * kg + q4;
  const int p2a0 = C::IMG + p2row * 128 + (((8 * qpair + p4) ^ img_sw((8 * kg + q4) & 15)) << 3);
  const int p2a1 = C::IMG + (p2row + 4) * 128 + (((8 * qpair + p4) ^ img_sw((8 * kg + q4 + 4) & 15)) << 3);

  // ---- per-lane address parts of `units` (byte offsets into the LDS, fixed for the whole kernel; `opaque` keeps the compiler
  // from re-deriving them from the lane id inside the tile loop).  T::sw() reads row bits 1..3 only, so neither the sub-tile
  // (+32 rows) nor the k-step of a transposed read (+16 rows) changes a swizzle.
  static_assert((T::BYTES % 128) == 0 && ((32 * T::ROWB) % 128) == 0 && (C::IMG % 128) == 0 && (C::IMG_BYTES % 128) == 0 && (C::CR % 128) == 0,
                "slot / sub-tile / region offsets must leave the low 7 address bits alone (XOR chunk selectors)");
  const int tq_ = (lane >> 2) & 3, tp_ = lane & 3, tgi = (lane >> 4) & 1;
  const unsigned a_const = opaque(lds0 + (unsigned)(C::CR + 16 * h));                                   // row constants: + 32 G (+ 256: delta)
  const unsigned a_row = opaque(lds0 + (unsigned)(C::QR + T::off(r, h)));                                // row fragment, k-step 0
  const unsigned a_trlo = opaque(lds0 + (unsigned)(C::QR + T::off(4 * h + tq_, 2 * tgi + (tp_ >> 1)) + (tp_ & 1) * 8));
  const unsigned a_trhi = opaque(lds0 + (unsigned)(C::QR + T::off(4 * h + tq_ + 8, 2 * tgi + (tp_ >> 1)) + (tp_ & 1) * 8));
  const unsigned a_img = opaque(lds0 + (unsigned)(C::IMG + img_off(wid * C::KW + r, h)));               // dS image, chunk h of this lane's key row

  for (int kb = 0; kb < nkb; ++kb) {
    const int key0 = kb * C::KB;
    // ---- stage this block's K rows in image 0 for the transposed reads of the loop-invariant K^T fragments
    {
      constexpr int CH = HD / 8;                 // 16-byte chunks per row
      constexpr int TOTAL = C::KB * CH;
#pragma unroll
      for (int i = 0; i < TOTAL / 512; ++i) {
        const int c = tid + 512 * i;
        const int row = c / CH, cc = c % CH;
        const u32x4 v = *reinterpret_cast<const u32x4*>(kb_ + (size_t)(key0 + row) * rs + 8 * cc);
        *reinterpret_cast<u32x2*>(smem + C::IMG + C::kst_off(row, 2 * cc)) = u32x2{v[0], v[1]};
        *reinterpret_cast<u32x2*>(smem + C::IMG + C::kst_off(row, 2 * cc + 1)) = u32x2{v[2], v[3]};
      }
    }
    // ---- this wave's keys: B operands of S (pre-scaled) and dP
    bf16x8 kS[NG][KS], vS[NG][KS];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const size_t krow = (size_t)(key0 + wid * C::KW + 32 * g + r) * rs;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const u32x4 kv = *reinterpret_cast<const u32x4*>(kb_ + krow + 16 * s + 8 * h);
        const u32x4 vv = *reinterpret_cast<const u32x4*>(vb_ + krow + 16 * s + 8 * h);
        kS[g][s] = scale_frag(kv, sc2);
        vS[g][s] = __builtin_bit_cast(bf16x8, vv);
      }
    }
    f32x16 dk[NG][DB], dv[NG][DB];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[g][d][e] = 0.f; dv[g][d][e] = 0.f; }
    __syncthreads();                              // K staging visible
    // A operand of dQ^T = K^T dS^T: K^T[row d = 16 dt + c16][k = key 32 ks + 8 kg + j], loop invariant, read once per block
    bf16x8 kT[C::QSTEPS];
#pragma unroll
    for (int ks = 0; ks < C::QSTEPS; ++ks)
      kT[ks] = cat4(lds_tr_read(smem + C::IMG + C::kst_off(p2row, 4 * dt + p4) + ks * 32 * HD * 2),
                    lds_tr_read(smem + C::IMG + C::kst_off(p2row + 4, 4 * dt + p4) + ks * 32 * HD * 2));
    wait_lgkm0();
    __syncthreads();                              // every wave has its K^T fragments: image 0 is free

    // ---- ring prologue
    issue(0);
    issue(1);
    wait_vm<PD + 1>();                            // tile 0 landed
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();

    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;   // dQ^T partial tiles (qt = 2 qpair, 2 qpair + 1) of the tile last multiplied
    const char* myold = smem + C::OLD + wid * NT * 1024 + lane * 16;

    // request the workspace values of tile tp's output tile(s) into this wave's OLD slot (NT operations; LDS-DMA, because a
    // register load would make the compiler drain the whole DMA ring at its first use)
    auto oldreq = [&](int tp) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const unsigned m0o = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + C::OLD + (wid * NT + i) * 1024));
        lds_dma16(m0o, (C::PAIR ? wsown : wsoff[i]) + (unsigned)tp * ws_tile, rsW);
      }
    };
    // finish(tp): tile tp's dQ^T -> workspace (NT stores).  Its partial sums are in acc0 / acc1 and, PAIR, in the partner's
    // exchange slot; YOUNGER operations of this wave allowed to stay in flight while OLD(tp) is waited for.
    auto finish = [&](int tp, auto younger_) {
      wait_vm<decltype(younger_)::value>();
      const f32x4_t o0 = *reinterpret_cast<const f32x4_t*>(myold);
      const f32x4_t o1 = *reinterpret_cast<const f32x4_t*>(myold + (NT - 1) * 1024);
      if (C::PAIR) {
        const f32x4_t other = *reinterpret_cast<const f32x4_t*>(smem + C::XCH + (wid ^ 1) * 1024 + lane * 16);
        f32x4_t v = (own ? acc1 : acc0) + other;
        if (kb > 0) v += o0;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsWs, wsown + (unsigned)tp * ws_tile, 0, 0);
      } else {
        f32x4_t v0 = acc0, v1 = acc1;
        if (kb > 0) {
          v0 += o0;
          v1 += o1;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), rsWs, wsoff[0] + (unsigned)tp * ws_tile, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), rsWs, wsoff[1] + (unsigned)tp * ws_tile, 0, 0);
      }
      wait_lgkm0();                               // the OLD / exchange reads have returned before anything rewrites them
    };
    // units(t): S / dP / P / dS of tile t against this wave's keys, dS -> image t % NIMG, dV^T / dK^T accumulate.  Per 32-query
    // sub-tile the row constants and the Q / dO fragments are read ONCE and shared by the wave's key groups.
    auto units = [&](int t, auto&& between) {
      const int slot = t % NB;
#pragma unroll 1
      for (int u = 0; u < 2; ++u) {
        // Addresses: every per-lane part is one of the kernel-lifetime values a_* (see their definition); what varies here --
        // ring slot and sub-tile -- is a multiple of 128 bytes and goes in with ONE add per address family, the k-step / head-dim
        // block / query-chunk selectors are XOR masks below 128 or immediate offsets.  (Left to the compiler this block took 38
        // address instructions per sub-tile in a loop that is bound by vector issue.)
        const unsigned X = (unsigned)(slot * T::BYTES + u * 32 * T::ROWB);
        const unsigned pc = a_const + (unsigned)(slot * 512 + u * 128);
        const unsigned prow = a_row + X, ptl = a_trlo + X, pth = a_trhi + X;
        const unsigned pw = (a_img + (unsigned)((t % C::NIMG) * C::IMG_BYTES)) ^ (unsigned)(u * 64);
        f32x16 lse_t, dlt_t;                      // row constants: register 4G+e <-> query row 32u + 8G + 4h + e
#pragma unroll
        for (int G = 0; G < 4; ++G) {
          const f32x4 a = lds_ld<f32x4>(pc + 32 * G);
          const f32x4 d = lds_ld<f32x4>(pc + 256 + 32 * G);
#pragma unroll
          for (int e = 0; e < 4; ++e) { lse_t[4 * G + e] = a[e]; dlt_t[4 * G + e] = d[e]; }
        }
        bf16x8 qrow[KS], orow[KS], qT[2][DB], oT[2][DB];
#pragma unroll
        for (int s = 0; s < KS; ++s) {            // T::row_frag(tile, 32 u, s): chunk 2 s + h  ->  address ^ 32 s
          const unsigned pr = prow ^ (unsigned)(32 * s);
          qrow[s] = lds_ld<bf16x8>(pr);
          orow[s] = lds_ld<bf16x8>(pr + (C::OR_ - C::QR));
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int d = 0; d < DB; ++d) {          // T::tr_frag(tile, 32 u, s, 32 d): rows + 16 s (immediate), chunk + 4 d -> ^ 64 d
            const unsigned pl = (ptl ^ (unsigned)(64 * d)) + s * 16 * T::ROWB;
            const unsigned ph = (pth ^ (unsigned)(64 * d)) + s * 16 * T::ROWB;
            qT[s][d] = cat4(lds_tr_ld(pl), lds_tr_ld(ph));
            oT[s][d] = cat4(lds_tr_ld(pl + (C::OR_ - C::QR)), lds_tr_ld(ph + (C::OR_ - C::QR)));
          }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          f32x16 sa = mfma32(qrow[0], kS[g][0], lse_t);                  // row constants as the initial accumulators
#pragma unroll
          for (int s = 1; s < KS; ++s) sa = mfma32(qrow[s], kS[g][s], sa);
          f32x16 dp = mfma32(orow[0], vS[g][0], dlt_t);
#pragma unroll
          for (int s = 1; s < KS; ++s) dp = mfma32(orow[s], vS[g][s], dp);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float p = fast_exp2(sa[e]);
            sa[e] = p;
            dp[e] = p * dp[e];                    // dS / scale
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const bf16x8 pf = acc_to_frag(sa, s);
            const bf16x8 dsf = acc_to_frag(dp, s);
            const u32x4 w = __builtin_bit_cast(u32x4, dsf);
            // registers 8s..8s+3 <-> queries 32u + 16s + 4h + 0..3 ; 8s+4..8s+7 <-> 32u + 16s + 8 + 4h + 0..3: image chunk
            // 8 u + 4 s + 2 k + h of key row wid * KW + 32 g + r  ->  address ^ (32 s + 16 k), + 32 g rows (immediate)
            lds_st<u32x2>((pw ^ (unsigned)(32 * s)) + g * 32 * 128, u32x2{w[0], w[1]});
            lds_st<u32x2>((pw ^ (unsigned)(32 * s + 16)) + g * 32 * 128, u32x2{w[2], w[3]});
#pragma unroll
            for (int d = 0; d < DB; ++d) {
              dv[g][d] = mfma32(oT[s][d], pf, dv[g][d]);
              dk[g][d] = mfma32(qT[s][d], dsf, dk[g][d]);
            }
          }
        }
        if (u == 0) between();
      }
    };
    // dqmul(tq): this wave's dQ^T tiles of tile tq over its KH keys of the image -> acc0 / acc1; PAIR: the tile this wave
    // does not keep goes to its exchange slot, where the partner picks it up
    auto dqmul = [&](int tq) {
      const int ib = (tq % C::NIMG) * C::IMG_BYTES;
      const char* a00 = smem + p2a0 + ib;
      const char* a01 = smem + p2a1 + ib;
      const char* a10 = smem + (p2a0 ^ 32) + ib;
      const char* a11 = smem + (p2a1 ^ 32) + ib;
      acc0 = f32x4_t{0.f, 0.f, 0.f, 0.f};
      acc1 = acc0;
#pragma unroll
      for (int ks = 0; ks < C::QSTEPS; ++ks) {
        const bf16x8 b0 = cat4(lds_tr_read(a00 + ks * 32 * 128), lds_tr_read(a01 + ks * 32 * 128));
        const bf16x8 b1 = cat4(lds_tr_read(a10 + ks * 32 * 128), lds_tr_read(a11 + ks * 32 * 128));
        acc0 = mfma16(kT[ks], b0, acc0);
        acc1 = mfma16(kT[ks], b1, acc1);
      }
      if (C::PAIR) *reinterpret_cast<f32x4_t*>(smem + C::XCH + wid * 1024 + lane * 16) = own ? acc0 : acc1;
    };
    using ICD = std::integral_constant<int, PD + 1>;

#ifdef BWD_STAMP
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast = stamp();
#endif
    for (int t = 0; t < ntiles; ++t) {
      // Where the tile's memory requests go.  head_dim 64: between the two sub-tiles of `units` -- issued among VALU work an
      // LDS-DMA piece costs 25-60 cycles, in a segment of their own 100-185 (MI355X_MICROARCH.md): same-box 6.18 -> 5.65 ms
      // (16 x 16 x 5121 x 64).  head_dim 32, whose units are VALU-bound already, keeps them at the head of the tile (neutral
      // there, and 6 more spilled registers).  Moving finish(t-1) there as well: 10 spilled registers at head_dim 64, +10 % time.
      if (BWD_DMA_MID(HD)) {
        if (t > 0) finish(t - 1, ICD{});
        STAMP(0);
        units(t, [&]() { oldreq(t); issue(t + 2); });
      } else {
        if (t > 0) finish(t - 1, ICD{});
        oldreq(t);
        issue(t + 2);
        STAMP(0);
        units(t, []() {});
      }
      STAMP(1);
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();               // A: the image of tile t is complete
      STAMP(2);
      dqmul(t);
      STAMP(3);
      if (t == 0) wait_vm<NT + PD + 1>();         // tile t+1 landed (younger: this tile's NT loads, tile t+2's DMAs ...
      else wait_vm<2 * NT + PD + 1>();            //  ... and, from the second tile on, the NT stores of finish(t-1))
      wait_lgkm0();
      STAMP(4);
      __builtin_amdgcn_s_barrier();               // B: image and ring slot free, exchange slots visible
      STAMP(5);
    }
#ifdef BWD_STAMP
    if (kb == 0 && lane == 0 && blockIdx.x < 512)
      for (int i = 0; i < 6; ++i) g_bwd_stamp[(blockIdx.x * 8 + wid) * 8 + i] = seg[i];
#endif
    finish(ntiles - 1, ICD{});
    // ---- dK, dV of this wave's keys
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      bf16_t* drow = dqkv + ((size_t)b * N + key0 + wid * C::KW + 32 * g + r) * rs + (size_t)head * HD;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int G = 0; G < 4; ++G) {
          const u32x2 wk = {pack2bf(dk[g][d][4 * G] * scale, dk[g][d][4 * G + 1] * scale),
                            pack2bf(dk[g][d][4 * G + 2] * scale, dk[g][d][4 * G + 3] * scale)};
          const u32x2 wv = {pack2bf(dv[g][d][4 * G], dv[g][d][4 * G + 1]), pack2bf(dv[g][d][4 * G + 2], dv[g][d][4 * G + 3])};
          *reinterpret_cast<u32x2*>(drow + (size_t)H * HD + d * 32 + 8 * G + 4 * h) = wk;
          *reinterpret_cast<u32x2*>(drow + (size_t)2 * H * HD + d * 32 + 8 * G + 4 * h) = wv;
        }
    }
    // the next block's workspace read-modify-write of a row is done by the same lane as this block's: program order;
    // drain everything (stale DMA of tiles past the end included) before the K staging reuses image 0
    wait_vm<0>();
    __syncthreads();
  }
}

// =====================================================================================================
// tail kernel: the keys past the last full block (query tiles split over the waves), then dQ workspace -> bf16
// =====================================================================================================
template <int HD>
__global__ __launch_bounds__(512, 2) void attn_bwd_tail_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                              const float* __restrict__ rowc, float* __restrict__ dq_ws,
                                                              bf16_t* __restrict__ dqkv, int N, int NPAD, int H, int key_start,
                                                              float scale) {
  constexpr int KS = HD / 16, DB = HD / 32, ROWB = HD * 2;
  constexpr int NW = 8;
  // LDS: K group [32][HD] | V group [32][HD] | per wave: Q sub-tile [32][HD], dO sub-tile [32][HD], dS image [32 keys][32 q]
  //      | reduction scratch [NW][64][16] f32
  constexpr int KG = 0, VG = KG + 32 * ROWB, WV = VG + 32 * ROWB, PERW = 2 * 32 * ROWB + 32 * 64, RED = WV + NW * PERW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, gi = (lane >> 4) & 1;
  const int bh = blockIdx.x;
  const int b = bh / H, head = bh % H;
  const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)head * HD;
  const bf16_t* kb_ = qb + (size_t)H * HD;
  const bf16_t* vb_ = qb + (size_t)2 * H * HD;
  const bf16_t* dob = dout + (size_t)b * N * os + (size_t)head * HD;
  const float* rc_l = rowc + (size_t)bh * NPAD;
  const float* rc_d = rowc + (size_t)gridDim.x * NPAD + (size_t)bh * NPAD;
  float* wsb = dq_ws + (size_t)bh * N * HD;
  const float sc2 = scale * LOG2E;
  const int nsub = (N + 31) / 32;
  const int ngroups = (N - key_start + 31) / 32;          // 0: nothing but the conversion
  char* myQ = smem + WV + wid * PERW;
  char* myO = myQ + 32 * ROWB;
  char* myS = myO + 32 * ROWB;

  // dQ rows of this lane for sub-tile j: query 32 j + r, head-dim runs 32 db + 8 G + 4 h .. + 3
  auto convert_only = [&]() {
    for (int j = wid; j < nsub; j += NW) {
      const int q = 32 * j + r;
      if (q >= N) continue;
      bf16_t* drow = dqkv + ((size_t)b * N + q) * rs + (size_t)head * HD;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int G = 0; G < 4; ++G) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(wsb + (size_t)q * HD + 32 * d + 8 * G + 4 * h);
          *reinterpret_cast<u32x2*>(drow + 32 * d + 8 * G + 4 * h) =
              u32x2{pack2bf(v[0] * scale, v[1] * scale), pack2bf(v[2] * scale, v[3] * scale)};
        }
    }
  };
  if (ngroups == 0) {
    convert_only();
    return;
  }

  for (int gk = 0; gk < ngroups; ++gk) {
    const int key0 = key_start + 32 * gk;
    const bool last = (gk == ngroups - 1);
    const bool have_ws = (key_start > 0) || (gk > 0);     // the workspace already holds a partial sum
    // ---- stage the K and V rows of this group (rows >= N: zeros)
    {
      constexpr int CH = HD / 8;
      for (int c = tid; c < 2 * 32 * CH; c += 512) {
        const int which = c / (32 * CH), cc = c % (32 * CH);
        const int row = cc / CH, col = cc % CH;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (key0 + row < N) v = *reinterpret_cast<const u32x4*>((which ? vb_ : kb_) + (size_t)(key0 + row) * rs + 8 * col);
        *reinterpret_cast<u32x4*>(smem + (which ? VG : KG) + row * ROWB + col * 16) = v;
      }
    }
    __syncthreads();
    bf16x8 kS[KS], vS[KS], kT[2][DB];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kS[s] = scale_frag(*reinterpret_cast<const u32x4*>(smem + KG + r * ROWB + (16 * s + 8 * h) * 2), sc2);
      vS[s] = *reinterpret_cast<const bf16x8*>(smem + VG + r * ROWB + (16 * s + 8 * h) * 2);
    }
    // A operand of dQ^T[d][q] = K^T dS^T (32x32x16, natural k order): A[row d = 32 db + r][k = key 16 s + 8 h + j]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int d = 0; d < DB; ++d) kT[s][d] = tr_pair(smem + KG, ROWB, 16 * s + 8 * h, 32 * d + 16 * gi, 4, lane);
    const bool key_live = (key0 + r) < N;
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk[d][e] = 0.f; dv[d][e] = 0.f; }

    // The sub-tile after the current one is requested (Q, dO rows and the row constants, into registers) before the current
    // one is computed: a wave walks ~N/256 sub-tiles back to back and nothing else hides the global-load latency
    // (without the prefetch this kernel took 330 us for ONE key at B = 32, N = 5121).
    constexpr int CHT = HD / 8, NCH = 32 * CHT / 64;
    u32x4 nq[NCH], no[NCH];
    f32x4 na[4], nd[4];
    auto prefetch = [&](int j) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        const int row = c / CHT, col = c % CHT;
        nq[i] = u32x4{0u, 0u, 0u, 0u};
        no[i] = nq[i];
        if (j < nsub && 32 * j + row < N) {
          nq[i] = *reinterpret_cast<const u32x4*>(qb + (size_t)(32 * j + row) * rs + 8 * col);
          no[i] = *reinterpret_cast<const u32x4*>(dob + (size_t)(32 * j + row) * os + 8 * col);
        }
      }
      const int jj = j < nsub ? j : wid;            // any valid sub-tile: NPAD covers every one
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        na[G] = *reinterpret_cast<const f32x4*>(rc_l + 32 * jj + 8 * G + 4 * h);
        nd[G] = *reinterpret_cast<const f32x4*>(rc_d + 32 * jj + 8 * G + 4 * h);
      }
    };
    prefetch(wid);
    for (int j = wid; j < nsub; j += NW) {
      // ---- this wave's own Q / dO sub-tile -> its private LDS region (rows >= N: zeros)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        const int row = c / CHT, col = c % CHT;
        *reinterpret_cast<u32x4*>(myQ + row * ROWB + col * 16) = nq[i];
        *reinterpret_cast<u32x4*>(myO + row * ROWB + col * 16) = no[i];
      }
      f32x16 sa, dp;
#pragma unroll
      for (int G = 0; G < 4; ++G)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sa[4 * G + e] = na[G][e]; dp[4 * G + e] = nd[G][e]; }
      prefetch(j + NW);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(myQ + r * ROWB + (16 * s + 8 * h) * 2);
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(myO + r * ROWB + (16 * s + 8 * h) * 2);
        sa = mfma32(qf, kS[s], sa);
        dp = mfma32(of, vS[s], dp);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p = fast_exp2(sa[e]);
        if (!key_live) p = 0.f;
        sa[e] = p;
        dp[e] = p * dp[e];
      }
      f32x16 dq[DB];
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[d][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = acc_to_frag(sa, s);
        const bf16x8 dsf = acc_to_frag(dp, s);
        const u32x4 w = __builtin_bit_cast(u32x4, dsf);
        // private dS image [key r][32 queries] (64-byte rows): queries 16 s + 4 h + 0..3 and 16 s + 8 + 4 h + 0..3
        *reinterpret_cast<u32x2*>(myS + r * 64 + (16 * s + 4 * h) * 2) = u32x2{w[0], w[1]};
        *reinterpret_cast<u32x2*>(myS + r * 64 + (16 * s + 8 + 4 * h) * 2) = u32x2{w[2], w[3]};
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          // A = dO^T / Q^T in the accumulator's k order: element j = T[16 s + 8 (j>>2) + 4 h + (j&3)][32 d + r]
          dv[d] = mfma32(tr_pair(myO, ROWB, 16 * s + 4 * h, 32 * d + 16 * gi, 8, lane), pf, dv[d]);
          dk[d] = mfma32(tr_pair(myQ, ROWB, 16 * s + 4 * h, 32 * d + 16 * gi, 8, lane), dsf, dk[d]);
        }
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]: B[k = key 16 s + 8 h + j][col q = r] from the private image
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bfrag = tr_pair(myS, 64, 16 * s + 8 * h, 16 * gi, 4, lane);
#pragma unroll
        for (int d = 0; d < DB; ++d) dq[d] = mfma32(kT[s][d], bfrag, dq[d]);
      }
      const int q = 32 * j + r;
      if (q < N) {
        bf16_t* drow = dqkv + ((size_t)b * N + q) * rs + (size_t)head * HD;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int G = 0; G < 4; ++G) {
            float* wp = wsb + (size_t)q * HD + 32 * d + 8 * G + 4 * h;
            f32x4 v = {dq[d][4 * G], dq[d][4 * G + 1], dq[d][4 * G + 2], dq[d][4 * G + 3]};
            if (have_ws) v += *reinterpret_cast<const f32x4*>(wp);
            if (last)
              *reinterpret_cast<u32x2*>(drow + 32 * d + 8 * G + 4 * h) =
                  u32x2{pack2bf(v[0] * scale, v[1] * scale), pack2bf(v[2] * scale, v[3] * scale)};
            else
              *reinterpret_cast<f32x4*>(wp) = v;
          }
      }
    }
    // ---- dK, dV of the group: sum of the 8 waves' partial tiles in a fixed order, written by wave 0
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + RED);
        const f32x16& src = which ? dv[d] : dk[d];
#pragma unroll
        for (int G = 0; G < 4; ++G)
          *reinterpret_cast<f32x4*>(red + ((wid * 64 + lane) * 16 + 4 * G)) = f32x4{src[4 * G], src[4 * G + 1], src[4 * G + 2], src[4 * G + 3]};
        __syncthreads();
        if (wid == 0 && key_live) {
          bf16_t* drow = dqkv + ((size_t)b * N + key0 + r) * rs + (size_t)head * HD + (size_t)(which ? 2 : 1) * H * HD;
          const float mul = which ? 1.f : scale;
#pragma unroll
          for (int G = 0; G < 4; ++G) {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w) s4 += *reinterpret_cast<const f32x4*>(red + ((w * 64 + lane) * 16 + 4 * G));
            *reinterpret_cast<u32x2*>(drow + 32 * d + 8 * G + 4 * h) =
                u32x2{pack2bf(s4[0] * mul, s4[1] * mul), pack2bf(s4[2] * mul, s4[3] * mul)};
          }
        }
      }
    __syncthreads();                               // K / V staging and the scratch are free for the next group
  }
}

// =====================================================================================================
// tail of exactly ONE key (N = nkb * KB + 1: 5121 = 10 * 512 + 1 and 1281 = 5 * 256 + 1, the token that the cls embedding adds),
// then dQ workspace -> bf16.  The general tail kernel spends whole 32-key MFMA tiles on that key (1.07 ms per decoder layer,
// 0.63 ms per encoder layer at micro-batch 128); one key is a rank-1 update -- a few fused multiply-adds per element on rows this
// pass has to stream anyway -- so this kernel runs at the rate of its memory traffic.
// One workgroup per (batch, head), HD / 8 lanes per query row (one 16-byte chunk each: a load instruction covers 8-16 whole
// rows of Q / dO and 1-2 KB of contiguous workspace), the next pass's rows requested before the current one is computed.
// Same rounding points as the MFMA path: K * scale * log2e, P and dS rounded to bf16 where that path feeds them to an MFMA; fp32 sums.
// =====================================================================================================
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_tail1_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                             const float* __restrict__ rowc, const float* __restrict__ dq_ws,
                                                             bf16_t* __restrict__ dqkv, int N, int NPAD, int H, int key, int have_ws,
                                                             float scale) {
  __shared__ float red[256 * 8];
  attn_bwd_tail1_body<HD, 1>(qkv, dout, rowc, dq_ws, dqkv, N, NPAD, H, key, have_ws, scale, xcd_remap((int)blockIdx.x, (int)gridDim.x),
                             (int)gridDim.x, red, (int)threadIdx.x);
}

// one wave per SIMD, head_dim 32 (attn_bwd1w.hip)
int launch_attn_bwd_fused1w(const bf16_t* qkv, const bf16_t* dout, const float* rowc, float* dq_ws, bf16_t* dqkv, int B, int N, int NPAD,
                            int H, int nkb, float scale, int tail_key, hipStream_t st);
// one wave per SIMD, head_dim 64 (attn_bwd1w64.hip)
int launch_attn_bwd_fused1w64(const bf16_t* qkv, const bf16_t* dout, const float* rowc, float* dq_ws, bf16_t* dqkv, int B, int N, int NPAD,
                              int H, int nkb, float scale, int tail_key, hipStream_t st);
std::atomic<int> g_attn_bwd_tail_fused{1};      // the single-key tail inside the one-wave main kernels (0: its own launch)
extern std::atomic<int> g_wgrad_stagger;    // defined in gemm.hip: staggered split-K slices of the weight-gradient kernel
extern std::atomic<int> g_wgrad_s1_atomic;  // defined in gemm.hip: fp32 atomics in the epilogue of an unsplit weight-gradient launch
extern std::atomic<int> g_gemm_small;       // defined in gemm.hip: the small-launch kernel (gemm128d_kernel) where the cost model picks it
extern std::atomic<int> g_small_launches, g_small_split_launches, g_small_wgrad_launches;
std::atomic<int> g_attn_bwd_hd32_form{1};
std::atomic<int> g_attn_bwd_hd64_form{1};

template <int HD>
static int run_fused(const bf16_t* qkv, const bf16_t* o, const bf16_t* dout, const float* lse, float* ws, bf16_t* dqkv, int B, int N,
                     int H, float scale, hipStream_t st, const float* delta = nullptr) {
  using C = BwdCfg<HD>;
  // one all-padding tile of row constants behind the last query tile: (-1e30, 0), i.e. P = 0 -- the one-wave-per-SIMD kernel
  // drains its software pipeline on it
  const int NPAD = (N + 63) / 64 * 64 + 64;
  float* dq_ws = ws;
  float* rowc = ws + (size_t)B * H * N * HD;
  {
    int blocks = B * (NPAD / 64);
    if (blocks > 8192) blocks = 8192;
    if (delta != nullptr) {   // delta came with dO (octmae_linear_dgrad_delta): transpose + pad only
      hipLaunchKernelGGL(attn_rowconst_from_delta_kernel, dim3(blocks), dim3(256), 0, st, delta, lse, rowc, B, N, NPAD, H);
    } else {
      hipLaunchKernelGGL(attn_rowconst_pad_kernel<HD>, dim3(blocks), dim3(256), 0, st, o, dout, lse, rowc, B, N, NPAD, H);
    }
    OCTMAE_LAUNCH_CHECK();
  }
  const int nkb = N / C::KB;
  // one key past the last full block: the one-wave main kernels take it (and the workspace -> bf16 conversion) at their end
  const bool one_key = N - nkb * C::KB == 1;
  bool tail_done = false;
  const int tail_key = (one_key && g_attn_bwd_tail_fused.load(std::memory_order_relaxed)) ? nkb * C::KB : -1;
  if (nkb > 0 && HD == 32 && g_attn_bwd_hd32_form.load(std::memory_order_relaxed) == 1) {
    if (int rc = launch_attn_bwd_fused1w(qkv, dout, rowc, dq_ws, dqkv, B, N, NPAD, H, nkb, scale, tail_key, st)) return rc;
    tail_done = tail_key >= 0;
  } else if (nkb > 0 && HD == 64 && g_attn_bwd_hd64_form.load(std::memory_order_relaxed) == 1) {
    if (int rc = launch_attn_bwd_fused1w64(qkv, dout, rowc, dq_ws, dqkv, B, N, NPAD, H, nkb, scale, tail_key, st)) return rc;
    tail_done = tail_key >= 0;
  } else if (nkb > 0) {
    static DynLdsOnce once;
    if (int rc = once.ensure(reinterpret_cast<const void*>(attn_bwd_fused_kernel<HD>), C::LDS)) return rc;
    hipLaunchKernelGGL(attn_bwd_fused_kernel<HD>, dim3(B * H), dim3(512), C::LDS, st, qkv, dout, rowc, dq_ws, dqkv, N, NPAD, H, nkb, scale);
    OCTMAE_LAUNCH_CHECK();
  }
  if (tail_done) return 0;
  if (one_key) {
    hipLaunchKernelGGL(attn_bwd_tail1_kernel<HD>, dim3(B * H), dim3(256), 0, st, qkv, dout, rowc, dq_ws, dqkv, N, NPAD, H, nkb * C::KB,
                       nkb > 0 ? 1 : 0, scale);
    OCTMAE_LAUNCH_CHECK();
  } else {
    constexpr int ROWB = HD * 2;
    constexpr int TAIL_LDS = 2 * 32 * ROWB + 8 * (2 * 32 * ROWB + 32 * 64) + 8 * 64 * 16 * 4;
    static DynLdsOnce once;
    if (int rc = once.ensure(reinterpret_cast<const void*>(attn_bwd_tail_kernel<HD>), TAIL_LDS)) return rc;
    hipLaunchKernelGGL(attn_bwd_tail_kernel<HD>, dim3(B * H), dim3(512), TAIL_LDS, st, qkv, dout, rowc, dq_ws, dqkv, N, NPAD, H,
                       nkb * C::KB, scale);
    OCTMAE_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace octmae
using namespace octmae;

#ifdef BWD_STAMP
extern "C" int octmae_debug_bwd_stamps(void* host, int nbytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_stamp), (size_t)nbytes);
}
#endif

extern "C" int octmae_set_option(const char* key, int value) {
  if (key == nullptr) return -1;
  if (__builtin_strcmp(key, "attn_bwd_hd32_form") == 0) return g_attn_bwd_hd32_form.exchange(value ? 1 : 0);
  if (__builtin_strcmp(key, "attn_bwd_hd64_form") == 0) return g_attn_bwd_hd64_form.exchange(value ? 1 : 0);
  if (__builtin_strcmp(key, "attn_bwd_tail_fused") == 0) return g_attn_bwd_tail_fused.exchange(value ? 1 : 0);
  if (__builtin_strcmp(key, "wgrad_stagger") == 0) return g_wgrad_stagger.exchange(value > 0 ? value : 0);
  if (__builtin_strcmp(key, "wgrad_s1_atomic") == 0) return g_wgrad_s1_atomic.exchange(value < 0 ? 0 : value > 2 ? 2 : value);
  if (__builtin_strcmp(key, "gemm_small") == 0) return g_gemm_small.exchange(value ? 1 : 0);      // csrc/gemm.hip
  if (__builtin_strcmp(key, "gemm_small_launches") == 0) return g_small_launches.load();      // read-only counters (tests)
  if (__builtin_strcmp(key, "gemm_small_split_launches") == 0) return g_small_split_launches.load();
  if (__builtin_strcmp(key, "gemm_small_wgrad_launches") == 0) return g_small_wgrad_launches.load();
  return -1;
}

extern "C" int octmae_attn_bwd_fused_ws_kib(int B, int N, int H, int HD) {
  if (B <= 0 || N <= 0 || H <= 0 || (HD != 32 && HD != 64)) return -1;
  const size_t npad = (size_t)(N + 63) / 64 * 64 + 64;
  const size_t bytes = ((size_t)B * H * N * HD + 2 * (size_t)B * H * npad) * 4;
  const size_t kib = (bytes + 1023) / 1024;
  return kib > 0x7fffffffull ? -2 : (int)kib;
}

extern "C" int octmae_attn_bwd_fused(const void* qkv, const void* o, const void* dout, const float* lse, void* ws, void* dqkv, int B,
                                     int N, int H, int HD, float scale, void* stream) {
  OCTMAE_CHECK_ARG(qkv && o && dout && lse && ws && dqkv && B > 0 && N > 0 && H > 0 && (HD == 64 || HD == 32));
  OCTMAE_CHECK_ARG(((size_t)N * 3 * H * HD * 2) < 0xFFFFFFFFull);      // one sample's qkv rows within a 32-bit buffer range
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* q = reinterpret_cast<const bf16_t*>(qkv);
  const bf16_t* oo = reinterpret_cast<const bf16_t*>(o);
  const bf16_t* dd = reinterpret_cast<const bf16_t*>(dout);
  bf16_t* out = reinterpret_cast<bf16_t*>(dqkv);
  float* w = reinterpret_cast<float*>(ws);
  return HD == 64 ? run_fused<64>(q, oo, dd, lse, w, out, B, N, H, scale, st) : run_fused<32>(q, oo, dd, lse, w, out, B, N, H, scale, st);
}

// The same with delta = -rowsum(dO * O) per token row and head (fp32 [B * N][H]) supplied by the caller -- written by the epilogue of
// the GEMM that produced dO (octmae_linear_dgrad_delta) -- instead of a pass over O and dO.
extern "C" int octmae_attn_bwd_fused_delta(const void* qkv, const void* dout, const float* lse, const float* delta, void* ws, void* dqkv, int B,
                                     int N, int H, int HD, float scale, void* stream) {
  OCTMAE_CHECK_ARG(qkv && delta && dout && lse && ws && dqkv && H <= 64 && B > 0 && N > 0 && H > 0 && (HD == 64 || HD == 32));
  OCTMAE_CHECK_ARG(((size_t)N * 3 * H * HD * 2) < 0xFFFFFFFFull);      // one sample's qkv rows within a 32-bit buffer range
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* q = reinterpret_cast<const bf16_t*>(qkv);
  const bf16_t* oo = nullptr;
  const bf16_t* dd = reinterpret_cast<const bf16_t*>(dout);
  bf16_t* out = reinterpret_cast<bf16_t*>(dqkv);
  float* w = reinterpret_cast<float*>(ws);
  return HD == 64 ? run_fused<64>(q, oo, dd, lse, w, out, B, N, H, scale, st, delta) : run_fused<32>(q, oo, dd, lse, w, out, B, N, H, scale, st, delta);
}
