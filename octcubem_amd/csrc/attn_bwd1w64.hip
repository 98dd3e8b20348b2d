// Fused (single-pass) attention backward, head_dim 64, ONE WAVE PER SIMD: the main kernel of octmae_attn_bwd_fused at head_dim 64
// (encoder: 16 heads x 64, N = 1281).  Same algorithm, inputs, outputs and rounding points as attn_bwd.hip's
// attn_bwd_fused_kernel<64>; the structure is attn_bwd1w.hip's (head_dim 32) with the shapes of head_dim 64:
//   * a workgroup is 4 waves with 512 registers each; a wave owns 64 keys (two 32-key groups) of the 256-key block: dK^T / dV^T
//     of those keys (128 accumulator registers), their pre-scaled K and V fragments (64) and the K^T fragments of ITS 16 head
//     dims over all 256 keys (32) stay in registers for the whole sweep over the queries;
//   * per 32-query sub-tile a wave does S, dP (4 k-steps each), exp2, dS, dV^T, dK^T (2 head-dim blocks x 2 k-steps each) for
//     its two key groups -- 16 32x32x16 MFMAs per group for 16 exp2 per lane, twice the matrix work per exp of head_dim 32 --
//     and writes dS into the workgroup's [256 keys][32 queries] image;
//   * one sub-step later (behind the workgroup barrier) it multiplies K^T of its 16 head dims with the whole image: its
//     16 x 32 slice of dQ^T, as two 16 x 16 tiles over 8 k-steps of 32 keys -- complete sums, nothing crosses waves in fp32 --
//     adds the workspace value of the previous key blocks (brought in by LDS-DMA a tile ahead) and stores, another sub-step
//     later;
//   * Q / dO / row-constant tiles arrive by LDS-DMA through a 4-deep ring, retired by counted vmcnt.
// The body of the tile loop is generated (tools/gen_attn_bwd1w.py, sub_step64): it fixes the instruction order.
// Reference op: backward of softmax((q k^T) scale) v, Pre-training/custom_util/video_vit.py:130-134 under autograd.
#include "attn_bwd1w.hpp"
#include "attn_bwd_tail1.hpp"
#ifndef BWD1W_TAIL_DEPTH
#define BWD1W_TAIL_DEPTH 4      // passes of rows in flight in the single-key tail (8: measured equal)
#endif
#include "../../include/octmae.h"

namespace octmae {

namespace bwd1w64 {

constexpr int HD = 64, NW = 4, KW = 64, NG = 2, KB = NW * KW;
constexpr int LA = 3;                                   // tiles requested ahead of the one being computed
constexpr int NB = LA + 1;                              // Q / dO / constants ring depth
constexpr int NOLD = 3;                                 // workspace-value buffers: requested one tile before their tile, read one after
using T = Tile<HD>;                                     // 64 rows x 128 B, XOR-swizzled 16-byte chunks
constexpr int QR = 0;                                   // Q ring      [NB][8192]
constexpr int OR_ = QR + NB * T::BYTES;                 // dO ring     [NB][8192]
constexpr int CR = OR_ + NB * T::BYTES;                 // constants   [NB][2][64] f32
constexpr int IMG = CR + NB * 512;                      // dS images [2 sub-steps][256 keys][32 queries] bf16 (64-byte rows); wave w
constexpr int IMG_BUF = KB * 64;                        //   writes rows 64 w .. 64 w + 63, every wave reads all rows (transposed)
constexpr int OLD = IMG + 2 * IMG_BUF;                  // workspace values   [NOLD tiles][2 sub-tiles][2 query tiles][NW][64 lanes] f32x4
constexpr int OLD_TILE = 2 * 2 * NW * 1024;
constexpr int LDS = OLD + NOLD * OLD_TILE;
constexpr int STG = IMG;                                // K rows of the block [256][128 B], staged once per block for the K^T reads
static_assert(KB * HD * 2 <= 2 * IMG_BUF, "K staging must fit the two images");
static_assert(LDS <= 160 * 1024, "LDS budget");
static_assert((CR % 128) == 0 && (IMG % 128) == 0 && (OLD % 128) == 0, "XOR chunk selectors act on address bits 0..6");
static_assert((OR_ - QR) % 128 == 0 && T::BYTES % 128 == 0, "slot / region offsets leave the low 7 address bits alone");

using namespace bwd1w_util;

// 8-byte chunk c (of 16) of staged K row `row` (128-byte rows): conflict-free transposed reads of rows {0..3, 8..11} /
// {4..7, 12..15} (attn_bwd.hip, BwdCfg<64>::kst_off)
__device__ __forceinline__ int kst_off(int row, int c) {
  const int sw = (row & 1) | (((row >> 2) & 1) << 1) | (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 3);
  return row * 128 + ((c ^ sw) << 3);
}

// ---- diagnostic build (-DBWD1W_STAMP, make stamp): s_memtime at the start and the middle of both group-steps, before the
// end-of-sub-step wait and behind the barrier; per-wave sums of the 6 intervals of each sub-step (tools/attn_bwd1w_stamps.py).
#ifdef BWD1W_STAMP
__device__ unsigned g_bwd1w64_stamp[512 * 4 * 2 * 8];      // [6]: the vmcnt wait at the top of the tile (sub-step 0 only)
#define STAMP(k) asm volatile("s_memtime %0" : "=s"(st_[k]))
#define STAMP_ACCUM(s)                                                                                                         \
  do {                                                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_[0]), "+s"(st_[1]), "+s"(st_[2]), "+s"(st_[3]), "+s"(st_[4]), "+s"(st_[5])); \
    _Pragma("unroll") for (int k_ = 0; k_ < 5; ++k_) acc_[s][k_] += (unsigned)st_[k_ + 1] - (unsigned)st_[k_];                \
    acc_[s][5] += (unsigned)st_[0] - last_;                                                                                    \
    last_ = (unsigned)st_[5];                                                                                                  \
  } while (0)
#else
#define STAMP(k)
#define STAMP_ACCUM(s)
#endif

}  // namespace bwd1w64

__global__ __launch_bounds__(256, 1) void attn_bwd_fused1w64_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                    const float* __restrict__ rowc, float* __restrict__ dq_ws,
                                                                    bf16_t* __restrict__ dqkv, int N, int NPAD, int H, int nkb,
                                                                    float scale, int tail_key) {
  using namespace bwd1w64;
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int bh = xcd_remap((int)blockIdx.x, (int)gridDim.x);      // an XCD gets whole samples: their Q / dO rows share lines
  const int b = bh / H, head = bh % H;
  const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)head * HD;
  const bf16_t* kb_ = qb + (size_t)H * HD;
  const bf16_t* vb_ = qb + (size_t)2 * H * HD;
  const bf16_t* dob = dout + (size_t)b * N * os + (size_t)head * HD;
  const float sc2 = scale * LOG2E;
  const bool half_drain = ((N - 1) & 63) < 32;    // the last tile's rows 32 .. 63 are all >= N
  const int ntiles = (N + 63) / 64;               // NPAD = 64 (ntiles + 1): one all-padding tile of row constants behind the last

  // ---- LDS-DMA plan: a tile is 8 Q pieces + 8 dO pieces of 1 KiB; waves 0, 1 bring Q, waves 2, 3 dO (4 pieces each) and every
  // wave one row of constants (-lse*log2e: even waves, -delta: odd)
  const bool isq = wid < 2;                                       // wave-uniform
  const i32x4_t rsD = isq ? make_rsrc(qb, (unsigned)(((size_t)(N - 1) * rs + HD) * 2)) : make_rsrc(dob, (unsigned)(((size_t)(N - 1) * os + HD) * 2));
  unsigned dvoff[4], dlds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = (wid & 1) * 4 + i;
    const int q = piece * 64 + lane;
    const int row = q / T::CHUNKS, c = (q % T::CHUNKS) ^ T::sw(row);
    dvoff[i] = (unsigned)(((size_t)row * (isq ? rs : os) + c * 8) * 2);
    dlds[i] = (unsigned)((isq ? QR : OR_) + piece * 1024);
  }
  const unsigned dstride = (unsigned)(64 * (isq ? rs : os) * 2);
  const i32x4_t rsC = make_rsrc(rowc + (size_t)(wid & 1) * gridDim.x * NPAD + (size_t)bh * NPAD, (unsigned)((size_t)NPAD * 4));
  // Q / dO of tile `dt` and the constants of tile `ct` -> ring slot   (5 operations)
  auto issue = [&](int dt, int ct, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + dlds[i] + (unsigned)(slot * T::BYTES)));
      lds_dma16(m0v, dvoff[i] + (unsigned)dt * dstride, rsD);
    }
    const unsigned m0c = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + CR + slot * 512 + (wid & 1) * 256));
    lds_dma4(m0c, (unsigned)((ct * 64 + lane) * 4), rsC);
  };

  // ---- dQ workspace of this (batch, head): fp32 [N][64]; rows >= N fall outside the descriptor (loads 0, stores dropped).
  // Wave w owns head dims 16 w .. 16 w + 15 of every sub-step's dQ^T[64 head dims][32 queries] as two 16 x 16 tiles (queries
  // 0..15, 16..31): lane = (query c16, dims 4 g16 .. + 3), the C / D layout of the 16x16x32 MFMA.
  float* wsb = dq_ws + (size_t)bh * N * HD;
  const i32x4_t rsW = make_rsrc(wsb, (unsigned)((size_t)N * HD * 4));
  const __amdgpu_buffer_rsrc_t rsWs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)(unsigned)rsW[1] << 32) | (unsigned)rsW[0]), 0, rsW[2], 0x00020000);
  const int g16 = lane >> 4, c16 = lane & 15;
#ifdef ABL_TILED_WS
  const unsigned wsoff = (unsigned)(wid * 1024 + lane * 16);
#else
  const unsigned wsoff = (unsigned)((c16 * HD + 16 * wid + 4 * g16) * 4);
#endif
  constexpr unsigned WS_QT = 16 * HD * 4, WS_SUB = 32 * HD * 4, WS_TILE = 64 * HD * 4, DROP = 0x80000000u;
  static_assert(WS_QT == 4096, "the generated body stores the second query tile at + 4096");
  // workspace values of this wave's quads of `tile` -> OLD buffer   (4 operations; `base` = DROP in the first key block: zeros)
  auto oldreq = [&](int tile, int buf, unsigned base) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned m0o = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + OLD + buf * OLD_TILE + (u * NW + wid) * 1024));
      lds_dma16_ws(m0o, wsoff + base + (unsigned)tile * WS_TILE + (unsigned)(u >> 1) * WS_SUB + (unsigned)(u & 1) * WS_QT, rsW);
    }
  };

  // ---- per-lane LDS address parts, fixed for the whole kernel (opaque: not re-derived from the lane id inside the loops)
  const int tq_ = (lane >> 2) & 3, tp_ = lane & 3, tgi = (lane >> 4) & 1;
  const unsigned a_const = opaque(lds0 + (unsigned)(CR + 16 * h));
  const unsigned a_row = opaque(lds0 + (unsigned)(QR + T::off(r, h)));
  const unsigned a_trlo = opaque(lds0 + (unsigned)(QR + T::off(4 * h + tq_, 2 * tgi + (tp_ >> 1)) + (tp_ & 1) * 8));
  const unsigned a_trhi = opaque(lds0 + (unsigned)(QR + T::off(4 * h + tq_ + 8, 2 * tgi + (tp_ >> 1)) + (tp_ & 1) * 8));
  // dS image: this lane's key row 64 wid + r (+ 32 g), chunk h (^ 32 half + 16 k for the query chunk 4 half + 2 k + h) (+ image buffer)
  const unsigned a_imgw = opaque(lds0 + (unsigned)(IMG + img_off(wid * KW + r, h)));
  // transposed reads of 4-key x 16-query blocks for the 16x16x32 B operand dS^T[k = key 32 ks + 8 g16 + e][col q = 16 qt + c16]:
  // keys 8 g16 + tq_ (+ 4) (+ 32 ks), query chunk tp_ (^ 32: query tile 1) (+ image buffer)
  const unsigned a_imglo = opaque(lds0 + (unsigned)(IMG + img_off(8 * g16 + tq_, tp_)));
  const unsigned a_imghi = opaque(lds0 + (unsigned)(IMG + img_off(8 * g16 + tq_ + 4, tp_)));
  const unsigned a_old = opaque(lds0 + (unsigned)(OLD + wid * 1024 + lane * 16));
  f32x16 zero16;
#pragma unroll
  for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

#ifdef BWD1W_STAMP
  unsigned long long k0_, r0_;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(k0_), "=s"(r0_));
#endif
  for (int kb = 0; kb < nkb; ++kb) {
    const int key0 = kb * KB;
    const unsigned oldbase = kb > 0 ? 0u : DROP;
    // ---- ring prologue: tiles 0 .. 2, workspace values of tile 0 -- requested first, their latency runs beside the K staging.
    // (The ring, the constants and the workspace-value buffers are not the image region; LDS-DMA writes of one wave land in
    // issue order, so the surplus tiles the previous block's loop left in flight need no drain.)
    issue(0, 0, 0);
    issue(1, 1, 1);
    issue(2, 2, 2);
    oldreq(0, 0, oldbase);
    // ---- stage this block's K rows for the transposed reads of the loop-invariant K^T fragments
    {
#pragma unroll
      for (int i = 0; i < KB * 8 / 256; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 3, cc = c & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(kb_ + (size_t)(key0 + row) * rs + 8 * cc);
        *reinterpret_cast<u32x2*>(smem + STG + kst_off(row, 2 * cc)) = u32x2{v[0], v[1]};
        *reinterpret_cast<u32x2*>(smem + STG + kst_off(row, 2 * cc + 1)) = u32x2{v[2], v[3]};
      }
    }
    // ---- this wave's keys: B operands of S (pre-scaled) and dP
    bf16x8 kS[NG][4], vS[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const size_t krow = (size_t)(key0 + wid * KW + 32 * g + r) * rs;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4 kv = *reinterpret_cast<const u32x4*>(kb_ + krow + 16 * s + 8 * h);
        const u32x4 vv = *reinterpret_cast<const u32x4*>(vb_ + krow + 16 * s + 8 * h);
        kS[g][s] = scale_frag(kv, sc2);
        vS[g][s] = __builtin_bit_cast(bf16x8, vv);
      }
    }
    // dk{g}{d} / dv{g}{d}: key group g, head dims 32 d .. 32 d + 31
    f32x16 dk00 = zero16, dk01 = zero16, dk10 = zero16, dk11 = zero16, dv00 = zero16, dv01 = zero16, dv10 = zero16, dv11 = zero16;
    __syncthreads();                              // K staging visible
    // A operand of dQ^T = K^T dS^T (16x16x32): K^T[row d = 16 wid + c16][k = key 32 ks + 8 g16 + e], all 256 keys, loop invariant
    bf16x8 kT[KB / 32];
#pragma unroll
    for (int ks = 0; ks < KB / 32; ++ks)
      kT[ks] = cat4(lds_tr_read(smem + STG + kst_off(8 * g16 + tq_, 4 * wid + tp_) + ks * 32 * 128),
                    lds_tr_read(smem + STG + kst_off(8 * g16 + tq_ + 4, 4 * wid + tp_) + ks * 32 * 128));
    __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0)
    __syncthreads();                              // every wave has its K^T fragments: the image region is free

    // (every load the compiler tracks has returned: without this it carries "loads pending" into the tile loop and waits there
    // with a vmcnt that also drains the hand-counted LDS-DMA ring; the ring prologue, requested before the K staging, is
    // complete with them)
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    // ---- pipeline prologue: fragments of sub-step (0, 0), S / dP of its first key group; everything the first C1 / D / write-out
    // of the loop consume without a producer is zero or is dropped (the dQ^T tiles of "sub-steps -2, -1": stores outside the range)
    f32x16 lse_t, dlt_t, saA, dpA, saB = zero16, dpB = zero16;
    f32x4 dqA0 = zero4, dqA1 = zero4, dqB0 = zero4, dqB1 = zero4;
    bf16x8 qrow0, qrow1, qrow2, qrow3, orow0, orow1, orow2, orow3, qT0d0, qT0d1, oT0d0, oT0d1, qT1d0, qT1d1, oT1d0, oT1d1;
    u32x4 pf0 = {0u, 0u, 0u, 0u}, pf1 = pf0, dsf0 = pf0, dsf1 = pf0;
    {
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const f32x4 a = lds_ld<f32x4>(a_const + 32 * G);
        const f32x4 d = lds_ld<f32x4>(a_const + 256 + 32 * G);
#pragma unroll
        for (int e = 0; e < 4; ++e) { lse_t[4 * G + e] = a[e]; dlt_t[4 * G + e] = d[e]; }
      }
      qrow0 = lds_ld<bf16x8>(a_row);
      orow0 = lds_ld<bf16x8>(a_row + (OR_ - QR));
      qrow1 = lds_ld<bf16x8>(a_row ^ 32u);
      orow1 = lds_ld<bf16x8>((a_row ^ 32u) + (OR_ - QR));
      qrow2 = lds_ld<bf16x8>(a_row ^ 64u);
      orow2 = lds_ld<bf16x8>((a_row ^ 64u) + (OR_ - QR));
      qrow3 = lds_ld<bf16x8>(a_row ^ 96u);
      orow3 = lds_ld<bf16x8>((a_row ^ 96u) + (OR_ - QR));
      qT0d0 = cat4(lds_tr_ld(a_trlo), lds_tr_ld(a_trhi));
      oT0d0 = cat4(lds_tr_ld(a_trlo + (OR_ - QR)), lds_tr_ld(a_trhi + (OR_ - QR)));
      qT0d1 = cat4(lds_tr_ld(a_trlo ^ 64u), lds_tr_ld(a_trhi ^ 64u));
      oT0d1 = cat4(lds_tr_ld((a_trlo ^ 64u) + (OR_ - QR)), lds_tr_ld((a_trhi ^ 64u) + (OR_ - QR)));
      qT1d0 = qT1d1 = oT1d0 = oT1d1 = __builtin_bit_cast(bf16x8, pf0);
      saA = mfma32(qrow0, kS[0][0], lse_t);
      dpA = mfma32(orow0, vS[0][0], dlt_t);
      saA = mfma32(qrow1, kS[0][1], saA);
      dpA = mfma32(orow1, vS[0][1], dpA);
      saA = mfma32(qrow2, kS[0][2], saA);
      dpA = mfma32(orow2, vS[0][2], dpA);
      saA = mfma32(qrow3, kS[0][3], saA);
      dpA = mfma32(orow3, vS[0][3], dpA);
    }

    // Vector-memory operations of this wave in issue order (vmcnt retires in order): prologue 5 + 5 + 5 + 4; iteration t:
    // [2 stores of the write-out, sub-step 0] [tile t+3: 5] [workspace values of tile t+1: 4] [2 stores, sub-step 1] = 13.
    // Iteration t (tile t; t = ntiles is the all-padding tile that drains the pipeline: P = 0 there) needs, before its
    // mid-tile barrier, tile t+1 complete, and in its write-outs the workspace values of tile t-1: both were requested in
    // iteration t-2 or earlier, so "all but the last iteration's 13" covers them.
#ifdef BWD1W_STAMP
    unsigned long long st_[6], sw_[2];
    unsigned acc_[2][8] = {};
    unsigned last_ = 0;
    STAMP(5);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_[5]));
    last_ = (unsigned)st_[5];
#endif
    // per-iteration scalars (ring slots, write-out offsets) and the request of tile t + 3 / the workspace values of tile t + 1
#define BWD1W_TILE_SETUP                                                                                                        \
  [[maybe_unused]] const int slot = t & (NB - 1), slotn = (t + 1) & (NB - 1);                                                                    \
  [[maybe_unused]] const unsigned s_x0 = (unsigned)(slot * T::BYTES), s_x1 = s_x0 + 32 * T::ROWB, s_xn = (unsigned)(slotn * T::BYTES);           \
  [[maybe_unused]] const unsigned s_pc1 = (unsigned)(slot * 512 + 128), s_pcn = (unsigned)(slotn * 512);                                         \
  [[maybe_unused]] const unsigned s_oldr = (unsigned)(((t + 2) % NOLD) * OLD_TILE);                                                              \
  [[maybe_unused]] const unsigned s_redoff0 = t > 0 ? (unsigned)(t - 1) * WS_TILE : DROP;                                                        \
  [[maybe_unused]] const unsigned s_redoff1 = t > 0 ? (unsigned)(t - 1) * WS_TILE + WS_SUB : DROP;                                               \
  [[maybe_unused]] auto issue_tile = [&]() {                                                                                                     \
    const int tn = t + LA;                                                                                                      \
    issue(BWD1W_DMA_TILE(tn), tn < ntiles ? tn : ntiles, tn & (NB - 1));                                                        \
    oldreq(t + 1, (t + 1) % NOLD, BWD1W_OLD_BASE);                                                                              \
  };
#ifdef ABL_NO_TILE_DMA
#define BWD1W_DMA_TILE(tn) 0
#else
#define BWD1W_DMA_TILE(tn) ((tn) < ntiles ? (tn) : (tn) - ntiles)
#endif
#ifdef ABL_NO_OLDREQ
#define BWD1W_OLD_BASE DROP
#else
#define BWD1W_OLD_BASE oldbase
#endif
    // The pipeline drains behind the last real sub-step X of the block -- (ntiles - 1, 0) when rows 32 .. 63 of the last tile are
    // all >= N (N = 64 m + 1 .. 64 m + 32: the cls token makes the model's lengths 64 m + 1), else (ntiles - 1, 1): one more
    // sub-step's worth of dV^T / dK^T (X's last group, second half) and the dQ^T product of X, and two write-outs.  Those run as
    // straight-line code behind the loop, reduced to exactly that (SLIM / TINY, generated from the same bundle table), instead of
    // as padding sub-steps in full: ~1.7 sub-steps less per key block (of 43 at N = 1281, of 163 at N = 5121).
    // (the drain code works from opaque copies of the per-lane address constants: derived addresses are then computed there,
    // not hoisted above the tile loop and kept in registers across it)
    const unsigned o_imglo = a_imglo, o_imghi = a_imghi, o_old = a_old, o_wsoff = wsoff;
#define BWD1W_DRAIN_ADDRS \
  [[maybe_unused]] const unsigned a_imglo = opaque(o_imglo), a_imghi = opaque(o_imghi), a_old = opaque(o_old), wsoff = opaque(o_wsoff);
    // (here only for lengths whose last tile has no row in its second half -- every length of this model; other lengths drain through
    // two padding iterations in full: a second copy of the drain code costs this kernel register spills)
    // (as in attn_bwd1w.hip: the two write-out stores that end iteration t - 2 may stay pending at the top of iteration t)
#ifndef BWD1W_VMWAIT_EXTRA
#define BWD1W_VMWAIT_EXTRA 2
#endif
#define BWD1W_WAIT(t_) do { if ((t_) > 1) wait_vm<13 + BWD1W_VMWAIT_EXTRA>(); else wait_vm<13>(); } while (0)
    const int nloop = half_drain ? ntiles - 1 : ntiles + 1;
    for (int t = 0; t < nloop; ++t) {
#ifdef BWD1W_STAMP
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sw_[0]));
#endif
      if (t > 0) BWD1W_WAIT(t);
#ifdef BWD1W_STAMP
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sw_[1]));
      acc_[0][6] += (unsigned)sw_[1] - (unsigned)sw_[0];
#endif
      BWD1W_TILE_SETUP
#include "attn_bwd1w_body_hd64.inc"
    }
    if (half_drain) {
      {
        const int t = ntiles - 1;
        BWD1W_WAIT(t);
        BWD1W_TILE_SETUP
#define BWD1W_ONLY_SUBSTEP0
#include "attn_bwd1w_body_hd64.inc"
#undef BWD1W_ONLY_SUBSTEP0
      {
        BWD1W_DRAIN_ADDRS
#define BWD1W_DRAIN_SLIM1
#include "attn_bwd1w_drain_hd64.inc"
#undef BWD1W_DRAIN_SLIM1
      }
      }
      {
        const int t = ntiles;
        BWD1W_WAIT(t);
        BWD1W_TILE_SETUP
        BWD1W_DRAIN_ADDRS
#define BWD1W_DRAIN_TINY0
#include "attn_bwd1w_drain_hd64.inc"
#undef BWD1W_DRAIN_TINY0
      }
    }
#undef BWD1W_TILE_SETUP
#undef BWD1W_DRAIN_ADDRS
#ifdef BWD1W_STAMP
    if (kb == 1 && lane == 0 && blockIdx.x < 512)
      for (int s_ = 0; s_ < 2; ++s_)
        for (int k_ = 0; k_ < 8; ++k_) g_bwd1w64_stamp[((blockIdx.x * 4 + wid) * 2 + s_) * 8 + k_] = acc_[s_][k_];
#endif
    // ---- dK, dV of this wave's keys (the last MFMAs into them are more than a sub-step behind; the nops keep the read-out of
    // the accumulators clear of them whatever the compiler places here)
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dk00), "+a"(dk01), "+a"(dk10), "+a"(dk11), "+a"(dv00), "+a"(dv01), "+a"(dv10), "+a"(dv11));
#ifdef ABL_NO_EPI
    if (N < 0)
#endif
    {
      // through a wave-private [64 keys][128 B] LDS image (the dS image region: free behind the loop's last barrier), so that a
      // store instruction writes 8 whole 128-byte rows of dK (dV) instead of 8 bytes per lane on 32 rows.  8-byte slots XORed
      // with the low 4 row bits: the 16 lanes of a ds_write_b64 group (16 consecutive rows, one slot) cover all banks.
      const f32x16* dks[2][2] = {{&dk00, &dk01}, {&dk10, &dk11}};
      const f32x16* dvs[2][2] = {{&dv00, &dv01}, {&dv10, &dv11}};
      // (addresses derived from an opaque copy of the lane id: computed here, once per key block, instead of living in registers
      // across the tile loop)
      const int le = (int)opaque((unsigned)lane);
      const int re = le & 31, he = le >> 5;
      char* stg = smem + IMG + wid * (KW * 128);
      bf16_t* dbase = dqkv + ((size_t)b * N + key0 + wid * KW) * rs + (size_t)head * HD;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const f32x16& acc = which ? *dvs[g][d] : *dks[g][d];
            const float sc = which ? 1.0f : scale;
            const int row = 32 * g + re;
#pragma unroll
            for (int G = 0; G < 4; ++G) {
              const u32x2 w = {pack2bf(acc[4 * G] * sc, acc[4 * G + 1] * sc), pack2bf(acc[4 * G + 2] * sc, acc[4 * G + 3] * sc)};
              *reinterpret_cast<u32x2*>(stg + row * 128 + (((8 * d + 2 * G + he) ^ (row & 15)) << 3)) = w;
            }
          }
#pragma unroll
        for (int i = 0; i < KW / 8; ++i) {
          const int row = 8 * i + (le >> 3), j = le & 7, x = row & 15;
          const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((j ^ (x >> 1)) << 4));
          const u32x4 o = (x & 1) ? u32x4{v[2], v[3], v[0], v[1]} : v;
          *reinterpret_cast<u32x4*>(dbase + (size_t)row * rs + (size_t)(1 + which) * H * HD + 8 * j) = o;
        }
      }
    }
    // the next block's K staging overwrites the image region: every wave is done reading its part of it
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    // next block: its workspace read-modify-write of a row is done by the same lane as this block's (program order: a wave's
    // vector-memory operations complete in issue order); the dK / dV stores above are nobody's input
  }
  wait_vm<0>();
  // ---- the key past the last full block (N = nkb * KB + 1: the cls token) and the workspace -> bf16 conversion, for this
  // (batch, head), by the workgroup that has just written the workspace rows and streamed Q / dO (attn_bwd_tail1.hpp)
  if (tail_key >= 0) {
    __syncthreads();
    // (the thread id re-derived from the lane count: nothing of the tail's per-lane state lives in registers across the tile loop)
    const int lane_t = lane_id_fresh();
    attn_bwd_tail1_body<HD, BWD1W_TAIL_DEPTH>(qkv, dout, rowc, dq_ws, dqkv, N, NPAD, H, tail_key, 1, scale, bh, (int)gridDim.x, reinterpret_cast<float*>(smem),
                               wid * 64 + lane_t);
  }
#ifdef BWD1W_STAMP
  {
    unsigned long long k1_, r1_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(k1_), "=s"(r1_));
    if (lane == 0 && blockIdx.x < 512) {      // whole-kernel shader-clock ticks and 100 MHz ticks of this wave
      g_bwd1w64_stamp[((blockIdx.x * 4 + wid) * 2 + 0) * 8 + 7] = (unsigned)(k1_ - k0_);
      g_bwd1w64_stamp[((blockIdx.x * 4 + wid) * 2 + 1) * 8 + 7] = (unsigned)(r1_ - r0_);
    }
  }
#endif
}

}  // namespace octmae
#ifdef BWD1W_STAMP
extern "C" int octmae_debug_bwd1w64_stamps(void* host, int nbytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(octmae::bwd1w64::g_bwd1w64_stamp), (size_t)nbytes);
}
#endif
namespace octmae {

// launcher used by attn_bwd.hip's run_fused<64>
int launch_attn_bwd_fused1w64(const bf16_t* qkv, const bf16_t* dout, const float* rowc, float* dq_ws, bf16_t* dqkv, int B, int N, int NPAD,
                              int H, int nkb, float scale, int tail_key, hipStream_t st) {
  static DynLdsOnce once;
  if (int rc = once.ensure(reinterpret_cast<const void*>(attn_bwd_fused1w64_kernel), bwd1w64::LDS)) return rc;
  hipLaunchKernelGGL(attn_bwd_fused1w64_kernel, dim3(B * H), dim3(256), bwd1w64::LDS, st, qkv, dout, rowc, dq_ws, dqkv, N, NPAD, H, nkb, scale, tail_key);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace octmae
