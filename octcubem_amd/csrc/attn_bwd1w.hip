// Fused (single-pass) attention backward, head_dim 32, ONE WAVE PER SIMD: the main kernel of octmae_attn_bwd_fused at head_dim 32
// (decoder: 16 heads x 32, N = 5121).  Same algorithm, inputs, outputs and rounding points as attn_bwd.hip's
// attn_bwd_fused_kernel<32> (5 matrix products and 1 exp per score, no atomics, bit-reproducible), re-structured for the issue
// port: that kernel ran two waves per SIMD with two workgroup barriers per 64-query tile and spent ~880-1050 cycles per 32 x 32
// score block against ~400 of instruction issue; here
//   * a workgroup is 4 waves with 512 registers each (one per SIMD); a wave owns 128 keys of the 512-key block: dK^T / dV^T of
//     those keys (128 accumulator registers), their pre-scaled K and V fragments (64) and their K^T fragments (32) stay in
//     registers for the whole sweep over the queries;
//   * per 32-query sub-tile a wave does S, dP, exp2, dS, dV^T, dK^T for its four 32-key groups AND the dQ^T product over its
//     OWN 128 keys: its dS goes through a wave-private LDS image (no barrier: a wave's LDS operations execute in order) and
//     comes back through transposed reads as the B operand of 8 more 32x32x16 MFMAs;
//   * the four waves' dQ^T partial tiles (fp32, 4 KB each) are summed through LDS in a fixed order -- wave w sums register quad
//     w of all four, adds the workspace value of the previous key blocks (brought in by LDS-DMA a tile ahead) and stores -- one
//     tile LATER, behind the single workgroup barrier of the tile, so no wave ever waits for another's arithmetic;
//   * Q / dO / row-constant tiles arrive by LDS-DMA through a 3-deep ring two tiles ahead, retired by counted vmcnt.
// Reference op: backward of softmax((q k^T) scale) v, Pre-training/custom_util/video_vit.py:130-134 under autograd.
#include "attn_bwd1w.hpp"
#include "attn_bwd_tail1.hpp"
#ifndef BWD1W_TAIL_DEPTH
#define BWD1W_TAIL_DEPTH 4      // passes of rows in flight in the single-key tail (8: measured equal)
#endif
#include "../../include/octmae.h"

namespace octmae {

namespace bwd1w {

constexpr int HD = 32, NW = 4, KW = 128, NG = 4, KB = NW * KW;
constexpr int LA = 3;                                   // tiles requested ahead of the one being computed
constexpr int NB = LA + 1;                              // Q / dO / constants ring depth
constexpr int NOLD = 3;                                 // workspace-value buffers: requested one tile before their tile, read one after
using T = Tile<HD>;                                     // 64 rows x 64 B, XOR-swizzled 16-byte chunks
constexpr int QR = 0;                                   // Q ring      [NB][4096]
constexpr int OR_ = QR + NB * T::BYTES;                 // dO ring     [NB][4096]
constexpr int CR = OR_ + NB * T::BYTES;                 // constants   [NB][2][64] f32
constexpr int IMG = CR + NB * 512;                      // dS images [2 sub-steps][512 keys][32 queries] bf16 (64-byte rows); wave w writes
                                                        //   rows 128 w .. 128 w + 127, every wave reads all rows (transposed)
constexpr int IMG_BUF = KB * 64;
constexpr int OLD = IMG + 2 * IMG_BUF;                  // workspace values   [NOLD tiles][2 sub-tiles][NW][64 lanes] f32x4
constexpr int LDS = OLD + NOLD * 2 * NW * 1024;
constexpr int STG = IMG;                                // K rows of the block [512][64 B], staged once per block for the K^T reads
static_assert(LDS <= 160 * 1024, "LDS budget");
static_assert((CR % 128) == 0 && (IMG % 128) == 0 && (OLD % 128) == 0, "XOR chunk selectors act on address bits 0..6");

using namespace bwd1w_util;

// ---- diagnostic build (-DBWD1W_STAMP, make stamp): s_memtime at the start and the middle of every group-step, before the
// end-of-sub-step wait and behind the barrier; per-wave sums of the 10 intervals of each sub-step (tools/attn_bwd1w_stamps.py).
// The stamps go to a buffer of their own and no output depends on them.
#ifdef BWD1W_STAMP
__device__ unsigned g_bwd1w_stamp[512 * 4 * 2 * 10];
#define STAMP(k) asm volatile("s_memtime %0" : "=s"(st_[k]))
#define STAMP_ACCUM(s)                                                                                           \
  do {                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_[0]), "+s"(st_[1]), "+s"(st_[2]), "+s"(st_[3]), "+s"(st_[4]), \
                 "+s"(st_[5]), "+s"(st_[6]), "+s"(st_[7]), "+s"(st_[8]), "+s"(st_[9]));                          \
    _Pragma("unroll") for (int k_ = 0; k_ < 9; ++k_) acc_[s][k_] += (unsigned)st_[k_ + 1] - (unsigned)st_[k_];  \
    acc_[s][9] += (unsigned)st_[0] - last_;                                                                      \
    last_ = (unsigned)st_[9];                                                                                    \
  } while (0)
#else
#define STAMP(k)
#define STAMP_ACCUM(s)
#endif

}  // namespace bwd1w

__global__ __launch_bounds__(256, 1) void attn_bwd_fused1w_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                  const float* __restrict__ rowc, float* __restrict__ dq_ws,
                                                                  bf16_t* __restrict__ dqkv, int N, int NPAD, int H, int nkb,
                                                                  float scale, int tail_key) {
  using namespace bwd1w;
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

  // ---- LDS-DMA plan: a tile is 4 Q pieces + 4 dO pieces of 1 KiB; waves 0, 1 bring Q, waves 2, 3 dO (2 pieces each) and every
  // wave one row of constants (-lse*log2e: even waves, -delta: odd)
  const bool isq = wid < 2;                                       // wave-uniform
  const i32x4_t rsD = isq ? make_rsrc(qb, (unsigned)(((size_t)(N - 1) * rs + HD) * 2)) : make_rsrc(dob, (unsigned)(((size_t)(N - 1) * os + HD) * 2));
  unsigned dvoff[2], dlds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = (wid & 1) * 2 + i;
    const int q = piece * 64 + lane;
    const int row = q / T::CHUNKS, c = (q % T::CHUNKS) ^ T::sw(row);
    dvoff[i] = (unsigned)(((size_t)row * (isq ? rs : os) + c * 8) * 2);
    dlds[i] = (unsigned)((isq ? QR : OR_) + piece * 1024);
  }
  const unsigned dstride = (unsigned)(64 * (isq ? rs : os) * 2);
  const i32x4_t rsC = make_rsrc(rowc + (size_t)(wid & 1) * gridDim.x * NPAD + (size_t)bh * NPAD, (unsigned)((size_t)NPAD * 4));
  // Q / dO of tile `dt` and the constants of tile `ct` -> ring slot   (3 operations)
  auto issue = [&](int dt, int ct, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + dlds[i] + (unsigned)(slot * T::BYTES)));
      lds_dma16(m0v, dvoff[i] + (unsigned)dt * dstride, rsD);
    }
    const unsigned m0c = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + CR + slot * 512 + (wid & 1) * 256));
    lds_dma4(m0c, (unsigned)((ct * 64 + lane) * 4), rsC);
  };

  // ---- dQ workspace of this (batch, head): fp32 [N][32]; rows >= N fall outside the descriptor (loads 0, stores dropped).
  // Wave w owns one 16 x 16 tile of every sub-step's dQ^T[32 head dims][32 queries]: head dims 16 (w & 1) + 4 (lane >> 4) .. + 3
  // of query 16 (w >> 1) + (lane & 15) of the sub-tile (the C / D layout of the 16x16x32 MFMA).
  float* wsb = dq_ws + (size_t)bh * N * HD;
  const i32x4_t rsW = make_rsrc(wsb, (unsigned)((size_t)N * HD * 4));
  const __amdgpu_buffer_rsrc_t rsWs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)(unsigned)rsW[1] << 32) | (unsigned)rsW[0]), 0, rsW[2], 0x00020000);
  const int dt = wid & 1, qt = wid >> 1, g16 = lane >> 4, c16 = lane & 15;
  const unsigned wsoff = (unsigned)(((16 * qt + c16) * HD + 16 * dt + 4 * g16) * 4);
  constexpr unsigned WS_SUB = 32 * HD * 4, WS_TILE = 64 * HD * 4, DROP = 0x80000000u;   // DROP: an offset outside every descriptor
  // workspace values of this wave's quads of `tile` -> OLD buffer   (2 operations; `base` = DROP in the first key block: zeros)
  auto oldreq = [&](int tile, int buf, unsigned base) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned m0o = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + OLD + ((buf * 2 + u) * NW + wid) * 1024));
      lds_dma16_ws(m0o, wsoff + base + (unsigned)tile * WS_TILE + (unsigned)u * WS_SUB, rsW);
    }
  };

  // ---- per-lane LDS address parts, fixed for the whole kernel (opaque: not re-derived from the lane id inside the loops)
  const int tq_ = (lane >> 2) & 3, tp_ = lane & 3, tgi = (lane >> 4) & 1;
  const unsigned a_const = opaque(lds0 + (unsigned)(CR + 16 * h));
  const unsigned a_row = opaque(lds0 + (unsigned)(QR + T::off(r, h)));
  const unsigned a_trlo = opaque(lds0 + (unsigned)(QR + T::off(4 * h + tq_, 2 * tgi + (tp_ >> 1)) + (tp_ & 1) * 8));
  const unsigned a_trhi = opaque(lds0 + (unsigned)(QR + T::off(4 * h + tq_ + 8, 2 * tgi + (tp_ >> 1)) + (tp_ & 1) * 8));
  // dS image: this lane's key row 128 wid + r (+ 32 g), chunk h (^ 32 s + 16 k for the query chunk 4 s + 2 k + h) (+ image buffer)
  const unsigned a_imgw = opaque(lds0 + (unsigned)(IMG + img_off(wid * KW + r, h)));
  // transposed reads of 4-key x 16-query blocks for the 16x16x32 B operand dS^T[k = key 32 ks + 8 g16 + e][col q = 16 qt + c16]:
  // keys 8 g16 + tq_ (+ 4) (+ 32 ks), query chunk 4 qt + tp_ (+ image buffer)
  const unsigned a_imglo = opaque(lds0 + (unsigned)(IMG + img_off(8 * g16 + tq_, 4 * qt + tp_)));
  const unsigned a_imghi = opaque(lds0 + (unsigned)(IMG + img_off(8 * g16 + tq_ + 4, 4 * qt + tp_)));
  const unsigned a_old = opaque(lds0 + (unsigned)(OLD + wid * 1024 + lane * 16));
  f32x16 zero16;
#pragma unroll
  for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

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
      for (int i = 0; i < KB * 4 / 256; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 2, cc = c & 3;
        const u32x4 v = *reinterpret_cast<const u32x4*>(kb_ + (size_t)(key0 + row) * rs + 8 * cc);
        *reinterpret_cast<u32x2*>(smem + STG + img_off(row, 2 * cc)) = u32x2{v[0], v[1]};
        *reinterpret_cast<u32x2*>(smem + STG + img_off(row, 2 * cc + 1)) = u32x2{v[2], v[3]};
      }
    }
    // ---- this wave's keys: B operands of S (pre-scaled) and dP
    bf16x8 kS[NG][2], vS[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const size_t krow = (size_t)(key0 + wid * KW + 32 * g + r) * rs;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x4 kv = *reinterpret_cast<const u32x4*>(kb_ + krow + 16 * s + 8 * h);
        const u32x4 vv = *reinterpret_cast<const u32x4*>(vb_ + krow + 16 * s + 8 * h);
        kS[g][s] = scale_frag(kv, sc2);
        vS[g][s] = __builtin_bit_cast(bf16x8, vv);
      }
    }
    f32x16 dk0 = zero16, dk1 = zero16, dk2 = zero16, dk3 = zero16, dv0 = zero16, dv1 = zero16, dv2 = zero16, dv3 = zero16;
    __syncthreads();                              // K staging visible
    // A operand of dQ^T = K^T dS^T (16x16x32): K^T[row d = 16 dt + c16][k = key 32 ks + 8 g16 + e], all 512 keys, loop invariant
    bf16x8 kT[KB / 32];
#pragma unroll
    for (int ks = 0; ks < KB / 32; ++ks)
      kT[ks] = cat4(lds_tr_read(smem + STG + img_off(8 * g16 + tq_, 4 * dt + tp_) + ks * 2048),
                    lds_tr_read(smem + STG + img_off(8 * g16 + tq_ + 4, 4 * dt + tp_) + ks * 2048));
    __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0)
    __syncthreads();                              // every wave has its K^T fragments: the image region is free

    // (every load the compiler tracks has returned: without this it carries "loads pending" into the tile loop and waits there
    // with a vmcnt that also drains the hand-counted LDS-DMA ring; the ring prologue, requested before the K staging, is
    // complete with them)
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    // ---- pipeline prologue: fragments of sub-step (0, 0), S / dP of its first key group; everything the first C1 / D / reduce
    // of the loop consume without a producer is zero or is dropped (the dQ^T tile of "sub-step -1": a store outside the range)
    f32x16 lse_a, dlt_a, lse_b = zero16, dlt_b = zero16, saA, dpA, saB, dpB;
    f32x4 dqA = zero4, dqB = zero4;
    bf16x8 qrowa0, qrowa1, orowa0, orowa1, qrowb0, qrowb1, orowb0, orowb1, qTa0, qTa1, oTa0, oTa1, qTb0, qTb1, oTb0, oTb1;
    u32x4 pf0 = {0u, 0u, 0u, 0u}, pf1 = pf0, dsf0 = pf0, dsf1 = pf0;
    {
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const f32x4 a = lds_ld<f32x4>(a_const + 32 * G);
        const f32x4 d = lds_ld<f32x4>(a_const + 256 + 32 * G);
#pragma unroll
        for (int e = 0; e < 4; ++e) { lse_a[4 * G + e] = a[e]; dlt_a[4 * G + e] = d[e]; }
      }
      qrowa0 = lds_ld<bf16x8>(a_row);
      orowa0 = lds_ld<bf16x8>(a_row + (OR_ - QR));
      qrowa1 = lds_ld<bf16x8>(a_row ^ 32u);
      orowa1 = lds_ld<bf16x8>((a_row ^ 32u) + (OR_ - QR));
      qTa0 = cat4(lds_tr_ld(a_trlo), lds_tr_ld(a_trhi));
      oTa0 = cat4(lds_tr_ld(a_trlo + (OR_ - QR)), lds_tr_ld(a_trhi + (OR_ - QR)));
      qTa1 = cat4(lds_tr_ld(a_trlo + 16 * 64), lds_tr_ld(a_trhi + 16 * 64));
      oTa1 = cat4(lds_tr_ld(a_trlo + 16 * 64 + (OR_ - QR)), lds_tr_ld(a_trhi + 16 * 64 + (OR_ - QR)));
      qTb0 = qTb1 = oTb0 = oTb1 = qrowb0 = qrowb1 = orowb0 = orowb1 = __builtin_bit_cast(bf16x8, pf0);
      saA = mfma32(qrowa0, kS[0][0], lse_a);
      dpA = mfma32(orowa0, vS[0][0], dlt_a);
      saA = mfma32(qrowa1, kS[0][1], saA);
      dpA = mfma32(orowa1, vS[0][1], dpA);
      saB = zero16; dpB = zero16;
    }

    // Vector-memory operations of this wave in issue order (vmcnt retires in order): prologue 3 + 3 + 3 + 2; iteration t:
    // [store of reduce, sub-step 0] [tile t+3: 3] [workspace values of tile t+1: 2] [store of reduce, sub-step 1] = 7.
    // Iteration t (tile t; t = ntiles is the all-padding tile that drains the pipeline: P = 0 there) needs, before its
    // mid-tile barrier, tile t+1 complete, and in its reduces the workspace values of tile t-1: both were requested in
    // iteration t-2 or earlier, so "all but the last iteration's 7" covers them.
#ifdef BWD1W_STAMP
    unsigned long long st_[10];
    unsigned acc_[2][10] = {};
    unsigned last_ = 0;
    STAMP(9);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_[9]));
    last_ = (unsigned)st_[9];
#endif
    // per-iteration scalars (ring slots, write-out offsets) and the request of tile t + 3 / the workspace values of tile t + 1
#define BWD1W_TILE_SETUP                                                                                            \
  [[maybe_unused]] const int slot = t & (NB - 1), slotn = (t + 1) & (NB - 1);                                                        \
  [[maybe_unused]] const unsigned s_pc1 = (unsigned)(slot * 512 + 128), s_x1 = (unsigned)(slot * T::BYTES + 32 * T::ROWB);           \
  [[maybe_unused]] const unsigned s_pcn = (unsigned)(slotn * 512), s_xn = (unsigned)(slotn * T::BYTES);                              \
  [[maybe_unused]] const unsigned s_oldr = (unsigned)(((t + 2) % NOLD) * 2 * NW * 1024);                                             \
  [[maybe_unused]] const unsigned s_redoff0 = t > 0 ? (unsigned)(t - 1) * WS_TILE : DROP;                                            \
  [[maybe_unused]] const unsigned s_redoff1 = t > 0 ? (unsigned)(t - 1) * WS_TILE + WS_SUB : DROP;                                   \
  [[maybe_unused]] auto issue_tile = [&]() {                                                                                         \
    const int tn = t + LA;                                                                                          \
    issue(tn < ntiles ? tn : tn - ntiles, tn < ntiles ? tn : ntiles, tn & (NB - 1));                                \
    oldreq(t + 1, (t + 1) % NOLD, oldbase);                                                                         \
  };
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
    // The wait at the top of iteration t retires the vector-memory operations of iteration t - 2 and earlier -- except that
    // iteration's LAST operation, the write-out store of its sub-step 1: nothing issued behind it is needed yet, and vmcnt
    // retires in order, so letting that store (HBM write latency under load) stay pending costs no correctness and gives every
    // store one more iteration to complete (BWD1W_VMWAIT_EXTRA = 1).  t = 1 keeps the plain count: the prologue's last request,
    // the workspace values of tile 0, is read in that iteration.
#ifndef BWD1W_VMWAIT_EXTRA
#define BWD1W_VMWAIT_EXTRA 1
#endif
#define BWD1W_WAIT(t_) do { if ((t_) > 1) wait_vm<7 + BWD1W_VMWAIT_EXTRA>(); else wait_vm<7>(); } while (0)
    const int nloop = half_drain ? ntiles - 1 : ntiles;
    for (int t = 0; t < nloop; ++t) {
      if (t > 0) BWD1W_WAIT(t);
      BWD1W_TILE_SETUP
#include "attn_bwd1w_body.inc"
    }
    if (half_drain) {
      {
        const int t = ntiles - 1;
        BWD1W_WAIT(t);
        BWD1W_TILE_SETUP
#define BWD1W_ONLY_SUBSTEP0
#include "attn_bwd1w_body.inc"
#undef BWD1W_ONLY_SUBSTEP0
      {
        BWD1W_DRAIN_ADDRS
#define BWD1W_DRAIN_SLIM1
#include "attn_bwd1w_drain.inc"
#undef BWD1W_DRAIN_SLIM1
      }
      }
      {
        const int t = ntiles;
        BWD1W_WAIT(t);
        BWD1W_TILE_SETUP
        BWD1W_DRAIN_ADDRS
#define BWD1W_DRAIN_TINY0
#include "attn_bwd1w_drain.inc"
#undef BWD1W_DRAIN_TINY0
      }
    } else {
      const int t = ntiles;
      BWD1W_WAIT(t);
      BWD1W_TILE_SETUP
      BWD1W_DRAIN_ADDRS
#define BWD1W_DRAIN_SLIM0
#include "attn_bwd1w_drain.inc"
#undef BWD1W_DRAIN_SLIM0
#define BWD1W_DRAIN_TINY1
#include "attn_bwd1w_drain.inc"
#undef BWD1W_DRAIN_TINY1
    }
#undef BWD1W_TILE_SETUP
#undef BWD1W_DRAIN_ADDRS
#ifdef BWD1W_STAMP
    if (kb == 1 && lane == 0 && blockIdx.x < 512)
      for (int s_ = 0; s_ < 2; ++s_)
        for (int k_ = 0; k_ < 10; ++k_) g_bwd1w_stamp[((blockIdx.x * 4 + wid) * 2 + s_) * 10 + k_] = acc_[s_][k_];
#endif
    // ---- dK, dV of this wave's keys (the last MFMAs into them are more than a sub-step behind; the nops keep the read-out of
    // the accumulators clear of them whatever the compiler places here)
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dk0), "+a"(dk1), "+a"(dk2), "+a"(dk3), "+a"(dv0), "+a"(dv1), "+a"(dv2), "+a"(dv3));
    {
      const f32x16* dks[4] = {&dk0, &dk1, &dk2, &dk3};
      const f32x16* dvs[4] = {&dv0, &dv1, &dv2, &dv3};
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        bf16_t* drow = dqkv + ((size_t)b * N + key0 + wid * KW + 32 * g + r) * rs + (size_t)head * HD;
        const f32x16& dkg = *dks[g];
        const f32x16& dvg = *dvs[g];
#pragma unroll
        for (int G = 0; G < 4; ++G) {
          const u32x2 wk = {pack2bf(dkg[4 * G] * scale, dkg[4 * G + 1] * scale), pack2bf(dkg[4 * G + 2] * scale, dkg[4 * G + 3] * scale)};
          const u32x2 wv = {pack2bf(dvg[4 * G], dvg[4 * G + 1]), pack2bf(dvg[4 * G + 2], dvg[4 * G + 3])};
          *reinterpret_cast<u32x2*>(drow + (size_t)H * HD + 8 * G + 4 * h) = wk;
          *reinterpret_cast<u32x2*>(drow + (size_t)2 * H * HD + 8 * G + 4 * h) = wv;
        }
      }
    }
    // next block: its workspace read-modify-write of a row is done by the same lane as this block's (program order: a wave's
    // vector-memory operations complete in issue order), the dK / dV stores above are nobody's input; its K staging overwrites
    // the image region, which every wave has finished reading behind the loop's last barrier
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
}

}  // namespace octmae

using namespace octmae;

#ifdef BWD1W_STAMP
extern "C" int octmae_debug_bwd1w_stamps(void* host, int nbytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(octmae::bwd1w::g_bwd1w_stamp), (size_t)nbytes);
}
#endif

// launcher used by attn_bwd.hip's run_fused<32>
namespace octmae {
int launch_attn_bwd_fused1w(const bf16_t* qkv, const bf16_t* dout, const float* rowc, float* dq_ws, bf16_t* dqkv, int B, int N, int NPAD,
                            int H, int nkb, float scale, int tail_key, hipStream_t st) {
  static DynLdsOnce once;
  if (int rc = once.ensure(reinterpret_cast<const void*>(attn_bwd_fused1w_kernel), bwd1w::LDS)) return rc;
  hipLaunchKernelGGL(attn_bwd_fused1w_kernel, dim3(B * H), dim3(256), bwd1w::LDS, st, qkv, dout, rowc, dq_ws, dqkv, N, NPAD, H, nkb, scale, tail_key);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}
}  // namespace octmae
