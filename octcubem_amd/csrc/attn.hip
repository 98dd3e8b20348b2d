// Flash-style non-causal attention forward / backward for gfx950, head_dim 64 (encoder) and 32 (decoder),
// arbitrary sequence length (1281 and 5121 are both 1 mod 64).  Replaces the reference's
// softmax((q k^T) * hd^-0.5) v  (Pre-training/custom_util/video_vit.py:130-134; flash path: flash-attn MHA
// built at models_mae_joint_res_flash_attn.py:131-149) without materialising the (B,H,N,N) scores.
//
// Layout: qkv bf16 [B][N][3][H][HD] (the fused Wqkv GEMM output), o bf16 [B][N][H][HD], lse fp32 [B][H][N].
//
// MFMA orientation (v_mfma_f32_32x32x16_bf16; D register g = row (g&3)+8(g>>2)+4h, lane&31 = column):
//   S^T[key][query] = K . Q^T        keys on the registers, the QUERY ON THE LANE: the softmax row
//                                    reduction is over a lane's own registers plus one exchange with
//                                    lane^32 -- no LDS, no 32-lane shuffles.
//   O^T[d][query]  += V^T . P^T      P^T is the previous accumulator re-used as the B operand with no lane
//                                    movement (registers 8s..8s+7 -> k-step s); V^T comes from the row-major V
//                                    tile in LDS through ds_read_b64_tr_b16.
// Backward is two deterministic kernels (no atomics): dQ walks key tiles with the same orientation; dK/dV
// keeps the KEY on the lane and walks query tiles, so dK^T/dV^T accumulate in registers.
// K/V (or Q/dO) tiles are 64 rows and arrive by LDS-DMA in a 3-4 deep ring (TileDma below), one barrier per tile behind a
// counted s_waitcnt; the XOR swizzle (applied to the DMA's SOURCE address) makes both the ds_read_b128 row reads and the
// transposed reads bank-conflict-free.  Workgroups are mapped XCD-aware (attn_block_coord).
#include <type_traits>

#include "attn_tile.hpp"
#include "../../include/octmae.h"

namespace octmae {

#ifndef ATT_PIPELINE
// S_{t+1} issued under tile t's exps (two score tiles live: +32 VGPRs).  head_dim 64 had it off to stay under the registers of a
// 4th wave per SIMD -- which its 48 KB of LDS per workgroup never allowed anyway: on, +0.7 % at 1281 tokens, +3.2 % at 5121.
#define ATT_PIPELINE(HD) 1
#endif
#ifndef ATT_OCC_FWD32
#define ATT_OCC_FWD32 3
#endif
#ifndef ATT_OCC_FWD64
#define ATT_OCC_FWD64 3
#endif
#ifndef ATT_OCC_DQ32
#define ATT_OCC_DQ32 3
#endif
#ifndef FWD_RING
#define FWD_RING(HD) ((HD) == 32 ? 4 : 3)      // K/V ring depth of the forward kernel (LDS: 2 * depth * tile bytes)
#endif
#ifndef DKV_RING
#define DKV_RING(HD) ((HD) == 32 ? 4 : 3)
#endif
#ifndef DQ_RING
#define DQ_RING(HD) ((HD) == 32 ? 4 : 3)
#endif
#ifndef ATT_OCC_DKV32
#define ATT_OCC_DKV32 3
#endif
// =====================================================================================================
// forward
// =====================================================================================================
// FAST = optimistic variant: no running max at all (reference point 0 in the scaled log2 domain).  Floating point is
// scale-free, so exp2(s) / sum exp2(s) is exactly as accurate as the max-subtracted form as long as nothing overflows
// or underflows -- in the row sums OR in the un-normalised accumulators of O -- and the logits of a healthy pre-norm ViT are
// nowhere near that (a diverging one's are: see the give-up test at the end of the kernel).  The kernel raises `*flag` unless every
// row sum lies in [2^-100, 2^100] and every O accumulator is finite; the host ALWAYS enqueues the safe
// (online-max) kernel right after, which returns immediately unless the flag is set and otherwise recomputes everything.
// No host synchronisation, and the result is always the safe one when it matters.  What the fast variant saves is VALU
// work, the limiter of this kernel: no max chain, no rescale, and the score accumulators start from the inline constant 0
// (no per-tile broadcast of -max): per score only v_exp_f32 + v_cvt_pk remain.
template <int HD, bool FAST>
__global__ __launch_bounds__(256, (HD == 32 ? ATT_OCC_FWD32 : ATT_OCC_FWD64)) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                          float* __restrict__ lse, int N, int H, float scale, int* __restrict__ flag) {
  if (!FAST && flag != nullptr && *flag == 0) return;   // safe kernel: only runs when the optimistic one gave up
  constexpr int KS = HD / 16;   // k-steps over the head dimension
  constexpr int DB = HD / 32;   // 32-wide blocks of the head dimension
  using T = Tile<HD>;
  extern __shared__ __attribute__((aligned(128))) char smem[];   // 128: XOR chunk selectors act on address bits 0..6
  // LDS: K ring | V ring

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const BlockCoord bc = attn_block_coord();
  const int head = bc.head, b = bc.b;
  const size_t rs = (size_t)3 * H * HD;
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)head * HD;
  const bf16_t* kb_ = qb + (size_t)H * HD;
  const bf16_t* vb_ = qb + (size_t)2 * H * HD;
  const int q0 = bc.x * 128 + wid * 32;
  const int qrow = q0 + r;

  // Row sums on the VALU (4 add chains per lane).  An all-ones MFMA operand can produce them instead (kept below, off): that
  // was the better trade while the softmax was the clear limiter, but with the optimistic forward and pre-scaled Q the kernel
  // time is ~ MFMA + VALU cycles, and 4 extra MFMAs per 64-key step (128 cycles) cost more than 32 v_add (64 cycles):
  // 2 306 -> 2 219 us on the decoder shape (32 x 16 x 5121 x 32), same box.
  constexpr bool ONES_SUM = false;
  const float sc2 = scale * LOG2E;
  bf16x8 qf[KS];   // Q^T fragments, pre-scaled by scale*log2(e): S accumulates directly in the exp2 domain
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (qrow < N) v = *reinterpret_cast<const u32x4*>(qb + (size_t)qrow * rs + 16 * s + 8 * h);
    qf[s] = scale_frag(v, sc2);
  }
  const u32x4 ones_w = {OCTMAE_LP_ONE_PAIR, OCTMAE_LP_ONE_PAIR, OCTMAE_LP_ONE_PAIR, OCTMAE_LP_ONE_PAIR};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_w);

  f32x16 oacc[DB], lacc;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 16; ++g) oacc[d][g] = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) lacc[g] = 0.f;
  float m_s = 0.f, l_run = 0.f;   // running max (scaled log2 domain) and, for head_dim 64, the per-lane partial row sum

  // Software pipeline across key tiles: S_{t+1} = K_{t+1} Q^T is issued (MFMA pipe) right after the rescale decision of
  // tile t and runs under tile t's exp / convert (VALU); P_t V_t follows.  K is therefore staged one tile ahead of V:
  //   iteration t reads Kbuf[(t+1)&1] (K_{t+1}) and Vbuf[t&1] (V_t); at its end V_{t+1} -> Vbuf[(t+1)&1], K_{t+2} -> Kbuf[t&1].
  const int ntiles = (N + 63) / 64;
  // K/V rings (LDS-DMA, see TileDma): step t issues K_{t+NB} and V_{t+NB-1} into the slots K_t / V_{t-1} left in step t-1
  // and, before its barrier, waits for everything but the loads of the NB-2 youngest steps: K_{t+2} and V_{t+1}, which
  // step t+1 reads, have then landed.
  constexpr int NB = FWD_RING(HD);
  char* const Kbuf = smem;
  char* const Vbuf = smem + NB * T::BYTES;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  TileDma<HD, 4> dk, dv;
  dk.init(kb_, rs, N, wid, lane);
  dv.init(vb_, rs, N, wid, lane);
  constexpr int DMA_CNT = 2 * TileDma<HD, 4>::PER_WAVE * (NB - 2);
  auto issue = [&](int tk, int tv) {      // K tile tk, V tile tv (a negative tv: a dummy load that keeps the count uniform)
    dk.load(tk, lds0 + (unsigned)((tk % NB) * T::BYTES));
    dv.load(tv < 0 ? ntiles + NB : tv, lds0 + (unsigned)((NB + (tv < 0 ? NB - 1 : tv % NB)) * T::BYTES));
  };
#pragma unroll
  for (int s = -NB; s < 0; ++s) issue(s + NB, s + NB - 1);
  dma_wait_barrier<DMA_CNT>();             // K_0, K_1, V_0 landed and visible

  // Hand-placed LDS addressing (attn_tile.hpp): the per-lane parts of T::row_frag / T::tr_frag are fixed for the kernel, the
  // ring slot goes in with one add per family, k-step / head-dim block are XOR masks below 128 or immediate offsets
  // (T::sw() reads row bits 1..3 only: +32 and +16 rows leave the swizzle alone).  Left to the compiler the hd-64 loop spent
  // 45 of its 205 instructions on these addresses.
  static_assert(T::BYTES % 128 == 0, "ring slots must leave the low 7 address bits alone");
  const int tq_ = (lane >> 2) & 3, tp_ = lane & 3, tgi = (lane >> 4) & 1;
  constexpr bool HAND = (HD == 64);     // head_dim 32 runs at the 168-register limit of 3 waves per SIMD: three more live values spill in its loop
  const unsigned a_row = !HAND ? 0u : opaque(lds0 + (unsigned)T::off(r, h));                                          // K row fragment, k-step 0
  const unsigned a_trlo = !HAND ? 0u : opaque(lds0 + (unsigned)(NB * T::BYTES + T::off(4 * h + tq_, 2 * tgi + (tp_ >> 1)) + (tp_ & 1) * 8));   // V^T
  const unsigned a_trhi = !HAND ? 0u : opaque(lds0 + (unsigned)(NB * T::BYTES + T::off(4 * h + tq_ + 8, 2 * tgi + (tp_ >> 1)) + (tp_ & 1) * 8));
  auto qk = [&](int tk, f32x16 (&sa)[2]) {                      // S^T of key tile tk
    // row constant -m_s as the initial accumulator (a persistent 16-register tile of it would save the broadcast but
    // pushes the kernel past the 168-VGPR budget of 3 waves per SIMD: measured 2.4x slower from spills)
    const float neg_m = FAST ? 0.f : -m_s;                      // FAST: literal 0 -> the MFMA's inline-constant C operand
    const unsigned pk = a_row + (unsigned)((tk % NB) * T::BYTES);
    const char* cK = Kbuf + (tk % NB) * T::BYTES;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int g = 0; g < 16; ++g) sa[kb][g] = neg_m;
#pragma unroll
      for (int s = 0; s < KS; ++s)
        sa[kb] = mfma32(HAND ? lds_ld<bf16x8>((pk ^ (unsigned)(32 * s)) + kb * 32 * T::ROWB) : T::row_frag(cK, kb * 32, s, lane), qf[s], sa[kb]);
    }
  };

  // tile t: scur = S_t (already computed), snext <- S_{t+1}
  auto step = [&](int t, f32x16 (&scur)[2], f32x16 (&snext)[2], auto last_tag) {
    constexpr bool last = decltype(last_tag)::value;             // compile-time: the masking code must not leak into the loop
    const unsigned pvl = a_trlo + (unsigned)((t % NB) * T::BYTES), pvh = a_trhi + (unsigned)((t % NB) * T::BYTES);
    const char* cV = Vbuf + (t % NB) * T::BYTES;
    issue(t + NB, t + NB - 1);
    if (last) {                                                  // only the last tile can hold keys >= N
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int key = t * 64 + kb * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
          if (key >= N) scur[kb][g] = -INFINITY;
        }
    }
    if (!FAST) {
      // excess of this tile's scores over the running max (two chains for ILP)
      float e0 = scur[0][0], e1 = scur[1][0];
#pragma unroll
      for (int g = 1; g < 16; ++g) { e0 = fmaxf(e0, scur[0][g]); e1 = fmaxf(e1, scur[1][g]); }
      float ex = fmaxf(e0, e1);
      ex = fmaxf(ex, __shfl_xor(ex, 32, 64));
      // Deferred rescale (wave-uniform, rare after the first tiles): decided BEFORE this tile's P is formed, so everything
      // accumulated so far is at the old max and is scaled exactly once; the pending scores are shifted by the same amount.
      if (t == 0 || !__all(ex <= RESCALE_SLACK)) {
        const float d = (t == 0) ? ex : fmaxf(ex, 0.f);
        const float alpha = (t == 0) ? 1.f : fast_exp2(-d);   // t == 0: nothing accumulated yet (and d may be hugely negative)
        m_s += d;
        l_run *= alpha;
        lacc[0] *= alpha;
#pragma unroll
        for (int dd = 0; dd < DB; ++dd)
#pragma unroll
          for (int g = 0; g < 16; ++g) oacc[dd][g] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int g = 0; g < 16; ++g) scur[kb][g] -= d;
      }
    }
    const bool alias = (&scur[0] == &snext[0]);
    if (!last && !alias) qk(t + 1, snext);     // MFMA pipe works on S_{t+1} under the exps below
    // (four add chains per lane; in the optimistic kernel as two v_pk_add_f32 chains: the kernel is bound by VALU cycles -- PMC: the
    // VALU is busy 82 % of the kernel, the matrix pipe 42 % -- and a packed add costs the cycles of one scalar add; same sums)
    f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int g = 0; g < 16; g += 4) {
        const float p0 = fast_exp2(scur[kb][g]), p1 = fast_exp2(scur[kb][g + 1]);
        const float p2 = fast_exp2(scur[kb][g + 2]), p3 = fast_exp2(scur[kb][g + 3]);
        scur[kb][g] = p0; scur[kb][g + 1] = p1; scur[kb][g + 2] = p2; scur[kb][g + 3] = p3;
        if (!ONES_SUM && FAST) { s01 += f32x2{p0, p1}; s23 += f32x2{p2, p3}; }
        if (!ONES_SUM && !FAST) { s0 += p0; s1 += p1; s2 += p2; s3 += p3; }
      }
    if (!ONES_SUM) l_run += FAST ? (s01[0] + s01[1]) + (s23[0] + s23[1]) : (s0 + s1) + (s2 + s3);
    // O^T += V^T P^T   (and, head_dim 32, row sums += 1^T P^T on the MFMA pipe)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = acc_to_frag(scur[kb], s);
#pragma unroll
        for (int d = 0; d < DB; ++d) {           // T::tr_frag(V tile, 32 kb, s, 32 d)
          const unsigned ro = (unsigned)((kb * 32 + s * 16) * T::ROWB);
          oacc[d] = mfma32(HAND ? cat4(lds_tr_ld((pvl ^ (unsigned)(64 * d)) + ro), lds_tr_ld((pvh ^ (unsigned)(64 * d)) + ro))
                                : T::tr_frag(cV, kb * 32, s, d * 32, lane), pf, oacc[d]);
        }
        if (ONES_SUM) lacc = mfma32(ones, pf, lacc);
      }
    if (!last && alias) qk(t + 1, snext);
    dma_wait_barrier<DMA_CNT>();
  };

  // A wave whose 32 queries all lie past N (the last workgroup of a sequence of 128 k + 1 tokens keeps one live wave of four)
  // only stages tiles and keeps the barriers, in a loop of its own: the SIMD time it would have spent on masked work goes to
  // the other workgroups resident on the CU.
  if (__builtin_amdgcn_readfirstlane(q0) >= N) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < ntiles; ++t) {
      issue(t + NB, t + NB - 1);
      dma_wait_barrier<DMA_CNT>();
    }
    return;
  }
  f32x16 sA[2], sB[2];
  qk(0, sA);
  // S_0 is the only score tile computed outside a step: step 0 restages slot 0 of the K ring (K_NB) right away, so every
  // wave's reads of K_0 must have returned first (found by tools/stress_attn_race.py: without this barrier 0.3 % of the
  // decoder-shape launches differed in a few rows)
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  int t = 0;
  if (ATT_PIPELINE(HD)) {
    for (; t + 2 < ntiles; t += 2) {
      step(t, sA, sB, std::false_type{});
      step(t + 1, sB, sA, std::false_type{});
    }
    if (t + 1 < ntiles) {
      step(t, sA, sB, std::false_type{});
      step(t + 1, sB, sA, std::true_type{});
    } else {
      step(t, sA, sB, std::true_type{});
    }
  } else {
    // same data flow without keeping two score tiles live (32 fewer VGPRs -> one more wave per SIMD): S_{t+1} is still
    // issued before tile t's exps, but into the SAME registers after P_t has been packed -- i.e. after the PV MFMAs
    for (; t + 1 < ntiles; ++t) step(t, sA, sA, std::false_type{});
    step(t, sA, sA, std::true_type{});
  }

  // every row of the ones-MFMA accumulator holds the full row sum (both lane halves); the VALU form is per half
  const float l_tot = ONES_SUM ? lacc[0] : (l_run + __shfl_xor(l_run, 32, 64));
  const float inv = 1.0f / l_tot;
  if (FAST) {
    // Give up (the safe kernel, enqueued behind this one, then recomputes everything) unless every row sum lies in [2^-100, 2^100] AND
    // every accumulator of O is finite.  A finite row sum is not enough: O accumulates p * v before the normalisation, so a row whose
    // largest score is ~ 2^126 (a sum just under FLT_MAX) overflows in O for |v| > 4 -- met in practice: a ViT-L decoder driven at 85 x
    // its recipe's learning rate reached logits of 87 (natural units) after 790 steps, the row sums passed the old "finite and positive"
    // test and O came out +inf (tools/train_sanity.py; tests: "o_overflow").  2^100 leaves 2^27 for |v| and for the number of keys;
    // the lower bound keeps the terms within 2^-26 of the largest one normal numbers (bf16 P has fp32's exponent range).
    bool bad = !(l_tot > 7.9e-31f && l_tot < 1.2e30f);      // also catches NaN
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 16; ++g) bad |= !(__builtin_fabsf(oacc[d][g]) < 3.0e38f);
    bad = bad && (qrow < N);
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
  }
  if (qrow < N) {
    bf16_t* orow = o + ((size_t)b * N + qrow) * (size_t)(H * HD) + (size_t)head * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w = {pack2bf(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv),
                   pack2bf(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv)};
        *reinterpret_cast<u32x2*>(orow + d * 32 + 8 * g + 4 * h) = w;
      }
    if (h == 0) lse[((size_t)b * H + head) * N + qrow] = (m_s + __builtin_amdgcn_logf(l_tot)) * LN2;
  }
}

// =====================================================================================================
// backward pre-pass: delta[b][h][q] = sum_d dO[q][h][d] * O[q][h][d]
// =====================================================================================================
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         const float* __restrict__ lse, float* __restrict__ rowc, int BN, int N,
                                                         int H) {
  // rowc: [2][B*H*N] -- the two per-query constants the backward kernels start their accumulators from:
  //   rowc[0] = -lse * log2(e)   (S accumulates in the exp2 domain)       rowc[1] = -delta = -rowsum(dO * O)
  const size_t plane = (size_t)(BN / N) * H * N;
  constexpr int LPH = HD / 8;  // lanes per head
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  const int D = H * HD, nchunk = D / 8;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < BN; row += nwaves) {
    const int b = row / N, q = row % N;
    for (int c0 = 0; c0 < nchunk; c0 += 64) {
      const int c = c0 + lane;
      float s = 0.f;
      if (c < nchunk) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(o + (size_t)row * D + 8 * c);
        const u32x4 d = *reinterpret_cast<const u32x4*>(dout + (size_t)row * D + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += bflo(a[e]) * bflo(d[e]) + bfhi(a[e]) * bfhi(d[e]);
      }
#pragma unroll
      for (int m = 1; m < LPH; m <<= 1) s += __shfl_xor(s, m, 64);
      if (c < nchunk && (lane % LPH) == 0) {
        const size_t i = ((size_t)b * H + c / LPH) * N + q;
        rowc[i] = -lse[i] * LOG2E;
        rowc[plane + i] = -s;
      }
    }
  }
}

// =====================================================================================================
// backward dQ: same orientation as forward, queries on the lane, walks key tiles
//   S^T = K Q^T ; P^T = exp(S^T*scale - lse) ; dP^T = V dO^T ; dS^T = P^T (dP^T - delta) * scale
//   dQ^T[d][query] += K^T[d][key] dS^T[key][query]
// =====================================================================================================
template <int HD>
__global__ __launch_bounds__(256, (HD == 32 ? ATT_OCC_DQ32 : 2)) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                             float* __restrict__ rowc, bf16_t* __restrict__ dqkv, int N,
                                                             int H, float scale, const bf16_t* __restrict__ o_in,
                                                             const float* __restrict__ lse_in) {
  constexpr int KS = HD / 16, DB = HD / 32;
  using T = Tile<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: K0 K1 V0 V1

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const BlockCoord bc = attn_block_coord();
  const int head = bc.head, b = bc.b;
  const size_t rs = (size_t)3 * H * HD;
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)head * HD;
  const bf16_t* kb_ = qb + (size_t)H * HD;
  const bf16_t* vb_ = qb + (size_t)2 * H * HD;
  const int qrow = bc.x * 128 + wid * 32 + r;
  const size_t ostride = (size_t)H * HD;
  const float sc2 = scale * LOG2E;

  bf16x8 qf[KS], dof[KS];
  float dpart = 0.f;                  // fused row constants: this lane's half of rowsum(dO * O)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    u32x4 v = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
    if (qrow < N) {
      v = *reinterpret_cast<const u32x4*>(qb + (size_t)qrow * rs + 16 * s + 8 * h);
      w = *reinterpret_cast<const u32x4*>(dout + ((size_t)b * N + qrow) * ostride + (size_t)head * HD + 16 * s + 8 * h);
      if (o_in != nullptr) {
        const u32x4 ov = *reinterpret_cast<const u32x4*>(o_in + ((size_t)b * N + qrow) * ostride + (size_t)head * HD + 16 * s + 8 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) dpart += bflo(ov[e]) * bflo(w[e]) + bfhi(ov[e]) * bfhi(w[e]);
      }
    }
    qf[s] = scale_frag(v, sc2);                 // pre-scaled: S^T accumulates in the exp2 domain (used for S only)
    dof[s] = __builtin_bit_cast(bf16x8, w);
  }
  // -lse*log2e and -delta of this lane's query: read from rowc (written by attn_delta_kernel), or -- o_in / lse_in given --
  // computed here from this wave's own O and dO rows and WRITTEN to rowc for the dK/dV kernel that follows: the separate
  // pre-pass (one more read of O and dO, 29 ms per two steps) goes away.
  float nlse2 = 0.f, ndlt = 0.f;
  const size_t rc_i = ((size_t)b * H + head) * N + qrow, rc_plane = (size_t)gridDim.z * H * N;
  if (o_in != nullptr) {
    dpart += __shfl_xor(dpart, 32, 64);
    if (qrow < N) {
      nlse2 = -lse_in[rc_i] * LOG2E;
      ndlt = -dpart;
      if (h == 0) { rowc[rc_i] = nlse2; rowc[rc_plane + rc_i] = ndlt; }
    }
  } else if (qrow < N) {
    nlse2 = rowc[rc_i];
    ndlt = rowc[rc_plane + rc_i];
  }

  f32x16 dq[DB], lse_t, dlt_t;   // row constants replicated over an accumulator tile: out-of-place C operands, set once
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 16; ++g) dq[d][g] = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) { lse_t[g] = nlse2; dlt_t[g] = ndlt; }

  const int ntiles = (N + 63) / 64;
  // K/V rings filled by LDS-DMA (TileDma): step t issues tile t+NB-1 into the slot tile t-1 left, and waits before its
  // barrier for everything but the loads of the NB-2 youngest steps, i.e. for tile t+1.
  constexpr int NB = DQ_RING(HD);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  TileDma<HD, 4> dmk, dmv;
  dmk.init(kb_, rs, N, wid, lane);
  dmv.init(vb_, rs, N, wid, lane);
  constexpr int DMA_CNT = 2 * TileDma<HD, 4>::PER_WAVE * (NB - 2);
  auto issue = [&](int tt) {
    dmk.load(tt, lds0 + (unsigned)((tt % NB) * T::BYTES));
    dmv.load(tt, lds0 + (unsigned)((NB + tt % NB) * T::BYTES));
  };
#pragma unroll
  for (int s = 1 - NB; s < 0; ++s) issue(s + NB - 1);
  dma_wait_barrier<DMA_CNT>();

  auto tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const char* cK = (smem + (t % NB) * T::BYTES);
    const char* cV = (smem + (NB + t % NB) * T::BYTES);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 sa, dp;
      sa = mfma32(T::row_frag(cK, kb * 32, 0, lane), qf[0], lse_t);     // row constants as the initial accumulators
#pragma unroll
      for (int s = 1; s < KS; ++s) sa = mfma32(T::row_frag(cK, kb * 32, s, lane), qf[s], sa);
      dp = mfma32(T::row_frag(cV, kb * 32, 0, lane), dof[0], dlt_t);
#pragma unroll
      for (int s = 1; s < KS; ++s) dp = mfma32(T::row_frag(cV, kb * 32, s, lane), dof[s], dp);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float p = fast_exp2(sa[g]);
        if (TAIL) {
          const int key = t * 64 + kb * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
          if (key >= N) p = 0.f;
        }
        sa[g] = p * dp[g];                                             // dS / scale; scale is applied to dQ once
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 dsf = acc_to_frag(sa, s);
#pragma unroll
        for (int d = 0; d < DB; ++d) dq[d] = mfma32(T::tr_frag(cK, kb * 32, s, d * 32, lane), dsf, dq[d]);
      }
    }
  };
  if (__builtin_amdgcn_readfirstlane(bc.x * 128 + wid * 32) >= N) {   // no live query in this wave: stage and synchronise only
    for (int t = 0; t + 1 < ntiles; ++t) {
      issue(t + NB - 1);
      dma_wait_barrier<DMA_CNT>();
    }
    return;
  }
  for (int t = 0; t + 1 < ntiles; ++t) {
    issue(t + NB - 1);
    tile(t, std::false_type{});
    dma_wait_barrier<DMA_CNT>();
  }
  tile(ntiles - 1, std::true_type{});   // last tile: the only one that can hold keys >= N
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 16; ++g) dq[d][g] *= scale;
  if (qrow < N) {
    bf16_t* drow = dqkv + ((size_t)b * N + qrow) * rs + (size_t)head * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w = {pack2bf(dq[d][4 * g], dq[d][4 * g + 1]), pack2bf(dq[d][4 * g + 2], dq[d][4 * g + 3])};
        *reinterpret_cast<u32x2*>(drow + d * 32 + 8 * g + 4 * h) = w;
      }
  }
}

// =====================================================================================================
// backward dK/dV: keys on the lane (32 per wave, 128 per block), walks 64-query tiles of Q and dO
//   S' = Q (K*scale*log2e)^T - lse*log2e (row constant as the initial accumulator) ; P = exp2(S')
//   dV^T[d][key] += dO^T[d][query] P[query][key]
//   dP = dO V^T - delta ; dS = P dP scale ; dK^T[d][key] += Q^T[d][query] dS[query][key]
// =====================================================================================================
template <int HD>
__global__ __launch_bounds__(256, (HD == 32 ? ATT_OCC_DKV32 : 2)) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                              const float* __restrict__ rowc, bf16_t* __restrict__ dqkv, int N,
                                                              int H, float scale) {
  constexpr int KS = HD / 16, DB = HD / 32;
  constexpr int NB = DKV_RING(HD);
  using T = Tile<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS rings (LDS-DMA, see TileDma): Q [NB] | dO [NB] | row constants [NB][2: -lse*log2e, -delta][64] f32
  const float* ldsC = reinterpret_cast<const float*>(smem + 2 * NB * T::BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const BlockCoord bc = attn_block_coord();
  const int head = bc.head, b = bc.b;
  const size_t rs = (size_t)3 * H * HD;
  const size_t ostride = (size_t)H * HD;
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)head * HD;
  const bf16_t* kb_ = qb + (size_t)H * HD;
  const bf16_t* vb_ = qb + (size_t)2 * H * HD;
  const bf16_t* dob = dout + (size_t)b * N * ostride + (size_t)head * HD;
  const int krow = bc.x * 128 + wid * 32 + r;
  const float sc2 = scale * LOG2E;

  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    u32x4 v = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
    if (krow < N) {
      v = *reinterpret_cast<const u32x4*>(kb_ + (size_t)krow * rs + 16 * s + 8 * h);
      w = *reinterpret_cast<const u32x4*>(vb_ + (size_t)krow * rs + 16 * s + 8 * h);
    }
    kf[s] = scale_frag(v, sc2);                 // pre-scaled: S accumulates in the exp2 domain (used for S only)
    vf[s] = __builtin_bit_cast(bf16x8, w);
  }

  f32x16 dk[DB], dv[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 16; ++g) { dk[d][g] = 0.f; dv[d][g] = 0.f; }

  const int ntiles = (N + 63) / 64;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  TileDma<HD, 4> dmq, dmo;
  dmq.init(qb, rs, N, wid, lane);
  dmo.init(dob, ostride, N, wid, lane);
  // row constants: one 256-byte dword DMA per wave and tile -- waves 0/2 fetch the -lse*log2e row, waves 1/3 the -delta row
  // (2 and 3 repeat 0 and 1 so that every wave has the same number of loads in flight); queries >= N read as zero
  const int uw = __builtin_amdgcn_readfirstlane(wid);
  const unsigned long long ca = (unsigned long long)(rowc + (size_t)(uw & 1) * gridDim.z * H * N + ((size_t)b * H + head) * N);
  const i32x4_t rc = {(int)(unsigned)ca, (int)(unsigned)(ca >> 32), (int)(unsigned)((size_t)N * 4), 0x00020000};
  constexpr int DMA_CNT = (2 * TileDma<HD, 4>::PER_WAVE + 1) * (NB - 2);
  auto issue = [&](int tt) {
    dmq.load(tt, lds0 + (unsigned)((tt % NB) * T::BYTES));
    dmo.load(tt, lds0 + (unsigned)((NB + tt % NB) * T::BYTES));
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(2 * NB * T::BYTES + (tt % NB) * 512 + (uw & 1) * 256)));
    const unsigned off = (unsigned)((tt * 64 + lane) * 4);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(m0v), "v"(off), "s"(rc) : "memory", "m0");
#pragma clang diagnostic pop
  };
#pragma unroll
  for (int s = 1 - NB; s < 0; ++s) issue(s + NB - 1);
  dma_wait_barrier<DMA_CNT>();

  if (__builtin_amdgcn_readfirstlane(bc.x * 128 + wid * 32) >= N) {   // no live key in this wave: stage and synchronise only
    for (int t = 0; t < ntiles; ++t) {
      issue(t + NB - 1);
      dma_wait_barrier<DMA_CNT>();
    }
    return;
  }
  for (int t = 0; t < ntiles; ++t) {
    issue(t + NB - 1);
    const char* cQ = (smem + (t % NB) * T::BYTES);
    const char* cO = (smem + (NB + t % NB) * T::BYTES);
    const float* cC = ldsC + (t % NB) * 128;
#pragma unroll
    for (int qb32 = 0; qb32 < 2; ++qb32) {
      f32x16 sa, dp;
      // row constants: register 4G+e <-> query row qb32*32 + 8G + 4h + e
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(cC + qb32 * 32 + 8 * G + 4 * h);
        const f32x4 d = *reinterpret_cast<const f32x4*>(cC + 64 + qb32 * 32 + 8 * G + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sa[4 * G + e] = a[e]; dp[4 * G + e] = d[e]; }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) sa = mfma32(T::row_frag(cQ, qb32 * 32, s, lane), kf[s], sa);
#pragma unroll
      for (int s = 0; s < KS; ++s) dp = mfma32(T::row_frag(cO, qb32 * 32, s, lane), vf[s], dp);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float p = fast_exp2(sa[g]);
        sa[g] = p;
        dp[g] = p * dp[g];                                             // dS / scale; scale is applied to dK once
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pf = acc_to_frag(sa, s);
        const bf16x8 dsf = acc_to_frag(dp, s);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          dv[d] = mfma32(T::tr_frag(cO, qb32 * 32, s, d * 32, lane), pf, dv[d]);
          dk[d] = mfma32(T::tr_frag(cQ, qb32 * 32, s, d * 32, lane), dsf, dk[d]);
        }
      }
    }
    dma_wait_barrier<DMA_CNT>();
  }
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 16; ++g) dk[d][g] *= scale;
  if (krow < N) {
    bf16_t* drow = dqkv + ((size_t)b * N + krow) * rs + (size_t)head * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 wk = {pack2bf(dk[d][4 * g], dk[d][4 * g + 1]), pack2bf(dk[d][4 * g + 2], dk[d][4 * g + 3])};
        u32x2 wv = {pack2bf(dv[d][4 * g], dv[d][4 * g + 1]), pack2bf(dv[d][4 * g + 2], dv[d][4 * g + 3])};
        *reinterpret_cast<u32x2*>(drow + (size_t)H * HD + d * 32 + 8 * g + 4 * h) = wk;
        *reinterpret_cast<u32x2*>(drow + (size_t)2 * H * HD + d * 32 + 8 * g + 4 * h) = wv;
      }
  }
}

template <int HD>
static int run_fwd(const bf16_t* qkv, bf16_t* o, float* lse, int B, int N, int H, float scale, int* flag, hipStream_t st) {
  const int lds = 2 * FWD_RING(HD) * Tile<HD>::BYTES;
  dim3 grid((N + 127) / 128, H, B);
  if (flag != nullptr) {      // optimistic kernel, then the safe one (a no-op unless the flag was raised)
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((attn_fwd_kernel<HD, true>), grid, dim3(256), lds, st, qkv, o, lse, N, H, scale, flag);
    OCTMAE_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL((attn_fwd_kernel<HD, false>), grid, dim3(256), lds, st, qkv, o, lse, N, H, scale, flag);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

template <int HD>
static int run_delta(const bf16_t* o, const bf16_t* dout, const float* lse, float* rowc, int B, int N, int H, hipStream_t st) {
  int blocks = (B * N + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(attn_delta_kernel<HD>, dim3(blocks), dim3(256), 0, st, o, dout, lse, rowc, B * N, N, H);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}
template <int HD>
static int run_dq(const bf16_t* qkv, const bf16_t* dout, float* rowc, bf16_t* dqkv, int B, int N, int H, float scale,
                  hipStream_t st, const bf16_t* o_in = nullptr, const float* lse_in = nullptr) {
  dim3 grid((N + 127) / 128, H, B);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<HD>, grid, dim3(256), 2 * DQ_RING(HD) * Tile<HD>::BYTES, st, qkv, dout, rowc, dqkv, N, H, scale,
                     o_in, lse_in);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}
template <int HD>
static int run_dkv(const bf16_t* qkv, const bf16_t* dout, const float* rowc, bf16_t* dqkv, int B, int N, int H, float scale,
                   hipStream_t st) {
  dim3 grid((N + 127) / 128, H, B);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<HD>, grid, dim3(256), DKV_RING(HD) * (2 * Tile<HD>::BYTES + 512), st, qkv, dout, rowc, dqkv,
                     N, H, scale);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace octmae
using namespace octmae;

extern "C" int octmae_attn_fwd(const void* qkv, void* o, float* lse, int* flag_ws, int B, int N, int H, int HD, float scale,
                               void* stream) {
  OCTMAE_CHECK_ARG(qkv && o && lse && B > 0 && N > 0 && H > 0);
  OCTMAE_CHECK_ARG(HD == 64 || HD == 32);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (HD == 64) return run_fwd<64>(reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<bf16_t*>(o), lse, B, N, H, scale, flag_ws, st);
  return run_fwd<32>(reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<bf16_t*>(o), lse, B, N, H, scale, flag_ws, st);
}

#define BFP(x) reinterpret_cast<const bf16_t*>(x)
extern "C" int octmae_attn_bwd_rowconst(const void* o, const void* dout, const float* lse, float* rowc, int B, int N, int H, int HD,
                                        void* stream) {
  OCTMAE_CHECK_ARG(o && dout && lse && rowc && B > 0 && N > 0 && H > 0 && (HD == 64 || HD == 32));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return HD == 64 ? run_delta<64>(BFP(o), BFP(dout), lse, rowc, B, N, H, st) : run_delta<32>(BFP(o), BFP(dout), lse, rowc, B, N, H, st);
}
extern "C" int octmae_attn_bwd_dq(const void* qkv, const void* dout, const float* rowc, void* dqkv, int B, int N, int H, int HD,
                                  float scale, void* stream) {
  OCTMAE_CHECK_ARG(qkv && dout && rowc && dqkv && B > 0 && N > 0 && H > 0 && (HD == 64 || HD == 32));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  bf16_t* d = reinterpret_cast<bf16_t*>(dqkv);
  float* rc = const_cast<float*>(rowc);   // only read on this path
  return HD == 64 ? run_dq<64>(BFP(qkv), BFP(dout), rc, d, B, N, H, scale, st) : run_dq<32>(BFP(qkv), BFP(dout), rc, d, B, N, H, scale, st);
}
extern "C" int octmae_attn_bwd_dkv(const void* qkv, const void* dout, const float* rowc, void* dqkv, int B, int N, int H, int HD,
                                   float scale, void* stream) {
  OCTMAE_CHECK_ARG(qkv && dout && rowc && dqkv && B > 0 && N > 0 && H > 0 && (HD == 64 || HD == 32));
  OCTMAE_CHECK_ARG((size_t)N * 4 < 0xFFFFFFFFull);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  bf16_t* d = reinterpret_cast<bf16_t*>(dqkv);
  return HD == 64 ? run_dkv<64>(BFP(qkv), BFP(dout), rowc, d, B, N, H, scale, st) : run_dkv<32>(BFP(qkv), BFP(dout), rowc, d, B, N, H, scale, st);
}
// convenience: the three launches above, in order
extern "C" int octmae_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* rowc_ws, void* dqkv,
                               int B, int N, int H, int HD, float scale, void* stream) {
  int rc = octmae_attn_bwd_dq_rowconst(qkv, o, dout, lse, rowc_ws, dqkv, B, N, H, HD, scale, stream);
  if (rc) return rc;
  return octmae_attn_bwd_dkv(qkv, dout, rowc_ws, dqkv, B, N, H, HD, scale, stream);
}
extern "C" int octmae_attn_bwd_dq_rowconst(const void* qkv, const void* o, const void* dout, const float* lse, float* rowc,
                                           void* dqkv, int B, int N, int H, int HD, float scale, void* stream) {
  OCTMAE_CHECK_ARG(qkv && o && dout && lse && rowc && dqkv && B > 0 && N > 0 && H > 0 && (HD == 64 || HD == 32));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  bf16_t* d = reinterpret_cast<bf16_t*>(dqkv);
  return HD == 64 ? run_dq<64>(BFP(qkv), BFP(dout), rowc, d, B, N, H, scale, st, BFP(o), lse)
                  : run_dq<32>(BFP(qkv), BFP(dout), rowc, d, B, N, H, scale, st, BFP(o), lse);
}
