// The single-key tail of the fused attention backward (attn_bwd.hip: the comment block above attn_bwd_tail1_kernel) as a device
// function: the stand-alone kernel calls it with one pass of rows in flight; the one-wave-per-SIMD main kernels (attn_bwd1w.hip,
// attn_bwd1w64.hip) call it at their end for their own (batch, head) with DEPTH passes in flight -- a workgroup that has a CU to
// itself needs that many requests outstanding to keep its share of the memory system busy -- and find the workspace rows, Q and
// dO they have just streamed in the caches instead of in HBM.
#pragma once
#include "attn_tile.hpp"

namespace octmae {

// One workgroup of 256 threads per (batch, head) `bh` of `nbh`; `red`: 8 KiB of LDS.  HD / 8 lanes per query row (one 16-byte
// chunk each: a load instruction covers 8-16 whole rows of Q / dO and 1-2 KB of contiguous workspace).
// Same rounding points as the MFMA path: K * scale * log2e, P and dS rounded to bf16 where that path feeds them to an MFMA; fp32 sums.
template <int HD, int DEPTH>
__device__ __forceinline__ void attn_bwd_tail1_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                    const float* __restrict__ rowc, const float* __restrict__ dq_ws,
                                                    bf16_t* __restrict__ dqkv, int N, int NPAD, int H, int key, int have_ws, float scale,
                                                    int bh, int nbh, float* red, int tid) {
  constexpr int LPR = HD / 8, RPP = 256 / LPR;               // lanes per query row (8 head dims = one 16-byte chunk each), rows per pass
  const int c = tid % LPR, rlane = tid / LPR;
  const int b = bh / H, head = bh % H;
  const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)head * HD + 8 * c;
  const bf16_t* kb_ = qb + (size_t)H * HD;
  const bf16_t* vb_ = qb + (size_t)2 * H * HD;
  const bf16_t* dob = dout + (size_t)b * N * os + (size_t)head * HD + 8 * c;
  const float* wsb = dq_ws + (size_t)bh * N * HD + 8 * c;
  bf16_t* dqb = dqkv + (size_t)b * N * rs + (size_t)head * HD + 8 * c;
  const float* rc_l = rowc + (size_t)bh * NPAD;
  const float* rc_d = rowc + (size_t)nbh * NPAD + (size_t)bh * NPAD;
  const float sc2 = scale * LOG2E;

  // this lane's 8 dims of the key: K (raw, for dQ), K * scale * log2e rounded to bf16 (for S), V
  float kf[8], ksf[8], vf[8];
  {
    const u32x4 kw = *reinterpret_cast<const u32x4*>(kb_ + (size_t)key * rs);
    const u32x4 vw = *reinterpret_cast<const u32x4*>(vb_ + (size_t)key * rs);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kf[2 * e] = bflo(kw[e]); kf[2 * e + 1] = bfhi(kw[e]);
      vf[2 * e] = bflo(vw[e]); vf[2 * e + 1] = bfhi(vw[e]);
      const unsigned ks = pack2bf(bflo(kw[e]) * sc2, bfhi(kw[e]) * sc2);
      ksf[2 * e] = bflo(ks); ksf[2 * e + 1] = bfhi(ks);
    }
  }
  float dk[8], dv[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) { dk[d] = 0.f; dv[d] = 0.f; }

  // DEPTH passes ahead: the loads of pass i + DEPTH are requested right after pass i is computed
  struct Row { u32x4 qw, ow; f32x4 w0, w1; float cl, cd; };
  auto ld = [&](int q, Row& r) {
    const int qq = q < N ? q : N - 1;
    r.qw = *reinterpret_cast<const u32x4*>(qb + (size_t)qq * rs);
    r.ow = *reinterpret_cast<const u32x4*>(dob + (size_t)qq * os);
    r.w0 = f32x4{0.f, 0.f, 0.f, 0.f}; r.w1 = r.w0;
    if (have_ws) {
      r.w0 = *reinterpret_cast<const f32x4*>(wsb + (size_t)qq * HD);
      r.w1 = *reinterpret_cast<const f32x4*>(wsb + (size_t)qq * HD + 4);
    }
    r.cl = rc_l[qq]; r.cd = rc_d[qq];
  };
  Row ring[DEPTH];
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) ld(j * RPP + rlane, ring[j]);
  for (int q0 = 0; q0 < N; q0 += RPP * DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int q = q0 + j * RPP + rlane;
      const Row cur = ring[j];
      ld(q + RPP * DEPTH, ring[j]);
      float qf[8], of[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qf[2 * e] = bflo(cur.qw[e]); qf[2 * e + 1] = bfhi(cur.qw[e]);
        of[2 * e] = bflo(cur.ow[e]); of[2 * e + 1] = bfhi(cur.ow[e]);
      }
      float sp = 0.f, dpp = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) { sp = fmaf(qf[d], ksf[d], sp); dpp = fmaf(of[d], vf[d], dpp); }
#pragma unroll
      for (int m = 1; m < LPR; m <<= 1) { sp += __shfl_xor(sp, m, 64); dpp += __shfl_xor(dpp, m, 64); }
      const bool live = q < N;
      const float p = live ? fast_exp2(sp + cur.cl) : 0.f;
      const float ds = p * (dpp + cur.cd);
      const unsigned pr = pack2bf(p, ds);                       // the MFMA path rounds P (for dV) and dS (for dK, dQ) to bf16
      const float pb = bflo(pr), dsb = bfhi(pr);
#pragma unroll
      for (int d = 0; d < 8; ++d) { dv[d] = fmaf(pb, of[d], dv[d]); dk[d] = fmaf(dsb, qf[d], dk[d]); }
      if (live) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = fmaf(dsb, kf[2 * e], e < 2 ? cur.w0[2 * e] : cur.w1[2 * e - 4]);
          const float a1 = fmaf(dsb, kf[2 * e + 1], e < 2 ? cur.w0[2 * e + 1] : cur.w1[2 * e - 3]);
          w[e] = pack2bf(a0 * scale, a1 * scale);
        }
        *reinterpret_cast<u32x4*>(dqb + (size_t)q * rs) = w;
      }
    }
  }
  // ---- dK, dV of the key: the 256 / LPR row lanes summed in a fixed order
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 8; ++d) red[d * 256 + tid] = which ? dv[d] : dk[d];
    __syncthreads();
    if (tid < HD) {
      const int dd = tid % 8;
      float acc = 0.f;
      for (int t = tid / 8; t < 256; t += LPR) acc += red[dd * 256 + t];
      bf16_t* dst = dqkv + ((size_t)b * N + key) * rs + (size_t)(which ? 2 : 1) * H * HD + (size_t)head * HD + tid;
      *dst = (bf16_t)(pack2bf(which ? acc : acc * scale, 0.f) & 0xffffu);
    }
  }
}

}  // namespace octmae
