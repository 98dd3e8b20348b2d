// LayerNorm forward / backward over the fp32 residual stream (reference K5: nn.LayerNorm eps=1e-6,
// Pre-training/models_mae_joint_res_flash_attn.py:799; applied at video_vit.py:181-184, :489, :592).
//
// HBM-bound: one wave owns one token row, lane t holds float4 chunks t, t+64, ... (16-byte coalesced
// loads), statistics by wave64 butterfly shuffles, bf16 output packed 8 bytes per chunk.
//   fwd bytes/element: 4 (x) read + 2 (y) write
//   bwd bytes/element: 2 (dy) + 4 (x) read, 4 (dx) [+2 (dx bf16)] write; dgamma/dbeta/column sums are
//   accumulated per lane across the rows a wave walks, reduced over the block in LDS, written as per-block
//   partials to a caller-provided workspace and folded by a second small kernel (deterministic, no atomics).
#include <cstdlib>
#include "common.hpp"
#include "../../include/octmae.h"

// The row operands (x, dy, the residual gradient) are read exactly once: their loads are non-temporal, like the stores of the outputs.
// Measured same box (profiles/r04_layernorm_nt_loads.txt): forward 5.85 -> 6.10 TB/s, backward 5.91 -> 6.12 at D = 1024 inside the
// training step, +0.17 % on the step.  -DLN_PLAIN_LOADS: the ordinary loads (A/B builds).
#ifndef LN_PLAIN_LOADS
#define LN_LD(T, p) __builtin_nontemporal_load(reinterpret_cast<const T*>(p))
#else
#define LN_LD(T, p) (*reinterpret_cast<const T*>(p))
#endif

namespace octmae {

constexpr int LN_MAXC = 8;  // float4 chunks per lane -> D <= 2048

template <int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                     float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int nchunk = D >> 2;
  const float invD = 1.0f / (float)D;
  f32x4 g[NC], b[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ci = lane + 64 * c;
    if (ci < nchunk) {
      g[c] = *reinterpret_cast<const f32x4*>(gamma + 4 * ci);
      b[c] = *reinterpret_cast<const f32x4*>(beta + 4 * ci);
    }
  }
  // the next row of the wave is requested before the current one is reduced (as in the backward kernel)
  f32x4 vn[NC];
  auto issue = [&](int row) {
    const float* xr = x + (size_t)row * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) vn[c] = LN_LD(f32x4, xr + 4 * ci);
    }
  };
  if (wave < M) issue(wave);
  for (int row = wave; row < M; row += nwaves) {
    f32x4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = vn[c];
    if (row + nwaves < M) issue(row + nwaves);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[c][e] - mu;
          q = fmaf(d, d, q);
        }
      }
    }
    const float var = wave_sum(q) * invD;
    const float rs = rsqrtf(var + eps);
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
    bf16_t* yr = y + (size_t)row * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf((v[c][e] - mu) * rs, g[c][e], b[c][e]);
        u32x2 w = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x2*>(yr + 4 * ci));
      }
    }
  }
}

// dx = [dres +] rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * gamma
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16_t* __restrict__ dxb,
                                                     float* __restrict__ partial, bool want_dxsum, int M, int D) {
  __shared__ float red[3][4][64 * 4 + 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + w;
  const int nwaves = gridDim.x * 4;
  const int nchunk = D >> 2;
  const float invD = 1.0f / (float)D;
  f32x4 g[NC], ag[NC], ab[NC], as[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ci = lane + 64 * c;
    if (ci < nchunk) g[c] = *reinterpret_cast<const f32x4*>(gamma + 4 * ci);
#pragma unroll
    for (int e = 0; e < 4; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; as[c][e] = 0.f; }
  }
  // Software-pipelined over rows: the loads of the NEXT row are issued before the current row is reduced, so every
  // wave keeps two rows (12-20 KB) in flight -- with one row per wave the kernel sat at ~35 % of HBM bandwidth.
  // The incoming residual-stream gradient is part of the prefetch: vmcnt retires in order, so a load issued after the
  // next row's requests and needed now would drain them all and leave one row in flight.
  f32x4 xn[NC], rn[NC];
  u32x2 dn[NC];
  float mu_n = 0.f, rs_n = 0.f;
  auto issue = [&](int row) {
    mu_n = mean[row];
    rs_n = rstd[row];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
        xn[c] = LN_LD(f32x4, x + (size_t)row * D + 4 * ci);
        dn[c] = LN_LD(u32x2, dy + (size_t)row * D + 4 * ci);
        if (dres != nullptr) rn[c] = LN_LD(f32x4, dres + (size_t)row * D + 4 * ci);
      }
    }
  };
  if (wave < M) issue(wave);
  for (int row = wave; row < M; row += nwaves) {
    const float mu = mu_n, rs = rs_n;
    f32x4 xh[NC], rv[NC];
    u32x2 dw[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { xh[c] = xn[c]; dw[c] = dn[c]; rv[c] = rn[c]; }
    if (row + nwaves < M) issue(row + nwaves);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
        const float d[4] = {bflo(dw[c][0]), bfhi(dw[c][0]), bflo(dw[c][1]), bfhi(dw[c][1])};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xhat = (xh[c][e] - mu) * rs;
          const float gy = d[e] * g[c][e];
          xh[c][e] = xhat;
          s1 += gy;
          s2 = fmaf(gy, xhat, s2);
          ag[c][e] = fmaf(d[e], xhat, ag[c][e]);
          ab[c][e] += d[e];
        }
      }
    }
    const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
        const float d[4] = {bflo(dw[c][0]), bfhi(dw[c][0]), bflo(dw[c][1]), bfhi(dw[c][1])};
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (d[e] * g[c][e] - m1 - xh[c][e] * m2);
        if (dres != nullptr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += rv[c][e];
        }
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dx + (size_t)row * D + 4 * ci));
        if (dxb != nullptr) {
          u32x2 wv = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
          __builtin_nontemporal_store(wv, reinterpret_cast<u32x2*>(dxb + (size_t)row * D + 4 * ci));
        }
        if (want_dxsum) {
#pragma unroll
          for (int e = 0; e < 4; ++e) as[c][e] += o[e];
        }
      }
    }
  }
  // block reduction of the per-lane column partials, one chunk slot at a time; each block writes its [3][D] partial
  // sums to the workspace (a second tiny kernel folds them): 1024 workgroups adding atomically into the same 3 x D
  // words ran at the contended-atomic rate and took 3/4 of this kernel's time.
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ci = lane + 64 * c;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][w][lane * 4 + e] = ag[c][e];
      red[1][w][lane * 4 + e] = ab[c][e];
      red[2][w][lane * 4 + e] = as[c][e];
    }
    __syncthreads();
    if (w < 3 && ci < nchunk) {
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        t[e] = (red[w][0][lane * 4 + e] + red[w][1][lane * 4 + e]) + (red[w][2][lane * 4 + e] + red[w][3][lane * 4 + e]);
      *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 3 + w) * D + 4 * ci) = t;
    }
  }
}

// out_k[c] += sum_b partial[b][k][c] for k = 0 (dgamma), 1 (dbeta), 2 (dxsum); 64 columns x 16 block-groups per workgroup,
// fixed summation order (the result does not depend on scheduling)
__global__ __launch_bounds__(1024) void ln_bwd_finish_kernel(const float* __restrict__ partial, int nblocks, int D,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ dxsum) {
  __shared__ float red[16][64];
  const int k = blockIdx.y;
  float* dst = (k == 0) ? dgamma : (k == 1) ? dbeta : dxsum;
  if (dst == nullptr) return;
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < D) {
    const float* src = partial + (size_t)k * D + c;
    const size_t stride = (size_t)3 * D;
    int b = grp;
    for (; b + 48 < nblocks; b += 64) {
      s0 += src[(size_t)b * stride];
      s1 += src[(size_t)(b + 16) * stride];
      s2 += src[(size_t)(b + 32) * stride];
      s3 += src[(size_t)(b + 48) * stride];
    }
    for (; b < nblocks; b += 16) s0 += src[(size_t)b * stride];
  }
  red[grp][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && c < D) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][cl];
    dst[c] += t;
  }
}

// Grid: whole multiples of the 256 CUs, and FEW of them -- two workgroups per CU forward (8 waves per CU, two rows in flight each), one
// backward.  With every wave slot filled (2048 / 512 blocks: rounds 1-3) the same kernels read 5.0-5.5 TB/s; with 512 / 256 blocks
// 5.75-5.99 (round 4, in the step: ln_bwd d512 1034 -> 887 us, d1024 486 -> 455; ln_fwd d512 405 -> 350, d1024 185 -> 173; step
// +0.55 %): a wave streams one contiguous row at a time, and fewer concurrent row streams keep more DRAM pages open.  Grids that
// are not multiples of 256 (768, 384, 192) lose to imbalance; one block per CU forward starves the D = 512 rows (3.96 TB/s).
// OCTMAE_LN_FWD_GRID / OCTMAE_LN_BWD_GRID override the caps for A/B runs (profiles/r04_layernorm_grid.txt).
static inline int ln_grid(int M, bool bwd = false) {
  static const int capf = getenv("OCTMAE_LN_FWD_GRID") ? atoi(getenv("OCTMAE_LN_FWD_GRID")) : 512;
  static const int capb = getenv("OCTMAE_LN_BWD_GRID") ? atoi(getenv("OCTMAE_LN_BWD_GRID")) : 256;
  const int cap = bwd ? capb : capf;
  int blocks = (M + 3) / 4;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return blocks;
}

}  // namespace octmae
using namespace octmae;

extern "C" int octmae_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                                    float* rstd, int M, int D, float eps, void* stream) {
  OCTMAE_CHECK_ARG(x && gamma && beta && y_bf16 && mean && rstd);
  OCTMAE_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAXC);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nc = (D / 4 + 63) / 64;
  bf16_t* y = reinterpret_cast<bf16_t*>(y_bf16);
  dim3 grid(ln_grid(M)), blk(256);
  switch (nc) {
    case 1: hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, blk, 0, st, x, gamma, beta, y, mean, rstd, M, D, eps); break;
    case 2: hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, blk, 0, st, x, gamma, beta, y, mean, rstd, M, D, eps); break;
    case 3: case 4: hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, blk, 0, st, x, gamma, beta, y, mean, rstd, M, D, eps); break;
    default: hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, blk, 0, st, x, gamma, beta, y, mean, rstd, M, D, eps); break;
  }
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_layernorm_bwd_ws_floats(int M, int D) {
  int blocks = ln_grid(M, true);
  if (blocks > 512) blocks = 512;
  return blocks * 3 * D;
}

extern "C" int octmae_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                    const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgamma,
                                    float* dbeta, float* dxsum, float* partial_ws, int M, int D, void* stream) {
  OCTMAE_CHECK_ARG(dy_bf16 && x && mean && rstd && gamma && dx && partial_ws);
  OCTMAE_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAXC);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nc = (D / 4 + 63) / 64;
  const bf16_t* dy = reinterpret_cast<const bf16_t*>(dy_bf16);
  bf16_t* dxb = reinterpret_cast<bf16_t*>(dx_bf16);
  int blocks = ln_grid(M, true);
  if (blocks > 512) blocks = 512;
  dim3 grid(blocks), blk(256);
  const bool ws = dxsum != nullptr;
  switch (nc) {
    case 1: hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, blk, 0, st, dy, x, mean, rstd, gamma, dres, dx, dxb, partial_ws, ws, M, D); break;
    case 2: hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, blk, 0, st, dy, x, mean, rstd, gamma, dres, dx, dxb, partial_ws, ws, M, D); break;
    case 3: case 4: hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, blk, 0, st, dy, x, mean, rstd, gamma, dres, dx, dxb, partial_ws, ws, M, D); break;
    default: hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, blk, 0, st, dy, x, mean, rstd, gamma, dres, dx, dxb, partial_ws, ws, M, D); break;
  }
  OCTMAE_LAUNCH_CHECK();
  if (dgamma != nullptr || dbeta != nullptr || dxsum != nullptr) {
    hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((D + 63) / 64, 3), dim3(1024), 0, st, partial_ws, blocks, D, dgamma, dbeta, dxsum);
    OCTMAE_LAUNCH_CHECK();
  }
  return 0;
}
