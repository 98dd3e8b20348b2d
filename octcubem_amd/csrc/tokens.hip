// HBM-bound token plumbing of the 3-D MAE path (reference K1 gather side, K3, K9, K10):
//   cast / column sums, patch gather (im2col of kept tokens only), encoder / decoder sequence assembly,
//   fused patchify + masked-MSE forward and backward.
// All kernels move 16 bytes per lane per access along the contiguous axis.
#include "common.hpp"
#include "../../include/octmae.h"

namespace octmae {

// ---------------------------------------------------------------------------------------------
// fp32 -> bf16 cast (weight arena refresh, gradient cast).  8 elements per thread.
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  const size_t n8 = n >> 3;
  // four independent 32-byte loads per lane in flight (a grid-stride apart), then the four stores
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n8; i += 4 * stride) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const f32x4*>(src + 8 * (i + u * stride));
      b[u] = *reinterpret_cast<const f32x4*>(src + 8 * (i + u * stride) + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x4 w = {pack2bf(a[u][0], a[u][1]), pack2bf(a[u][2], a[u][3]), pack2bf(b[u][0], b[u][1]), pack2bf(b[u][2], b[u][3])};
      *reinterpret_cast<u32x4*>(dst + 8 * (i + u * stride)) = w;
    }
  }
  for (; i < n8; i += stride) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + 8 * i);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + 8 * i + 4);
    u32x4 w = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    *reinterpret_cast<u32x4*>(dst + 8 * i) = w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const size_t i = (n8 << 3) + threadIdx.x;
    dst[i] = f2bf(src[i]);
  }
}

// dst bf16 [R][D] = rowscale[r / rows_per_scale] * src f32 [R][D]   (stochastic-depth backward: the branch gradient)
__global__ __launch_bounds__(256) void cast_rowscale_kernel(const float* __restrict__ src, const float* __restrict__ rowscale,
                                                            bf16_t* __restrict__ dst, size_t R, int D, int rows_per_scale) {
  const int d8 = D >> 3;
  const size_t n8 = R * (size_t)d8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const float sc = rowscale[(i / d8) / rows_per_scale];
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + 8 * i) * sc;
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + 8 * i + 4) * sc;
    u32x4 w = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    *reinterpret_cast<u32x4*>(dst + 8 * i) = w;
  }
}

// ---------------------------------------------------------------------------------------------
// out[c] += sum_rows in[r][c]   (bias gradients).  Block = 32 row-lanes x 8 column-lanes of 8 columns.
template <bool IN_BF16>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ in, float* __restrict__ out, int M, int N,
                                                     int ld, int rows_per_block) {
  __shared__ float red[32][65];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cl * 8;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c0 < N) {
    for (int r = r0 + rl; r < r1; r += 32) {
      if (IN_BF16) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(in) + (size_t)r * ld + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += bflo(w[e]); acc[2 * e + 1] += bfhi(w[e]); }
      } else {
        const float* p = reinterpret_cast<const float*>(in) + (size_t)r * ld + c0;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += a[e]; acc[4 + e] += b[e]; }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cl * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) s += red[r][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < N) unsafeAtomicAdd(out + c, s);
  }
}

// ---------------------------------------------------------------------------------------------
// Patch gather = im2col restricted to the kept tokens (the reference embeds all L tokens and gathers
// afterwards, video_vit.py:80-82 then models_mae…:363; embedding only the kept rows is output-identical).
// out[row = b*nkeep + i][ (c, u, py, px) ] = imgs[b][c][t*tp + u][hy*p + py][wx*p + px],  token id -> (t, hy, wx).
// One thread = 8 consecutive px (32 B fp32 read -> 16 B bf16 write).
template <typename IdxT>
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ imgs, const IdxT* __restrict__ ids,
                                                           bf16_t* __restrict__ out, int B, int C, int T, int H, int W,
                                                           int tp, int p, int nkeep, int L) {
  const int gh = H / p, gw = W / p;
  const int pc = p >> 3;                   // 8-px chunks per patch row
  const int kdim = C * tp * p * p;
  const int chunks_per_tok = kdim >> 3;
  const size_t total = (size_t)B * nkeep * chunks_per_tok;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
    const int ch = (int)(q % chunks_per_tok);
    const size_t row = q / chunks_per_tok;
    const int b = (int)(row / nkeep), i = (int)(row % nkeep);
    const int id = ids ? (int)ids[(size_t)b * nkeep + i] : i;
    const int t = id / (gh * gw), hy = (id / gw) % gh, wx = id % gw;
    const int px8 = ch % pc;
    int rest = ch / pc;
    const int py = rest % p; rest /= p;
    const int u = rest % tp;
    const int c = rest / tp;
    const float* src = imgs + ((((size_t)b * C + c) * T + (t * tp + u)) * H + (hy * p + py)) * W + wx * p + px8 * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
    const f32x4 d = *reinterpret_cast<const f32x4*>(src + 4);
    u32x4 w = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(d[0], d[1]), pack2bf(d[2], d[3])};
    *reinterpret_cast<u32x4*>(out + row * kdim + (size_t)ch * 8) = w;
  }
}

// ---------------------------------------------------------------------------------------------
// Encoder sequence assembly (models_mae…:409-478): x[b][0] = cls + pos_cls; x[b][1+i] = tok[b][i] + pos[ids_keep[b][i]].
template <typename IdxT>
__global__ __launch_bounds__(256) void enc_assemble_kernel(const bf16_t* __restrict__ tok, const float* __restrict__ pos,
                                                           const float* __restrict__ cls, const float* __restrict__ pos_cls,
                                                           const IdxT* __restrict__ ids_keep, float* __restrict__ x, int B,
                                                           int nkeep, int D) {
  const int d4 = D >> 2;
  const size_t total = (size_t)B * (nkeep + 1) * d4;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
    const int c = (int)(q % d4);
    const size_t row = q / d4;
    const int b = (int)(row / (nkeep + 1)), i = (int)(row % (nkeep + 1));
    f32x4 o;
    if (i == 0) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(cls + 4 * c);
      const f32x4 d = *reinterpret_cast<const f32x4*>(pos_cls + 4 * c);
      o = a + d;
    } else {
      const int id = (int)ids_keep[(size_t)b * nkeep + i - 1];
      const u32x2 w = *reinterpret_cast<const u32x2*>(tok + ((size_t)b * nkeep + i - 1) * D + 4 * c);
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pos + (size_t)id * D + 4 * c);
      o[0] = bflo(w[0]) + pv[0]; o[1] = bfhi(w[0]) + pv[1]; o[2] = bflo(w[1]) + pv[2]; o[3] = bfhi(w[1]) + pv[3];
    }
    *reinterpret_cast<f32x4*>(x + row * D + 4 * c) = o;
  }
}

// Decoder sequence assembly (models_mae…:511-573): x[b][0] = dcls + dpos_cls;
// x[b][1+j] = (ids_restore[b][j] < nkeep ? emb[b][ids_restore[b][j]] : mask_token) + dpos[j].
template <typename IdxT>
__global__ __launch_bounds__(256) void dec_assemble_kernel(const bf16_t* __restrict__ emb, const float* __restrict__ mask_token,
                                                           const float* __restrict__ dpos, const float* __restrict__ dcls,
                                                           const float* __restrict__ dpos_cls, const IdxT* __restrict__ ids_restore,
                                                           float* __restrict__ x, int B, int nkeep, int L, int D, int emb_has_cls) {
  // emb_has_cls (2-D MAE, OCTCube/models_mae.py:175-178): emb holds 1 + nkeep rows per sample and row 0 (the encoder's cls
  // token after decoder_embed) becomes the decoder cls row; otherwise the cls row is the parameter `dcls`.
  const int erows = nkeep + (emb_has_cls ? 1 : 0);
  const int d4 = D >> 2;
  const size_t total = (size_t)B * (L + 1) * d4;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
    const int c = (int)(q % d4);
    const size_t row = q / d4;
    const int b = (int)(row / (L + 1)), j = (int)(row % (L + 1));
    f32x4 o;
    if (j == 0) {
      const f32x4 pc = *reinterpret_cast<const f32x4*>(dpos_cls + 4 * c);
      if (emb_has_cls) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(emb + ((size_t)b * erows) * D + 4 * c);
        o[0] = bflo(w[0]) + pc[0]; o[1] = bfhi(w[0]) + pc[1]; o[2] = bflo(w[1]) + pc[2]; o[3] = bfhi(w[1]) + pc[3];
      } else {
        o = *reinterpret_cast<const f32x4*>(dcls + 4 * c) + pc;
      }
    } else {
      const int r = (int)ids_restore[(size_t)b * L + j - 1];
      const f32x4 pv = *reinterpret_cast<const f32x4*>(dpos + (size_t)(j - 1) * D + 4 * c);
      if (r < nkeep) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(emb + ((size_t)b * erows + (emb_has_cls ? 1 : 0) + r) * D + 4 * c);
        o[0] = bflo(w[0]) + pv[0]; o[1] = bfhi(w[0]) + pv[1]; o[2] = bflo(w[1]) + pv[2]; o[3] = bfhi(w[1]) + pv[3];
      } else {
        o = *reinterpret_cast<const f32x4*>(mask_token + 4 * c) + pv;
      }
    }
    *reinterpret_cast<f32x4*>(x + row * D + 4 * c) = o;
  }
}

// Row gather with fp32 -> bf16 cast: out[b*n + i][:] = src[b][1 + ids[b][i]][:]  (src rows include the cls slot).
// Backward of both assemblies w.r.t. the token matrix: ids = ids_keep (a permutation prefix), so a pure gather.
// ids == nullptr means identity (out[b*n+i] = src[b][1+i]).
// ---------------------------------------------------------------------------------------------
// Backward of a row gather as a GATHER (scatter_add_rows): out[l][:] (+)= sum over samples b with ids_restore[b][l] < nkeep of
// src[b][row0 + ids_restore[b][l]][:].  A kept token's source row is its rank in the shuffle, so the positional table's
// gradient needs no atomics: one workgroup per table row l collects the samples that kept l (deterministic compaction, ascending
// b) and sums their rows in that order, four row loads in flight.  Replaces ATen's index_add_ (fp32 atomics in arrival order).
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, const long long* __restrict__ ids_restore,
                                                               float* __restrict__ out, int B, int nkeep, int L, int D,
                                                               int src_rows, int row0, int accumulate) {
  __shared__ int rows[1024];                 // compact list: source row index (b * src_rows + row0 + rank) of every keeper
  __shared__ int wave_cnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int d4 = D >> 2;
  for (int l = blockIdx.x; l < L; l += gridDim.x) {
    int total = 0;
    for (int b0 = 0; b0 < B; b0 += 256) {    // B <= 1024 (checked by the launcher)
      const int b = b0 + tid;
      int rank = nkeep;
      if (b < B) rank = (int)ids_restore[(size_t)b * L + l];
      const bool keep = rank < nkeep;
      const unsigned long long m = __ballot(keep);
      if (lane == 0) wave_cnt[wid] = __popcll(m);
      __syncthreads();
      int base = total;
      for (int w = 0; w < wid; ++w) base += wave_cnt[w];
      if (keep) rows[base + __popcll(m & ((1ull << lane) - 1ull))] = b * src_rows + row0 + rank;
      total += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      __syncthreads();
    }
    for (int c = tid; c < d4; c += 256) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (accumulate) acc = *reinterpret_cast<const f32x4*>(out + (size_t)l * D + 4 * c);
      int i = 0;
      for (; i + 4 <= total; i += 4) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (size_t)rows[i] * D + 4 * c);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (size_t)rows[i + 1] * D + 4 * c);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (size_t)rows[i + 2] * D + 4 * c);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (size_t)rows[i + 3] * D + 4 * c);
        acc += v0; acc += v1; acc += v2; acc += v3;
      }
      for (; i < total; ++i) acc += *reinterpret_cast<const f32x4*>(src + (size_t)rows[i] * D + 4 * c);
      *reinterpret_cast<f32x4*>(out + (size_t)l * D + 4 * c) = acc;
    }
    __syncthreads();                         // the list is rebuilt for the next row
  }
}

// Backward of the decoder assembly w.r.t. the positional table and the mask token in ONE pass over the gradient
// (models_mae_joint_res_flash_attn.py:515-573 under autograd): ddpos[l][:] = sum_b dx[b][1 + l][:] and, per table row,
// dmask_part[l][:] = sum over the samples in which token l was masked (ids_restore[b][l] >= nkeep); the caller folds dmask_part
// over l with octmae_colsum_accum.  (ATen: a [B, L, D] multiply by the mask, two reductions and the copies between them.)
__global__ __launch_bounds__(256) void dec_assemble_bwd_kernel(const float* __restrict__ dx, const long long* __restrict__ ids_restore,
                                                               float* __restrict__ ddpos, float* __restrict__ dmask_part, int B,
                                                               int nkeep, int L, int D) {
  __shared__ unsigned char masked[1024];
  const int tid = threadIdx.x;
  const int d4 = D >> 2;
  for (int l = blockIdx.x; l < L; l += gridDim.x) {
    for (int b = tid; b < B; b += 256) masked[b] = ids_restore[(size_t)b * L + l] >= nkeep ? 1 : 0;
    __syncthreads();
    for (int c = tid; c < d4; c += 256) {
      f32x4 all = {0.f, 0.f, 0.f, 0.f}, msk = all;
      const float* col = dx + ((size_t)1 + l) * D + 4 * c;
      const size_t bs = (size_t)(L + 1) * D;
      int b = 0;
      for (; b + 4 <= B; b += 4) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(col + (size_t)b * bs);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(col + (size_t)(b + 1) * bs);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(col + (size_t)(b + 2) * bs);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(col + (size_t)(b + 3) * bs);
        all += v0; all += v1; all += v2; all += v3;
        if (masked[b]) msk += v0;
        if (masked[b + 1]) msk += v1;
        if (masked[b + 2]) msk += v2;
        if (masked[b + 3]) msk += v3;
      }
      for (; b < B; ++b) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(col + (size_t)b * bs);
        all += v;
        if (masked[b]) msk += v;
      }
      *reinterpret_cast<f32x4*>(ddpos + (size_t)l * D + 4 * c) = all;
      *reinterpret_cast<f32x4*>(dmask_part + (size_t)l * D + 4 * c) = msk;
    }
    __syncthreads();
  }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void gather_rows_cast_kernel(const float* __restrict__ src, const IdxT* __restrict__ ids,
                                                               bf16_t* __restrict__ out, int B, int n, int src_rows, int D) {
  const int d8 = D >> 3;
  const size_t total = (size_t)B * n * d8;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
    const int c = (int)(q % d8);
    const size_t row = q / d8;
    const int b = (int)(row / n), i = (int)(row % n);
    const int id = ids ? (int)ids[(size_t)b * n + i] : i;
    const float* s = src + ((size_t)b * src_rows + 1 + id) * D + 8 * c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(s);
    const f32x4 d = *reinterpret_cast<const f32x4*>(s + 4);
    u32x4 w = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(d[0], d[1]), pack2bf(d[2], d[3])};
    *reinterpret_cast<u32x4*>(out + row * D + 8 * c) = w;
  }
}

// ---------------------------------------------------------------------------------------------
// Fused patchify + masked MSE (models_mae…:289-314, :613-667).  One wave per token.
// target[(u, py, px, c)] = imgs[b][c][fi(t*u_sz + u)][hy*p + py][wx*p + px]; fi = linspace frame select (identity
// when pred_t_dim == T).  pred rows live in a [B][L+1][PD] buffer (row 0 of each sample = cls, skipped).
// MODE 0: loss_tok[b][l] = mean_e (pred - target)^2           (forward)
// MODE 1: dpred[b][1+l][e] = coef * mask[b][l] * (pred - target), dpred[b][0][:] = 0, bf16 (backward)
template <int MODE>
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ imgs,
                                                  const int* __restrict__ frame_idx, const float* __restrict__ mask,
                                                  const float* __restrict__ coef_ptr, float* __restrict__ loss_tok,
                                                  bf16_t* __restrict__ dpred, int B, int C, int T, int H, int W, int u_sz,
                                                  int p, int L, int norm_pix) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int gh = H / p, gw = W / p;
  const int PD = u_sz * p * p * C;
  const int nq = PD >> 2;  // float4 groups along (u,py,px,c); contiguous in pred
  const float coef = (MODE == 1) ? *coef_ptr : 0.f;
  const int rows = (MODE == 1) ? B * (L + 1) : B * L;
  for (int row = wave; row < rows; row += nwaves) {
    int b, l;
    if (MODE == 1) {
      b = row / (L + 1);
      l = row % (L + 1) - 1;
      if (l < 0) {
        for (int q = lane; q < (PD >> 3); q += 64) {
          u32x4 z = {0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(dpred + (size_t)row * PD + 8 * q) = z;
        }
        continue;
      }
    } else {
      b = row / L;
      l = row % L;
    }
    const float mk = (MODE == 1) ? mask[(size_t)b * L + l] : 0.f;
    const int t = l / (gh * gw), hy = (l / gw) % gh, wx = l % gw;
    const float* pr = pred + ((size_t)b * (L + 1) + 1 + l) * PD;
    // target statistics for norm_pix_loss (unbiased variance, torch.var default)
    float mu = 0.f, inv = 1.f;
    if (norm_pix) {
      float s = 0.f, s2 = 0.f;
      for (int e = lane; e < PD; e += 64) {
        const int c = e % C; int rest = e / C;
        const int px = rest % p; rest /= p;
        const int py = rest % p; const int u = rest / p;
        const int f = frame_idx ? frame_idx[t * u_sz + u] : t * u_sz + u;
        const float v = imgs[((((size_t)b * C + c) * T + f) * H + hy * p + py) * W + wx * p + px];
        s += v;
      }
      mu = wave_sum(s) / (float)PD;
      for (int e = lane; e < PD; e += 64) {
        const int c = e % C; int rest = e / C;
        const int px = rest % p; rest /= p;
        const int py = rest % p; const int u = rest / p;
        const int f = frame_idx ? frame_idx[t * u_sz + u] : t * u_sz + u;
        const float v = imgs[((((size_t)b * C + c) * T + f) * H + hy * p + py) * W + wx * p + px] - mu;
        s2 = fmaf(v, v, s2);
      }
      const float var = wave_sum(s2) / (float)(PD - 1);
      inv = rsqrtf(var + 1.0e-6f);
    }
    float acc = 0.f;
    for (int q = lane; q < nq; q += 64) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pr + 4 * q);
      float tg[4];
      if (C == 1) {  // 4 consecutive px of one patch row: a 16-byte image read
        const int e = 4 * q;
        const int px = e % p; int rest = e / p;
        const int py = rest % p; const int u = rest / p;
        const int f = frame_idx ? frame_idx[t * u_sz + u] : t * u_sz + u;
        const f32x4 iv = *reinterpret_cast<const f32x4*>(imgs + (((size_t)b * T + f) * H + hy * p + py) * W + wx * p + px);
        tg[0] = iv[0]; tg[1] = iv[1]; tg[2] = iv[2]; tg[3] = iv[3];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * q + k;
          const int c = e % C; int rest = e / C;
          const int px = rest % p; rest /= p;
          const int py = rest % p; const int u = rest / p;
          const int f = frame_idx ? frame_idx[t * u_sz + u] : t * u_sz + u;
          tg[k] = imgs[((((size_t)b * C + c) * T + f) * H + hy * p + py) * W + wx * p + px];
        }
      }
      float d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d[k] = pv[k] - (tg[k] - mu) * inv;
        acc = fmaf(d[k], d[k], acc);
      }
      if (MODE == 1) {
        const float s = coef * mk;
        u32x2 w = {pack2bf(s * d[0], s * d[1]), pack2bf(s * d[2], s * d[3])};
        *reinterpret_cast<u32x2*>(dpred + (size_t)row * PD + 4 * q) = w;
      }
    }
    if (MODE == 0) {
      acc = wave_sum(acc);
      if (lane == 0) loss_tok[(size_t)b * L + l] = acc / (float)PD;
    }
  }
}

static inline int grid_for(size_t work_items, int per_block = 256, int cap = 4096) {
  size_t g = (work_items + per_block - 1) / per_block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace octmae
using namespace octmae;

extern "C" int octmae_cast_f32_bf16(const float* src, void* dst_bf16, long long n, void* stream) {
  OCTMAE_CHECK_ARG(src && dst_bf16 && n >= 0);
  if (n == 0) return 0;
  // 512 workgroups (two per CU, four loads in flight per lane): 5.4 TB/s on the 332 M-parameter arena against 4.8-5.2 with 1024-4096
  // workgroups and 2.7-4.0 for the un-unrolled loop on 4096 (tools/cast_bench.py) -- the "few row streams" rule of layernorm.hip
  static const int cast_cap = getenv("OCTMAE_CAST_GRID") ? atoi(getenv("OCTMAE_CAST_GRID")) : 512;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for((size_t)n / 8 + 1, 256, cast_cap)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, reinterpret_cast<bf16_t*>(dst_bf16), (size_t)n);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_cast_rowscale_f32_bf16(const float* src, const float* rowscale, void* dst_bf16, long long R, int D,
                                             int rows_per_scale, void* stream) {
  OCTMAE_CHECK_ARG(src && rowscale && dst_bf16 && R >= 0 && D > 0 && D % 8 == 0 && rows_per_scale > 0);
  if (R == 0) return 0;
  hipLaunchKernelGGL(cast_rowscale_kernel, dim3(grid_for((size_t)R * (D / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src, rowscale, reinterpret_cast<bf16_t*>(dst_bf16), (size_t)R, D, rows_per_scale);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_colsum_accum(const void* in, int in_is_bf16, float* out, int M, int N, int ld, void* stream) {
  OCTMAE_CHECK_ARG(in && out && M > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0);
  int splits = (M + 255) / 256;
  if (splits > 128) splits = 128;
  int rpb = (M + splits - 1) / splits;
  rpb = (rpb + 31) / 32 * 32;
  splits = (M + rpb - 1) / rpb;
  dim3 grid((N + 63) / 64, splits);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (in_is_bf16) hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(256), 0, st, in, out, M, N, ld, rpb);
  else hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(256), 0, st, in, out, M, N, ld, rpb);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_patch_gather(const float* imgs, const void* ids, int ids_is_i64, void* out_bf16, int B, int C, int T,
                                   int H, int W, int tp, int p, int nkeep, void* stream) {
  OCTMAE_CHECK_ARG(imgs && out_bf16 && B > 0 && C > 0 && nkeep > 0);
  OCTMAE_CHECK_ARG(p % 8 == 0 && H % p == 0 && W % p == 0 && T % tp == 0 && W % 4 == 0);
  const int L = (T / tp) * (H / p) * (W / p);
  OCTMAE_CHECK_ARG(nkeep <= L);
  const size_t total = (size_t)B * nkeep * (C * tp * p * p / 8);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  bf16_t* out = reinterpret_cast<bf16_t*>(out_bf16);
  if (ids_is_i64)
    hipLaunchKernelGGL(patch_gather_kernel<long long>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, imgs,
                       reinterpret_cast<const long long*>(ids), out, B, C, T, H, W, tp, p, nkeep, L);
  else
    hipLaunchKernelGGL(patch_gather_kernel<int>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, imgs,
                       reinterpret_cast<const int*>(ids), out, B, C, T, H, W, tp, p, nkeep, L);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_enc_assemble(const void* tok_bf16, const float* pos, const float* cls, const float* pos_cls,
                                   const long long* ids_keep, float* x, int B, int nkeep, int D, void* stream) {
  OCTMAE_CHECK_ARG(tok_bf16 && pos && cls && pos_cls && ids_keep && x && B > 0 && nkeep > 0 && D % 4 == 0);
  const size_t total = (size_t)B * (nkeep + 1) * (D / 4);
  hipLaunchKernelGGL(enc_assemble_kernel<long long>, dim3(grid_for(total, 256, 8192)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16_t*>(tok_bf16), pos, cls, pos_cls,
                     ids_keep, x, B, nkeep, D);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_dec_assemble(const void* emb_bf16, const float* mask_token, const float* dpos, const float* dcls,
                                   const float* dpos_cls, const long long* ids_restore, float* x, int B, int nkeep, int L,
                                   int D, int emb_has_cls, void* stream) {
  OCTMAE_CHECK_ARG(emb_bf16 && mask_token && dpos && (dcls || emb_has_cls) && dpos_cls && ids_restore && x);
  OCTMAE_CHECK_ARG(B > 0 && nkeep > 0 && L >= nkeep && D % 4 == 0);
  const size_t total = (size_t)B * (L + 1) * (D / 4);
  hipLaunchKernelGGL(dec_assemble_kernel<long long>, dim3(grid_for(total, 256, 8192)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16_t*>(emb_bf16), mask_token, dpos,
                     dcls, dpos_cls, ids_restore, x, B, nkeep, L, D, emb_has_cls);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_gather_rows_cast(const float* src, const long long* ids, void* out_bf16, int B, int n, int src_rows,
                                       int D, void* stream) {
  OCTMAE_CHECK_ARG(src && out_bf16 && B > 0 && n > 0 && src_rows >= n + 1 && D % 8 == 0);
  const size_t total = (size_t)B * n * (D / 8);
  hipLaunchKernelGGL(gather_rows_cast_kernel<long long>, dim3(grid_for(total, 256, 8192)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, ids, reinterpret_cast<bf16_t*>(out_bf16), B, n, src_rows, D);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_scatter_add_rows(const float* src, const long long* ids_restore, float* out, int B, int nkeep, int L, int D,
                                       int src_rows, int row0, int accumulate, void* stream) {
  OCTMAE_CHECK_ARG(src && ids_restore && out && B > 0 && B <= 1024 && nkeep >= 0 && L > 0 && D > 0 && (D & 3) == 0 && src_rows >= row0 + nkeep && row0 >= 0);
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(L < 65535 ? L : 65535), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, ids_restore, out,
                     B, nkeep, L, D, src_rows, row0, accumulate);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_dec_assemble_bwd(const float* dx, const long long* ids_restore, float* ddpos, float* dmask_part, int B, int nkeep,
                                       int L, int D, void* stream) {
  OCTMAE_CHECK_ARG(dx && ids_restore && ddpos && dmask_part && B > 0 && B <= 1024 && nkeep >= 0 && L > 0 && D > 0 && (D & 3) == 0);
  hipLaunchKernelGGL(dec_assemble_bwd_kernel, dim3(L < 65535 ? L : 65535), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dx, ids_restore, ddpos,
                     dmask_part, B, nkeep, L, D);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_mse_fwd(const float* pred, const float* imgs, const int* frame_idx, float* loss_tok, int B, int C,
                              int T, int H, int W, int u_sz, int p, int L, int norm_pix, void* stream) {
  OCTMAE_CHECK_ARG(pred && imgs && loss_tok && B > 0 && L > 0);
  OCTMAE_CHECK_ARG(p % 4 == 0 && (u_sz * p * p * C) % 8 == 0 && W % 4 == 0);
  hipLaunchKernelGGL(mse_kernel<0>, dim3(grid_for((size_t)B * L, 4, 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     pred, imgs, frame_idx, loss_tok /*mask unused: any valid pointer*/, nullptr, loss_tok, nullptr, B, C, T, H,
                     W, u_sz, p, L, norm_pix);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_mse_bwd(const float* pred, const float* imgs, const int* frame_idx, const float* mask,
                              const float* coef, void* dpred_bf16, int B, int C, int T, int H, int W, int u_sz, int p, int L,
                              int norm_pix, void* stream) {
  OCTMAE_CHECK_ARG(pred && imgs && mask && coef && dpred_bf16 && B > 0 && L > 0);
  OCTMAE_CHECK_ARG(p % 4 == 0 && (u_sz * p * p * C) % 8 == 0 && W % 4 == 0);
  hipLaunchKernelGGL(mse_kernel<1>, dim3(grid_for((size_t)B * (L + 1), 4, 4096)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), pred, imgs, frame_idx, mask, coef, nullptr,
                     reinterpret_cast<bf16_t*>(dpred_bf16), B, C, T, H, W, u_sz, p, L, norm_pix);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}
