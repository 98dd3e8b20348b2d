// Optimizer-side kernels (reference K11): multi-tensor gradient L2 norm (get_grad_norm_,
// Pre-training/custom_util/misc.py:356-373), gradient clip scale, and fused multi-tensor AdamW
// (torch.optim._multi_tensor.AdamW built at main_pretrain_oph_joint_2d512_flash_attn.py:451).
// HBM-bound: AdamW touches 16 B read + 12 B written per parameter; one launch per parameter group.
//
// Multi-tensor addressing: a device table of tensors {p, g, m, v, n} plus a chunk table mapping every
// 65536-element chunk to (tensor, offset); one workgroup per chunk, 16-byte accesses.
#include "common.hpp"
#include "../../include/octmae.h"

namespace octmae {

constexpr int CHUNK = 65536;

struct TensorDesc {
  float* p;
  float* g;
  float* m;
  float* v;
  long long n;
};

// sumsq[t] += sum(g_t^2) for every tensor t of the table
__global__ __launch_bounds__(256) void mt_sumsq_kernel(const TensorDesc* __restrict__ tensors, const int* __restrict__ chunk_tensor,
                                                       const long long* __restrict__ chunk_off, float* __restrict__ sumsq) {
  __shared__ float red[4];
  const int t = chunk_tensor[blockIdx.x];
  const long long off = chunk_off[blockIdx.x];
  const TensorDesc d = tensors[t];
  long long n = d.n - off;
  if (n > CHUNK) n = CHUNK;
  const float* g = d.g + off;
  float s = 0.f;
  const bool vec = ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
  if (vec) {
    const long long n4 = n >> 2;
    for (long long i = threadIdx.x; i < n4; i += 256) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(g + 4 * i);
      s += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
    }
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  } else {
    for (long long i = threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(sumsq + t, (red[0] + red[1]) + (red[2] + red[3]));
}

// total_norm = sqrt(sum_t sumsq[t])  (== 2-norm of the stack of per-tensor 2-norms);
// clip_coef = min(1, max_norm / (total_norm + 1e-6)) when max_norm > 0 (torch clip_grad_norm_), else 1
__global__ void mt_finish_norm_kernel(const float* __restrict__ sumsq, int nt, float max_norm, float* __restrict__ out_norm,
                                      float* __restrict__ out_coef) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nt; i += 64) s += sumsq[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(s);
    *out_norm = nrm;
    float c = 1.f;
    if (max_norm > 0.f) {
      c = max_norm / (nrm + 1e-6f);
      if (c > 1.f) c = 1.f;
    }
    *out_coef = c;
  }
}

struct AdamArgs {
  float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;  // bc1 = 1 - beta1^t, bc2_sqrt = sqrt(1 - beta2^t)
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  p *= (1.f - a.lr * a.wd);
  m = a.beta1 * m + (1.f - a.beta1) * g;
  v = a.beta2 * v + (1.f - a.beta2) * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= (a.lr / a.bc1) * (m / denom);
}

// g is multiplied by *gscale (clip coefficient and/or 1/loss-scale) before use.
// LP: also write the 16-bit operand copy of the updated parameter (lp_table[t], the arena's mirror: the per-forward cast pass over
// the whole arena -- 6 B per parameter -- becomes 2 B written here).  SQ: also accumulate sumsq[t] += sum(g^2) of the RAW gradient
// (get_grad_norm_ without its own pass over the gradients; only when no clip coefficient has to be known before the update).
template <bool LP, bool SQ>
__global__ __launch_bounds__(256) void mt_adamw_kernel(const TensorDesc* __restrict__ tensors, const int* __restrict__ chunk_tensor,
                                                       const long long* __restrict__ chunk_off, const float* __restrict__ gscale,
                                                       bf16_t* const* __restrict__ lp_table, float* __restrict__ sumsq,
                                                       const AdamArgs a) {
  __shared__ float red[4];
  const int t = chunk_tensor[blockIdx.x];
  const long long off = chunk_off[blockIdx.x];
  const TensorDesc d = tensors[t];
  long long n = d.n - off;
  if (n > CHUNK) n = CHUNK;
  float* p = d.p + off;
  const float* g = d.g + off;
  float* m = d.m + off;
  float* v = d.v + off;
  bf16_t* lp = LP ? lp_table[t] : nullptr;
  if (LP && lp) lp += off;
  const float gs = gscale ? *gscale : 1.f;
  float s = 0.f;
  const bool vec = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                      reinterpret_cast<uintptr_t>(v)) & 15) == 0) && (!LP || (reinterpret_cast<uintptr_t>(lp) & 7) == 0);
  long long done = 0;
  if (vec) {
    const long long n4 = n >> 2;
    for (long long i = threadIdx.x; i < n4; i += 256) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + 4 * i);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + 4 * i);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + 4 * i);
      f32x4 vv = *reinterpret_cast<f32x4*>(v + 4 * i);
      if (SQ) s += (gv[0] * gv[0] + gv[1] * gv[1]) + (gv[2] * gv[2] + gv[3] * gv[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pv[e], me = mv[e], ve = vv[e];
        adam_one(pe, gv[e] * gs, me, ve, a);
        pv[e] = pe; mv[e] = me; vv[e] = ve;
      }
      *reinterpret_cast<f32x4*>(p + 4 * i) = pv;
      *reinterpret_cast<f32x4*>(m + 4 * i) = mv;
      *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
      if (LP && lp) {
        u32x2 w = {pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])};
        *reinterpret_cast<u32x2*>(lp + 4 * i) = w;
      }
    }
    done = n4 << 2;
  }
  for (long long i = done + threadIdx.x; i < n; i += 256) {
    float pe = p[i], me = m[i], ve = v[i];
    const float ge = g[i];
    if (SQ) s += ge * ge;
    adam_one(pe, ge * gs, me, ve, a);
    p[i] = pe; m[i] = me; v[i] = ve;
    if (LP && lp) lp[i] = f2bf(pe);
  }
  if (SQ) {
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(sumsq + t, (red[0] + red[1]) + (red[2] + red[3]));
  }
}

}  // namespace octmae
using namespace octmae;

extern "C" int octmae_mt_chunk_elems(void) { return CHUNK; }

extern "C" int octmae_mt_sumsq(const void* tensor_table, const int* chunk_tensor, const long long* chunk_off, int nchunks,
                               float* sumsq, void* stream) {
  OCTMAE_CHECK_ARG(tensor_table && chunk_tensor && chunk_off && sumsq && nchunks > 0);
  hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const TensorDesc*>(tensor_table), chunk_tensor, chunk_off, sumsq);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_mt_finish_norm(const float* sumsq, int ntensors, float max_norm, float* out_norm, float* out_coef,
                                     void* stream) {
  OCTMAE_CHECK_ARG(sumsq && out_norm && out_coef && ntensors > 0);
  hipLaunchKernelGGL(mt_finish_norm_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sumsq, ntensors, max_norm,
                     out_norm, out_coef);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_mt_adamw_fused(const void* tensor_table, const int* chunk_tensor, const long long* chunk_off, int nchunks,
                                     const float* gscale, void* const* lp_table, float* sumsq, float lr, float beta1, float beta2,
                                     float eps, float weight_decay, int step, void* stream) {
  OCTMAE_CHECK_ARG(tensor_table && chunk_tensor && chunk_off && nchunks > 0 && step >= 1);
  AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  const TensorDesc* tt = reinterpret_cast<const TensorDesc*>(tensor_table);
  bf16_t* const* lpt = reinterpret_cast<bf16_t* const*>(lp_table);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (lpt && sumsq) hipLaunchKernelGGL((mt_adamw_kernel<true, true>), dim3(nchunks), dim3(256), 0, st, tt, chunk_tensor, chunk_off, gscale, lpt, sumsq, a);
  else if (lpt) hipLaunchKernelGGL((mt_adamw_kernel<true, false>), dim3(nchunks), dim3(256), 0, st, tt, chunk_tensor, chunk_off, gscale, lpt, sumsq, a);
  else if (sumsq) hipLaunchKernelGGL((mt_adamw_kernel<false, true>), dim3(nchunks), dim3(256), 0, st, tt, chunk_tensor, chunk_off, gscale, lpt, sumsq, a);
  else hipLaunchKernelGGL((mt_adamw_kernel<false, false>), dim3(nchunks), dim3(256), 0, st, tt, chunk_tensor, chunk_off, gscale, lpt, sumsq, a);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_mt_adamw(const void* tensor_table, const int* chunk_tensor, const long long* chunk_off, int nchunks,
                               const float* gscale, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               void* stream) {
  return octmae_mt_adamw_fused(tensor_table, chunk_tensor, chunk_off, nchunks, gscale, nullptr, nullptr, lr, beta1, beta2, eps,
                               weight_decay, step, stream);
}
