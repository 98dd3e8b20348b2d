// random_masking index generation (reference K2/K3: models_mae_joint_res_flash_attn.py:336-372).
//   ids_shuffle = argsort(noise) (ties -> lower index), ids_restore = argsort(ids_shuffle) (= inverse permutation),
//   ids_keep = ids_shuffle[:, :len_keep], mask[b][l] = (ids_restore[b][l] >= len_keep).
// Integer work, bit-exact by construction: one workgroup sorts one row in LDS as 64-bit composite keys
// (order-preserving transform of the fp32 bits << 32 | index); the keys are unique, so a bitonic network
// yields exactly the stable order.  L <= 16384 (128 KiB of LDS); the path's L is 5120.
#include "common.hpp"
#include "../../include/octmae.h"

namespace octmae {

__device__ __forceinline__ uint32_t sortable_bits(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if (f != f) u = 0x7fc00000u;        // every NaN sorts last, like torch.sort
  if (u == 0x80000000u) u = 0u;       // -0.0 == +0.0 (a tie, resolved by index)
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void mask_sort_kernel(const float* __restrict__ noise, long long* __restrict__ ids_restore,
                                                         long long* __restrict__ ids_keep, long long* __restrict__ ids_shuffle,
                                                         float* __restrict__ mask, int L, int npad, int len_keep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = noise + (size_t)b * L;
  for (int i = tid; i < npad; i += 1024)
    keys[i] = (i < L) ? (((unsigned long long)sortable_bits(row[i]) << 32) | (unsigned)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (npad >> 1); t += 1024) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const unsigned long long a = keys[lo], c = keys[hi];
        if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < L; i += 1024) {
    const int src = (int)(keys[i] & 0xffffffffu);
    if (ids_shuffle) ids_shuffle[(size_t)b * L + i] = src;
    ids_restore[(size_t)b * L + src] = i;
    mask[(size_t)b * L + src] = (i < len_keep) ? 0.f : 1.f;
    if (i < len_keep) ids_keep[(size_t)b * len_keep + i] = src;
  }
}

}  // namespace octmae
using namespace octmae;

extern "C" int octmae_random_masking_ids(const float* noise, long long* ids_restore, long long* ids_keep,
                                         long long* ids_shuffle, float* mask, int B, int L, int len_keep, void* stream) {
  OCTMAE_CHECK_ARG(noise && ids_restore && mask && (ids_keep || len_keep == 0));
  OCTMAE_CHECK_ARG(B > 0 && L > 0 && L <= 16384 && len_keep >= 0 && len_keep <= L);
  int npad = 2;
  while (npad < L) npad <<= 1;
  const size_t lds = (size_t)npad * 8;
  static DynLdsOnce once;
  if (int rc = once.ensure(reinterpret_cast<const void*>(mask_sort_kernel), 16384 * 8)) return rc;
  hipLaunchKernelGGL(mask_sort_kernel, dim3(B), dim3(1024), lds, reinterpret_cast<hipStream_t>(stream), noise, ids_restore,
                     ids_keep, ids_shuffle, mask, L, npad, len_keep);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}
