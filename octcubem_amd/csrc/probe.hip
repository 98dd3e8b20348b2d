// Hardware layout probes: tests use them to pin the lane maps of v_mfma_f32_32x32x16_bf16 and
// ds_read_b64_tr_b16 that gemm.hip / attn.hip are written against.
#include "common.hpp"
#include "../../include/octmae.h"

namespace octmae {
// a_frag / b_frag: [64 lanes][8] bf16; d_regs: [64 lanes][16] f32
__global__ void probe_mfma32_kernel(const bf16_t* a, const bf16_t* b, float* d) {
  const int l = threadIdx.x;
  const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a + 8 * l);
  const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b + 8 * l);
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[g] = 0.f;
  acc = mfma32(fa, fb, acc);
#pragma unroll
  for (int g = 0; g < 16; ++g) d[16 * l + g] = acc[g];
}
// tile: 16 rows x 16 cols bf16 row-major (32-byte rows) copied to LDS; every 16-lane group reads rows 4g'..4g'+3
// where g' = group index, lane 4q+p supplying &tile[4g' + q][4p]; out: [64 lanes][4] bf16
__global__ void probe_trread_kernel(const bf16_t* tile, bf16_t* out) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = tile[i];
  __syncthreads();
  const int grp = l >> 4, q = (l >> 2) & 3, p = l & 3;
  const bf16x4 v = lds_tr_read(reinterpret_cast<const char*>(lds) + ((4 * grp + q) * 16 + 4 * p) * 2);
  *reinterpret_cast<bf16x4*>(out + 4 * l) = v;
}
}  // namespace octmae
using namespace octmae;

extern "C" int octmae_abi_version(void) { return OCTMAE_ABI_VERSION; }   // the number lives in include/octmae.h
extern "C" int octmae_lp_dtype(void) { return OCTMAE_LP_IS_F16; }

extern "C" int octmae_probe_mfma32(const void* a, const void* b, float* d, void* stream) {
  OCTMAE_CHECK_ARG(a && b && d);
  hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(a), reinterpret_cast<const bf16_t*>(b), d);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}
extern "C" int octmae_probe_trread(const void* tile, void* out, void* stream) {
  OCTMAE_CHECK_ARG(tile && out);
  hipLaunchKernelGGL(probe_trread_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(tile), reinterpret_cast<bf16_t*>(out));
  OCTMAE_LAUNCH_CHECK();
  return 0;
}
