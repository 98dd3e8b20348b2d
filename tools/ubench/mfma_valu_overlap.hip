// Do an MFMA-only wave and a VALU-only wave that share ONE SIMD overlap?  512-thread workgroups, one per CU: waves 0-3 and
// 4-7 are co-resident pairwise on the 4 SIMDs.  mode 0: all waves MFMA; 1: all waves v_exp; 2: waves 0-3 MFMA, 4-7 v_exp;
// 3: every wave interleaves 1 MFMA + 3 v_exp (same totals as mode 2 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode, int vmul) {
  const long long t_begin = __builtin_readcyclecounter();
  const int wid = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 7); b[i] = (short)(0x3f00 + threadIdx.x % 5); }
  f32x16 c0 = {0}, c1 = {0};
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + .5f, x5 = x0 + .25f;
  if (wid >= 4 && (mode == 2 || mode == 5 || mode == 7 || mode == 9)) iters *= vmul;
  if (wid < 4 && (mode == 10 || mode == 11)) iters *= vmul;
  const bool do_mfma = (mode == 0) || ((mode == 2 || mode == 5 || mode == 7 || mode == 9) && wid < 4) || ((mode == 10 || mode == 11) && wid >= 4);
  if (mode == 11 && wid >= 4) __builtin_amdgcn_s_setprio(3);
  const bool do_exp = (mode == 1) || (mode == 2 && wid >= 4);
  const bool do_fma = (mode == 4) || (mode == 5 && wid >= 4);
  const bool do_cvt = (mode == 6) || (mode == 7 && wid >= 4);
  const bool do_mix = (mode == 8) || (mode == 9 && wid >= 4) || ((mode == 10 || mode == 11) && wid < 4);
  if (mode == 3) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2" : "+v"(x0), "+v"(x1), "+v"(x2));
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2" : "+v"(x3), "+v"(x4), "+v"(x5));
      }
    }
  } else if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      }
    }
  } else if (do_exp) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5));
    }
  }
  else if (do_fma) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5));
    }
  } else if (do_cvt) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %0"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5));
    }
  } else if (do_mix) {   // softmax-like: 2 exp, 1 cvt_pk, 2 add, 1 mov per pair
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_cvt_pk_bf16_f32 %2, %0, %1\n v_add_f32 %3, %3, %0\n v_add_f32 %4, %4, %1\n v_mov_b32 %5, %2"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5));
    }
  }
  float s = x0 + x1 + x2 + x3 + x4 + x5;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  if (s == 1234.5f) out[0] = s;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) ((long long*)out)[1 + threadIdx.x / 256] = __builtin_readcyclecounter() - t_begin;
}
int main() {
  float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  const int iters = 2000;
  const char* names[12] = {"all waves MFMA (32 per iter per wave)", "all waves v_exp (96 per iter per wave)",
                          "waves 0-3 MFMA, 4-7 v_exp (one of each per SIMD)", "every wave: 16 MFMA + 48 v_exp interleaved",
    "all waves v_fma (96/iter)", "waves 0-3 MFMA, 4-7 v_fma", "all waves v_cvt_pk (96/iter)", "waves 0-3 MFMA, 4-7 v_cvt_pk", "all waves softmax-mix (96/iter)", "waves 0-3 MFMA, 4-7 softmax-mix",
    "waves 0-3 softmax-mix (OLDER), 4-7 MFMA", "same, MFMA waves at s_setprio 3"};
  for (int vmul = 1; vmul <= 4; vmul *= 2)
  for (int mode = 0; mode < 12; ++mode) {
    if (vmul > 1 && !(mode == 9 || mode == 10 || mode == 11)) continue;
    if (vmul == 1 && mode < 8) continue;
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode, vmul);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode, vmul);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long ticks[3]; hipMemcpy(ticks, d, 24, hipMemcpyDeviceToHost);
    printf("vmul %d mode %d %-52s %.3f ms   ticks wave0 %lld  wave4 %lld\n", vmul, mode, names[mode], ms, ticks[1], ticks[2]);
  }
  return 0;
}
