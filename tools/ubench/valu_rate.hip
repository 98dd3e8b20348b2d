// VALU issue-rate micro-benchmark for gfx950: cycles per wave-instruction for a few opcodes at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
template <int OP>
__global__ void k(float* out, int iters, float a0) {
  float x0 = a0 + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (OP == 0) {  // v_fma_f32 x8 independent
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                     "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 1) {  // v_pk_fma_f32 x4 (8 elements) then again x4
        asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                     "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      } else if (OP == 2) {  // v_exp_f32 x8
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 3) {  // v_max3_f32 x8
        asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n"
                     "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 4) {  // v_cvt_pk_bf16_f32 x8
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                     "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 5) {  // v_add_f32 dependent chain x8
        asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4\n"
                     "v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %6\n v_add_f32 %0, %0, %7\n v_add_f32 %0, %0, %1\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 7) {  // v_pk_fma_f16 x8
        asm volatile("v_pk_fma_f16 %0, %0, %0, %0\n v_pk_fma_f16 %1, %1, %1, %1\n v_pk_fma_f16 %2, %2, %2, %2\n v_pk_fma_f16 %3, %3, %3, %3\n"
                     "v_pk_fma_f16 %4, %4, %4, %4\n v_pk_fma_f16 %5, %5, %5, %5\n v_pk_fma_f16 %6, %6, %6, %6\n v_pk_fma_f16 %7, %7, %7, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 8) {  // v_pk_add_u16 x8
        asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %4\n"
                     "v_pk_add_u16 %4, %4, %5\n v_pk_add_u16 %5, %5, %6\n v_pk_add_u16 %6, %6, %7\n v_pk_add_u16 %7, %7, %0\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 9) {  // v_pk_lshlrev_b16 x8
        asm volatile("v_pk_lshlrev_b16 %0, 3, %0\n v_pk_lshlrev_b16 %1, 3, %1\n v_pk_lshlrev_b16 %2, 3, %2\n v_pk_lshlrev_b16 %3, 3, %3\n"
                     "v_pk_lshlrev_b16 %4, 3, %4\n v_pk_lshlrev_b16 %5, 3, %5\n v_pk_lshlrev_b16 %6, 3, %6\n v_pk_lshlrev_b16 %7, 3, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 10) {  // v_cvt_pkrtz_f16_f32 x8
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %1, %1, %2\n v_cvt_pkrtz_f16_f32 %2, %2, %3\n v_cvt_pkrtz_f16_f32 %3, %3, %4\n"
                     "v_cvt_pkrtz_f16_f32 %4, %4, %5\n v_cvt_pkrtz_f16_f32 %5, %5, %6\n v_cvt_pkrtz_f16_f32 %6, %6, %7\n v_cvt_pkrtz_f16_f32 %7, %7, %0\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 11) {  // v_exp_f16 x8
        asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n"
                     "v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 12) {  // v_ldexp_f32 x8
        asm volatile("v_ldexp_f32 %0, %0, %1\n v_ldexp_f32 %1, %1, %2\n v_ldexp_f32 %2, %2, %3\n v_ldexp_f32 %3, %3, %4\n"
                     "v_ldexp_f32 %4, %4, %5\n v_ldexp_f32 %5, %5, %6\n v_ldexp_f32 %6, %6, %7\n v_ldexp_f32 %7, %7, %0\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 13) {  // v_perm_b32 x8
        asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %4, %5\n"
                     "v_perm_b32 %4, %4, %5, %6\n v_perm_b32 %5, %5, %6, %7\n v_perm_b32 %6, %6, %7, %0\n v_perm_b32 %7, %7, %0, %1\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (OP == 6) {  // v_pk_mul_f32 x8
        asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0\n"
                     "v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      }
    }
  }
  long long t1 = clock64();
  float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}
template <int OP>
void run(const char* name) {
  float* d; hipMalloc(&d, 64);
  const int iters = 2000;
  for (int waves : {1, 2, 4}) {   // waves per SIMD: block of 256*waves/... use blocks of 256 threads (1 wave per SIMD each), `waves` blocks per CU
    float h[2];
    hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    double insts = (double)iters * REP;          // per wave
    printf("%-22s waves/SIMD %d: %.2f clk/inst/wave (s_memtime ticks), wall %.3f ms -> %.2f ns per inst per SIMD-slot\n", name, waves,
           h[1] / insts, ms, ms * 1e6 / (insts * waves));
  }
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("v_exp_f32"); run<3>("v_max3_f32"); run<4>("v_cvt_pk_bf16_f32");
  run<5>("v_add_f32 (dep chain)"); run<6>("v_pk_mul_f32");
  run<7>("v_pk_fma_f16"); run<8>("v_pk_add_u16"); run<9>("v_pk_lshlrev_b16"); run<10>("v_cvt_pkrtz_f16_f32"); run<11>("v_exp_f16");
  run<12>("v_ldexp_f32"); run<13>("v_perm_b32");
  return 0;
}
