// Does the ORDER of the MFMAs inside a GEMM phase change what they cost?  The step is power-bound (profiles/r04_power_probe.txt), a
// phase's MFMAs are independent of each other's order, and the orders differ in what stays put between consecutive instructions:
//   s-major (shipped gemm256p phase): (s0,i0) (s0,i1) (s1,i0) (s1,i1) ...   B fragment shared by pairs, accumulators alternate
//   i-major:                          (i0: s0 s1 s2 s3) (i1: s0 s1 s2 s3)     one accumulator for 4 consecutive MFMAs, A and B change
// Register-only, random bf16 operands, every CU busy, 1 or 2 waves per SIMD.  Prints wall time and in-kernel clock.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_order.hip -o tools/ubench/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define SB __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x16 mf(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mq(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int MODE, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void k(float* out, long long* ticks, int iters) {
  const int t = threadIdx.x;
  bf16x8 A[2][4], B[4];                 // a quadrant: 2 a-tiles x 4 k-steps, 1 b-tile x 4 k-steps
  for (int i = 0; i < 2; ++i)
    for (int s = 0; s < 4; ++s)
      for (int e = 0; e < 8; ++e) {
        const unsigned h = (t * 2654435761u + (i * 4 + s) * 40503u + e * 977u + blockIdx.x * 131u);
        A[i][s][e] = (short)(0x3c00 + (h & 0x3ff) + ((h >> 11 & 1) << 15));
      }
  for (int s = 0; s < 4; ++s)
    for (int e = 0; e < 8; ++e) {
      const unsigned h = (t * 40503u + s * 2654435761u + e * 1231u + blockIdx.x * 17u);
      B[s][e] = (short)(0x3c00 + (h >> 3 & 0x3ff) + ((h >> 15 & 1) << 15));
    }
  f32x16 acc[2][4];                     // 4 quadrants per iteration (as the 4 phases of a k-tile), 2 accumulators each
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 2; ++i)
      for (int e = 0; e < 16; ++e) acc[i][q][e] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MODE == 0) {                  // s-major
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i) { acc[i][q] = mf(A[i][s], B[s], acc[i][q]); SB; }
      } else if (MODE == 1) {           // i-major
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < 4; ++s) { acc[i][q] = mf(A[i][s], B[s], acc[i][q]); SB; }
      } else {                          // nothing shared: both operands and the accumulator change every instruction
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i) { acc[i][(q + s) & 3] = mf(A[i][s], B[(s + i) & 3], acc[i][(q + s) & 3]); SB; }
      }
    }
    // keep the accumulators bounded (and the data moving): fold them back every iteration
    if ((it & 63) == 63)
      for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 2; ++i)
          for (int e = 0; e < 16; ++e) acc[i][q][e] *= 1.0f / 4096.0f;
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 2; ++i)
      for (int e = 0; e < 16; ++e) s += acc[i][q][e];
  if (s == 1234.5f) out[t] = s;
  if (t == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE, int WPS>
void run(const char* name, float* d, long long* tk, int iters) {
  hipLaunchKernelGGL((k<MODE, WPS>), dim3(256), dim3(256 * WPS), 0, 0, d, tk, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, WPS>), dim3(256), dim3(256 * WPS), 0, 0, d, tk, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[512]; (void)hipMemcpy(h, tk, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  const double flops = 256.0 * 4 * WPS * (double)iters * 32 * 2.0 * 32 * 32 * 16;
  printf("%-46s waves/SIMD %d  %8.3f ms  %7.1f TFLOP/s  cycles/MFMA %.2f  clock %.2f GHz\n", name, WPS, ms, flops / ms / 1e9,
         cyc / 256 / ((double)iters * 32), cyc / rt / 10.0);
}

int main() {
  float* d; long long* tk;
  (void)hipMalloc(&d, 8192); (void)hipMalloc(&tk, 512 * 8);
  const int iters = 40000;
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 1>("s-major (B shared by pairs; shipped)", d, tk, iters);
    run<1, 1>("i-major (accumulator stationary x4)", d, tk, iters);
    run<2, 1>("nothing shared", d, tk, iters);
    run<0, 2>("s-major (B shared by pairs; shipped)", d, tk, iters / 2);
    run<1, 2>("i-major (accumulator stationary x4)", d, tk, iters / 2);
    run<2, 2>("nothing shared", d, tk, iters / 2);
  }
  return 0;
}
