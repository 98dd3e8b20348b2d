// What do the operand reads from LDS cost a power-bound MFMA loop?  Register-only MFMAs on random data sustain ~1.75 PFLOP/s at the
// 1400 W cap (tools/ubench/mfma_order.hip); the 256-tile GEMM reads 0.75 ds_read_b128 per v_mfma_f32_32x32x16_bf16 (8 waves of
// 128 x 64: 16 A + 8 B fragment reads per 32 MFMAs), a 4-wave kernel with 128 x 128 per wave 0.5.  This loop issues R reads per 8
// MFMAs (R = 0, 2, 4, 6, 8, 12), each read landing in the fragment registers the following MFMAs consume (so the data keep changing),
// from a 64 KiB LDS image of random bf16, conflict-free addresses, 2 waves per SIMD like the GEMM.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_lds_energy.hip -o tools/ubench/mfma_lds_energy
#include <hip/hip_runtime.h>
#include <glob.h>
#include <atomic>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>
// board power while a kernel runs: every amdgpu hwmon power1_input under /sys is sampled from a host thread, the busiest card reported
struct PowerSampler {
  std::vector<std::string> files; std::vector<double> sum; std::vector<double> fsum; int n = 0; std::atomic<bool> stop{false}; std::thread th;
  PowerSampler() {
    glob_t g; if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", 0, nullptr, &g) == 0) { for (size_t i = 0; i < g.gl_pathc; ++i) files.push_back(g.gl_pathv[i]); globfree(&g); }
    sum.assign(files.size(), 0.0); fsum.assign(files.size(), 0.0);
  }
  static double rd(const std::string& f) { FILE* fp = fopen(f.c_str(), "r"); double v = 0; if (fp) { if (fscanf(fp, "%lf", &v) != 1) v = 0; fclose(fp); } return v; }
  void start() { th = std::thread([this] { while (!stop) { for (size_t i = 0; i < files.size(); ++i) { sum[i] += rd(files[i]) * 1e-6; std::string f = files[i]; f.replace(f.rfind("power1_input"), 12, "freq1_input"); fsum[i] += rd(f) * 1e-6; } ++n; std::this_thread::sleep_for(std::chrono::milliseconds(5)); } }); }
  void finish(double* watts, double* mhz) { stop = true; th.join(); *watts = 0; *mhz = 0; for (size_t i = 0; i < files.size(); ++i) if (n && sum[i] / n > *watts) { *watts = sum[i] / n; *mhz = fsum[i] / n; } }
};
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define SB __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x16 mf(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

template <int R>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* ticks, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 64 KiB
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int i = t; i < 65536 / 4; i += 512) {
    const unsigned h = (i * 2654435761u + blockIdx.x * 40503u);
    reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (h & 0x03ff03ffu) + ((h >> 3) & 0x80008000u);
  }
  __syncthreads();
  bf16x8 A[2][4], B[4];
  // conflict-free b128 pattern: lane l reads 16 B at (row l & 31) * 128 + ((chunk ^ (row >> 1 & 7)) << 4), as the GEMM's k-contiguous image
  const int r = lane & 31, hh = lane >> 5;
  auto rd = [&](int rb, int s) {
    const int row = (rb + r) & 255, c = (2 * s + hh) & 7;
    return *reinterpret_cast<const bf16x8*>(smem + ((w & 1) * 32768) + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  };
  for (int i = 0; i < 2; ++i) for (int s = 0; s < 4; ++s) A[i][s] = rd(32 * i, s);
  for (int s = 0; s < 4; ++s) B[s] = rd(64 + 32 * (w & 3), s);
  f32x16 acc[2][4];
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][q][e] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  int rot = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // R fragment reads ahead of this phase's 8 MFMAs (they replace fragments the phase is about to use: data keep changing)
      rot = (rot + 32) & 255;
#pragma unroll
      for (int n = 0; n < R; ++n) {
        if (n < 8) A[n >> 2][n & 3] = rd(rot + 32 * (n >> 2), n & 3);
        else B[n & 3] = rd(rot + 96, n & 3);
      }
      SB;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][q] = mf(A[i][s], B[s], acc[i][q]);
      SB;
    }
    if ((it & 63) == 63)
      for (int q = 0; q < 4; ++q) for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][q][e] *= 1.0f / 4096.0f;
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) s += acc[i][q][e];
  if (s == 1234.5f) out[t] = s;
  if (t == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int R>
void run(float* d, long long* tk, int iters) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<R>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k<R>, dim3(256), dim3(512), 65536, 0, d, tk, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  PowerSampler ps; ps.start();
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k<R>, dim3(256), dim3(512), 65536, 0, d, tk, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  double watts, mhz; ps.finish(&watts, &mhz);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  long long h[512]; (void)hipMemcpy(h, tk, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  const double flops = 256.0 * 8 * (double)iters * 32 * 2.0 * 32 * 32 * 16;
  printf("ds_read_b128 per MFMA %.3f   %8.3f ms  %7.1f TFLOP/s  in-kernel clock %.2f GHz   board %6.0f W  sclk %4.0f MHz   %.3f pJ/flop dynamic (board - 311 W)\n",
         R / 8.0, ms, flops / ms / 1e9, cyc / rt / 10.0, watts, mhz, (watts - 311.0) * ms * 1e-3 / flops * 1e12);
}

int main() {
  float* d; long long* tk;
  (void)hipMalloc(&d, 8192); (void)hipMalloc(&tk, 512 * 8);
  const int iters = 20000;
  for (int rep = 0; rep < 1; ++rep) {
    run<0>(d, tk, iters); run<2>(d, tk, iters); run<4>(d, tk, iters); run<6>(d, tk, iters); run<8>(d, tk, iters); run<12>(d, tk, iters);
  }
  return 0;
}
