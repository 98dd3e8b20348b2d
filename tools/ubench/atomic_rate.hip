// fp32 atomic-add throughput into an L2-resident accumulator, in the access shape a merged attention backward would have:
// (b,h) pairs x key blocks; every workgroup walks all 160 query tiles and adds one 32x32 fp32 tile (4 KiB) per tile.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/atomic_rate.hip -o tools/ubench/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void k(float* dq, int nkb, int ntiles, int mode) {
  const int bh = blockIdx.x / nkb;
  float* base = dq + (size_t)bh * ntiles * 1024;
  const int tid = threadIdx.x;
  const float v = 1.0f + tid;
  // key blocks of one (b,h) start at different tiles so that they do not hit the same lines at the same time
  const int kb = blockIdx.x % nkb;
  for (int i = 0; i < ntiles; ++i) {
    int t = i + kb * (ntiles / nkb); if (t >= ntiles) t -= ntiles;
    float* p = base + (size_t)t * 1024;
    if (mode == 0) {            // 512 threads x 2 dwords
      unsafeAtomicAdd(p + tid, v);
      unsafeAtomicAdd(p + 512 + tid, v);
    } else if (mode == 1) {     // waves 0-3 only, 4 dwords per lane (lane-contiguous 16 B)
      if (tid < 256) { for (int e = 0; e < 4; ++e) unsafeAtomicAdd(p + tid * 4 + e, v); }
    } else {                    // plain stores for reference
      p[tid] = v; p[512 + tid] = v;
    }
  }
}
int main(int argc, char** argv) {
  const int BH = 1024, ntiles = 160;
  float* dq; hipMalloc(&dq, (size_t)BH * ntiles * 1024 * 4);
  hipMemset(dq, 0, (size_t)BH * ntiles * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nkb : {10, 20, 40}) for (int mode = 0; mode < 3; ++mode) {
    k<<<BH * nkb, 512>>>(dq, nkb, ntiles, mode);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) k<<<BH * nkb, 512>>>(dq, nkb, ntiles, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)BH * nkb * ntiles * 4096;
    printf("key blocks %2d mode %d: %.3f ms  %.1f GB of adds  %.2f TB/s\n", nkb, mode, ms, bytes / 1e9, bytes / ms / 1e9);
  }
  return 0;
}
