// What does the MFMA SHAPE cost the fused attention backward at head_dim 32, one wave per SIMD?  
// One 32 x 32 score block = S, dP (VALU reads the results), exp2 / multiply / two bf16 packs per score, dV^T, dK^T (accumulators
// only MFMAs touch), dQ^T (16x16x32).  Mode 0: S, dP, dV^T, dK^T on v_mfma_f32_32x32x16_bf16 (the shipped kernel: 12 MFMA
// instructions per block, 4 of them with 16 VGPR results); mode 1: all on v_mfma_f32_16x16x32_bf16 (20 instructions per block,
// 8 of them with 4 VGPR results).  Same VALU work (16 v_exp_f32, 16 v_mul_f32, 16 v_cvt_pk_bf16_f32 per lane and block), same
// software pipeline (S / dP of block i + 1 issued beside the VALU work of block i), fillers dealt evenly to the MFMA slots, no LDS
// and no memory traffic: the instruction-issue side alone.  Prints shader cycles per block (s_memtime) and wall time with every
// CU busy (256 workgroups x 256 threads, launch_bounds(256, 1) => one wave per SIMD), on non-trivial data.
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize tools/ubench/bwd_block_shapes.hip -o tools/ubench/bwd_block_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define SB __builtin_amdgcn_sched_barrier(0)
// (operands made opaque once per block: otherwise the loop-invariant S / dP products are hoisted out of the loop)
#define OPQ(x) asm volatile("" : "+v"(x))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define MUL(d, s) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(d) : "v"(s))
#define CVT(d, a, b) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define ACC32(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define ACC16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// ---- mode 0: one block with 32x32x16.  cur = (s, dp) of this block (already computed), nxt = the next block's, produced here.
#define BLOCK32(S, DP, SN, DPN)                                                                                            \
  do {                                                                                                                     \
    OPQ(q0); OPQ(o0); OPQ(q1); OPQ(o1);                                                                                    \
    SN = mfma32(q0, k0, lse); SB; EXP(S[0]); EXP(S[1]); EXP(S[2]); EXP(S[3]); SB;                                          \
    DPN = mfma32(o0, v0, dlt); SB; EXP(S[4]); EXP(S[5]); EXP(S[6]); EXP(S[7]); SB;                                         \
    SN = mfma32(q1, k1, SN); SB; MUL(DP[0], S[0]); MUL(DP[1], S[1]); MUL(DP[2], S[2]); MUL(DP[3], S[3]); SB;               \
    DPN = mfma32(o1, v1, DPN); SB; MUL(DP[4], S[4]); MUL(DP[5], S[5]); MUL(DP[6], S[6]); MUL(DP[7], S[7]); SB;             \
    /* bundle 5..8: dQ^T (16x16x32) */                                                                                     \
    dq = mfma16(kt0, __builtin_bit_cast(bf16x8, pfp), dq); SB; CVT(pf[0], S[0], S[1]); CVT(pf[1], S[2], S[3]); CVT(pf[2], S[4], S[5]); CVT(pf[3], S[6], S[7]); SB; \
    dq = mfma16(kt1, __builtin_bit_cast(bf16x8, dsp), dq); SB; CVT(ds[0], DP[0], DP[1]); CVT(ds[1], DP[2], DP[3]); CVT(ds[2], DP[4], DP[5]); CVT(ds[3], DP[6], DP[7]); SB; \
    dq = mfma16(kt0, __builtin_bit_cast(bf16x8, dsp), dq); SB; EXP(S[8]); EXP(S[9]); EXP(S[10]); EXP(S[11]); SB;           \
    dq = mfma16(kt1, __builtin_bit_cast(bf16x8, pfp), dq); SB; EXP(S[12]); EXP(S[13]); EXP(S[14]); EXP(S[15]); SB;         \
    /* bundle 9..12: dV^T, dK^T of the first and second 16 queries */                                                      \
    ACC32(dv, o0, pf); SB; MUL(DP[8], S[8]); MUL(DP[9], S[9]); MUL(DP[10], S[10]); MUL(DP[11], S[11]); SB;                 \
    ACC32(dk, q0, ds); SB; MUL(DP[12], S[12]); MUL(DP[13], S[13]); MUL(DP[14], S[14]); MUL(DP[15], S[15]); SB;             \
    ACC32(dv, o1, pfp); SB; CVT(pfp[0], S[8], S[9]); CVT(pfp[1], S[10], S[11]); CVT(pfp[2], S[12], S[13]); CVT(pfp[3], S[14], S[15]); SB; \
    ACC32(dk, q1, dsp); SB; CVT(dsp[0], DP[8], DP[9]); CVT(dsp[1], DP[10], DP[11]); CVT(dsp[2], DP[12], DP[13]); CVT(dsp[3], DP[14], DP[15]); SB; \
  } while (0)

// ---- mode 1: the same block with 16x16x32: S, dP are four 16 x 16 tiles each (one k-step over the whole head dimension)
#define BLOCK16(S, DP, SN, DPN)                                                                                            \
  do {                                                                                                                     \
    OPQ(q0); OPQ(o0); OPQ(q1); OPQ(o1);                                                                                    \
    SN[0] = mfma16(q0, k0, l4); SB; EXP(S[0][0]); EXP(S[0][1]); SB;                                                        \
    SN[1] = mfma16(q1, k0, l4); SB; EXP(S[0][2]); EXP(S[0][3]); EXP(S[1][0]); SB;                                          \
    SN[2] = mfma16(q0, k1, l4); SB; EXP(S[1][1]); EXP(S[1][2]); SB;                                                        \
    SN[3] = mfma16(q1, k1, l4); SB; EXP(S[1][3]); EXP(S[2][0]); EXP(S[2][1]); SB;                                          \
    DPN[0] = mfma16(o0, v0, d4); SB; EXP(S[2][2]); EXP(S[2][3]); SB;                                                       \
    DPN[1] = mfma16(o1, v0, d4); SB; EXP(S[3][0]); EXP(S[3][1]); EXP(S[3][2]); SB;                                         \
    DPN[2] = mfma16(o0, v1, d4); SB; EXP(S[3][3]); MUL(DP[0][0], S[0][0]); MUL(DP[0][1], S[0][1]); SB;                     \
    DPN[3] = mfma16(o1, v1, d4); SB; MUL(DP[0][2], S[0][2]); MUL(DP[0][3], S[0][3]); SB;                                   \
    dq = mfma16(kt0, __builtin_bit_cast(bf16x8, pfp), dq); SB; MUL(DP[1][0], S[1][0]); MUL(DP[1][1], S[1][1]); MUL(DP[1][2], S[1][2]); SB; \
    dq = mfma16(kt1, __builtin_bit_cast(bf16x8, dsp), dq); SB; MUL(DP[1][3], S[1][3]); MUL(DP[2][0], S[2][0]); SB;         \
    dq = mfma16(kt0, __builtin_bit_cast(bf16x8, dsp), dq); SB; MUL(DP[2][1], S[2][1]); MUL(DP[2][2], S[2][2]); MUL(DP[2][3], S[2][3]); SB; \
    dq = mfma16(kt1, __builtin_bit_cast(bf16x8, pfp), dq); SB; MUL(DP[3][0], S[3][0]); MUL(DP[3][1], S[3][1]); SB;         \
    ACC16(dv0, o0, pfp); SB; MUL(DP[3][2], S[3][2]); MUL(DP[3][3], S[3][3]); CVT(pf[0], S[0][0], S[0][1]); SB;             \
    ACC16(dv1, o1, pfp); SB; CVT(pf[1], S[0][2], S[0][3]); CVT(pf[2], S[1][0], S[1][1]); SB;                               \
    ACC16(dk0, q0, dsp); SB; CVT(pf[3], S[1][2], S[1][3]); CVT(ds[0], DP[0][0], DP[0][1]); CVT(ds[1], DP[0][2], DP[0][3]); SB; \
    ACC16(dk1, q1, dsp); SB; CVT(ds[2], DP[1][0], DP[1][1]); CVT(ds[3], DP[1][2], DP[1][3]); SB;                           \
    ACC16(dv2, o0, pf); SB; CVT(pfp[0], S[2][0], S[2][1]); CVT(pfp[1], S[2][2], S[2][3]); CVT(pfp[2], S[3][0], S[3][1]); SB; \
    ACC16(dv3, o1, pf); SB; CVT(pfp[3], S[3][2], S[3][3]); CVT(dsp[0], DP[2][0], DP[2][1]); SB;                            \
    ACC16(dk2, q0, ds); SB; CVT(dsp[1], DP[2][2], DP[2][3]); CVT(dsp[2], DP[3][0], DP[3][1]); CVT(dsp[3], DP[3][2], DP[3][3]); SB; \
    ACC16(dk3, q1, ds); SB;                                                                                                \
  } while (0)

__global__ __launch_bounds__(256, 1) void k(float* out, long long* ticks, int iters, int mode) {
  const int t = threadIdx.x;
  bf16x8 q0, q1, o0, o1, k0, k1, v0, v1, kt0, kt1;
  for (int i = 0; i < 8; ++i) {
    const unsigned h = (t * 2654435761u + i * 40503u + blockIdx.x * 97u);
    q0[i] = (short)(0x3e00 + (h & 0x1ff) + ((h >> 9 & 1) << 15)); q1[i] = (short)(0x3e00 + (h >> 3 & 0x1ff) + ((h >> 13 & 1) << 15));
    o0[i] = (short)(0x3d80 + (h >> 5 & 0x1ff) + ((h >> 15 & 1) << 15)); o1[i] = (short)(0x3d80 + (h >> 7 & 0x1ff) + ((h >> 17 & 1) << 15));
    k0[i] = (short)(0x3e00 + (h >> 2 & 0x1ff) + ((h >> 11 & 1) << 15)); k1[i] = (short)(0x3e00 + (h >> 4 & 0x1ff) + ((h >> 19 & 1) << 15));
    v0[i] = (short)(0x3d80 + (h >> 6 & 0x1ff) + ((h >> 21 & 1) << 15)); v1[i] = (short)(0x3d80 + (h >> 8 & 0x1ff) + ((h >> 23 & 1) << 15));
    kt0[i] = (short)(0x3e00 + (h >> 1 & 0x1ff) + ((h >> 25 & 1) << 15)); kt1[i] = (short)(0x3e00 + (h >> 10 & 0x1ff) + ((h >> 27 & 1) << 15));
  }
  u32x4 pf = {0x3e003e80u, 0x3d803e00u, 0x3e803d00u, 0x3e003e00u}, ds = pf, pfp = pf, dsp = pf;
  f32x4 dq = {0.f, 0.f, 0.f, 0.f};
  long long t0 = 0, t1 = 0;
  if (mode == 0) {
    f32x16 lse, dlt, sa, da, sb, db, dv, dk;
    for (int e = 0; e < 16; ++e) { lse[e] = -3.f - 0.01f * ((t + e) & 15); dlt[e] = 0.01f * ((t + 3 * e) & 7); dv[e] = 0.f; dk[e] = 0.f; }
    sa = mfma32(q0, k0, lse); da = mfma32(o0, v0, dlt); sb = sa; db = da;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      BLOCK32(sa, da, sb, db);
      BLOCK32(sb, db, sa, da);
    }
    t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dv), "+a"(dk));
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += dv[e] + dk[e] + sa[e] + da[e];
    if (s == 1234.5f) out[t] = s + dq[0];
  } else {
    f32x4 l4, d4, sa[4], da[4], sb[4], db[4], dv0, dv1, dv2, dv3, dk0, dk1, dk2, dk3;
    for (int e = 0; e < 4; ++e) { l4[e] = -3.f - 0.01f * ((t + e) & 15); d4[e] = 0.01f * ((t + 3 * e) & 7); }
    dv0 = dv1 = dv2 = dv3 = dk0 = dk1 = dk2 = dk3 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 4; ++j) { sa[j] = mfma16(q0, k0, l4); da[j] = mfma16(o0, v0, d4); sb[j] = sa[j]; db[j] = da[j]; }
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      BLOCK16(sa, da, sb, db);
      BLOCK16(sb, db, sa, da);
    }
    t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dv0), "+a"(dv1), "+a"(dv2), "+a"(dv3), "+a"(dk0), "+a"(dk1), "+a"(dk2), "+a"(dk3));
    float s = 0.f;
    for (int e = 0; e < 4; ++e) s += dv0[e] + dv1[e] + dv2[e] + dv3[e] + dk0[e] + dk1[e] + dk2[e] + dk3[e] + sa[0][e] + da[3][e];
    if (s == 1234.5f) out[t] = s + dq[0];
  }
  if ((t & 63) == 0) ticks[blockIdx.x * 4 + (t >> 6)] = t1 - t0;
}

int main() {
  float* d; long long* tk;
  hipMalloc(&d, 4096); hipMalloc(&tk, 256 * 4 * 8);
  const int iters = 20000;
  const char* names[2] = {"32x32x16 (S, dP, dV, dK) + 16x16x32 (dQ): 12 MFMA / block", "16x16x32 everywhere: 20 MFMA / block"};
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, tk, iters, mode);
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, tk, iters, mode);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[1024]; hipMemcpy(h, tk, sizeof(h), hipMemcpyDeviceToHost);
      double sum = 0; for (int i = 0; i < 1024; ++i) sum += (double)h[i];
      const double cyc = sum / 1024 / (2.0 * iters);
      printf("mode %d %-62s %8.3f ms  %7.1f cycles / block  (MFMA pipe 320)  clock %.2f GHz\n", mode, names[mode], ms, cyc,
             cyc * 2.0 * iters / (ms * 1e6));
    }
  return 0;
}
