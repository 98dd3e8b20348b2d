// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of the 3-D MAE hot path.
// wave = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace octmae {

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) short bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- 16-bit operand type <-> f32 -------------------------------------------------------------
// The 16-bit MFMA operand type is bfloat16 (the shipped library).  `make F16=1` (-DOCTMAE_F16) builds the SAME kernels on IEEE half:
// every conversion and every MFMA of the library goes through the helpers below (and MFMA_ACC of attn_bwd1w.hpp), so the switch
// is this block.  That build is the verification build of DESIGN.md section 2 (the reference's own default arithmetic is fp16 +
// GradScaler, main_pretrain_oph_joint_2d512_flash_attn.py:259-263): 3 more mantissa bits, same kernels, same rounding points.
// The names keep "bf" in both builds: bf16_t = the raw bits of the 16-bit operand type.
#ifdef OCTMAE_F16
#define OCTMAE_LP_IS_F16 1
#define OCTMAE_LP_ONE_PAIR 0x3c003c00u       // two 1.0 of the operand type in one dword
typedef __attribute__((ext_vector_type(8))) _Float16 lp8_t;
__device__ __forceinline__ bf16_t f2bf(float x) {
  _Float16 h = (_Float16)x;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 v = {lo, hi};
  h2 r = __builtin_convertvector(v, h2);      // round-to-nearest-even (v_cvt_pk_f16_f32 / two v_cvt_f16_f32), never the rtz pack
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bflo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float bfhi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
#else
#define OCTMAE_LP_IS_F16 0
#define OCTMAE_LP_ONE_PAIR 0x3f803f80u
// Plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN).
__device__ __forceinline__ bf16_t f2bf(float x) {
  __bf16 h = (__bf16)x;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
  return __builtin_bit_cast(float, ((uint32_t)h) << 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 v = {lo, hi};
  bf2 r = __builtin_convertvector(v, bf2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bflo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
#endif

// ---- wave reductions (64 lanes) -------------------------------------------------------------
// Wave-wide reductions on DPP (every lane of the wave must be active: all call sites sit in wave-uniform control flow).  hipcc lowers
// __shfl_xor to ds_bpermute -- an LDS round trip per step, six dependent ones per reduction (two reductions per LayerNorm row) -- where
// the 16-lane row steps are one VALU instruction each: quad_perm [1,0,3,2] and [2,3,0,1] leave every lane of a quad with the quad's
// value; row_half_mirror (lane i <-> 7 - i of an 8-lane half) then adds the other quad's, row_mirror (i <-> 15 - i) the other half's;
// the four row results are combined from four v_readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) { return x + dpp_mov<CTRL>(x); }
__device__ __forceinline__ float lane_bcast(float x, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
#ifdef OCTMAE_WAVE_SUM_SHFL      // the __shfl_xor form of rounds 1-3 (A/B builds only)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#else
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
#endif
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// ---- GELU (timm Mlp act_layer=nn.GELU, exact-erf form x * Phi(x)) ---------------------------------------------
// Phi(x) = 0.5 + u * P(u^2), u = clamp(x, +-4.2): odd minimax polynomial of degree 19, |Phi error| <= 1.4e-5 as evaluated (fp32
// Horner); the bf16-rounded activation differs from the erf form's by 2-4e-5 relative L2 on N(0, 0.5 .. 4) pre-activations
// (the degree-15 fit used until round 3, |Phi error| 7.8e-5, cost 7e-4 there: the largest term of the "polynomial vs erf" ledger).  No transcendental: the epilogue of the fc1 GEMM is VALU
// work that the MFMA pipe cannot hide at one workgroup per CU (the A&S erf form with v_rcp + v_exp cost 2.4x more).
// The backward evaluates gelu'(x) = Phi(x) + x phi(x) from the erf form itself (dgelu_f below, |error| <= 3e-7): its epilogue
// waits for HBM (it streams the saved pre-activation), so two transcendentals per element are hidden there.
__device__ __forceinline__ float gelu_phi(float x) {
  const float u = __builtin_amdgcn_fmed3f(x, -4.2f, 4.2f);
  const float t = u * u;
  float p = -2.306640613e-12f;
  p = fmaf(p, t, 2.495095069e-10f);
  p = fmaf(p, t, -1.216585654e-08f);
  p = fmaf(p, t, 3.568801260e-07f);
  p = fmaf(p, t, -7.116507187e-06f);
  p = fmaf(p, t, 1.035454878e-04f);
  p = fmaf(p, t, -1.148414365e-03f);
  p = fmaf(p, t, 9.898752642e-03f);
  p = fmaf(p, t, -6.641823237e-02f);
  p = fmaf(p, t, 3.989180135e-01f);
  return fmaf(u, p, 0.5f);
}
__device__ __forceinline__ float gelu_f(float x) { return x * gelu_phi(x); }
// gelu'(x) = Phi(x) + x phi(x), nn.GELU's exact (erf) derivative (models_mae_joint_res_flash_attn.py:141 act_layer under
// autograd).  With a = |x|: Phi(a) = 1 - phi(a) (b1 t + ... + b5 t^5), t = 1 / (1 + p a) (Abramowitz & Stegun 26.2.17,
// |error| < 7.5e-8), so gelu'(a) = 1 - g with g = phi(a) (poly(t) - a), and gelu'(-a) = g.  1 / sqrt(2 pi) is folded into the
// coefficients; phi through the hardware exp2.  Measured against float64 over [-12, 12]: |error| <= 3e-7 (tests/test_gpu_kernels.py).
// (Until round 3 this was an odd degree-19 polynomial, 0.5 + u Q(u^2): its fp32 Horner evaluation cancels, |error| 4.4e-4,
// which cost up to 3.3e-3 on q / k weight gradients -- kept as dgelu_poly_f for the A/B, build with -DOCTMAE_DGELU_POLY.)
__device__ __forceinline__ float dgelu_exact_f(float x) {
  const float a = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.2316419f, a, 1.0f));
  const float e = __builtin_amdgcn_exp2f(a * a * -0.72134752044448170f);        // exp(-a^2 / 2)
  float w = 0.53070271f;                                                          // b5 / sqrt(2 pi)
  w = fmaf(w, t, -0.72657601f);
  w = fmaf(w, t, 0.71070687f);
  w = fmaf(w, t, -0.14224837f);
  w = fmaf(w, t, 0.12741479f);
  w = w * t;
  const float g = e * fmaf(-0.39894228040143268f, a, w);
  return 0.5f + __builtin_copysignf(0.5f - g, x);
}
__device__ __forceinline__ float dgelu_poly_f(float x) {
  const float u = __builtin_amdgcn_fmed3f(x, -5.0f, 5.0f);
  const float t = u * u;
  float q = -8.945184002e-12f;
  q = fmaf(q, t, 1.221804868e-09f);
  q = fmaf(q, t, -7.286091231e-08f);
  q = fmaf(q, t, 2.499930865e-06f);
  q = fmaf(q, t, -5.482182127e-05f);
  q = fmaf(q, t, 8.080908045e-04f);
  q = fmaf(q, t, -8.191250186e-03f);
  q = fmaf(q, t, 5.702680522e-02f);
  q = fmaf(q, t, -2.631234724e-01f);
  q = fmaf(q, t, 7.970332990e-01f);
  return fmaf(u, q, 0.5f);
}

// Two-at-a-time forms on v_pk_fma_f32 / v_pk_mul_f32.  A packed op costs the same 4 cycles as two scalar ones, but the GEMM
// epilogue runs with one or two waves per SIMD, where the limit is the ISSUE rate (one VALU instruction per ~5 cycles from one
// wave, ~2.6 from two: tools/ubench/valu_rate.hip) -- half the instructions is up to twice the elements per cycle.
// Same coefficients and evaluation order per element as gelu_f / dgelu_f: results are bit-identical.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float c) { return f32x2{c, c}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 gelu_f2(f32x2 x) {
  const f32x2 u = {__builtin_amdgcn_fmed3f(x[0], -4.2f, 4.2f), __builtin_amdgcn_fmed3f(x[1], -4.2f, 4.2f)};
  const f32x2 t = u * u;
  f32x2 p = splat2(-2.306640613e-12f);
  p = pk_fma(p, t, splat2(2.495095069e-10f));
  p = pk_fma(p, t, splat2(-1.216585654e-08f));
  p = pk_fma(p, t, splat2(3.568801260e-07f));
  p = pk_fma(p, t, splat2(-7.116507187e-06f));
  p = pk_fma(p, t, splat2(1.035454878e-04f));
  p = pk_fma(p, t, splat2(-1.148414365e-03f));
  p = pk_fma(p, t, splat2(9.898752642e-03f));
  p = pk_fma(p, t, splat2(-6.641823237e-02f));
  p = pk_fma(p, t, splat2(3.989180135e-01f));
  return x * pk_fma(u, p, splat2(0.5f));
}
__device__ __forceinline__ f32x2 dgelu_poly_f2(f32x2 x) {
  const f32x2 u = {__builtin_amdgcn_fmed3f(x[0], -5.0f, 5.0f), __builtin_amdgcn_fmed3f(x[1], -5.0f, 5.0f)};
  const f32x2 t = u * u;
  f32x2 q = splat2(-8.945184002e-12f);
  q = pk_fma(q, t, splat2(1.221804868e-09f));
  q = pk_fma(q, t, splat2(-7.286091231e-08f));
  q = pk_fma(q, t, splat2(2.499930865e-06f));
  q = pk_fma(q, t, splat2(-5.482182127e-05f));
  q = pk_fma(q, t, splat2(8.080908045e-04f));
  q = pk_fma(q, t, splat2(-8.191250186e-03f));
  q = pk_fma(q, t, splat2(5.702680522e-02f));
  q = pk_fma(q, t, splat2(-2.631234724e-01f));
  q = pk_fma(q, t, splat2(7.970332990e-01f));
  return pk_fma(u, q, splat2(0.5f));
}
__device__ __forceinline__ f32x2 dgelu_exact_f2(f32x2 x) {
  const f32x2 a = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  const f32x2 d = pk_fma(splat2(0.2316419f), a, splat2(1.0f));
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f32x2 m = a * a * splat2(-0.72134752044448170f);
  const f32x2 e = {__builtin_amdgcn_exp2f(m[0]), __builtin_amdgcn_exp2f(m[1])};
  f32x2 w = splat2(0.53070271f);
  w = pk_fma(w, t, splat2(-0.72657601f));
  w = pk_fma(w, t, splat2(0.71070687f));
  w = pk_fma(w, t, splat2(-0.14224837f));
  w = pk_fma(w, t, splat2(0.12741479f));
  w = w * t;
  const f32x2 g = e * pk_fma(splat2(-0.39894228040143268f), a, w);
  const f32x2 hmg = splat2(0.5f) - g;
  return f32x2{0.5f + __builtin_copysignf(hmg[0], x[0]), 0.5f + __builtin_copysignf(hmg[1], x[1])};
}
#ifdef OCTMAE_DGELU_POLY
__device__ __forceinline__ float dgelu_f(float x) { return dgelu_poly_f(x); }
__device__ __forceinline__ f32x2 dgelu_f2(f32x2 x) { return dgelu_poly_f2(x); }
#else
__device__ __forceinline__ float dgelu_f(float x) { return dgelu_exact_f(x); }
__device__ __forceinline__ f32x2 dgelu_f2(f32x2 x) { return dgelu_exact_f2(x); }
#endif

// ---- MFMA wrappers ---------------------------------------------------------------------------
// D(32x32) += A(32x16) * B(16x32).  lane l: r = l & 31, h = l >> 5.
//   A fragment element j = A[row r][k = 8h + j]      B fragment element j = B[k = 8h + j][col r]
//   D register g      = D[row (g&3) + 8*(g>>2) + 4h][col r]
#ifdef OCTMAE_F16
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(lp8_t, a), __builtin_bit_cast(lp8_t, b), c, 0, 0, 0);
}
// D(16x16) += A(16x32) * B(32x16)
__device__ __forceinline__ f32x4 mfma16x16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lp8_t, a), __builtin_bit_cast(lp8_t, b), c, 0, 0, 0);
}
#define OCTMAE_MFMA32_ASM "v_mfma_f32_32x32x16_f16"
#else
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// D(16x16) += A(16x32) * B(32x16)
__device__ __forceinline__ f32x4 mfma16x16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#define OCTMAE_MFMA32_ASM "v_mfma_f32_32x32x16_bf16"
#endif

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3
// of a 4x16 block of 16-bit elements; lane i of the group receives column i (rows 0..3).
__device__ __forceinline__ bf16x4 lds_tr_read(const void* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(lds_ptr));
}

__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// XCD-aware block remap (8 XCDs, round-robin dispatch): blocks that land on one XCD get a contiguous
// range of logical tile ids, so neighbouring tiles share that XCD's L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, r = n & 7, x = bid & 7, w = bid >> 3;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + w;
}

}  // namespace octmae

// ---- hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) --------------------------------------------
// One object per kernel (a function-local static in its launcher).  The attribute is per device, so the "done" state is a bit
// per device ordinal, set with an atomic OR: concurrent first calls from the forward thread and autograd's backward thread
// both set the (idempotent) attribute and both succeed -- no thread-local or unsynchronised state behind the C ABI.
#include <atomic>
namespace octmae {
struct DynLdsOnce {
  std::atomic<unsigned long long> done{0};
  int ensure(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
  }
};
}  // namespace octmae

// ---- host-side error plumbing for the C ABI ---------------------------------------------------
#define OCTMAE_CHECK_ARG(cond) \
  do {                         \
    if (!(cond)) return -1;    \
  } while (0)
#define OCTMAE_LAUNCH_CHECK()                  \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)
