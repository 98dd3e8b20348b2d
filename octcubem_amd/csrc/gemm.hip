// bf16 MFMA GEMM for gfx950 with the three operand layouts the transformer needs and fused epilogues.
//
//   X[a][b] = sum_k A[a][k] * B[b][k]          (A: NA rows, B: NB rows, reduction K)
//
// Each operand is either "k-contiguous" (memory [rows][K], the nn.Linear layout) or "k-strided"
// (memory [K][rows]); a k-strided tile is staged row-major in LDS and handed to the MFMA through
// ds_read_b64_tr_b16 (hardware transpose), so no operand is ever transposed in HBM:
//   forward  y = x W^T      : A = W [N][K] kc,  B = x  [M][K] kc    -> C[M][N]   (reference K6 GEMMs,
//   dgrad    dx = dy W      : A = W [N][K] ks,  B = dy [M][N] kc    -> C[M][K]    video_vit.py:114-135,
//   wgrad    dW = dy^T x    : A = dy[M][N] ks,  B = x  [M][K] ks    -> C[N][K]    timm Mlp fc1/fc2)
// Output orientation: OUT_BA writes C[b][a] (lane = row b, 4 consecutive a per register quad: packed
// 8/16-byte stores); OUT_AB writes C[a][b] (lane-contiguous 128-B segments: the shape fp32 atomics
// run at full rate with, used by split-K wgrad).
//
// Tile 128(a) x 128(b) x 64(k), 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles,
// register-staged double-buffered LDS (global loads for tile t+2 are in flight during the MFMAs of tile
// t; LDS write of tile t+1 after the MFMAs; one barrier per k-tile), XOR-swizzled LDS images that are
// conflict-free for ds_read_b128 (k-contiguous) and for the transposed reads (k-strided).
// That is the SMALL-problem kernel (gemm_kernel).  Problems with at least one full 256-tile each way run the 256 x 256 x 64
// LDS-DMA kernels further down: gemm256p_kernel (phased main loop with staggered wave groups; the default for forward, dgrad and
// wgrad) and gemm256_kernel (plain two-stage main loop: the round-1 forward kernel, kept selectable for tests and A/B runs).
#include "common.hpp"
#include "../../include/octmae.h"

namespace octmae {

constexpr int TA = 128, TB = 128, TK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // one operand tile, either layout: 16 KiB

enum Epi : int {
  EPI_BF16 = 0,      // C bf16 = X (+bias[a])
  EPI_F32 = 1,       // C f32  = X (+bias[a])
  EPI_GELU = 2,      // C bf16 = X+bias (pre-activation), C2 bf16 = gelu(X+bias)
  EPI_RESID = 3,     // C f32  = aux_f32 + X + bias
  EPI_DGELU = 4,     // C bf16 = X * gelu'(aux_bf16)
  EPI_ACCUM = 5,     // C f32 += X   (OUT_AB; atomics when split-K > 1)
  EPI_DELTA = 6,     // C bf16 = X, C2 f32 [NB][ldc2]: C2[b][a / hd] = -sum_{j < hd} bf16(X[b][a0 + j]) * aux_bf16[b][a0 + j]
                     //   (the attention backward's per-query delta = rowsum(dO * O), from the proj dgrad that produces dO;
                     //   256-tile LDS-transposing epilogue only)
};

constexpr unsigned EPI_OOB_ANY = 0xfffffff0u;   // a byte offset past any buffer range: masks a lane of a raw buffer access

struct GemmParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  void* C2;
  const float* bias;
  const void* aux;
  int NA, NB, K;
  int lda, ldb, ldc, ldaux;
  int tiles_a, tiles_b;
  int ktiles, ktiles_per_split;
  const float* rowscale;   // EPI_RESID only: C = aux + rowscale[b / rows_per_scale] * (X + bias)   (stochastic depth); NULL = 1
  int rows_per_scale;
  int cgroup;              // 256-tile kernels: column tiles (a) per group of the tile order, see tile_coord()
  int hd;                  // EPI_DELTA: head dimension (32 or 64: a head is 4 or 8 lanes of the epilogue's read side)
  int kstagger;            // split-K: 0 = equal k slices; d > 0 = slice z is d / 256 k-tiles longer than slice z - 1 (split_range)
  int ldc2;                // EPI_DGELU column sums: 0 = C2 is an fp32 [NA] vector (atomics); > 0 = C2 is fp32 [NB / 64][ldc2],
                           // one row of partial sums per 64-row slab, plain stores (folded by a second launch)
  int dgelu_stored = 0;    // EPI_GELU: C receives gelu'(pre) instead of pre; EPI_DGELU: aux holds gelu'(pre) already (variant bit 15)
  int atomic1 = 0;         // EPI_ACCUM, UNSPLIT launch: 0 guarded read-modify-write, 1 fp32 atomics, 2 batched buffer read-modify-write (g_wgrad_s1_atomic)
};

// k-tile range of slice kz of an S-way split.  Equal slices end together and their fp32-atomic epilogues (one dword per L2
// channel and clock, ~1.35 TB/s: 256 workgroups x 256 KiB = 50 us) then run with every k-loop already over.  With kstagger the
// slice lengths rise linearly with kz -- bound(z) = z ktiles / S - d z (S - z) / 2 -- so the slices finish one after the other and
// the atomics of slice z overlap the k-loops of slices z + 1 ... (the host picks d = the atomic time of one slice's tiles).
__host__ __device__ __forceinline__ void split_range_of(int ktiles, int ktiles_per_split, int kstagger, int kz, int S, int& kt0, int& kt1) {
  if (kstagger == 0) {
    kt0 = kz * ktiles_per_split;
    kt1 = kt0 + ktiles_per_split;
    if (kt1 > ktiles) kt1 = ktiles;
  } else {
    auto bound = [&](int z) { return (int)(((long long)z * ktiles) / S) - (int)(((long long)kstagger * z * (S - z)) >> 9); };
    kt0 = bound(kz);
    kt1 = bound(kz + 1);
  }
}
__device__ __forceinline__ void split_range(const GemmParams& p, int kz, int S, int& kt0, int& kt1) {
  split_range_of(p.ktiles, p.ktiles_per_split, p.kstagger, kz, S, kt0, kt1);
}

// Tile order of the 256-tile kernels.  xcd_remap hands every XCD one contiguous range of t.  Inside a group of `cgroup`
// column tiles the order is a-fastest (neighbouring workgroups share the activation row panel B), and a group is swept over
// ALL row tiles before the next group starts.  With cgroup == tiles_a that is the plain a-fastest order: the 32 workgroups of
// an XCD touch up to 16 weight panels at once, 8 MB for a [4096 x 1024] weight against a 4 MB L2, and every round of
// workgroups fetches the whole weight matrix again (1.3 GB per fc1 forward, 8x the activations it reads).  With a group
// whose weight panels fit the L2 (<= 2 MB) they are fetched once per XCD and stay; the row panels are then read once per
// group instead of once.
__device__ __forceinline__ void tile_coord(const GemmParams& p, int t, int& ta, int& tb) {
  const int c = p.cgroup;
  if (c >= p.tiles_a) {
    tb = t / p.tiles_a; ta = t - tb * p.tiles_a;
  } else {
    const int per = c * p.tiles_b;           // cgroup divides tiles_a (host side)
    const int cg = t / per, r = t - cg * per;
    tb = r / c; ta = cg * c + (r - tb * c);
  }
}

// ---- LDS images ------------------------------------------------------------------------------
// k-contiguous tile: [128 rows][64 k] bf16, 128-B rows, 16-B chunk c of row r at r*128 + ((c ^ ((r>>1)&7))<<4)
__device__ __forceinline__ int kc_off(int r, int c) { return r * 128 + (((c ^ ((r >> 1) & 7))) << 4); }
// k-strided tile: [64 k][128 rows] bf16, 256-B rows, chunk c of k-row kr at kr*256 + ((c ^ ((kr&3)<<2))<<4)
__device__ __forceinline__ int ks_off(int kr, int c) { return kr * 256 + ((c ^ ((kr & 3) << 2)) << 4); }

template <bool KS>
struct TileLoader {
  // each of the 256 threads moves 4 x 16 B of a 16 KiB tile
  const bf16_t* base;  // operand base
  int ld;              // leading dimension (elements)
  int row0;            // first tile row (a or b index)
  int nrows;           // valid rows of the operand
  int K;
  u32x4 regs[4];

  __device__ __forceinline__ void issue(int k0, int tid) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int q = tid + 256 * n;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (!KS) {
        const int r = q >> 3, c = q & 7;
        const int gr = row0 + r, gk = k0 + c * 8;
        if (gr < nrows && gk < K) v = *reinterpret_cast<const u32x4*>(base + (size_t)gr * ld + gk);
      } else {
        const int kr = q >> 4, c = q & 15;
        const int gk = k0 + kr, gr = row0 + c * 8;
        if (gk < K && gr < nrows) v = *reinterpret_cast<const u32x4*>(base + (size_t)gk * ld + gr);
      }
      regs[n] = v;
    }
  }
  __device__ __forceinline__ void commit(char* lds, int tid) const {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int q = tid + 256 * n;
      int off;
      if (!KS) off = kc_off(q >> 3, q & 7);
      else off = ks_off(q >> 4, q & 15);
      *reinterpret_cast<u32x4*>(lds + off) = regs[n];
    }
  }
};

// fragment of 32 rows starting at rb for k-step s (16 k) of a staged tile
template <bool KS>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int rb, int s, int lane) {
  if (!KS) {
    const int r = rb + (lane & 31), c = 2 * s + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(lds + kc_off(r, c));
  } else {
    const int h = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const int c = (rb >> 3) + 2 * gi + (p >> 1);
    const int kr0 = 16 * s + 8 * h + q;
    const bf16x4 lo = lds_tr_read(lds + ks_off(kr0, c) + (p & 1) * 8);
    const bf16x4 hi = lds_tr_read(lds + ks_off(kr0 + 4, c) + (p & 1) * 8);
    return cat4(lo, hi);
  }
}

// Epilogue shared by both tile shapes.  acc[i][j] is the 32x32 block at rows a_base + 32 i, columns b_base + 32 j
// of X[a][b]; register 4g+e of a lane (r = lane&31, h = lane>>5) is X[a_base + 32 i + 8g + 4h + e][b_base + 32 j + r].
template <int EPI, bool OUT_AB, int MI, int NJ>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[MI][NJ], int a_base, int b_base, int lane,
                                              bool atomic) {
  const int r = lane & 31, h = lane >> 5;
  if (!OUT_AB) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int b = b_base + j * 32 + r;
      if (b >= p.NB) continue;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int a = a_base + i * 32 + 8 * g + 4 * h;
          if (a >= p.NA) continue;  // NA is a multiple of 4: whole quad in or out
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
          if (EPI != EPI_DGELU && EPI != EPI_ACCUM && p.bias != nullptr) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + a);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
          }
          const size_t o = (size_t)b * p.ldc + a;
          if (EPI == EPI_BF16) {
            u32x2 w = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + o) = w;
          } else if (EPI == EPI_F32) {
            f32x4 w = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + o) = w;
          } else if (EPI == EPI_GELU) {
            u32x2 w = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            if (p.dgelu_stored) {
              const u32x2 wd = {pack2bf(dgelu_f(bflo(w[0])), dgelu_f(bfhi(w[0]))), pack2bf(dgelu_f(bflo(w[1])), dgelu_f(bfhi(w[1])))};
              *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + o) = wd;
            } else {
              *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + o) = w;
            }
            // activation of the bf16-ROUNDED pre-activation, so backward (which only has the
            // rounded value) differentiates exactly the function the forward evaluated
            float y[4];
            y[0] = gelu_f(bflo(w[0])); y[1] = gelu_f(bfhi(w[0]));
            y[2] = gelu_f(bflo(w[1])); y[3] = gelu_f(bfhi(w[1]));
            u32x2 w2 = {pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C2) + o) = w2;
          } else if (EPI == EPI_RESID) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.aux) + (size_t)b * p.ldaux + a);
            const float sc = p.rowscale ? p.rowscale[b / p.rows_per_scale] : 1.0f;
            f32x4 w = {sc * v[0] + rv[0], sc * v[1] + rv[1], sc * v[2] + rv[2], sc * v[3] + rv[3]};
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + o) = w;
          } else if (EPI == EPI_DGELU) {
            const u32x2 pre = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.aux) + (size_t)b * p.ldaux + a);
            if (p.dgelu_stored) {
              v[0] *= bflo(pre[0]); v[1] *= bfhi(pre[0]); v[2] *= bflo(pre[1]); v[3] *= bfhi(pre[1]);
            } else {
              v[0] *= dgelu_f(bflo(pre[0])); v[1] *= dgelu_f(bfhi(pre[0]));
              v[2] *= dgelu_f(bflo(pre[1])); v[3] *= dgelu_f(bfhi(pre[1]));
            }
            u32x2 w = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + o) = w;
          }
        }
      }
    }
  } else {
    // C[a][b] += X[a][b]: one register across a half-wave is 32 consecutive b of one row a: two 128-B row segments per
    // wave instruction (the full-rate fp32 atomic shape).
    float* C = reinterpret_cast<float*>(p.C);
    if (!atomic && p.atomic1 == 2) {
      // unsplit launch, plain read-modify-write in batches of 16 registers through a buffer descriptor: out-of-range elements are
      // clipped by the offset (no branches), 16 loads are in flight before the first add (the guarded form below makes one
      // dependent round trip per register: 35 us of a 75 us launch at 21 k-tiles, tools/gemm_small_fit.py)
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(C, 0, (unsigned)((size_t)p.NA * p.ldc * 4), 0x00020000);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int b = b_base + j * 32 + r;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          unsigned off[16];
          float v[16];
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int a = a_base + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
            off[g] = (a < p.NA && b < p.NB) ? (unsigned)(((size_t)a * p.ldc + b) * 4) : EPI_OOB_ANY;
            v[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, off[g], 0, 0));
          }
#pragma unroll
          for (int g = 0; g < 16; ++g)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[g] + acc[i][j][g]), rc, off[g], 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int b = b_base + j * 32 + r;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int a = a_base + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
          if (a < p.NA && b < p.NB) {
            float* dst = C + (size_t)a * p.ldc + b;
            if (atomic) unsafeAtomicAdd(dst, acc[i][j][g]);
            else *dst += acc[i][j][g];
          }
        }
      }
    }
  }
}

template <bool A_KS, bool B_KS, int EPI, bool OUT_AB>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // buffers: A0 A1 B0 B1

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wa = wid >> 1, wb = wid & 1;

  // a-fastest tile order inside an XCD chunk: neighbouring blocks share the B (activation) panel
  const int nt = p.tiles_a * p.tiles_b;
  const int t = xcd_remap(blockIdx.x, nt);
  const int tb = t / p.tiles_a, ta = t - tb * p.tiles_a;
  const int a0 = ta * TA, b0 = tb * TB;

  int kt0, kt1;
  split_range(p, (int)blockIdx.z, (int)gridDim.z, kt0, kt1);
  const int nk = kt1 - kt0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileLoader<A_KS> la{p.A, p.lda, a0, p.NA, p.K, {}};
  TileLoader<B_KS> lb{p.B, p.ldb, b0, p.NB, p.K, {}};

  if (nk > 0) {
    la.issue(kt0 * TK, tid);
    lb.issue(kt0 * TK, tid);
    la.commit(smem, tid);
    lb.commit(smem + 2 * TILE_BYTES, tid);
    if (nk > 1) {
      la.issue((kt0 + 1) * TK, tid);
      lb.issue((kt0 + 1) * TK, tid);
    }
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
      const char* cA = smem + (it & 1) * TILE_BYTES;
      const char* cB = smem + (2 + (it & 1)) * TILE_BYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 fa[2], fb[2];
        fa[0] = read_frag<A_KS>(cA, wa * 64, s, lane);
        fa[1] = read_frag<A_KS>(cA, wa * 64 + 32, s, lane);
        fb[0] = read_frag<B_KS>(cB, wb * 64, s, lane);
        fb[1] = read_frag<B_KS>(cB, wb * 64 + 32, s, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);  // D[a][b]: lane = b, registers = a
          }
      }
      if (it + 1 < nk) {
        la.commit(smem + ((it + 1) & 1) * TILE_BYTES, tid);
        lb.commit(smem + (2 + ((it + 1) & 1)) * TILE_BYTES, tid);
        if (it + 2 < nk) {
          la.issue((kt0 + it + 2) * TK, tid);
          lb.issue((kt0 + it + 2) * TK, tid);
        }
      }
      __syncthreads();
    }
  }

  gemm_epilogue<EPI, OUT_AB, 2, 2>(p, acc, a0 + wa * 64, b0 + wb * 64, lane, gridDim.z > 1);
}

// =====================================================================================================
// 256 x 256 x 64 tile, 8 waves (2 x 4), each wave 128(a) x 64(b) = 4 x 2 MFMA blocks, operands brought in by
// LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write), two 64 KiB stages, the DMA of k-tile t+1
// in flight under the MFMAs of k-tile t.  The LDS images are the same XOR-swizzled ones as above; because a DMA
// instruction writes 1 KiB lane-linearly, the swizzle is applied to the per-lane SOURCE address instead.
// Out-of-range rows are zero-filled by the buffer descriptor's bounds check (rows past the end of the operand);
// requires K % 64 == 0 for k-contiguous operands (host falls back to the 128-tile kernel otherwise).
// =====================================================================================================
constexpr int T2 = 256;
constexpr int TILE2_BYTES = 256 * 64 * 2;  // 32 KiB per operand tile

__device__ __forceinline__ int kc2_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ int ks2_off(int kr, int c) { return kr * 512 + ((c ^ ((kr & 3) << 2)) << 4); }

template <bool KS>
__device__ __forceinline__ void dma_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int wid, int lane, int row0, int k0, int ld) {
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int j = wid * 4 + n;  // 1-KiB piece of the 32 KiB image (wave-uniform)
    unsigned voff;
    if (!KS) {
      const int row = 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      voff = (unsigned)(((size_t)(row0 + row) * ld + k0 + c * 8) * 2);
    } else {
      const int kr = 2 * j + (lane >> 5);
      const int c = (lane & 31) ^ ((kr & 3) << 2);
      voff = (unsigned)(((size_t)(k0 + kr) * ld + row0 + c * 8) * 2);
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds_tile + j * 1024), 16, voff, 0, 0, 0);
  }
}

template <bool KS>
__device__ __forceinline__ bf16x8 read_frag2(const char* lds, int rb, int s, int lane) {
  if (!KS) {
    const int r = rb + (lane & 31), c = 2 * s + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(lds + kc2_off(r, c));
  } else {
    const int h = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const int c = (rb >> 3) + 2 * gi + (p >> 1);
    const int kr0 = 16 * s + 8 * h + q;
    const bf16x4 lo = lds_tr_read(lds + ks2_off(kr0, c) + (p & 1) * 8);
    const bf16x4 hi = lds_tr_read(lds + ks2_off(kr0 + 4, c) + (p & 1) * 8);
    return cat4(lo, hi);
  }
}

// Coalesced epilogue of the 256-tile kernel.  In the accumulator a lane owns ONE output row (b) and 4 consecutive
// columns per register quad, so direct stores touch 32 rows x 16 B per instruction: 32 L2 write requests each, and the
// store path -- not the MFMAs -- bounded the fc1 / proj GEMMs.  Instead every wave transposes its 64(b) x 128(a) block
// through a wave-private 16 KiB LDS region (the k-loop stages are free by now) and writes whole 256-B row segments:
// 4 rows x 256 B per instruction, 16-byte lanes.  LDS rows are 256 B, 16-B chunks XOR-swizzled by (row & 15).
// No workgroup barrier: the region is private to the wave and LDS executes a wave's accesses in order.
__device__ __forceinline__ int epi_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

// Global accesses of this epilogue go through buffer descriptors that cover exactly the wave's <= 64 output rows: rows past
// NB and columns past NA fall outside num_records (loads return 0, stores are dropped), so the whole epilogue is one
// branch-free block.  That matters for the epilogues that READ (residual, pre-activation): with guarded plain loads LLVM
// sinks every load into its store's branch, and a wave then makes 16-32 dependent HBM round trips with nothing else in
// flight -- the fc2 / proj / dgelu GEMMs spent as long in the epilogue as in the k-loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* base, int b_base, int NB, int ld, int esize) {
  int rows = NB - b_base;
  rows = rows < 0 ? 0 : (rows > 64 ? 64 : rows);
  const unsigned long long q = reinterpret_cast<unsigned long long>(base) + (unsigned long long)b_base * ld * esize;
  // wave-uniform by construction; said explicitly so that the descriptor lives in SGPRs without a waterfall loop
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)q), hi = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(rows * ld * esize), 0x00020000);
}
constexpr unsigned EPI_OOB = 0x7ffffff0u;
#ifndef EPI_LOAD_AUX
#define EPI_LOAD_AUX 0    // cache policy of the epilogues' own streaming reads (residual stream, pre-activation, attention output)
#endif
#ifndef EPI_STORE_AUX
#define EPI_STORE_AUX 3   // sc0 nt: streaming stores (outputs are >= 100 MB and not re-read before they leave the L2); measured -5..6 % on the bf16 / GELU forwards
#endif   // byte offset past any num_records above: masks a lane whose columns are >= NA

// MI = 32-row blocks of a per wave: 4 (the 256-tile kernels: a wave's block is 128 a x 64 b) or 2 (gemm128d_kernel: 64 a x 64 b).
// acc is f32x16[MI][2], the 32x32 C layout above: a lane holds quads of 4 consecutive a of one output row b.
// bf16 image [64 b][32 MI a]: 256-B rows (epi_off) for MI = 4, 128-B rows with the chunk swizzled by (row >> 1) & 7 for MI = 2 (two
// rows per 64-bank wrap: the 16 rows a half-wave's quads of one a-chunk go to then spread over all banks); read side 4 MI lanes per
// row, 64 / (4 MI) rows per instruction.  fp32 image [64 b][64 a], 256-B rows, one half of the wave's columns at a time (MI / 2 halves).
__device__ __forceinline__ int epi_off128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int EPI, int MI = 4>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& p, f32x16 (&acc)[MI][2], int a_base, int b_base, int lane,
                                                  char* wl) {
  static_assert(MI == 4 || MI == 2, "wave block of 128 or 64 columns");
  const int r = lane & 31, h = lane >> 5;
  if (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_DGELU || EPI == EPI_DELTA) {
    // bf16 image [64 b][32 MI a].  EPI_DGELU multiplies by gelu'(pre) on the read side: the product of the bf16-rounded
    // dgrad with gelu' is rounded once more -- the numerics of an autocast GELU backward, which also reads a bf16 dgrad.
    constexpr int LPR = 4 * MI, RPI = 64 / LPR, NIT = 64 / RPI;     // lanes per row, rows per instruction, instructions
    auto off = [](int row, int chunk) { return MI == 4 ? epi_off(row, chunk) : epi_off128(row, chunk); };
    const int rrow = lane / LPR, rc = lane % LPR;   // read side
    const int a = a_base + rc * 8;
    const bool a_ok = a < p.NA;
    const unsigned o_out = a_ok ? (unsigned)(rrow * p.ldc + a) * 2u : EPI_OOB;
    u32x4 prev[NIT];
    if (EPI == EPI_DGELU || EPI == EPI_DELTA) {   // all pre-activation (attention-output) row segments requested before the transpose
      const __amdgpu_buffer_rsrc_t rx = epi_rsrc(p.aux, b_base, p.NB, p.ldaux, 2);
      const unsigned o_aux = a_ok ? (unsigned)(rrow * p.ldaux + a) * 2u : EPI_OOB;
#pragma unroll
      for (int it = 0; it < NIT; ++it) prev[it] = __builtin_amdgcn_raw_buffer_load_b128(rx, o_aux + it * RPI * p.ldaux * 2, 0, EPI_LOAD_AUX);
    }
    // All bias quads of the lane are requested together before the first store.  vmcnt retires in order: a bias load
    // issued after stores -- the second pass of EPI_GELU reloaded them -- waits for every store ahead of it to be
    // acknowledged, and 16 separately guarded loads were 16 dependent L2 round trips per pass (most of the 8.9 k / 21.6 k
    // cycles the plain / GELU epilogue took).
    f32x4 bias_q[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) bias_q[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI != EPI_DGELU && EPI != EPI_DELTA && p.bias != nullptr) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bias_q[i][g] = *reinterpret_cast<const f32x4*>(p.bias + min(a_base + i * 32 + 8 * g + 4 * h, p.NA - 4));
    }
#pragma unroll
    for (int pass = 0; pass < (EPI == EPI_GELU ? 2 : 1); ++pass) {
      auto put = [&](f32x4 v, const f32x4 bv, int row, int chunk, int half8) {
        u32x2 w = {pack2bf(v[0] + bv[0], v[1] + bv[1]), pack2bf(v[2] + bv[2], v[3] + bv[3])};
        if (pass == 1) {   // activation of the bf16-ROUNDED pre-activation (what backward differentiates)
          const f32x2 y0 = gelu_f2(f32x2{bflo(w[0]), bfhi(w[0])}), y1 = gelu_f2(f32x2{bflo(w[1]), bfhi(w[1])});
          w = u32x2{pack2bf(y0[0], y0[1]), pack2bf(y1[0], y1[1])};
        } else if (EPI == EPI_GELU && p.dgelu_stored) {   // ... or its derivative there, for the fc2 dgrad's epilogue to multiply with
          const f32x2 y0 = dgelu_f2(f32x2{bflo(w[0]), bfhi(w[0])}), y1 = dgelu_f2(f32x2{bflo(w[1]), bfhi(w[1])});
          w = u32x2{pack2bf(y0[0], y0[1]), pack2bf(y1[0], y1[1])};
        }
        *reinterpret_cast<u32x2*>(wl + off(row, chunk) + 8 * half8) = w;
      };
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            put(f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]}, bias_q[i][g], j * 32 + r,
                4 * i + g, h);
        }
      }
      __builtin_amdgcn_wave_barrier();
      const __amdgpu_buffer_rsrc_t ro = epi_rsrc(pass == 0 ? p.C : p.C2, b_base, p.NB, p.ldc, 2);
      float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // EPI_DGELU: column sums of what is stored (the fc1 bias gradient)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + rrow;
        u32x4 v = *reinterpret_cast<const u32x4*>(wl + off(row, rc));
        if (EPI == EPI_DGELU) {
          const u32x4 pre = prev[it];   // zeros outside the matrix: gelu'(0) * v is stored nowhere and summed nowhere
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 pv = f32x2{bflo(pre[e]), bfhi(pre[e])};
            const f32x2 y = f32x2{bflo(v[e]), bfhi(v[e])} * (p.dgelu_stored ? pv : dgelu_f2(pv));
            v[e] = pack2bf(y[0], y[1]);
          }
          if (p.C2 != nullptr && b_base + row < p.NB) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[2 * e] += bflo(v[e]); cs[2 * e + 1] += bfhi(v[e]); }
          }
        }
        if (EPI == EPI_DELTA) {
          // delta of this row and head from the bf16-ROUNDED dO the attention backward will read, summed in the order of
          // attn_rowconst_pad_kernel (8 elements per lane, then a butterfly over the hd / 8 lanes of the head)
          const u32x4 oo = prev[it];   // zeros outside the matrix
          float sum = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) sum += bflo(oo[e]) * bflo(v[e]) + bfhi(oo[e]) * bfhi(v[e]);
          // butterfly over the head's lanes on DPP (quad_perm [1,0,3,2], [2,3,0,1], then row_half_mirror: every lane of a quad
          // holds the quad's sum by then, so mirroring the 8-lane half adds the other quad's) -- the same additions as
          // __shfl_xor 1, 2, 4, which hipcc lowers to ds_bpermute (48 dependent LDS round trips per wave here)
          sum = dpp_add<0xB1>(sum);
          sum = dpp_add<0x4E>(sum);
          if (p.hd == 64) sum = dpp_add<0x141>(sum);
          const int lph = p.hd >> 3;
          if ((rc & (lph - 1)) == 0 && a_ok && b_base + row < p.NB)
            reinterpret_cast<float*>(p.C2)[(size_t)(b_base + row) * p.ldc2 + (a >> (p.hd == 64 ? 6 : 5))] = -sum;
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, ro, o_out + it * RPI * p.ldc * 2, 0, EPI_STORE_AUX);
      }
      if (EPI == EPI_DGELU && p.C2 != nullptr) {      // the row groups (lane / LPR) hold partial sums of the same 8 columns
        float* colsum = reinterpret_cast<float*>(p.C2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (MI == 2) cs[e] += __shfl_xor(cs[e], 8, 64);
          cs[e] += __shfl_xor(cs[e], 16, 64);
          cs[e] += __shfl_xor(cs[e], 32, 64);
        }
        if (p.ldc2 > 0) {
          // one row of partial sums per 64-row slab, 512 contiguous bytes per wave, plain stores: thousands of row slabs adding
          // to the SAME [NA] vector with atomics serialise in the L2 (~60 ns per add and address: +0.67 ms on the decoder's fc2 dgrad)
          if (rrow == 0 && a_ok) {
            float* dst = colsum + (size_t)(b_base >> 6) * p.ldc2 + a;
            *reinterpret_cast<f32x4*>(dst) = f32x4{cs[0], cs[1], cs[2], cs[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (rrow == 0 && a_ok) unsafeAtomicAdd(colsum + a + e, cs[e]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    // fp32 image [64 b][64 a], one half of the wave's columns at a time.  EPI_RESID: the residual row segments of a half (and
    // the per-row stochastic-depth scales) are requested before that half is transposed, half 1 while half 0 is stored.
    constexpr int NH = MI / 2;
    const int rrow = lane >> 4, rc = lane & 15;   // read side: 16 lanes per 256-B row, 4 rows per instruction
    const __amdgpu_buffer_rsrc_t ro = epi_rsrc(p.C, b_base, p.NB, p.ldc, 4);
    const __amdgpu_buffer_rsrc_t rx = epi_rsrc(EPI == EPI_RESID ? p.aux : p.C, b_base, p.NB, EPI == EPI_RESID ? p.ldaux : p.ldc, 4);
    u32x4 resv[NH][16];
    float rsc[16];
    auto load_res = [&](int half) {
      const int a = a_base + half * 64 + rc * 4;
      const unsigned o_aux = a < p.NA ? (unsigned)(rrow * p.ldaux + a) * 4u : EPI_OOB;
#pragma unroll
      for (int it = 0; it < 16; ++it) resv[half][it] = __builtin_amdgcn_raw_buffer_load_b128(rx, o_aux + it * 4 * p.ldaux * 4, 0, EPI_LOAD_AUX);
    };
    // the bias is added on the read side, where a lane keeps the same 4 columns for all 16 row segments of a half: one
    // load per lane and half, issued ahead of the residual requests (vmcnt retires in order)
    f32x4 bias4[NH];
#pragma unroll
    for (int half = 0; half < NH; ++half) bias4[half] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
      for (int half = 0; half < NH; ++half)
        bias4[half] = *reinterpret_cast<const f32x4*>(p.bias + min(a_base + half * 64 + rc * 4, p.NA - 4));
    }
    if (EPI == EPI_RESID) {
      load_res(0);
#pragma unroll
      for (int it = 0; it < 16; ++it) rsc[it] = 1.0f;
      if (p.rowscale) {
        // sample index of the wave's rows without a per-row division: one scalar quotient for the first row, then a
        // compare per row when a sample spans >= 64 rows (always, for token sequences; the division is the general case)
        const int rps = p.rows_per_scale;
        const int q0 = __builtin_amdgcn_readfirstlane(b_base / rps), rem0 = b_base - q0 * rps;
        const int qmax = __builtin_amdgcn_readfirstlane((p.NB - 1) / rps);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int x = rem0 + it * 4 + rrow;
          const int q = rps >= 64 ? q0 + (x >= rps ? 1 : 0) : q0 + x / rps;
          rsc[it] = p.rowscale[min(q, qmax)];
        }
      }
    }
#pragma unroll
    for (int half = 0; half < NH; ++half) {
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const int i = half * 2 + i2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x4 w = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            const int row = j * 32 + r;
            *reinterpret_cast<f32x4*>(wl + epi_off(row, i2 * 8 + 2 * g + h)) = w;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (EPI == EPI_RESID && half + 1 < NH) load_res(half + 1);
      const int a = a_base + half * 64 + rc * 4;
      const unsigned o_out = a < p.NA ? (unsigned)(rrow * p.ldc + a) * 4u : EPI_OOB;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + rrow;
        f32x4 v = *reinterpret_cast<const f32x4*>(wl + epi_off(row, rc)) + bias4[half];
        if (EPI == EPI_RESID) v = v * rsc[it] + __builtin_bit_cast(f32x4, resv[half][it]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, o_out + it * 4 * p.ldc * 4, 0, EPI_STORE_AUX);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Optional per-workgroup timeline (make -C octcubem_amd/csrc trace; tools/gemm_trace.py): 100 MHz wall-clock stamps at
// workgroup start, k-loop start, k-loop end and workgroup end, plus the hardware id, for the two-stage kernel.
#ifdef GEMM_TRACE
__device__ unsigned long long* g_trace = nullptr;   // [workgroup][6]
#define TRACE_MARK(slot) do { if (g_trace && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 6 + (slot)] = wall_clock64(); } while (0)
#else
#define TRACE_MARK(slot) do {} while (0)
#endif
template <bool A_KS, bool B_KS, int EPI, bool OUT_AB>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // A0 A1 B0 B1
  TRACE_MARK(0);
#ifdef GEMM_TRACE
  if (g_trace && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_trace[(size_t)blockIdx.x * 6 + 4] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wid >> 2, wb = wid & 3;

  const int nt = p.tiles_a * p.tiles_b;
  // Split-K launches (weight gradients) are ordered k-slice-major over the XCD-contiguous logical index: the ~32 workgroups
  // an XCD runs at a time then belong to ONE k slice and to neighbouring tiles, so the operand panels they stream are
  // shared through that XCD's L2 (with the slice index on blockIdx.z taken as is, every XCD held 8 tiles of EACH slice and
  // fetched 36 panel streams per 32 workgroups instead of 12-18; the kernel ran at the HBM ceiling, 6 TB/s).
  int t, kz;
  if (gridDim.z > 1) {
    const int L = xcd_remap((int)(blockIdx.x + nt * blockIdx.z), nt * (int)gridDim.z);
    kz = L / nt; t = L - kz * nt;
  } else {
    t = xcd_remap(blockIdx.x, nt); kz = 0;
  }
  int ta, tb;
  tile_coord(p, t, ta, tb);
  const int a0 = ta * T2, b0 = tb * T2;

  int kt0, kt1;
  split_range(p, kz, (int)gridDim.z, kt0, kt1);
  const int nk = kt1 - kt0;

  const unsigned a_bytes = (unsigned)((size_t)(A_KS ? p.K : p.NA) * p.lda * 2);
  const unsigned b_bytes = (unsigned)((size_t)(B_KS ? p.K : p.NB) * p.ldb * 2);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B), 0, b_bytes, 0x00020000);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) {
    dma_tile<A_KS>(ra, smem, wid, lane, a0, kt0 * TK, p.lda);
    dma_tile<B_KS>(rb, smem + 2 * TILE2_BYTES, wid, lane, b0, kt0 * TK, p.ldb);
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier
    TRACE_MARK(1);
    for (int it = 0; it < nk; ++it) {
      if (it + 1 < nk) {
        dma_tile<A_KS>(ra, smem + ((it + 1) & 1) * TILE2_BYTES, wid, lane, a0, (kt0 + it + 1) * TK, p.lda);
        dma_tile<B_KS>(rb, smem + (2 + ((it + 1) & 1)) * TILE2_BYTES, wid, lane, b0, (kt0 + it + 1) * TK, p.ldb);
      }
      const char* cA = smem + (it & 1) * TILE2_BYTES;
      const char* cB = smem + (2 + (it & 1)) * TILE2_BYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 fa[4], fb[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag2<A_KS>(cA, wa * 128 + i * 32, s, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = read_frag2<B_KS>(cB, wb * 64 + j * 32, s, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
      }
      __syncthreads();
    }
  }
  TRACE_MARK(2);
  if (OUT_AB || (p.NA & 7) != 0) {
    gemm_epilogue<EPI, OUT_AB, 4, 2>(p, acc, a0 + wa * 128, b0 + wb * 64, lane, gridDim.z > 1);
  } else {
    gemm_epilogue_lds<EPI>(p, acc, a0 + wa * 128, b0 + wb * 64, lane, smem + wid * 16384);
  }
  TRACE_MARK(3);
}
#ifdef GEMM_TRACE
extern "C" int octmae_debug_set_gemm_trace(void* ptr) {   // trace builds only; not part of include/octmae.h
  unsigned long long* q = reinterpret_cast<unsigned long long*>(ptr);
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &q, sizeof(q));
}
#endif

// =====================================================================================================
// Phased main loop for the same 256 x 256 x 64 tile (cdna_hip_programming.md section 5, "8-phase" structure, re-derived for
// 32x32x16 MFMAs and this kernel's operand layouts).  A k-tile is computed in 4 phases, one quadrant (64 a x 32 b) of the
// wave's 128 x 64 block per phase over the full K = 64 (8 MFMAs); a phase is [LDS-read segment | barrier | MFMA segment |
// barrier].  Waves 4-7 -- the SIMD partners of waves 0-3 -- run ONE barrier behind, so on every SIMD one wave's MFMA
// segment runs beside its partner's LDS-read segment: the matrix pipe never waits for fragment reads.  Fragments are
// reused across phases (A: quadrant rows loaded in phases 1 and 3; B: both column blocks kept), 24 reads per k-tile as before.
//   phase 1: read A[q0], B[q0]   stage A halves of k-tile t+1     MFMA (q0, q0)
//   phase 2: read B[q1]                                            MFMA (q0, q1)
//   phase 3: read A[q1]                                            MFMA (q1, q1)
//   phase 4:                     stage B halves of k-tile t+2     MFMA (q1, q0)     s_waitcnt vmcnt(4) before its barrier
// Operand tiles are four 16 KiB half-images per stage (A rows 0-127 / 128-255, B likewise), staged by LDS-DMA from inline
// asm (untracked: the only vmcnt waits are the counted ones), each half one phase-set AFTER its last reader has passed a
// barrier: A halves die in phase 3, B halves in phase 2.  The B halves are therefore prefetched a whole k-tile ahead and
// the A halves three phases ahead; only the 4 youngest loads are in flight across the phase-4 barrier.
// =====================================================================================================
typedef __attribute__((ext_vector_type(4))) int gi32x4;

__device__ __forceinline__ int kcH_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }          // [128 rows][64 k]
__device__ __forceinline__ int ksH_off(int kr, int c) { return kr * 256 + ((c ^ ((kr & 3) << 2)) << 4); }         // [64 k][128 rows]

// fragment of 32 rows starting at in-half row rb, k-step s, from one 16 KiB half-image
template <bool KS>
__device__ __forceinline__ bf16x8 read_fragH(const char* lds, int rb, int s, int lane) {
  if (!KS) {
    const int r = rb + (lane & 31), c = 2 * s + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(lds + kcH_off(r, c));
  } else {
    const int h = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    const int c = (rb >> 3) + 2 * gi + (pp >> 1);
    const int kr0 = 16 * s + 8 * h + q;
    const bf16x4 lo = lds_tr_read(lds + ksH_off(kr0, c) + (pp & 1) * 8);
    const bf16x4 hi = lds_tr_read(lds + ksH_off(kr0 + 4, c) + (pp & 1) * 8);
    return cat4(lo, hi);
  }
}

__device__ __forceinline__ void glds16(const gi32x4& rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is "reserved"; nothing else in this kernel lives in it
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
#pragma clang diagnostic pop
}

// the tile body: tile t of p, k-tiles kt0 .. kt1 - 1   (smem: [stage 2][operand 2][half 2] x 16 KiB); partial = split-K (atomics)
template <bool A_KS, bool B_KS, int EPI, bool OUT_AB>
__device__ __forceinline__ void gemm256p_body(const GemmParams& p, const int t, const int kt0, const int kt1, const bool partial,
                                              char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wid >> 2, wb = wid & 3;          // wa = 1: the late group (waves 4-7)
  int ta, tb;
  tile_coord(p, t, ta, tb);
  const int a0 = ta * T2, b0 = tb * T2;
  const int nk = kt1 - kt0;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) {
    const unsigned a_bytes = (unsigned)((size_t)(A_KS ? p.K : p.NA) * p.lda * 2);
    const unsigned b_bytes = (unsigned)((size_t)(B_KS ? p.K : p.NB) * p.ldb * 2);
    const unsigned long long pa = (unsigned long long)p.A, pb = (unsigned long long)p.B;
    const gi32x4 ra = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)a_bytes, 0x00020000};
    const gi32x4 rb = {(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)b_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // per-lane source offsets of this wave's two 1-KiB pieces of each half-image (k-tile 0), and the k-tile stride
    unsigned va[2][2], vb[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int j = 2 * wid + n;
        if (!A_KS) {
          const int r = 8 * j + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
          va[hf][n] = (unsigned)(((size_t)(a0 + hf * 128 + r) * p.lda + (size_t)kt0 * TK + c * 8) * 2);
        } else {
          const int kr = 4 * j + (lane >> 4), c = (lane & 15) ^ ((kr & 3) << 2);
          va[hf][n] = (unsigned)((((size_t)kt0 * TK + kr) * p.lda + a0 + hf * 128 + c * 8) * 2);
        }
        if (!B_KS) {
          const int r = 8 * j + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
          vb[hf][n] = (unsigned)(((size_t)(b0 + hf * 128 + r) * p.ldb + (size_t)kt0 * TK + c * 8) * 2);
        } else {
          const int kr = 4 * j + (lane >> 4), c = (lane & 15) ^ ((kr & 3) << 2);
          vb[hf][n] = (unsigned)((((size_t)kt0 * TK + kr) * p.ldb + b0 + hf * 128 + c * 8) * 2);
        }
      }
    const unsigned ka = (unsigned)(A_KS ? (size_t)TK * p.lda * 2 : (size_t)TK * 2);
    const unsigned kb = (unsigned)(B_KS ? (size_t)TK * p.ldb * 2 : (size_t)TK * 2);
    auto stage = [&](int op, int kt, int buf) {            // both halves of operand `op` of k-tile kt -> stage buf (4 loads)
      const unsigned base = lds0 + (unsigned)(buf * 65536 + op * 32768 + 2 * wid * 1024);
      const unsigned so = (unsigned)kt * (op ? kb : ka);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          glds16(op ? rb : ra, base + (unsigned)(hf * 16384 + n * 1024), op ? vb[hf][n] : va[hf][n], so);
    };
    // prologue: B(0), A(0), B(1) -- the order the loop keeps (B a k-tile ahead of A)
    stage(1, 0, 0);
    stage(0, 0, 0);
    stage(1, 1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F74);       // vmcnt(4) (lgkmcnt/expcnt untouched): A(0), B(0) have landed
    __builtin_amdgcn_s_barrier();
    if (wa) __builtin_amdgcn_s_barrier();     // the late group starts one barrier behind

    const int rowB = (wb & 1) * 64;           // this wave's 64 b-rows inside its B half
    bf16x8 fa[2][4], fb[2][4];
    // EPI_ACCUM with a C2 vector: column sums of the A operand (A = dY: the bias gradient) ride along as ONE extra MFMA per
    // k-step against an all-ones B fragment, in the tiles of the first b column only; the 4 waves that share an a-range take one
    // of its 32-row blocks each (block wb: phase 1 for wb < 2, phase 3 otherwise).  +12.5 % MFMAs in 1 / tiles_b of the tiles,
    // against a separate pass over dY (0.25 ms per qkv weight gradient at micro-batch 128).
    const bool cs_on = (EPI == EPI_ACCUM) && p.C2 != nullptr && tb == 0;
    const u32x4 ones_w = {OCTMAE_LP_ONE_PAIR, OCTMAE_LP_ONE_PAIR, OCTMAE_LP_ONE_PAIR, OCTMAE_LP_ONE_PAIR};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_w);
    f32x16 csum;
#pragma unroll
    for (int r = 0; r < 16; ++r) csum[r] = 0.f;
    for (int it = 0; it < nk; ++it) {
      const int buf = it & 1;
      const char* cA = smem + buf * 65536 + wa * 16384;                    // my A half: rows wa*128 ..
      const char* cB = smem + buf * 65536 + 32768 + (wb >> 1) * 16384;     // my B half
      // ---- phase 1
#pragma unroll
      for (int s = 0; s < 4; ++s) fb[0][s] = read_fragH<B_KS>(cB, rowB, s, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[i][s] = read_fragH<A_KS>(cA, i * 32, s, lane);
      stage(0, it + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][0] = mfma32(fa[i][s], fb[0][s], acc[i][0]);
      if (EPI == EPI_ACCUM && cs_on && wb < 2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) csum = mfma32(wb ? fa[1][s] : fa[0][s], ones, csum);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // ---- phase 2
#pragma unroll
      for (int s = 0; s < 4; ++s) fb[1][s] = read_fragH<B_KS>(cB, rowB + 32, s, lane);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][1] = mfma32(fa[i][s], fb[1][s], acc[i][1]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // ---- phase 3
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[i][s] = read_fragH<A_KS>(cA, 64 + i * 32, s, lane);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[2 + i][1] = mfma32(fa[i][s], fb[1][s], acc[2 + i][1]);
      if (EPI == EPI_ACCUM && cs_on && wb >= 2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) csum = mfma32(wb == 3 ? fa[1][s] : fa[0][s], ones, csum);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // ---- phase 4
      stage(1, it + 2, buf);
      __builtin_amdgcn_s_waitcnt(0x0F74);     // vmcnt(4): everything but the B halves just issued -- A(it+1), B(it+1) landed
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[2 + i][0] = mfma32(fa[i][s], fb[0][s], acc[2 + i][0]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    if (!wa) __builtin_amdgcn_s_barrier();    // the early group waits for the late one
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the trailing (unused) prefetches must not land in the epilogue's LDS
    __builtin_amdgcn_s_barrier();
    if (EPI == EPI_ACCUM && cs_on && (lane & 31) == 0) {
      // every column of the ones product holds the sums: column 0 = lanes 0 (rows 0-3, 8-11, ...) and 32 (rows 4-7, 12-15, ...)
      float* colsum = reinterpret_cast<float*>(p.C2);
      const int h = lane >> 5;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int a = a0 + wa * 128 + wb * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (a < p.NA) unsafeAtomicAdd(colsum + a, csum[g]);
      }
    }
  }
  if (OUT_AB || (p.NA & 7) != 0) {
    gemm_epilogue<EPI, OUT_AB, 4, 2>(p, acc, a0 + wa * 128, b0 + wb * 64, lane, partial);
  } else {
    gemm_epilogue_lds<EPI>(p, acc, a0 + wa * 128, b0 + wb * 64, lane, smem + wid * 16384);
  }
}

template <bool A_KS, bool B_KS, int EPI, bool OUT_AB>
__global__ __launch_bounds__(512, 1) void gemm256p_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = p.tiles_a * p.tiles_b;
  // Split-K launches (weight gradients) are ordered k-slice-major over the XCD-contiguous logical index: the ~32 workgroups
  // an XCD runs at a time then belong to ONE k slice and to neighbouring tiles, so the operand panels they stream are
  // shared through that XCD's L2 (with the slice index on blockIdx.z taken as is, every XCD held 8 tiles of EACH slice and
  // fetched 36 panel streams per 32 workgroups instead of 12-18; the kernel ran at the HBM ceiling, 6 TB/s).
  int t, kz;
  if (gridDim.z > 1) {
    const int L = xcd_remap((int)(blockIdx.x + nt * blockIdx.z), nt * (int)gridDim.z);
    kz = L / nt; t = L - kz * nt;
  } else {
    t = xcd_remap(blockIdx.x, nt); kz = 0;
  }
  int kt0, kt1;
  split_range(p, kz, (int)gridDim.z, kt0, kt1);
  gemm256p_body<A_KS, B_KS, EPI, OUT_AB>(p, t, kt0, kt1, gridDim.z > 1 || p.atomic1 == 1, smem);
}


// Two weight gradients with the same reduction length (the same token rows) in ONE launch: the tiles of both share the split, so the
// pair runs tiles0 + tiles1 tiles x S slices where each alone would run tiles x 2 S -- half the fp32-atomic epilogues (256 KiB per
// workgroup whatever the split) and k-loops twice as long (fc1 + fc2 of a Block: 128 tiles x 2 instead of 2 x (64 x 4)).
struct GemmPair {
  GemmParams p0, p1;
  int nt0, nt1, S;
};
__global__ __launch_bounds__(512, 1) void gemm256p_wgrad_pair_kernel(const GemmPair pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = pp.nt0 + pp.nt1;
  const int L = xcd_remap((int)blockIdx.x, nt * pp.S);          // k-slice-major, as above
  const int kz = L / nt;
  int t = L - kz * nt;
  const bool second = t >= pp.nt0;                              // workgroup-uniform
  if (second) t -= pp.nt0;
  int kt0, kt1;
  if (!second) {
    split_range(pp.p0, kz, pp.S, kt0, kt1);
    gemm256p_body<true, true, EPI_ACCUM, true>(pp.p0, t, kt0, kt1, pp.S > 1 || pp.p0.atomic1 == 1, smem);
  } else {
    split_range(pp.p1, kz, pp.S, kt0, kt1);
    gemm256p_body<true, true, EPI_ACCUM, true>(pp.p1, t, kt0, kt1, pp.S > 1 || pp.p1.atomic1 == 1, smem);
  }
}


// =====================================================================================================
// gemm128d_kernel (round 6): the SMALL-LAUNCH kernel of the forward / dgrad kinds.  A launch of the 256-tile kernels costs one
// tile time -- k-tiles x ~1.2 us -- however few of the 256 CUs it fills: one volume per step (the reference's shipped recipe,
// scripts/run_chunks_pretraining_vitl_oph_joint_flash_attn.sh:25-30) gives the encoder's proj / fc2 forwards 24 tiles and the
// GEMMs 49 % of the step (profiles/r05_batch1_kernel_stats.csv).  Here: tile 128(a) x 128(b) x 64(k), 4 waves (2 x 2, ONE per
// SIMD), each wave 64 x 64 = 2 x 2 v_mfma_f32_32x32x16 blocks; four times the workgroups, a quarter of the k-tile time each.
//   * operands by LDS-DMA into a ring of NST stages of (A 16 KiB | B 16 KiB) -- the half-images of the phased kernel, same
//     swizzles, same fragment reads -- NST - 1 k-tiles ahead, one counted vmcnt and ONE workgroup barrier per k-tile.  NST = 4
//     (128 KiB, one workgroup per CU: a wave per SIMD hides HBM / L2 latency only through the ring's depth) for launches of at
//     most one workgroup per CU; NST = 2 (64 KiB, two co-resident workgroups per CU) for launches between one and two per CU;
//   * trailing ring slots are requested with soffset = num_records: every lane out of range, zero-filled, no memory traffic;
//   * the coalescing LDS-transposing epilogue of the 256-tile kernels with a 64-column wave block (gemm_epilogue_lds<EPI, 2>),
//     every fused epilogue of the forward / dgrad kinds;
//   * split-K (S <= 4 slices of a tile's k range, chosen by the host for long reductions on few tiles: plan128 below),
//     DETERMINISTIC: every slice leaves its fp32 partial tile in a caller-lent workspace (16-byte write-through stores), one lane
//     adds to the tile's arrival counter after every storing wave's vmcnt(0) and a workgroup barrier, and the workgroup whose add
//     came LAST -- told by the value the add returned -- sums the S partials in slice order (its own from registers, the others
//     by sc1 loads after a barrier that lane joins: MI355X_MICROARCH.md, hand-offs measured with sc1 loads, first row), runs the
//     fused epilogue and resets the counter.  Nobody ever waits for another workgroup: no co-residency assumption, no spin.
// Requires NA % 8 == 0, K % 64 == 0 for k-contiguous operands, operands within a 32-bit buffer range (host: gemm128_ok).
// =====================================================================================================
struct SplitWs {
  float* slots;       // [tiles x S] fp32 partial tiles of 128 x 128 (S > 1)
  unsigned* ctr;      // [tiles] arrival counters: zero before the launch, zero again after it
};
constexpr int T1 = 128;
constexpr int D_STAGE = 2 * TILE_BYTES;      // 32 KiB: A image | B image
constexpr int D_SLOT_FLOATS = T1 * T1;

// main loop of the small-launch kernels: tile t of p, k slice kz of S -> acc (the wave's 64 x 64 block at a0 + wa 64, b0 + wb 64);
// returns with every ring request retired and a workgroup barrier passed (the LDS is free for the epilogue)
template <bool A_KS, bool B_KS, int NST>
__device__ __forceinline__ void gemm128d_mainloop(const GemmParams& p, const int t, const int kz, const int S, char* smem,
                                                  f32x16 (&acc)[2][2], int& a0, int& b0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wid >> 1, wb = wid & 1;
  int ta, tb;
  tile_coord(p, t, ta, tb);
  a0 = ta * T1; b0 = tb * T1;
  int kt0, kt1;
  split_range(p, kz, S, kt0, kt1);
  const int nk = kt1 - kt0;

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) {
    const unsigned a_bytes = (unsigned)((size_t)(A_KS ? p.K : p.NA) * p.lda * 2);
    const unsigned b_bytes = (unsigned)((size_t)(B_KS ? p.K : p.NB) * p.ldb * 2);
    const unsigned long long pa = (unsigned long long)p.A, pb = (unsigned long long)p.B;
    const gi32x4 ra = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)a_bytes, 0x00020000};
    const gi32x4 rb = {(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)b_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // per-lane source offsets of this wave's four 1-KiB pieces of each image (k-tile kt0), and the k-tile stride
    unsigned va[4], vb[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int j = 4 * wid + n;
      if (!A_KS) {
        const int r = 8 * j + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        va[n] = (unsigned)(((size_t)(a0 + r) * p.lda + (size_t)kt0 * TK + c * 8) * 2);
      } else {
        const int kr = 4 * j + (lane >> 4), c = (lane & 15) ^ ((kr & 3) << 2);
        va[n] = (unsigned)((((size_t)kt0 * TK + kr) * p.lda + a0 + c * 8) * 2);
      }
      if (!B_KS) {
        const int r = 8 * j + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        vb[n] = (unsigned)(((size_t)(b0 + r) * p.ldb + (size_t)kt0 * TK + c * 8) * 2);
      } else {
        const int kr = 4 * j + (lane >> 4), c = (lane & 15) ^ ((kr & 3) << 2);
        vb[n] = (unsigned)((((size_t)kt0 * TK + kr) * p.ldb + b0 + c * 8) * 2);
      }
    }
    const unsigned ka = (unsigned)(A_KS ? (size_t)TK * p.lda * 2 : (size_t)TK * 2);
    const unsigned kb = (unsigned)(B_KS ? (size_t)TK * p.ldb * 2 : (size_t)TK * 2);
    auto stage = [&](int it) {            // k-tile kt0 + it -> ring slot it % NST (8 loads per wave)
      const unsigned base = lds0 + (unsigned)((it % NST) * D_STAGE + 4 * wid * 1024);
      const bool live = it < nk;
      const unsigned sa = live ? (unsigned)it * ka : a_bytes, sb = live ? (unsigned)it * kb : b_bytes;
#pragma unroll
      for (int n = 0; n < 4; ++n) glds16(ra, base + (unsigned)(n * 1024), va[n], sa);
#pragma unroll
      for (int n = 0; n < 4; ++n) glds16(rb, base + (unsigned)(TILE_BYTES + n * 1024), vb[n], sb);
    };
#pragma unroll
    for (int it = 0; it < NST - 1; ++it) stage(it);
    for (int it = 0; it < nk; ++it) {
      // tile `it` has landed (this wave's pieces: the NST - 2 younger stages may still be in flight), then everybody's; the
      // barrier also says that every wave has finished reading tile it - 1, whose slot the next request overwrites
      if (NST == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (NST == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const char* cA = smem + (it % NST) * D_STAGE;
      const char* cB = cA + TILE_BYTES;
      // all 16 fragment reads of the k-tile first, into registers of their own (a wave is alone on its SIMD: 512 registers),
      // the 16 MFMAs behind them: the matrix pipe waits for the first k-step's reads only, the rest arrive under it; the ring's
      // next requests go out BEHIND the reads (their slot, tile it - 1's, is free since the barrier): their issue runs under the
      // reads' latency instead of in front of it
      bf16x8 fa[4][2], fb[4][2];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[s][i] = read_fragH<A_KS>(cA, wa * 64 + i * 32, s, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[s][j] = read_fragH<B_KS>(cB, wb * 64 + j * 32, s, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      stage(it + NST - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[s][i], fb[s][j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (zero-filling) requests must not land in the epilogue's LDS
    __builtin_amdgcn_s_barrier();
  }

}

template <bool A_KS, bool B_KS, int EPI, int NST>
__global__ __launch_bounds__(256, NST == 2 ? 2 : 1) void gemm128d_kernel(const GemmParams p, const SplitWs w, const int S) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned s_arrival;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wid >> 1, wb = wid & 1;
  const int nt = p.tiles_a * p.tiles_b;
  int t, kz;
  if (S > 1) {      // k-slice-major over the XCD-contiguous index, as the weight gradients: an XCD's workgroups share panels
    const int L = xcd_remap((int)blockIdx.x, nt * S);
    kz = L / nt; t = L - kz * nt;
  } else {
    t = xcd_remap((int)blockIdx.x, nt); kz = 0;
  }
  f32x16 acc[2][2];
  int a0, b0;
  gemm128d_mainloop<A_KS, B_KS, NST>(p, t, kz, S, smem, acc, a0, b0);

  if (S > 1) {
    {   // publish this slice's partial tile: lane-contiguous 16-byte write-through stores, 4 KiB per store instruction
      const unsigned long long q = reinterpret_cast<unsigned long long>(w.slots) + ((unsigned long long)t * S + kz) * (D_SLOT_FLOATS * 4);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(q), 0, D_SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (((i * 2 + j) * 4 + g) * 256 + tid) * 16, 0, 16 /* sc1 */);
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its stores have been written through
    __syncthreads();
    if (tid == 0) s_arrival = __hip_atomic_fetch_add(w.ctr + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_arrival != (unsigned)(S - 1)) return;               // not the last slice of this tile to arrive
    f32x16 tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    for (int z = 0; z < S; ++z) {                             // slice order, whoever arrived last: the sum is a function of the shape alone
      if (z == kz) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] += acc[i][j][r];
      } else {
        const unsigned long long q = reinterpret_cast<unsigned long long>(w.slots) + ((unsigned long long)t * S + z) * (D_SLOT_FLOATS * 4);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(q), 0, D_SLOT_FLOATS * 4, 0x00020000);
        u32x4 v[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) v[n] = __builtin_amdgcn_raw_buffer_load_b128(rs, (n * 256 + tid) * 16, 0, 16 /* sc1 */);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
          const f32x4 f = __builtin_bit_cast(f32x4, v[n]);
          const int i = n >> 3, j = (n >> 2) & 1, g = n & 3;
          tot[i][j][4 * g] += f[0]; tot[i][j][4 * g + 1] += f[1]; tot[i][j][4 * g + 2] += f[2]; tot[i][j][4 * g + 3] += f[3];
        }
      }
    }
    if (tid == 0) __hip_atomic_store(w.ctr + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch on this stream
    gemm_epilogue_lds<EPI, 2>(p, tot, a0 + wa * 64, b0 + wb * 64, lane, smem + wid * 16384);
    return;
  }
  gemm_epilogue_lds<EPI, 2>(p, acc, a0 + wa * 64, b0 + wb * 64, lane, smem + wid * 16384);
}


// The weight gradients of a small launch (round 6): the pair kernel above on 128 x 128 tiles -- the main loop of gemm128d_kernel with
// both operands k-strided, the fp32-accumulating OUT_AB epilogue (atomics when the k range is split, batched read-modify-write
// otherwise).  One volume per step gives the fc1 + fc2 pair 128 tiles of 256 x 256 over 21 k-tiles: half the CUs for 40 us; here
// 512 workgroups of a quarter the work, two per CU.  The bias gradient (column sums of dY) is a separate launch on this path.
template <int NST>
__global__ __launch_bounds__(256, NST == 2 ? 2 : 1) void gemm128d_wgrad_kernel(const GemmPair pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wid >> 1, wb = wid & 1;
  const int nt = pp.nt0 + pp.nt1;
  const int L = xcd_remap((int)blockIdx.x, nt * pp.S);          // k-slice-major
  const int kz = L / nt;
  int t = L - kz * nt;
  const bool second = t >= pp.nt0;                              // workgroup-uniform
  if (second) t -= pp.nt0;
  f32x16 acc[2][2];
  int a0, b0;
  if (!second) {
    gemm128d_mainloop<true, true, NST>(pp.p0, t, kz, pp.S, smem, acc, a0, b0);
    gemm_epilogue<EPI_ACCUM, true, 2, 2>(pp.p0, acc, a0 + wa * 64, b0 + wb * 64, lane, pp.S > 1 || pp.p0.atomic1 == 1);
  } else {
    gemm128d_mainloop<true, true, NST>(pp.p1, t, kz, pp.S, smem, acc, a0, b0);
    gemm_epilogue<EPI_ACCUM, true, 2, 2>(pp.p1, acc, a0 + wa * 64, b0 + wb * 64, lane, pp.S > 1 || pp.p1.atomic1 == 1);
  }
}

// Staggered split-K slices of the 256-tile weight-gradient kernel (split_range): v = the length step between neighbouring slices in
// 1/256 k-tiles PER OUTPUT TILE of the launch (the atomic time of a slice grows with its tile count: 256 KiB at 1.35 TB/s = 0.19 us
// per tile against 1.7 us per k-tile, i.e. v = 29); 0 = equal slices.  octmae_set_option("wgrad_stagger", v) / OCTMAE_WGRAD_STAGGER.
// Applied only to splits of >= 8 slices with <= 96 k-tiles each (the [C x C] proj gradients at <= 32 volumes per rank): measured
// (profiles/r04_wgrad_stagger.txt) -9 / -11 % there, and nothing or a loss for the 4- and 5-way splits
// and for every shape at 128 volumes -- their workgroups do not end together anyway.
std::atomic<int> g_wgrad_stagger{29};
// The epilogue of an UNSPLIT weight-gradient launch of the phased 256-tile kernel: 0 = plain read-modify-write (one guarded load + store
// per register), 1 = the same fire-and-forget fp32 atomics a split launch uses (still deterministic: one add per element and launch),
// 2 = read-modify-write in batches of 16 registers through a buffer descriptor.  "wgrad_s1_atomic" / OCTMAE_WGRAD_S1_ATOMIC.
std::atomic<int> g_wgrad_s1_atomic{2};

// CU count of the current device (cached per device)
static int device_cus() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int v = cached[dev & 63].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  cached[dev & 63].store(v, std::memory_order_relaxed);
  return v;
}

template <bool A_KS, bool B_KS, int EPI, bool OUT_AB>
static int launch256(const GemmParams& p, int splitk, hipStream_t st, bool phased) {
  auto kern = phased ? gemm256p_kernel<A_KS, B_KS, EPI, OUT_AB> : gemm256_kernel<A_KS, B_KS, EPI, OUT_AB>;
  static DynLdsOnce once[2];
  if (int rc = once[phased].ensure(reinterpret_cast<const void*>(kern), 4 * TILE2_BYTES)) return rc;
  dim3 grid(p.tiles_a * p.tiles_b, 1, splitk);
  hipLaunchKernelGGL(kern, grid, dim3(512), 4 * TILE2_BYTES, st, p);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

// ---- the small-launch kernel: workspace, plan, launch -------------------------------------------------------------------------
// Workspace (caller-owned, lent per call, one per stream): [D_WS_SLOTS partial tiles of 64 KiB][D_WS_TILES arrival counters].  The
// counters must be zero when the workspace is first lent; every launch leaves them zero.
constexpr int D_WS_SLOTS = 1024, D_WS_TILES = 4096;
static long long d_ws_bytes() { return (long long)D_WS_SLOTS * D_SLOT_FLOATS * 4 + (long long)D_WS_TILES * 4; }

std::atomic<int> g_gemm_small{1};          // octmae_set_option("gemm_small", 0 / 1); OCTMAE_GEMM_SMALL overrides
std::atomic<int> g_small_launches{0};      // how many launches took gemm128d_kernel ("gemm_small_launches": tests)
std::atomic<int> g_small_split_launches{0};
std::atomic<int> g_small_wgrad_launches{0};   // weight-gradient pairs that took gemm128d_wgrad_kernel ("gemm_small_wgrad_launches")

struct Plan128 {
  int use;      // 1: gemm128d_kernel
  int S;        // k slices per tile (1: no workspace needed)
  int nst;      // ring stages: 4 (one workgroup per CU) or 2 (two)
};
// Which kernel a forward / dgrad launch of NA x NB x K takes on G CUs (fitted on MI355X with tools/gemm_small_fit.py: device times of
// graph-replayed launches, profiles/r06_gemm_small_fit.txt).
//  * At most one round of 256-tiles (nt256 <= G, the regime the kernel was built for): a latency model in microseconds.  A workgroup of
//    the 256-tile kernel takes ~1.05 us per k-tile + ~6.5 us (prologue, epilogue, launch); one of this kernel ~0.5 us per k-tile + ~4.5 us
//    alone on its CU (4-stage ring) and ~0.85 us per k-tile + ~5.5 us in rounds of two per CU (2-stage ring); a k split adds the publish
//    and the last arriver's gather, ~5.5 us + ~1.5 us per slice beyond the second.
//  * More than one round: both kernels stream, and what differs is how much of their LAST round is empty.  Per 128 x 128 x 64 block of
//    work the 256-tile kernel costs 1.18 / 4 us and this kernel's 2-stage form 0.64 / 2 us; with eff = tiles / (rounds x slots) of each
//    (slots: G and 2 G) it is taken when 0.32 / eff128 < 0.9 x 0.295 / eff256 -- e.g. the decoder's [8 x 5121 rows] x K -> 512 dgrads:
//    322 tiles of 256 = 1.26 rounds of CUs against 1284 of 128 = 2.5 rounds of 512 slots, 82 against 96 us.  The 128-volume shapes of
//    the headline step sit at eff256 >= 0.83 and keep the 256-tile kernel (tests/test_cpu_host.py).
static Plan128 plan128(int NA, int NB, int K, int G, bool have_ws, bool big256_ok) {
  static const int env = getenv("OCTMAE_GEMM_SMALL") ? atoi(getenv("OCTMAE_GEMM_SMALL")) : -1;
  const int on = env >= 0 ? env : g_gemm_small.load(std::memory_order_relaxed);
  Plan128 pl{0, 1, 4};
  if (!on) return pl;
  const int ktiles = (K + TK - 1) / TK;
  const long long nt128 = (long long)((NA + T1 - 1) / T1) * ((NB + T1 - 1) / T1);
  const long long nt256 = (long long)((NA + T2 - 1) / T2) * ((NB + T2 - 1) / T2);
  if (big256_ok && nt256 > G) {
    const double eff256 = (double)nt256 / (double)(((nt256 + G - 1) / G) * G);
    const double eff128 = (double)nt128 / (double)(((nt128 + 2 * G - 1) / (2 * G)) * 2 * G);
    if (0.32 / eff128 < 0.9 * 0.295 / eff256) { pl.use = 1; pl.S = 1; pl.nst = 2; }
    return pl;
  }
  const double t256 = big256_ok ? 1.05 * ktiles + 6.5 : 1e30;
  double best = 1e30;
  for (int S = 1; S <= 4; ++S) {
    if (S > 1 && (!have_ws || ktiles / S < 8 || nt128 * S > D_WS_SLOTS || nt128 > D_WS_TILES)) break;
    const int kper = (ktiles + S - 1) / S;
    if (S > 1 && (long long)(S - 1) * kper >= ktiles) continue;          // an empty last slice
    const long long wg = nt128 * S;
    const double split = S > 1 ? 5.5 + 1.5 * (S - 2) : 0.0;
    const double t4 = (double)((wg + G - 1) / G) * (0.5 * kper + 4.5 + split);
    const double t2 = wg > G ? (double)((wg + 2 * G - 1) / (2 * G)) * (0.85 * kper + 5.5 + split) : 1e30;
    if (t4 < best) { best = t4; pl.S = S; pl.nst = 4; }
    if (t2 < best) { best = t2; pl.S = S; pl.nst = 2; }
  }
  pl.use = best < 0.92 * t256;
  if (!pl.use) { pl.S = 1; pl.nst = 4; }
  return pl;
}

// operands the small-launch kernel can take (the epilogue transposes 8-column chunks; LDS-DMA of whole k-tiles; 32-bit buffer ranges)
static bool gemm128_ok(int NA, int NB, int K, int lda, int ldb, bool a_ks, bool b_ks) {
  if ((NA & 7) != 0 || (a_ks && b_ks)) return false;
  if ((!a_ks || !b_ks) && (K % TK) != 0) return false;
  const size_t a_bytes = (size_t)(a_ks ? K : NA) * lda * 2, b_bytes = (size_t)(b_ks ? K : NB) * ldb * 2;
  return a_bytes < 0xFFF00000ull && b_bytes < 0xFFF00000ull;
}

template <bool A_KS, bool B_KS, int EPI>
static int launch128d(GemmParams p, const Plan128& pl, void* ws, hipStream_t st) {
  p.tiles_a = (p.NA + T1 - 1) / T1;
  p.tiles_b = (p.NB + T1 - 1) / T1;
  p.cgroup = p.tiles_a;
  p.kstagger = 0;
  p.ktiles_per_split = (p.ktiles + pl.S - 1) / pl.S;
  SplitWs w{nullptr, nullptr};
  if (pl.S > 1) {
    w.slots = reinterpret_cast<float*>(ws);
    w.ctr = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + (long long)D_WS_SLOTS * D_SLOT_FLOATS * 4);
    g_small_split_launches.fetch_add(1, std::memory_order_relaxed);
  }
  const int nt = p.tiles_a * p.tiles_b;
  if (pl.nst == 2) {
    auto kern = gemm128d_kernel<A_KS, B_KS, EPI, 2>;
    static DynLdsOnce once;
    if (int rc = once.ensure(reinterpret_cast<const void*>(kern), 2 * D_STAGE)) return rc;
    hipLaunchKernelGGL(kern, dim3(nt * pl.S, 1, 1), dim3(256), 2 * D_STAGE, st, p, w, pl.S);
  } else {
    auto kern = gemm128d_kernel<A_KS, B_KS, EPI, 4>;
    static DynLdsOnce once;
    if (int rc = once.ensure(reinterpret_cast<const void*>(kern), 4 * D_STAGE)) return rc;
    hipLaunchKernelGGL(kern, dim3(nt * pl.S, 1, 1), dim3(256), 4 * D_STAGE, st, p, w, pl.S);
  }
  OCTMAE_LAUNCH_CHECK();
  g_small_launches.fetch_add(1, std::memory_order_relaxed);
  return 0;
}

template <bool A_KS, bool B_KS, int EPI, bool OUT_AB>
static int launch(const GemmParams& p, int splitk, hipStream_t st) {
  auto kern = gemm_kernel<A_KS, B_KS, EPI, OUT_AB>;
  static DynLdsOnce once;
  if (int rc = once.ensure(reinterpret_cast<const void*>(kern), 4 * TILE_BYTES)) return rc;
  dim3 grid(p.tiles_a * p.tiles_b, 1, splitk);
  hipLaunchKernelGGL(kern, grid, dim3(256), 4 * TILE_BYTES, st, p);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace octmae

using namespace octmae;

// GemmParams::kstagger of an S-way split over `ktiles` k-tiles of a launch with `tiles` output tiles (0: equal slices)
static int wgrad_stagger_for(int ktiles, int splitk, int tiles) {
  static const int envs = getenv("OCTMAE_WGRAD_STAGGER") ? atoi(getenv("OCTMAE_WGRAD_STAGGER")) : -1;
  const int v = envs >= 0 ? envs : g_wgrad_stagger.load(std::memory_order_relaxed);
  if (!(v > 0 && splitk >= 8 && ktiles <= 96 * splitk)) return 0;
  // the shortest slice (ktiles / S - d (S - 1) / 2) keeps at least half the mean length and 8 k-tiles
  const long long mean_q8 = ((long long)ktiles << 8) / splitk;
  long long d = (long long)v * tiles;
  const long long dmax = (mean_q8 - (8 << 8) < mean_q8 / 2 ? mean_q8 - (8 << 8) : mean_q8 / 2) * 2 / (splitk - 1);
  if (d > dmax) d = dmax;
  return d > 0 ? (int)d : 0;
}

static int wgrad_s1_atomic() {
  static const int env = getenv("OCTMAE_WGRAD_S1_ATOMIC") ? atoi(getenv("OCTMAE_WGRAD_S1_ATOMIC")) : -1;
  return env >= 0 ? env : g_wgrad_s1_atomic.load(std::memory_order_relaxed);
}

extern "C" int octmae_colsum_accum(const void* in, int in_is_bf16, float* out, int M, int N, int ld, void* stream);

// C[b][a] (+epilogue) = sum_k A[a][k] B[b][k];  see include/octmae.h for the contract.
static int gemm_impl(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux,
                     int NA, int NB, int K, int lda, int ldb, int ldc, int ldaux, int a_kstrided,
                     int b_kstrided, int epilogue, int splitk, void* stream, const float* rowscale, int rows_per_scale,
                     float* colsum_ws = nullptr, void* split_ws = nullptr, long long split_bytes = 0) {
  // Variant bits of `epilogue` (tests and A/B runs exercise every kernel on the same problem): bit 8 forces the 128-tile
  // register-staged kernel, bit 9 the two-stage (un-phased) 256-tile main loop, bit 10 the phased one; bit 12 / 13 force the
  // small-launch kernel gemm128d_kernel with a 4- / 2-stage ring (its k split is then `splitk`, 1 .. 4, which otherwise only the
  // weight gradients read); bit 14 forbids it (the pre-round-6 choice); bits 16-18: the forced kernel's k split where the entry point has no `splitk`.
  const int variant = (epilogue >> 8) & 1;
  const int force128 = ((epilogue >> 12) & 1) ? 4 : ((epilogue >> 13) & 1) ? 2 : 0;
  const bool never128 = ((epilogue >> 14) & 1) != 0;
  const int dgelu_stored = (epilogue >> 15) & 1;        // bit 15: the fc1 forward stores gelu'(pre) / the fc2 dgrad multiplies with it
  // phased main loop by default (dgrad, wgrad: +10..25 % over the two-stage loop; forward, re-measured in round 2 after the
  // epilogue work: qkv -4 %, proj -11 %, fc2 -8 %, fc1 + GELU -1.5 %, decoder fc1 + GELU +0.6 % -- in round 1 the two-stage loop
  // had still been 12 % faster at K = 1024).
  const bool phased = ((epilogue >> 10) & 1) ? true : ((epilogue >> 9) & 1) ? false : true;
  const int forced_split = ((epilogue >> 16) & 7) ? ((epilogue >> 16) & 7) : splitk;     // bits 16-18: the forced kernel's k split
  epilogue &= 0xff;
  OCTMAE_CHECK_ARG(A && B && C);
  OCTMAE_CHECK_ARG(NA > 0 && NB > 0 && K > 0);
  OCTMAE_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0);
  OCTMAE_CHECK_ARG(epilogue >= 0 && epilogue <= 5);
  // contiguous-dimension granularity: 16-byte chunks of 8 bf16
  if (!a_kstrided) OCTMAE_CHECK_ARG(K % 8 == 0); else OCTMAE_CHECK_ARG(NA % 8 == 0);
  if (!b_kstrided) OCTMAE_CHECK_ARG(K % 8 == 0); else OCTMAE_CHECK_ARG(NB % 8 == 0);
  if (epilogue != EPI_ACCUM) OCTMAE_CHECK_ARG(NA % 4 == 0 && ldc % 4 == 0);
  if (epilogue == EPI_GELU) OCTMAE_CHECK_ARG(C2 != nullptr);
  if (epilogue == EPI_RESID || epilogue == EPI_DGELU) OCTMAE_CHECK_ARG(aux != nullptr && ldaux % 4 == 0);
  if (splitk < 1) splitk = 1;
  if (epilogue != EPI_ACCUM) splitk = 1;
  GemmParams p;
  p.A = reinterpret_cast<const bf16_t*>(A);
  p.B = reinterpret_cast<const bf16_t*>(B);
  p.C = C; p.C2 = C2; p.bias = bias; p.aux = aux;
  p.rowscale = rowscale; p.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  p.NA = NA; p.NB = NB; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
  p.ktiles = (K + TK - 1) / TK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  // 256-tile LDS-DMA kernel when the problem has at least one full tile each way, k-contiguous operands have whole
  // k-tiles, and each operand fits a 32-bit buffer descriptor; otherwise the 128-tile register-staged kernel.
  const size_t a_bytes = (size_t)(a_kstrided ? K : NA) * lda * 2, b_bytes = (size_t)(b_kstrided ? K : NB) * ldb * 2;
  bool big = NA >= T2 && NB >= T2 && a_bytes < 0xFFF00000ull && b_bytes < 0xFFF00000ull && (variant & 1) == 0;
  if ((!a_kstrided || !b_kstrided) && (K % TK) != 0) big = false;
  const int tile = big ? T2 : TA;
  p.tiles_a = (NA + tile - 1) / tile;
  p.tiles_b = (NB + tile - 1) / tile;
  // column-tile group of the tile order (tile_coord): groups of 4 column tiles when there are >= 16 of them and 4 weight
  // panels fit half the L2 (K <= 1024: the fc1 forward and the fc2 dgrad of ViT-L).  Measured same-box: fc1 forward
  // 891 -> 871 us, fc2 dgrad+dgelu 801 -> 790 us; narrower groups (K >= 3072, one panel per group) lose 3-5 % to the
  // activation re-reads and 12 column tiles (qkv) gain nothing, so those keep the plain order.
  p.cgroup = p.tiles_a;
  if (big && epilogue != EPI_ACCUM) {
    static const int force = getenv("OCTMAE_CGROUP") ? atoi(getenv("OCTMAE_CGROUP")) : 0;
    const size_t panel = (size_t)T2 * K * 2;
    if (force > 0) {
      if (p.tiles_a % force == 0) p.cgroup = force;
    } else if (p.tiles_a >= 16 && p.tiles_a % 4 == 0 && 4 * panel <= (2u << 20)) {
      p.cgroup = 4;
    }
  } else if (big && p.tiles_a * p.tiles_b > 32) {
    // weight gradients: the ~32 workgroups an XCD runs at a time should form a compact rectangle of tiles (8 x 4 rather than
    // 16 x 2 for fc1's 16 x 4 tiles): c column tiles x all row tiles, c = 32 / tiles_b rounded down to a divisor of tiles_a
    int c = 32 / p.tiles_b;
    if (c < 1) c = 1;
    while (c > 1 && p.tiles_a % c != 0) --c;
    p.cgroup = c;
  }
  if (splitk > p.ktiles) splitk = p.ktiles;
  p.ktiles_per_split = (p.ktiles + splitk - 1) / splitk;
  splitk = (p.ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split;

  // The small-launch kernel (forward / dgrad kinds): when the cost model says so (plan128), or forced.
  Plan128 pl{0, 1, 4};
  if (epilogue != EPI_ACCUM && !never128 && (variant & 1) == 0 && gemm128_ok(NA, NB, K, lda, ldb, a_kstrided != 0, b_kstrided != 0)) {
    const bool have_ws = split_ws != nullptr && split_bytes >= d_ws_bytes();
    if (force128) {
      pl.use = 1; pl.nst = force128;
      pl.S = forced_split < 1 ? 1 : forced_split > 4 ? 4 : forced_split;
      while (pl.S > 1 && (!have_ws || (long long)(pl.S - 1) * ((p.ktiles + pl.S - 1) / pl.S) >= p.ktiles ||
                          (long long)((NA + T1 - 1) / T1) * ((NB + T1 - 1) / T1) * pl.S > D_WS_SLOTS)) --pl.S;
    } else if (!((epilogue >> 9) & 1) && !((epilogue >> 10) & 1)) {
      pl = plan128(NA, NB, K, device_cus(), have_ws, big);
    }
  }

  // EPI_DGELU with C2: C2 is an fp32 [NA] vector that receives += the column sums of C (bias gradient of the Linear whose
  // activation is being differentiated).  Fused into the LDS-transposing epilogues (256-tile kernels, small-launch kernel); the
  // 128-tile register-staged kernel is followed by the stand-alone column-sum kernel.
  float* dgelu_colsum = (epilogue == EPI_DGELU) ? reinterpret_cast<float*>(C2) : nullptr;
  if (epilogue == EPI_DGELU && !((big || pl.use) && (NA & 7) == 0)) p.C2 = nullptr;
  p.ldc2 = 0; p.hd = 0;
  p.kstagger = (big && epilogue == EPI_ACCUM) ? wgrad_stagger_for(p.ktiles, splitk, p.tiles_a * p.tiles_b) : 0;
  p.atomic1 = (big && phased && epilogue == EPI_ACCUM) ? wgrad_s1_atomic() : 0;
  p.dgelu_stored = (epilogue == EPI_GELU || epilogue == EPI_DGELU) ? dgelu_stored : 0;
  // 64-row slabs that write a row of partial column sums: those of the grid that runs (octmae_dgelu_colsum_ws_rows covers both)
  const int ws_rows = pl.use ? 2 * ((NB + T1 - 1) / T1) : 4 * p.tiles_b;
  if (epilogue == EPI_DGELU && p.C2 != nullptr && colsum_ws != nullptr) { p.C2 = colsum_ws; p.ldc2 = NA; }
#define OCTMAE_GEMM_CASE(AKS, BKS, E, AB)                                  \
  if (a_kstrided == AKS && b_kstrided == BKS && epilogue == E) {           \
    int rc_;                                                               \
    if constexpr (!AB) {                                                   \
      rc_ = pl.use ? launch128d<AKS, BKS, E>(p, pl, split_ws, st)          \
                   : big ? launch256<AKS, BKS, E, AB>(p, splitk, st, phased) : launch<AKS, BKS, E, AB>(p, splitk, st);  \
    } else {                                                               \
      rc_ = big ? launch256<AKS, BKS, E, AB>(p, splitk, st, phased) : launch<AKS, BKS, E, AB>(p, splitk, st);           \
    }                                                                      \
    if (rc_ == 0 && E == EPI_DGELU && dgelu_colsum != nullptr && p.C2 == nullptr)                                       \
      rc_ = octmae_colsum_accum(C, 1, dgelu_colsum, NB, NA, ldc, stream);                                                \
    if (rc_ == 0 && E == EPI_DGELU && dgelu_colsum != nullptr && p.ldc2 > 0)                                             \
      rc_ = octmae_colsum_accum(colsum_ws, 0, dgelu_colsum, ws_rows, NA, NA, stream);                                     \
    return rc_;                                                                                                           \
  }
  // forward linears (nn.Linear layout both sides)
  OCTMAE_GEMM_CASE(0, 0, EPI_BF16, false)
  OCTMAE_GEMM_CASE(0, 0, EPI_F32, false)
  OCTMAE_GEMM_CASE(0, 0, EPI_GELU, false)
  OCTMAE_GEMM_CASE(0, 0, EPI_RESID, false)
  // dgrad (weight read k-strided)
  OCTMAE_GEMM_CASE(1, 0, EPI_BF16, false)
  OCTMAE_GEMM_CASE(1, 0, EPI_F32, false)
  OCTMAE_GEMM_CASE(1, 0, EPI_DGELU, false)
  // wgrad (both k-strided, fp32 accumulate, lane-contiguous output).  A non-NULL C2: fp32 [NA] += column sums of A over k (the
  // bias gradient, A = dY [K rows][NA]); fused into the phased 256-tile kernel, a separate pass over A otherwise.
  if (a_kstrided == 1 && b_kstrided == 1 && epilogue == EPI_ACCUM && C2 != nullptr && !(big && phased)) {
    p.C2 = nullptr;
    if (int rc = octmae_colsum_accum(A, 1, reinterpret_cast<float*>(C2), K, NA, lda, stream)) return rc;
  }
  OCTMAE_GEMM_CASE(1, 1, EPI_ACCUM, true)
#undef OCTMAE_GEMM_CASE
  return -2;  // layout / epilogue combination not built
}

// The split a weight-gradient launch of `tiles` 256 x 256 output tiles over M rows would use for a requested `splitk` (host-side
// arithmetic only; no GPU is touched): writes the number of slices to *slices and the first k-tile of slice z to bounds[z]
// (z = 0 .. slices, bounds[slices] = the number of 64-row k-tiles), and returns the length step kstagger (1/256 k-tiles; 0 = equal
// slices).  `bounds` needs splitk + 1 ints.  For tests of the planning code (tests/test_cpu_host.py).
extern "C" int octmae_wgrad_split_plan(int M, int splitk, int tiles, int* slices, int* bounds) {
  OCTMAE_CHECK_ARG(M > 0 && tiles > 0 && slices && bounds);
  const int ktiles = (M + TK - 1) / TK;
  if (splitk < 1) splitk = 1;
  if (splitk > ktiles) splitk = ktiles;
  const int per = (ktiles + splitk - 1) / splitk;
  splitk = (ktiles + per - 1) / per;
  const int d = wgrad_stagger_for(ktiles, splitk, tiles);
  *slices = splitk;
  for (int z = 0; z < splitk; ++z) {
    int a, b;
    split_range_of(ktiles, per, d, z, splitk, a, b);
    bounds[z] = a;
    bounds[z + 1] = b;
  }
  return d;
}

// gW0[N0][K0] += dY0[M][N0]^T X0[M][K0]  and  gW1[N1][K1] += dY1[M][N1]^T X1[M][K1]  (gB: fp32 [N] += column sums of dY, or NULL) in
// one launch of gemm256p_wgrad_pair_kernel.  Returns -2 when either problem does not take the 256-tile kernel (the caller then
// issues two octmae_gemm_bf16 calls).
extern "C" int octmae_wgrad_accum_pair(const void* dY0, const void* X0, float* gW0, float* gB0, int N0, int K0, int ldy0, int ldx0, int ldw0,
                                       const void* dY1, const void* X1, float* gW1, float* gB1, int N1, int K1, int ldy1, int ldx1, int ldw1,
                                       int M, int splitk, void* stream) {
  OCTMAE_CHECK_ARG(dY0 && X0 && gW0 && dY1 && X1 && gW1 && M > 0 && N0 > 0 && K0 > 0 && N1 > 0 && K1 > 0);
  OCTMAE_CHECK_ARG((ldy0 % 8) == 0 && (ldx0 % 8) == 0 && (ldy1 % 8) == 0 && (ldx1 % 8) == 0);
  OCTMAE_CHECK_ARG((N0 % 8) == 0 && (K0 % 8) == 0 && (N1 % 8) == 0 && (K1 % 8) == 0);
  GemmPair pp;
  auto fill = [&](GemmParams& p, const void* dY, const void* X, float* gW, float* gB, int N, int K, int ldy, int ldx, int ldw) {
    const size_t a_bytes = (size_t)M * ldy * 2, b_bytes = (size_t)M * ldx * 2;
    if (!(N >= T2 && K >= T2 && a_bytes < 0xFFF00000ull && b_bytes < 0xFFF00000ull)) return false;
    p.A = reinterpret_cast<const bf16_t*>(dY); p.B = reinterpret_cast<const bf16_t*>(X);
    p.C = gW; p.C2 = gB; p.bias = nullptr; p.aux = nullptr; p.rowscale = nullptr; p.rows_per_scale = 1;
    p.NA = N; p.NB = K; p.K = M; p.lda = ldy; p.ldb = ldx; p.ldc = ldw; p.ldaux = 0;
    p.ktiles = (M + TK - 1) / TK;
    p.tiles_a = (N + T2 - 1) / T2; p.tiles_b = (K + T2 - 1) / T2;
    p.cgroup = p.tiles_a;
    if (p.tiles_a * p.tiles_b > 32) {          // as in gemm_impl: a compact rectangle of tiles per XCD
      int c = 32 / p.tiles_b;
      if (c < 1) c = 1;
      while (c > 1 && p.tiles_a % c != 0) --c;
      p.cgroup = c;
    }
    p.ldc2 = 0; p.hd = 0; p.kstagger = 0;
    p.atomic1 = wgrad_s1_atomic();
    return true;
  };
  if (!fill(pp.p0, dY0, X0, gW0, gB0, N0, K0, ldy0, ldx0, ldw0) || !fill(pp.p1, dY1, X1, gW1, gB1, N1, K1, ldy1, ldx1, ldw1)) return -2;
  const int ktiles = pp.p0.ktiles;
  if (splitk < 1) splitk = 1;
  if (splitk > ktiles) splitk = ktiles;
  const int per = (ktiles + splitk - 1) / splitk;
  splitk = (ktiles + per - 1) / per;
  pp.p0.ktiles_per_split = pp.p1.ktiles_per_split = per;
  pp.nt0 = pp.p0.tiles_a * pp.p0.tiles_b; pp.nt1 = pp.p1.tiles_a * pp.p1.tiles_b; pp.S = splitk;
  pp.p0.kstagger = pp.p1.kstagger = wgrad_stagger_for(ktiles, splitk, pp.nt0 + pp.nt1);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // Small launches (round 6): 128 x 128 tiles when the cost model prices them faster.  Microseconds, fitted with tools/gemm_small_fit.py:
  // the 256-tile pair ~1.4 per k-tile of a slice + 0.22 per tile and slice of fp32 atomics (split) or ~10 of read-modify-write
  // (unsplit); a 128-tile workgroup (both operands through transposing LDS reads) ~0.6 per k-tile alone on its CU (4-stage ring), ~1.05
  // two per CU (2-stage), + ~8, + 0.055 per tile and slice of atomics.  Measured (profiles/r06_gemm_small_fit.txt): one volume, encoder
  // fc pair 40 -> 32 us, qkv + proj 40 -> 26, decoder 52 -> 39 and 40 -> 32; four volumes 113 -> 90 and 77 -> 64.  Short reductions only
  // (<= 96 k-tiles per launch): beyond, both kernels stream and the larger tile wins.
  {
    static const int env = getenv("OCTMAE_GEMM_SMALL") ? atoi(getenv("OCTMAE_GEMM_SMALL")) : -1;
    const int on = env >= 0 ? env : g_gemm_small.load(std::memory_order_relaxed);
    const int G = device_cus();
    const long long nt128_0 = (long long)((N0 + T1 - 1) / T1) * ((K0 + T1 - 1) / T1), nt128_1 = (long long)((N1 + T1 - 1) / T1) * ((K1 + T1 - 1) / T1);
    const long long nt128 = nt128_0 + nt128_1;
    static const int maxkt = getenv("OCTMAE_WGRAD128_MAXKT") ? atoi(getenv("OCTMAE_WGRAD128_MAXKT")) : 96;      // A/B runs
    if (on && ktiles <= maxkt && nt128 <= 2 * G) {
      const double c256 = 1.4 * per + (splitk > 1 ? 0.22 * (pp.nt0 + pp.nt1) * splitk : 10.0);
      double best = 1e30;
      int bS = 1, bN = 4;
      for (int S2 = 1; S2 <= 4 && S2 <= splitk; ++S2) {        // never more slices than the caller allows (splitk = 1: no atomics)
        const int kper = (ktiles + S2 - 1) / S2;
        if (S2 > 1 && (kper < 8 || (long long)(S2 - 1) * kper >= ktiles)) continue;
        const long long wg = nt128 * S2;
        const double at = S2 > 1 ? 0.055 * nt128 * S2 : 0.0;
        if (wg <= G && 0.6 * kper + 8.0 + at < best) { best = 0.6 * kper + 8.0 + at; bS = S2; bN = 4; }
        if (wg > G && wg <= 2 * G && 1.05 * kper + 8.0 + at < best) { best = 1.05 * kper + 8.0 + at; bS = S2; bN = 2; }
      }
      if (best < 0.9 * c256) {
        auto to128 = [&](GemmParams& p, int N, int K) {
          p.tiles_a = (N + T1 - 1) / T1; p.tiles_b = (K + T1 - 1) / T1; p.cgroup = p.tiles_a; p.kstagger = 0;
          p.ktiles_per_split = (ktiles + bS - 1) / bS;
        };
        // the bias gradients ride in the 256-tile kernel's main loop; on this path they are a pass of their own over dY
        if (gB0 != nullptr) { if (int rc = octmae_colsum_accum(dY0, 1, gB0, M, N0, ldy0, stream)) return rc; }
        if (gB1 != nullptr) { if (int rc = octmae_colsum_accum(dY1, 1, gB1, M, N1, ldy1, stream)) return rc; }
        pp.p0.C2 = nullptr; pp.p1.C2 = nullptr;
        to128(pp.p0, N0, K0); to128(pp.p1, N1, K1);
        pp.nt0 = (int)nt128_0; pp.nt1 = (int)nt128_1; pp.S = bS;
        if (bN == 2) {
          auto k2 = gemm128d_wgrad_kernel<2>;
          static DynLdsOnce once2;
          if (int rc = once2.ensure(reinterpret_cast<const void*>(k2), 2 * D_STAGE)) return rc;
          hipLaunchKernelGGL(k2, dim3((unsigned)(nt128 * bS), 1, 1), dim3(256), 2 * D_STAGE, st, pp);
        } else {
          auto k4 = gemm128d_wgrad_kernel<4>;
          static DynLdsOnce once4;
          if (int rc = once4.ensure(reinterpret_cast<const void*>(k4), 4 * D_STAGE)) return rc;
          hipLaunchKernelGGL(k4, dim3((unsigned)(nt128 * bS), 1, 1), dim3(256), 4 * D_STAGE, st, pp);
        }
        OCTMAE_LAUNCH_CHECK();
        g_small_wgrad_launches.fetch_add(1, std::memory_order_relaxed);
        return 0;
      }
    }
  }
  auto kern = gemm256p_wgrad_pair_kernel;
  static DynLdsOnce once;
  if (int rc = once.ensure(reinterpret_cast<const void*>(kern), 4 * TILE2_BYTES)) return rc;
  hipLaunchKernelGGL(kern, dim3((pp.nt0 + pp.nt1) * splitk, 1, 1), dim3(512), 4 * TILE2_BYTES, st, pp);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_gemm_bf16(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux,
                                int NA, int NB, int K, int lda, int ldb, int ldc, int ldaux, int a_kstrided,
                                int b_kstrided, int epilogue, int splitk, void* stream) {
  return gemm_impl(A, B, C, C2, bias, aux, NA, NB, K, lda, ldb, ldc, ldaux, a_kstrided, b_kstrided, epilogue, splitk, stream,
                   nullptr, 1);
}

// Which kernel a forward / dgrad launch of NA x NB x K would take on `cus` CUs under the current "gemm_small" option (host-side
// arithmetic only, no GPU is touched; for tests of the planning code): returns 1 for the small-launch kernel (gemm128d_kernel) and
// writes its k split to *slices and its ring depth to *stages, 0 for the 256-tile / register-staged kernels.  `have_ws`: a split
// workspace is lent; `big_ok`: the problem qualifies for the 256-tile kernel.
extern "C" int octmae_gemm_small_plan(int NA, int NB, int K, int cus, int have_ws, int big_ok, int* slices, int* stages) {
  OCTMAE_CHECK_ARG(NA > 0 && NB > 0 && K > 0 && cus > 0 && slices && stages);
  const Plan128 pl = plan128(NA, NB, K, cus, have_ws != 0, big_ok != 0);
  *slices = pl.S; *stages = pl.nst;
  return pl.use;
}

extern "C" int octmae_gemm_split_ws_kib(void) { return (int)((d_ws_bytes() + 1023) / 1024); }

extern "C" int octmae_gemm_bf16_ws(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux,
                                   int NA, int NB, int K, int lda, int ldb, int ldc, int ldaux, int a_kstrided,
                                   int b_kstrided, int epilogue, int splitk, void* split_ws, long long split_ws_bytes, void* stream) {
  return gemm_impl(A, B, C, C2, bias, aux, NA, NB, K, lda, ldb, ldc, ldaux, a_kstrided, b_kstrided, epilogue, splitk, stream,
                   nullptr, 1, nullptr, split_ws, split_ws_bytes);
}

extern "C" int octmae_linear_resid_rowscale(const void* W, const void* X, float* out, const float* bias, const float* res,
                                            const float* rowscale, int rows_per_scale, int N, int M, int K, int ldw, int ldx,
                                            int ldout, int ldres, int variant, void* split_ws, long long split_ws_bytes, void* stream) {
  OCTMAE_CHECK_ARG(rowscale != nullptr && rows_per_scale > 0);
  return gemm_impl(W, X, out, nullptr, bias, res, N, M, K, ldw, ldx, ldout, ldres, 0, 0, EPI_RESID | (variant & 0x77F00), 1,
                   stream, rowscale, rows_per_scale, nullptr, split_ws, split_ws_bytes);
}

// dX[M][K] bf16 = dY[M][N] @ W[N][K]  and  delta[M][H] f32 = -sum over each head's hd columns of dX * O  (O bf16 [M][K], K = H * hd):
// the proj dgrad of an attention block, whose output dO the attention backward multiplies with O row by row anyway.
// Returns -2 when the problem takes neither LDS-transposing kernel (the caller then uses octmae_gemm_bf16 + octmae_attn_bwd_fused).
extern "C" int octmae_linear_dgrad_delta(const void* W, const void* dY, void* dX, const void* O, float* delta, int M, int N, int K,
                                         int ldw, int ldy, int ldx, int ldo, int H, int hd, int variant, void* split_ws,
                                         long long split_ws_bytes, void* stream) {
  OCTMAE_CHECK_ARG(W && dY && dX && O && delta && M > 0 && N > 0 && K > 0);
  OCTMAE_CHECK_ARG((hd == 32 || hd == 64) && H > 0 && H * hd == K && (ldw % 8) == 0 && (ldy % 8) == 0 && (ldx % 4) == 0 && (ldo % 4) == 0);
  OCTMAE_CHECK_ARG(K % 8 == 0 && N % 8 == 0);
  const size_t a_bytes = (size_t)N * ldw * 2, b_bytes = (size_t)M * ldy * 2;
  const bool big = K >= T2 && M >= T2 && a_bytes < 0xFFF00000ull && b_bytes < 0xFFF00000ull && ((variant >> 8) & 1) == 0 &&
                   (N % TK) == 0 && (K & 7) == 0;
  GemmParams p;
  p.A = reinterpret_cast<const bf16_t*>(W); p.B = reinterpret_cast<const bf16_t*>(dY);
  p.C = dX; p.C2 = delta; p.bias = nullptr; p.aux = O; p.rowscale = nullptr; p.rows_per_scale = 1;
  p.NA = K; p.NB = M; p.K = N; p.lda = ldw; p.ldb = ldy; p.ldc = ldx; p.ldaux = ldo;
  p.ktiles = (N + TK - 1) / TK; p.ktiles_per_split = p.ktiles;
  p.tiles_a = (K + T2 - 1) / T2; p.tiles_b = (M + T2 - 1) / T2;
  p.cgroup = p.tiles_a;
  if (p.tiles_a >= 16 && p.tiles_a % 4 == 0 && 4 * (size_t)T2 * N * 2 <= (2u << 20)) p.cgroup = 4;
  p.hd = hd; p.ldc2 = H; p.kstagger = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int force128 = ((variant >> 12) & 1) ? 4 : ((variant >> 13) & 1) ? 2 : 0;
  if (((variant >> 8) & 1) == 0 && ((variant >> 14) & 1) == 0 && gemm128_ok(K, M, N, ldw, ldy, true, false)) {
    const bool have_ws = split_ws != nullptr && split_ws_bytes >= d_ws_bytes();
    Plan128 pl{0, 1, 4};
    if (force128) {
      pl.use = 1; pl.nst = force128;
      pl.S = (variant >> 16) & 7;
      if (pl.S < 1) pl.S = 1;
      if (pl.S > 4) pl.S = 4;
      while (pl.S > 1 && (!have_ws || (long long)(pl.S - 1) * ((p.ktiles + pl.S - 1) / pl.S) >= p.ktiles ||
                          (long long)((K + T1 - 1) / T1) * ((M + T1 - 1) / T1) * pl.S > D_WS_SLOTS)) --pl.S;
    } else {
      pl = plan128(K, M, N, device_cus(), have_ws, big);
    }
    if (pl.use) return launch128d<true, false, EPI_DELTA>(p, pl, split_ws, st);
  }
  if (!big) return -2;
  auto kern = gemm256p_kernel<true, false, EPI_DELTA, false>;
  static DynLdsOnce once;
  if (int rc = once.ensure(reinterpret_cast<const void*>(kern), 4 * TILE2_BYTES)) return rc;
  hipLaunchKernelGGL(kern, dim3(p.tiles_a * p.tiles_b, 1, 1), dim3(512), 4 * TILE2_BYTES, st, p);
  OCTMAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int octmae_dgelu_colsum_ws_rows(int M) { return M > 0 ? 4 * ((M + T2 - 1) / T2) : 0; }

extern "C" int octmae_linear_dgrad_dgelu(const void* W, const void* dY, void* dX, const void* pre, float* ws, float* bias_grad,
                                         int M, int N, int K, int ldw, int ldy, int ldx, int ldpre, int variant, void* split_ws,
                                         long long split_ws_bytes, void* stream) {
  OCTMAE_CHECK_ARG(bias_grad == nullptr || ws != nullptr);
  return gemm_impl(W, dY, dX, bias_grad, nullptr, pre, K, M, N, ldw, ldy, ldx, ldpre, 1, 0, EPI_DGELU | (variant & 0x7FF00), 1, stream, nullptr, 1,
                   bias_grad != nullptr ? ws : nullptr, split_ws, split_ws_bytes);
}
