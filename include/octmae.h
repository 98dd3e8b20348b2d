/* liboctmae -- C ABI of the MI355X-native 3-D MAE hot path (gfx950 only).
 *
 * The reference (ZucksLiu/OCTCubeM) has no FFI layer: every GPU operation it performs arrives through
 * PyTorch/ATen, cuDNN/cuBLAS and flash-attn from Python modules.  These entry points are what a binding
 * for that path binds instead; each one names the reference call site it replaces (paths relative to
 * /root/reference).  The Python host side (octcubem_amd/_lib.py, ctypes) is the only caller.
 *
 * Conventions
 *   - plain pointers and sizes, no torch types; all pointers are DEVICE pointers unless noted
 *   - the caller owns every buffer (incl. workspaces); nothing is allocated or retained by the library
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*); no host synchronisation
 *   - return value: 0 ok, <0 argument error (-1 bad argument, -2 unsupported combination), >0 hipError_t
 *   - "bf16" buffers are raw 16-bit bfloat16 values; row-major; leading dimensions in ELEMENTS
 *   - re-entrant, no thread-local state: forward runs on the main thread, backward on autograd's thread
 */
#ifndef OCTMAE_H_
#define OCTMAE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* The ONE place the ABI number lives: octmae_abi_version() returns it (csrc/probe.hip), octcubem_amd/_lib.py parses it
 * from this header and refuses a library that reports another number, __graft_entry__.build() and the tests compare the two.
 * 4: octmae_attn_bwd_dq_rowconst.  5: octmae_comm_* (RCCL), octmae_attn_bwd_fused + workspace query.  7: octmae_set_option, octmae_scatter_add_rows, octmae_dec_assemble_bwd.
 * 8: octmae_linear_dgrad_delta, octmae_attn_bwd_fused_delta.  9: octmae_wgrad_accum_pair, octmae_wgrad_split_plan.
 * 10: octmae_lp_dtype, octmae_comm_stream, octmae_mt_adamw_fused, octmae_gemm_bf16_ws + workspace arguments.
 * 11: the small-launch GEMM kernel and its split-K workspace (octmae_gemm_split_ws_kib, octmae_gemm_small_plan, "gemm_small");
 *     the stream-K and 16x16x32 variants of round 4 / 5 left the library (octmae_gemm_streamk_*, "gemm_mfma16", "gemm_streamk"). */
#define OCTMAE_ABI_VERSION 11
int octmae_abi_version(void);

/* The 16-bit operand type this library was built for: 0 = bfloat16 (liboctmae.so, the shipped build; BASELINE's headline type),
 * 1 = IEEE half (`make -C octcubem_amd/csrc F16=1` -> liboctmae_f16.so: the same kernels with the type, the conversions and the
 * MFMA opcodes switched in csrc/common.hpp).  The reference's own default arithmetic is fp16 autocast + GradScaler
 * (Pre-training/main_pretrain_oph_joint_2d512_flash_attn.py:259-263, custom_util/misc.py:311-312); the half build is how the
 * north star's 1e-3 on pred / gradients is shown directly (tests/test_gpu_f16_parity.py).  Every "bf16" buffer of this header is
 * a buffer of THIS type; the host side allocates with the matching torch dtype (octcubem_amd/ops.py: BF16). */
int octmae_lp_dtype(void);

/* Kernel-selection switches for same-process A/B measurements and for tests that cover both forms of a kernel (no reference
 * counterpart: the reference's kernels come from its libraries).  Returns the previous value, -1 for an unknown key.
 *   "attn_bwd_hd32_form"   1 (default): one wave per SIMD, 4 x 128 keys per workgroup (csrc/attn_bwd1w.hip)
 *                          0: two waves per SIMD, 8 x 64 keys (csrc/attn_bwd.hip) -- the round-2 kernel
 *   "attn_bwd_hd64_form"   1 (default): one wave per SIMD, 4 x 64 keys per workgroup (csrc/attn_bwd1w64.hip)
 *                          0: two waves per SIMD, 8 x 32 keys (csrc/attn_bwd.hip); the two forms agree bit for bit
 *   "attn_bwd_tail_fused"  1 (default): the one-wave kernels also take the single key past the last full key block and the
 *                          workspace -> bf16 conversion of their (batch, head); 0: the separate launch (same results)
 *   "gemm_small"           1 (default): forward / dgrad launches whose 256 x 256 tiles would leave most CUs idle take the
 *                          small-launch kernel (gemm128d_kernel: 128 x 128 tiles, deterministic split-K) where the cost model of
 *                          csrc/gemm.hip (plan128) prices it faster; 0: never (the choice before round 6).  Bits 12 / 13 (0x1000 /
 *                          0x2000) of octmae_gemm_bf16's `epilogue` argument force that kernel with a 4- / 2-stage ring for one
 *                          call (its k split is then `splitk`, 1 .. 4), bit 14 (0x4000) forbids it (bits 8-10: the other variants)
 *   "gemm_small_launches", "gemm_small_split_launches", "gemm_small_wgrad_launches"   read-only: how many GEMM launches of this process
 *                          took that kernel / took it with a k split / how many weight-gradient pairs took its 128-tile form (tests)
 *   "wgrad_s1_atomic"      the epilogue of an UNSPLIT weight-gradient launch: 2 (default) read-modify-write in batches of 16 registers
 *                          through a buffer descriptor, 1 the fp32 atomics of a split launch, 0 one guarded load + store per register
 *   "wgrad_stagger"        v >= 0 (default 29): split-K weight gradients of >= 8 slices with <= 96 k-tiles each run with slice
 *                          lengths rising by v / 256 k-tiles per output tile of the launch from one slice to the next, so that the
 *                          slices' fp32-atomic epilogues follow one another instead of colliding; 0: equal slices */
int octmae_set_option(const char* key, int value);

/* ---- GEMM with fused epilogues ------------------------------------------------------------------
 * X[a][b] = sum_k A[a][k] * B[b][k],  A has NA rows, B has NB rows, reduction length K.
 * a_kstrided / b_kstrided = 0: operand stored [rows][K] (nn.Linear layout); 1: stored [K][rows].
 * epilogue (output is C[b][a], i.e. NB rows x NA contiguous columns, unless noted):
 *   0  C bf16 = X + bias[a]                       nn.Linear forward        video_vit.py:114-135, timm Mlp fc1/fc2
 *   1  C f32  = X + bias[a]                       decoder_pred             models_mae_joint_res_flash_attn.py:595
 *   2  C bf16 = X + bias, C2 bf16 = gelu(C)       Mlp fc1 + nn.GELU        video_vit.py:174-179 (timm Mlp)
 *   3  C f32  = aux_f32[b][a] + X + bias[a]       proj / fc2 + residual    video_vit.py:182-183
 *   4  C bf16 = X * gelu'(aux_bf16[b][a])         backward of 2 (dgrad of fc2 fused with GELU'); a non-NULL C2 is an fp32 [NA]
 *                                                 vector that receives += the column sums of C (fc1's bias gradient)
 *   5  C f32[a][b] += X  (NA rows x NB columns; split-K over `splitk` workgroup slices, fp32 atomics
 *                         when splitk > 1)        weight gradients (autograd of nn.Linear)
 *                         a non-NULL C2 is an fp32 [NA] vector that receives += sum_k A[a][k]: with A = dY this is the
 *                         bias gradient of the same Linear, taken from the operand tiles the kernel stages anyway
 * bias may be NULL.  Requirements: lda, ldb multiples of 8; NA multiple of 4 (epilogues 0-4). */
int octmae_gemm_bf16(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux,
                     int NA, int NB, int K, int lda, int ldb, int ldc, int ldaux, int a_kstrided, int b_kstrided,
                     int epilogue, int splitk, void* stream);

/* octmae_gemm_bf16 with a split-K workspace lent for this one call (also the trailing `split_ws, split_ws_bytes` of the three
 * fused entry points below; NULL / 0 = no k split).  The reference's cuBLAS / hipBLASLt picks split-K / stream-K kernels by itself
 * for such shapes (nn.Linear forward and backward: video_vit.py:114-135, timm Mlp); its shipped recipe runs ONE volume per GPU
 * (scripts/run_chunks_pretraining_vitl_oph_joint_flash_attn.sh:25-30), where a [1281 x 4096] x [4096 x 1024] Linear is 24 tiles of
 * 256 x 256 for 256 CUs.  Forward / dgrad launches that the cost model prices faster that way run on 128 x 128 tiles
 * (gemm128d_kernel), long reductions on few tiles additionally split 2-4 ways over k: every slice leaves its fp32 partial tile in
 * `split_ws`, the slice that ARRIVES last (an arrival counter per tile) adds the partials in slice order and runs the SAME fused
 * epilogue -- deterministic, no workgroup ever waits for another.  split_ws: octmae_gemm_split_ws_kib() KiB of device memory
 * whose counters (the last 16 KiB) were ZERO when it was first lent; every launch leaves them zero; contents need not be preserved
 * between calls, but two launches that may run CONCURRENTLY (different streams) must not share one workspace. */
int octmae_gemm_split_ws_kib(void);     /* size of that workspace in KiB */
/* host-side arithmetic only (tests of the planner): 1 if a forward / dgrad launch of NA x NB x K on `cus` CUs takes the small-launch
 * kernel under the current "gemm_small" option (*slices = its k split, *stages = its ring depth, 4 or 2), 0 otherwise.
 * have_ws: a split workspace is lent; big_ok: the problem qualifies for the 256-tile kernels */
int octmae_gemm_small_plan(int NA, int NB, int K, int cus, int have_ws, int big_ok, int* slices, int* stages);
int octmae_gemm_bf16_ws(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux,
                        int NA, int NB, int K, int lda, int ldb, int ldc, int ldaux, int a_kstrided, int b_kstrided,
                        int epilogue, int splitk, void* split_ws, long long split_ws_bytes, void* stream);

/* Backward of epilogue 2 together with fc1's bias gradient, WITHOUT atomics (timm Mlp backward: fc2 dgrad, nn.GELU backward,
 * fc1.bias.grad; video_vit.py:174-179 under autograd):
 *   dX bf16 [M][K] = (dY[M][N] @ W[N][K]) * gelu'(pre[M][K]),   bias_grad[K] += column sums of dX   (bias_grad may be NULL).
 * Every 64-row slab of dX leaves its column sums as one row of `ws` (fp32 [octmae_dgelu_colsum_ws_rows(M)][K], caller-owned,
 * need not be initialised) with plain stores; a second launch (octmae_colsum_accum over `ws`) folds the rows into bias_grad:
 * at most 128 atomic adds per address instead of one per 64-row slab -- thousands of slabs adding to one [K] vector serialise
 * per address (~60 ns each: +0.67 ms on the decoder's fc2 at micro-batch 128, +0.13 ms on the encoder's).  Same result as
 * octmae_gemm_bf16(epilogue 4) up to the order of the fp32 additions.  Problems that take the 128-tile register-staged kernel ignore `ws`
 * and sum the columns of dX in a separate pass.  variant: kernel-selection bits as in octmae_linear_resid_rowscale, 0 = automatic. */
int octmae_dgelu_colsum_ws_rows(int M);
int octmae_linear_dgrad_dgelu(const void* W, const void* dY, void* dX, const void* pre, float* ws, float* bias_grad,
                              int M, int N, int K, int ldw, int ldy, int ldx, int ldpre, int variant, void* split_ws,
                              long long split_ws_bytes, void* stream);

/* Two weight gradients over the SAME token rows in one launch (the fc1 / fc2 and the qkv / proj Linears of a Block; backward of
 * video_vit.py:114-135 and timm Mlp under autograd):
 *   gW0 f32 [N0][K0] += dY0[M][N0]^T @ X0[M][K0],   gW1 f32 [N1][K1] += dY1[M][N1]^T @ X1[M][K1]      (dY, X bf16, row-major)
 *   gB0 / gB1: NULL, or f32 [N] += the column sums of dY (the bias gradient, as epilogue 5 of octmae_gemm_bf16 with C2)
 * The output tiles of both problems share one split over M: half the fp32-atomic epilogues of two separate launches and k-loops
 * twice as long.  splitk as in octmae_gemm_bf16.  Short reductions on few tiles (one or two volumes per step: M <= 96 k-tiles of 64 rows)
 * run on 128 x 128 tiles instead when the cost model of csrc/gemm.hip prices that faster ("gemm_small"; gB is then a launch of its own).
 * Returns -2 when either problem does not take the 256-tile kernel (N or K < 256, an operand beyond a 32-bit buffer range): the
 * caller then issues two octmae_gemm_bf16 calls. */
int octmae_wgrad_accum_pair(const void* dY0, const void* X0, float* gW0, float* gB0, int N0, int K0, int ldy0, int ldx0, int ldw0,
                            const void* dY1, const void* X1, float* gW1, float* gB1, int N1, int K1, int ldy1, int ldx1, int ldw1,
                            int M, int splitk, void* stream);

/* Planning arithmetic of the split-K weight gradients, host side only (no GPU is touched; for tests): the number of k slices a launch of
 * `tiles` output tiles over M token rows uses for a requested splitk (-> *slices), the first 64-row k-tile of every slice
 * (bounds[0 .. *slices], bounds[*slices] = ceil(M / 64); room for splitk + 1 ints), and -- the return value -- the length step between
 * neighbouring slices in 1/256 k-tiles (0 = equal slices; see "wgrad_stagger" above).  Negative: argument error. */
int octmae_wgrad_split_plan(int M, int splitk, int tiles, int* slices, int* bounds);

/* The proj dgrad of an attention block together with the attention backward's per-query constant delta (flash-attn's `dsoftmax_sum`,
 * the backward of video_vit.py:130-134 under autograd):
 *   dX bf16 [M][K] = dY[M][N] @ W[N][K],   delta f32 [M][H] = -sum over each head's hd columns of dX * O     (O bf16 [M][K], K = H hd)
 * dX is the gradient of the attention output O; the fused attention backward needs rowsum(dO * O) per query and head, which is a
 * full extra pass over O and dO when computed on its own (0.13 / 0.25 ms per call at the ViT-L shapes) and 8 multiply-adds per
 * lane in this GEMM's epilogue, whose tiles hold whole heads.  delta is computed from the bf16-ROUNDED dX (what the attention
 * backward reads).  Returns -2 when the problem takes neither LDS-transposing kernel (N % 64 != 0, K % 8 != 0; M or K < 256
 * with the small-launch kernel off): the caller then uses octmae_gemm_bf16 and octmae_attn_bwd_fused.  variant: kernel-selection
 * bits (bit 8 forces "not applicable", bits 12-14 as above). */
int octmae_linear_dgrad_delta(const void* W, const void* dY, void* dX, const void* O, float* delta, int M, int N, int K, int ldw,
                              int ldy, int ldx, int ldo, int H, int hd, int variant, void* split_ws, long long split_ws_bytes, void* stream);

/* Stochastic depth (timm DropPath around both Block branches, video_vit.py:181-184 with drop_path > 0; fine-tune drivers use
 * 0.1-0.2): out f32 [M][N] = res + rowscale[m / rows_per_scale] * (X[M][K] @ W[N][K]^T + bias) -- the per-sample keep mask
 * (0 or 1/keep_prob) applied to the branch inside the residual epilogue.  nn.Linear layouts; variant: the kernel-selection bits of `epilogue` above (0x100 ... 0x4000), 0 = automatic. */
int octmae_linear_resid_rowscale(const void* W, const void* X, float* out, const float* bias, const float* res,
                                 const float* rowscale, int rows_per_scale, int N, int M, int K, int ldw, int ldx, int ldout,
                                 int ldres, int variant, void* split_ws, long long split_ws_bytes, void* stream);

/* ---- LayerNorm over the fp32 residual stream ---------------------------------------------------
 * nn.LayerNorm(eps=1e-6): models_mae_joint_res_flash_attn.py:799, video_vit.py:161,172,181-184, :489, :592.
 * fwd: y bf16 = (x - mean) * rstd * gamma + beta; saves mean, rstd (fp32 [M]).
 * bwd: dx f32 = (dres ? dres : 0) + LN'(dy); optional bf16 copy of dx; dgamma/dbeta/dxsum (column sums of
 *      dx = bias gradient of the preceding Linear) are ACCUMULATED (+=) when non-NULL (two-stage, deterministic, through the
 *      caller-provided workspace partial_ws).  D % 4 == 0, D <= 2048. */
int octmae_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd,
                         int M, int D, float eps, void* stream);
int octmae_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* gamma,
                         const float* dres, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dxsum,
                         float* partial_ws, int M, int D, void* stream);
/* number of floats `partial_ws` must hold for (M, D) */
int octmae_layernorm_bwd_ws_floats(int M, int D);

/* ---- attention -----------------------------------------------------------------------------------
 * softmax(q k^T * scale) v, non-causal, no dropout: video_vit.py:130-134 (flash path: flash_attn MHA,
 * models_mae_joint_res_flash_attn.py:131-149).  qkv bf16 [B][N][3][H][HD]; o, dout bf16 [B][N][H][HD];
 * lse f32 [B][H][N] (natural log); rowc_ws f32 [2][B][H][N] workspace; dqkv bf16 like qkv.  HD in {32, 64}.
 * flag_ws: one int of device workspace, or NULL.  Non-NULL enables the optimistic forward: a kernel without running-max
 * tracking runs first and raises *flag_ws if any softmax row sum is not a finite positive number; the online-max kernel is
 * always enqueued behind it and returns immediately unless the flag is set (no host synchronisation). */
int octmae_attn_fwd(const void* qkv, void* o, float* lse, int* flag_ws, int B, int N, int H, int HD, float scale, void* stream);
int octmae_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* rowc_ws, void* dqkv, int B,
                    int N, int H, int HD, float scale, void* stream);
/* octmae_attn_bwd = octmae_attn_bwd_dq_rowconst + octmae_attn_bwd_dkv: the dQ kernel computes the per-query constants the
 * gradient kernels start their accumulators from, rowc[0] = -lse * log2(e), rowc[1] = -rowsum(dO * O), for its own query rows
 * and WRITES them to rowc [2][B][H][N] for the dK/dV kernel.  The separate pre-pass (rowconst) followed by octmae_attn_bwd_dq,
 * which only reads rowc, is the same computation in three launches. */
int octmae_attn_bwd_dq_rowconst(const void* qkv, const void* o, const void* dout, const float* lse, float* rowc, void* dqkv, int B,
                                int N, int H, int HD, float scale, void* stream);
int octmae_attn_bwd_rowconst(const void* o, const void* dout, const float* lse, float* rowc, int B, int N, int H, int HD,
                             void* stream);
int octmae_attn_bwd_dq(const void* qkv, const void* dout, const float* rowc, void* dqkv, int B, int N, int H, int HD, float scale,
                       void* stream);
int octmae_attn_bwd_dkv(const void* qkv, const void* dout, const float* rowc, void* dqkv, int B, int N, int H, int HD, float scale,
                        void* stream);

/* Fused backward (csrc/attn_bwd.hip): S, dP and exp once per score for all three gradients (5 matrix products and 1 exp per
 * score instead of 7 and 2), dQ summed over key blocks in an fp32 workspace by one workgroup per (batch, head) in program
 * order -- no atomics, bit-reproducible.  Same inputs and outputs as octmae_attn_bwd; `ws` is a device workspace of
 * octmae_attn_bwd_fused_ws_kib(B, N, H, HD) KiB (dQ fp32 [B][H][N][HD] + the padded per-query constants), contents
 * irrelevant on entry.  Three launches: per-query constants, full key blocks (512 keys at HD 32, 256 at HD 64), remaining keys
 * + conversion of dQ to bf16. */
int octmae_attn_bwd_fused_ws_kib(int B, int N, int H, int HD);
int octmae_attn_bwd_fused(const void* qkv, const void* o, const void* dout, const float* lse, void* ws, void* dqkv, int B, int N,
                          int H, int HD, float scale, void* stream);
/* octmae_attn_bwd_fused with delta (fp32 [B * N][H], from octmae_linear_dgrad_delta) supplied instead of O: the per-query constants are
 * transposed and padded from it, no pass over O and dO.  Same workspace, same outputs. */
int octmae_attn_bwd_fused_delta(const void* qkv, const void* dout, const float* lse, const float* delta, void* ws, void* dqkv, int B,
                                int N, int H, int HD, float scale, void* stream);

/* ---- random masking indices ---------------------------------------------------------------------
 * MaskedAutoencoderViT.random_masking index part, models_mae_joint_res_flash_attn.py:349-369:
 * ids_shuffle = argsort(noise) with ties -> lower index, ids_restore = its inverse, ids_keep = first len_keep,
 * mask = 0 keep / 1 remove in original order.  int64 outputs (torch index dtype).  ids_shuffle may be NULL.
 * L <= 16384. */
int octmae_random_masking_ids(const float* noise, long long* ids_restore, long long* ids_keep, long long* ids_shuffle,
                              float* mask, int B, int L, int len_keep, void* stream);

/* ---- token plumbing ------------------------------------------------------------------------------ */
int octmae_cast_f32_bf16(const float* src, void* dst_bf16, long long n, void* stream);
/* dst bf16 [R][D] = rowscale[r / rows_per_scale] * src f32 [R][D]: the gradient entering a stochastic-depth branch */
int octmae_cast_rowscale_f32_bf16(const float* src, const float* rowscale, void* dst_bf16, long long R, int D,
                                  int rows_per_scale, void* stream);
/* out[c] += sum_r in[r][c]  (nn.Linear bias gradient) */
int octmae_colsum_accum(const void* in, int in_is_bf16, float* out, int M, int N, int ld, void* stream);
/* im2col of the kept tokens for PatchEmbed's Conv3d(k = s = (tp,p,p)), video_vit.py:70-83 + the gather at
 * models_mae_joint_res_flash_attn.py:363: out bf16 [B*nkeep][C*tp*p*p] in conv-weight order (c,u,py,px).
 * ids: [B][nkeep] int64 (ids_is_i64=1) or int32, NULL = tokens 0..nkeep-1. */
int octmae_patch_gather(const float* imgs, const void* ids, int ids_is_i64, void* out_bf16, int B, int C, int T, int H,
                        int W, int tp, int p, int nkeep, void* stream);
/* encoder input: cls concat + gathered sep pos-embed add, models_mae_joint_res_flash_attn.py:409-478 */
int octmae_enc_assemble(const void* tok_bf16, const float* pos, const float* cls, const float* pos_cls,
                        const long long* ids_keep, float* x, int B, int nkeep, int D, void* stream);
/* decoder input: mask tokens + un-shuffle + cls + pos-embed, models_mae_joint_res_flash_attn.py:515-573.
 * emb_has_cls = 1 (2-D MAE, OCTCube/models_mae.py:175-178): emb has 1 + nkeep rows per sample, row 0 is the cls row. */
int octmae_dec_assemble(const void* emb_bf16, const float* mask_token, const float* dpos, const float* dcls,
                        const float* dpos_cls, const long long* ids_restore, float* x, int B, int nkeep, int L, int D,
                        int emb_has_cls, void* stream);
/* out bf16[b*n+i][:] = src f32[b][1+ids[b][i]][:]  (backward of both assemblies w.r.t. the token rows) */
int octmae_gather_rows_cast(const float* src, const long long* ids, void* out_bf16, int B, int n, int src_rows, int D,
                            void* stream);
/* backward of the keep-gather w.r.t. the positional table (autograd of torch.gather at models_mae_joint_res_flash_attn.py:442-447;
 * ATen: index_add_): out f32 [L][D] (+)= sum over samples b that kept token l (ids_restore[b][l] < nkeep) of
 * src f32 [b][row0 + ids_restore[b][l]][:], src has src_rows rows per sample.  Deterministic (ascending b), no atomics. */
int octmae_scatter_add_rows(const float* src, const long long* ids_restore, float* out, int B, int nkeep, int L, int D,
                            int src_rows, int row0, int accumulate, void* stream);
/* backward of the decoder assembly w.r.t. decoder_pos_embed and mask_token in one pass over dx f32 [B][1 + L][D]
 * (models_mae_joint_res_flash_attn.py:515-573 under autograd): ddpos [L][D] = sum_b dx[b][1 + l], dmask_part [L][D] = the same sum
 * over the samples in which token l was masked; the mask-token gradient is the column sum of dmask_part. */
int octmae_dec_assemble_bwd(const float* dx, const long long* ids_restore, float* ddpos, float* dmask_part, int B, int nkeep, int L,
                            int D, void* stream);
/* fused patchify + per-token MSE, models_mae_joint_res_flash_attn.py:289-314, :613-650.  pred f32 [B][L+1][PD]
 * (row 0 = cls, ignored); loss_tok f32 [B][L].  frame_idx int32 [pred_t_dim] or NULL (identity). */
int octmae_mse_fwd(const float* pred, const float* imgs, const int* frame_idx, float* loss_tok, int B, int C, int T, int H,
                   int W, int u_sz, int p, int L, int norm_pix, void* stream);
/* dpred bf16 [B][L+1][PD] = *coef * mask * (pred - target), cls rows zero (backward of :649-663) */
int octmae_mse_bwd(const float* pred, const float* imgs, const int* frame_idx, const float* mask, const float* coef,
                   void* dpred_bf16, int B, int C, int T, int H, int W, int u_sz, int p, int L, int norm_pix, void* stream);

/* ---- optimizer side --------------------------------------------------------------------------------
 * Multi-tensor tables: tensor_table = device array of {float* p, g, m, v; int64 n}; chunk_tensor/chunk_off map
 * each chunk of octmae_mt_chunk_elems() elements to (tensor, element offset).
 * get_grad_norm_ / clip_grad_norm_: custom_util/misc.py:328-336, :356-373.  AdamW: main_pretrain_oph_joint_2d512_flash_attn.py:451. */
int octmae_mt_chunk_elems(void);
int octmae_mt_sumsq(const void* tensor_table, const int* chunk_tensor, const long long* chunk_off, int nchunks, float* sumsq,
                    void* stream);
int octmae_mt_finish_norm(const float* sumsq, int ntensors, float max_norm, float* out_norm, float* out_coef, void* stream);
int octmae_mt_adamw(const void* tensor_table, const int* chunk_tensor, const long long* chunk_off, int nchunks,
                    const float* gscale, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    void* stream);
/* The same update with two optional by-products of the one pass it makes over the parameters and gradients (both may be NULL):
 *   lp_table  device array of one pointer per tensor of the table: where to write the 16-bit operand copy (octmae_lp_dtype) of the
 *             UPDATED parameter, NULL entries skipped -- replaces the per-forward cast of the whole parameter arena
 *             (the bf16 weights autocast re-derives per forward in the reference, engine_pretrain.py:110);
 *   sumsq     fp32 [ntensors], += sum of squares of the RAW gradient per tensor (before gscale): get_grad_norm_
 *             (custom_util/misc.py:356-373) without its own pass over the gradients; finish with octmae_mt_finish_norm. */
int octmae_mt_adamw_fused(const void* tensor_table, const int* chunk_tensor, const long long* chunk_off, int nchunks,
                          const float* gscale, void* const* lp_table, float* sumsq, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, void* stream);

/* ---- data-parallel exchange over RCCL (xGMI) -----------------------------------------------------------
 * What the reference gets from torch.distributed's NCCL backend on this path:
 *   init_process_group("nccl") + barrier                    Pre-training/custom_util/misc.py:283-296
 *   DistributedDataParallel: bucketed gradient mean          Pre-training/main_pretrain_oph_joint_2d512_flash_attn.py:434-439
 *   DDP's constructor broadcast of the parameters            (same wrap site)
 *   all_reduce_mean of the logged loss                       Pre-training/custom_util/misc.py:622-630
 *   open_clip's feature all-gather (+ its reduce-scatter backward)   retinal-COEM/src/open_clip/loss.py:51-63
 * One process per GPU.  Bootstrap: rank 0 calls octmae_comm_unique_id() into a HOST buffer of OCTMAE_COMM_ID_BYTES, the host
 * side hands those bytes to every rank (the launcher's key-value store: MASTER_ADDR/MASTER_PORT of torchrun), every rank calls
 * octmae_comm_init().  The handle is opaque and owns one communication stream; it is the one object this library retains
 * between calls (freed by octmae_comm_destroy).  Every *_async call enqueues ONE collective on the communication stream,
 * ordered behind everything `after_stream` (the caller's compute stream) holds at the time of the call, and returns without
 * synchronising the host; octmae_comm_wait() makes `stream` wait for every collective enqueued so far.  In-place (`buf`)
 * unless send/recv are given.  Calls may come from any host thread (autograd's worker thread reports finished gradient
 * slices); they are serialised per communicator.  Buffers are DEVICE pointers and must stay alive until a later
 * octmae_comm_wait()'s stream has passed it.
 * Return codes: as above, plus -3 = librccl not found at run time, and 10000 + ncclResult_t for an RCCL error. */
#define OCTMAE_COMM_ID_BYTES 128
enum { OCTMAE_COMM_F32 = 0, OCTMAE_COMM_BF16 = 1, OCTMAE_COMM_F64 = 2, OCTMAE_COMM_F16 = 3 /* IEEE half: the 16-bit tensors of liboctmae_f16.so */ };
enum { OCTMAE_COMM_SUM = 0, OCTMAE_COMM_AVG = 1, OCTMAE_COMM_MAX = 2 };
int octmae_comm_available(void);                       /* 1 when librccl could be loaded, 0 otherwise (never fails) */
int octmae_comm_unique_id(void* id_bytes_host);
int octmae_comm_init(void** comm_out, const void* id_bytes_host, int rank, int world, int device);
int octmae_comm_destroy(void* comm);
int octmae_comm_rank(void* comm);                      /* plain values, not status codes; -1 for a NULL handle */
int octmae_comm_world(void* comm);
int octmae_comm_allreduce_async(void* comm, void* buf, long long count, int dtype, int op, void* after_stream);
int octmae_comm_broadcast_async(void* comm, void* buf, long long count, int dtype, int root, void* after_stream);
int octmae_comm_allgather_async(void* comm, const void* send, void* recv, long long count_per_rank, int dtype,
                                void* after_stream);
int octmae_comm_reduce_scatter_async(void* comm, const void* send, void* recv, long long count_per_rank, int dtype, int op,
                                     void* after_stream);
int octmae_comm_wait(void* comm, void* stream);
/* The communication stream itself (a hipStream_t written to *stream_out), for MEASUREMENT only: the host side records timing
 * events on it around a collective (bench.py's exposed-communication fields; DDP offers the same through its logging hooks,
 * torch/nn/parallel/distributed.py `_get_ddp_logging_data`).  Nothing may be enqueued on it that a collective would wait for. */
int octmae_comm_stream(void* comm, void** stream_out);

/* ---- hardware layout probes (tests only: pin the MFMA / ds_read_b64_tr_b16 lane maps the kernels assume) */
int octmae_probe_mfma32(const void* a_frag_bf16, const void* b_frag_bf16, float* d_regs, void* stream);
int octmae_probe_trread(const void* tile_bf16_16x16, void* out_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCTMAE_H_ */
