"""TEST INFRASTRUCTURE (see oracle/__init__): the WHOLE 3-D MAE training step -- models_mae_joint_res_flash_attn.py:374-680
(forward_encoder, forward_decoder, forward_loss, forward) under autograd -- evaluated in float64 with a bfloat16 rounding
inserted at exactly the points where the HIP path (octcubem_amd.models_mae -> ops.* -> csrc/*.hip) rounds, forward AND backward.
It extends oracle/bf16_points.py (one Block) to everything around the Blocks:

  forward   patches = bf(pixels of the kept tokens) ; tok = bf(patches bf(Wpe)^T + b)                       (PatchEmbedFn)
            x = tok + pos[ids_keep] with the cls row (fp32: EncAssembleFn) ; Blocks (bf16_points.block_forward)
            latent = bf(LN(x))[:, 1:] ; emb = bf(latent bf(Wde)^T + b)                                       (LayerNormFn, LinearFn)
            x = [cls ; un-shuffled emb / mask tokens] + pos (fp32: DecAssembleFn) ; Blocks ; y = bf(LN(x))
            pred = y bf(Wpred)^T + b in fp32 (no rounding) ; per-token MSE and the masked mean in fp32
  backward  dpred = bf(dL/dpred) (octmae_mse_bwd writes bf16) ; every gradient entering a GEMM is bf16: the input gradients of
            decoder_pred / decoder_embed (linear_dgrad writes bf16), demb and dtok (octmae_gather_rows_cast), the bf16 copies
            inside the Blocks ; LayerNorm backward, positional / cls / mask-token sums, weight and bias gradients in fp32.

The float64 autograd graph carries two tiny Functions -- RoundFwd (value rounded, gradient passed) and RoundBwd (value passed,
gradient rounded) -- at those points, and one Function per Block that calls the hand-written forward / backward model of
bf16_points.py (the attention kernels round P and dS inside, which autograd cannot express).  What remains between this model and
the HIP result is accumulation order, the hardware exp2 and rare rounding flips at bf16 ties: tests/test_gpu_rounding_model.py
requires <= 1e-3 for the loss, pred and every gradient tensor (north star: "within 1e-3 rel"), while the plain fp32 oracle
(oracle/mae3d_ref.py) differs from both by the 5e-3 ... 2e-2 that bf16 operands cost element-wise."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import bf16_points as R
from . import mae3d_ref as O

D = torch.float64


def _bf(x):
    return R.bf(x)          # the one switchable rounding function (identity under bf16_points.exact_arithmetic())


class RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


BLOCK_KEYS = ["norm1.weight", "norm1.bias", "attn.q.weight", "attn.q.bias", "attn.k.weight", "attn.k.bias", "attn.v.weight",
              "attn.v.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
              "mlp.fc2.weight", "mlp.fc2.bias"]


class BlockRP(torch.autograd.Function):
    """One Block: bf16_points.block_forward / block_backward (float64 with the HIP roundings, fused attention backward for
    head_dim 32 and 64 as ops.ATTN_BWD_FUSED has it)."""

    @staticmethod
    def forward(ctx, x, num_heads, eps, *params):
        P = dict(zip(BLOCK_KEYS, params))
        x3, saved = R.block_forward(P, x, num_heads, eps)
        ctx.saved = saved
        return x3

    @staticmethod
    def backward(ctx, dx3):
        dx, G = R.block_backward(ctx.saved, dx3, fused_bwd=True)
        ctx.saved = None
        return (dx, None, None) + tuple(G[k] for k in BLOCK_KEYS)


def _ln_bf(x, w, b, eps):
    """LayerNormFn: y = bf16(LN(x)); its backward receives a bf16 gradient (dyb = bf16(dy))."""
    y = F.layer_norm(x, (x.shape[-1],), w, b, eps)
    return RoundBwd.apply(RoundFwd.apply(y))


def _linear_bf(x, w, b, out_f32=False):
    """LinearFn: bf16 operands; bf16 (or fp32) output; input gradient written as bf16; dy used as bf16."""
    y = RoundBwd.apply(x) @ RoundFwd.apply(w).T + b
    if not out_f32:
        y = RoundFwd.apply(y)
    return RoundBwd.apply(y)


def forward_backward(P: Dict[str, torch.Tensor], imgs: torch.Tensor, cfg: O.MAEConfig, mask_ratio: float, noise: torch.Tensor,
                     trace: Optional[dict] = None):
    """Returns (loss, pred [N, L, PD], mask, ids_restore, grads by state_dict key), all float64 (indices int64).
    `trace` (optional dict) receives the intermediate activations by stage name (diagnostics: tests/tool_rp_model_trace.py)."""
    def tr(name, v):
        if trace is not None:
            trace[name] = v.detach().clone()

    Pg = {k: v.detach().to(D).clone().requires_grad_(True) for k, v in P.items()}
    imgs = imgs.to(D)
    N, Cc, T, Hh, Ww = imgs.shape
    high_res = Hh == cfg.hr_grid[1] * cfg.patch_size
    pe = "high_res_patch_embed" if high_res else "patch_embed"
    tp, p = cfg.t_patch_size, cfg.patch_size
    t_actual = T // tp
    hw = cfg.hr_grid if high_res else cfg.grid
    L = t_actual * hw[1] * hw[2]
    _, ids_restore, ids_keep, mask = O.masking_indices(noise, mask_ratio)
    nkeep = ids_keep.shape[1]
    # ---- patch embedding of the kept tokens (Conv3d k = s = (tp, p, p) as a GEMM over [c, tp, p, p]-ordered patch vectors)
    x = imgs.reshape(N, Cc, t_actual, tp, hw[1], p, hw[2], p).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(N, L, Cc * tp * p * p)
    patches = _bf(torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, x.shape[-1])))
    w_pe = Pg[f"{pe}.proj.weight"].reshape(cfg.embed_dim, -1)
    tok = RoundBwd.apply(RoundFwd.apply(patches @ RoundFwd.apply(w_pe).T + Pg[f"{pe}.proj.bias"]))      # [N, nkeep, D]
    # ---- encoder assembly (fp32 in the HIP path: exact here)
    pos = O.sep_pos_table(Pg["pos_embed_spatial"], Pg["pos_embed_temporal"], cfg, high_res, t_actual).expand(N, -1, -1)
    pos = torch.gather(pos, 1, ids_keep.unsqueeze(-1).expand(-1, -1, cfg.embed_dim))
    x = torch.cat([(Pg["cls_token"] + Pg["pos_embed_class"]).expand(N, -1, -1), tok + pos], 1)
    tr("tok", tok); tr("enc_in", x)
    for i in range(cfg.depth):
        x = BlockRP.apply(x, cfg.num_heads, cfg.ln_eps, *[Pg[f"blocks.{i}.{k}"] for k in BLOCK_KEYS])
        tr(f"blocks.{i}", x)
    latent = _ln_bf(x, Pg["norm.weight"], Pg["norm.bias"], cfg.ln_eps)[:, 1:, :]
    # ---- decoder
    emb = _linear_bf(latent, Pg["decoder_embed.weight"], Pg["decoder_embed.bias"])
    Dd = cfg.decoder_embed_dim
    x_ = torch.cat([emb, Pg["mask_token"].expand(N, L - nkeep, -1)], 1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).expand(-1, -1, Dd))
    dpos = O.sep_pos_table(Pg["decoder_pos_embed_spatial"], Pg["decoder_pos_embed_temporal"], cfg, high_res, t_actual)
    x = torch.cat([(Pg["decoder_cls_token"] + Pg["decoder_pos_embed_class"]).expand(N, -1, -1), x_ + dpos], 1)
    tr("latent", latent); tr("emb", emb); tr("dec_in", x)
    for i in range(cfg.decoder_depth):
        x = BlockRP.apply(x, cfg.decoder_num_heads, cfg.ln_eps, *[Pg[f"decoder_blocks.{i}.{k}"] for k in BLOCK_KEYS])
        tr(f"decoder_blocks.{i}", x)
    y = _ln_bf(x, Pg["decoder_norm.weight"], Pg["decoder_norm.bias"], cfg.ln_eps)
    pred = _linear_bf(y, Pg["decoder_pred.weight"], Pg["decoder_pred.bias"], out_f32=True)[:, 1:, :]
    loss, _ = O.forward_loss(imgs, pred, mask.to(D), cfg)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    return loss.detach(), pred.detach(), mask, ids_restore, grads
