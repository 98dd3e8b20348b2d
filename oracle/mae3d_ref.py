"""CPU oracle for the 3-D MAE hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  ``octcubem_amd`` never does.

It is a plain-PyTorch fp32, functional (params-dict) restatement of the reference's
*non-flash* 3-D masked auto-encoder, following

  * Pre-training/models_mae_joint_res_flash_attn.py   (model: masking, pos-embeds, decoder, loss)
  * Pre-training/custom_util/video_vit.py              (PatchEmbed / Attention / Block)
  * timm ``Mlp`` (fc1 -> exact-erf GELU -> fc2; third-party, un-vendored; restated here)
  * Pre-training/custom_util/misc.py:308-373,678-696   (grad-norm, weight-decay grouping)
  * Pre-training/custom_util/lr_sched.py               (warm-up + half-cosine schedule)

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference from
/root/reference (build container only), runs it on seeded inputs and writes the golden
vectors in ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement against
those vectors (<=1e-5 rel).  The parameter dict uses exactly the reference's ``state_dict`` keys.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class MAEConfig:
    """Constructor arguments of the reference model (models_mae_joint_res_flash_attn.py:32-62)."""
    input_size: int = 256
    patch_size: int = 16
    in_chans: int = 1
    embed_dim: int = 1024
    depth: int = 24
    num_heads: int = 16
    decoder_embed_dim: int = 512
    decoder_depth: int = 8
    decoder_num_heads: int = 16
    mlp_ratio: float = 4.0
    num_frames: int = 60
    t_patch_size: int = 3
    pred_t_dim: int = 60
    high_res_input_size: int = 512
    norm_pix_loss: bool = False
    sep_pos_embed: bool = True
    cls_embed: bool = True
    ln_eps: float = 1e-6          # factories pass partial(nn.LayerNorm, eps=1e-6)  (:799,:827)

    # derived (video_vit.py:44-67)
    @property
    def grid(self) -> Tuple[int, int, int]:
        return (self.num_frames // self.t_patch_size, self.input_size // self.patch_size,
                self.input_size // self.patch_size)

    @property
    def hr_grid(self) -> Tuple[int, int, int]:
        return (self.num_frames // self.t_patch_size, self.high_res_input_size // self.patch_size,
                self.high_res_input_size // self.patch_size)

    @property
    def num_patches(self) -> int:
        t, h, w = self.grid
        return t * h * w

    @property
    def t_pred_patch_size(self) -> int:
        return self.t_patch_size * self.pred_t_dim // self.num_frames   # (:70)

    @property
    def patch_dim(self) -> int:
        return self.t_pred_patch_size * self.patch_size ** 2 * self.in_chans


VIT_L = MAEConfig()


def param_shapes(cfg: MAEConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys and shapes of the reference non-flash model (constructor, :75-241)."""
    D, Dd = cfg.embed_dim, cfg.decoder_embed_dim
    tp, p, c = cfg.t_patch_size, cfg.patch_size, cfg.in_chans
    T, _, _ = cfg.grid
    _, hh, hw = cfg.hr_grid
    s: Dict[str, Tuple[int, ...]] = {}
    for pe in ("patch_embed", "high_res_patch_embed"):
        s[f"{pe}.proj.weight"] = (D, c, tp, p, p)
        s[f"{pe}.proj.bias"] = (D,)
    s["cls_token"] = (1, 1, D)
    s["decoder_cls_token"] = (1, 1, Dd)
    s["pos_embed_spatial"] = (1, hh * hw, D)
    s["pos_embed_temporal"] = (1, T, D)
    s["pos_embed_class"] = (1, 1, D)

    def block(prefix, dim):
        hid = int(dim * cfg.mlp_ratio)
        s[f"{prefix}.norm1.weight"] = (dim,)
        s[f"{prefix}.norm1.bias"] = (dim,)
        for n in ("q", "k", "v", "proj"):
            s[f"{prefix}.attn.{n}.weight"] = (dim, dim)
            s[f"{prefix}.attn.{n}.bias"] = (dim,)
        s[f"{prefix}.norm2.weight"] = (dim,)
        s[f"{prefix}.norm2.bias"] = (dim,)
        s[f"{prefix}.mlp.fc1.weight"] = (hid, dim)
        s[f"{prefix}.mlp.fc1.bias"] = (hid,)
        s[f"{prefix}.mlp.fc2.weight"] = (dim, hid)
        s[f"{prefix}.mlp.fc2.bias"] = (dim,)

    for i in range(cfg.depth):
        block(f"blocks.{i}", D)
    s["norm.weight"] = (D,)
    s["norm.bias"] = (D,)
    s["decoder_embed.weight"] = (Dd, D)
    s["decoder_embed.bias"] = (Dd,)
    s["mask_token"] = (1, 1, Dd)
    s["decoder_pos_embed_spatial"] = (1, hh * hw, Dd)
    s["decoder_pos_embed_temporal"] = (1, T, Dd)
    s["decoder_pos_embed_class"] = (1, 1, Dd)
    for i in range(cfg.decoder_depth):
        block(f"decoder_blocks.{i}", Dd)
    s["decoder_norm.weight"] = (Dd,)
    s["decoder_norm.bias"] = (Dd,)
    s["decoder_pred.weight"] = (cfg.patch_dim, Dd)
    s["decoder_pred.bias"] = (cfg.patch_dim,)
    return s


def init_params(cfg: MAEConfig, seed: int = 0, bias_std: float = 0.0) -> Dict[str, torch.Tensor]:
    """Deterministic init with the reference's distributions (initialize_weights, :249-287):
    xavier-uniform matrices, LN 1/0, trunc-normal(0.02) pos-embeds / cls, normal(0.02) mask token.
    ``bias_std`` > 0 perturbs biases / LN affine so that parity tests exercise them (the reference
    zero-inits biases).  Not the reference's RNG stream -- tests load the SAME dict into both sides."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith("proj.weight") and len(shp) == 5 or (k.endswith(".weight") and len(shp) == 2):
            fan_out = shp[0]
            fan_in = int(math.prod(shp[1:]))
            a = math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shp, generator=g) * 2 - 1) * a
        elif k.endswith(".weight") and len(shp) == 1:      # LayerNorm gamma
            t = torch.ones(shp) + bias_std * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            t = bias_std * torch.randn(shp, generator=g)
        elif k == "mask_token":
            t = 0.02 * torch.randn(shp, generator=g)
        else:                                               # cls tokens, pos embeds
            t = (0.02 * torch.randn(shp, generator=g)).clamp_(-2.0, 2.0)
        out[k] = t.float()
    return out


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def patch_embed(x, w, b, cfg: MAEConfig):
    """video_vit.py:74-83 -- Conv3d(k=s=(tp,p,p)) then 'ncts->ntsc'.  Returns (N, T*S, C)."""
    k = (cfg.t_patch_size, cfg.patch_size, cfg.patch_size)
    y = F.conv3d(x, w, b, stride=k).flatten(3)            # N, C, T, S
    y = torch.einsum("ncts->ntsc", y)
    N, T, S, C = y.shape
    return y.reshape(N, T * S, C)


def stable_argsort_rows(noise: torch.Tensor) -> torch.Tensor:
    """argsort with ties broken by lower index.  The reference calls the default (unstable)
    torch.argsort (:356); on tie-free rows both agree exactly, which is what the bit-exact
    fixtures use."""
    return torch.argsort(noise, dim=1, stable=True)


def random_masking(x, mask_ratio: float, noise: Optional[torch.Tensor]):
    """models_mae_joint_res_flash_attn.py:336-372 (pre_mask branch unreachable from forward)."""
    N, L, D = x.shape
    len_keep = int(L * (1 - mask_ratio))
    if noise is None:
        noise = torch.rand(N, L) if mask_ratio > 0 else torch.arange(L).expand(N, L)
    ids_shuffle = stable_argsort_rows(noise)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :len_keep]
    x_masked = torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    mask = torch.ones(N, L)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return x_masked, mask, ids_restore, ids_keep


def masking_indices(noise: torch.Tensor, mask_ratio: float):
    """Index-only part of random_masking: (ids_shuffle, ids_restore, ids_keep, mask)."""
    N, L = noise.shape
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = stable_argsort_rows(noise)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    mask = torch.ones(N, L)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return ids_shuffle, ids_restore, ids_shuffle[:, :len_keep], mask


def sep_pos_table(spatial, temporal, cfg: MAEConfig, high_res: bool, t_actual: Optional[int] = None):
    """:415-441 / :532-557 -- bicubic 32x32 -> 16x16 of the spatial table, tiled over T, plus
    repeat_interleave'd temporal table.  Returns (1, T*h*w, C).  When the token grid has a single
    temporal slot (the 2-D/512 branch, T == 1) the reference adds NO temporal table
    (temp_pos_emb_type == 'none', :402-404,:437-440,:522-524,:554-557)."""
    T, h, w = cfg.grid
    _, hh, hw = cfg.hr_grid
    C = spatial.shape[-1]
    if not high_res:
        pe = F.interpolate(spatial.view(1, hh, hw, C).permute(0, 3, 1, 2), [h, w],
                           mode="bicubic", align_corners=False)
        pe = pe.permute(0, 2, 3, 1).reshape(1, h * w, C)
        ph, pw = h, w
    else:
        pe, ph, pw = spatial, hh, hw
    if t_actual == 1:
        return pe
    return pe.repeat(1, T, 1) + torch.repeat_interleave(temporal, ph * pw, dim=1)


def attention(x, P, prefix, num_heads):
    """video_vit.py:112-138."""
    B, N, C = x.shape
    hd = C // num_heads

    def lin(n):
        return F.linear(x, P[f"{prefix}.{n}.weight"], P[f"{prefix}.{n}.bias"])

    q = lin("q").reshape(B, N, num_heads, hd).permute(0, 2, 1, 3)
    k = lin("k").reshape(B, N, num_heads, hd).permute(0, 2, 1, 3)
    v = lin("v").reshape(B, N, num_heads, hd).permute(0, 2, 1, 3)
    attn = (q @ k.transpose(-2, -1)) * hd ** -0.5
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, P[f"{prefix}.proj.weight"], P[f"{prefix}.proj.bias"])


def block(x, P, prefix, num_heads, eps, drop_scales=None, final_residual=True):
    """video_vit.py:181-184 with timm Mlp (fc1 -> GELU(erf) -> fc2).
    final_residual=False: return the MLP branch only -- the ``hidden_states`` flash-attn 2.5.2's prenorm Block
    (flash_attn/modules/block.py; un-vendored, CUDA only: PARITY UNPINNED for this mode) returns next to ``residual``,
    which is all the reference's flash path keeps after its last block (models_mae_joint_res_flash_attn.py:480-489).
    drop_scales = (s1, s2): per-sample stochastic-depth factors, i.e. what timm's DropPath multiplies each branch by in
    training (floor(keep + U[0,1)) / keep, one draw per sample and branch); None = drop_path 0 / eval."""
    C = x.shape[-1]
    s1, s2 = (None, None) if drop_scales is None else drop_scales
    h = F.layer_norm(x, (C,), P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], eps)
    a = attention(h, P, f"{prefix}.attn", num_heads)
    x = x + (a if s1 is None else a * s1.view(-1, 1, 1))
    h = F.layer_norm(x, (C,), P[f"{prefix}.norm2.weight"], P[f"{prefix}.norm2.bias"], eps)
    h = F.linear(h, P[f"{prefix}.mlp.fc1.weight"], P[f"{prefix}.mlp.fc1.bias"])
    h = F.gelu(h)
    h = F.linear(h, P[f"{prefix}.mlp.fc2.weight"], P[f"{prefix}.mlp.fc2.bias"])
    if not final_residual:
        return h
    return x + (h if s2 is None else h * s2.view(-1, 1, 1))


def patchify(imgs, cfg: MAEConfig):
    """:289-314 -- (N,C,T,H,W) -> (N, t*h*w, u*p*p*C) with per-token order (u,p,q,c)."""
    N, C, T, H, W = imgs.shape
    p, u = cfg.patch_size, cfg.t_pred_patch_size
    assert W % p == 0 and H % p == 0 and T % u == 0
    h, w, t = H // p, W // p, T // u
    x = imgs.reshape(N, C, t, u, h, p, w, p)
    x = torch.einsum("nctuhpwq->nthwupqc", x)
    return x.reshape(N, t * h * w, p * p * u * C)


# --------------------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------------------
def forward_encoder(P, imgs, mask_ratio, noise, cfg: MAEConfig, flash_compat=False):
    """:374-497."""
    H = imgs.shape[-2]
    high_res = H == cfg.hr_grid[1] * cfg.patch_size
    pe = "high_res_patch_embed" if high_res else "patch_embed"
    x = patch_embed(imgs, P[f"{pe}.proj.weight"], P[f"{pe}.proj.bias"], cfg)
    N, _, C = x.shape
    x, mask, ids_restore, ids_keep = random_masking(x, mask_ratio, noise)
    x = torch.cat((P["cls_token"].expand(N, -1, -1), x), dim=1)             # cls first, (:409-412)
    t_actual = imgs.shape[2] // cfg.t_patch_size
    pos = sep_pos_table(P["pos_embed_spatial"], P["pos_embed_temporal"], cfg, high_res, t_actual)
    pos = pos.expand(N, -1, -1)
    pos = torch.gather(pos, 1, ids_keep.unsqueeze(-1).expand(-1, -1, C))
    pos = torch.cat([P["pos_embed_class"].expand(N, -1, -1), pos], 1)
    x = x + pos
    for i in range(cfg.depth):
        x = block(x, P, f"blocks.{i}", cfg.num_heads, cfg.ln_eps, final_residual=not (flash_compat and i == cfg.depth - 1))
    x = F.layer_norm(x, (C,), P["norm.weight"], P["norm.bias"], cfg.ln_eps)
    return x[:, 1:, :], mask, ids_restore


def forward_decoder(P, x, ids_restore, cfg: MAEConfig, high_res=False, flash_compat=False):
    """:499-606."""
    N = x.shape[0]
    x = F.linear(x, P["decoder_embed.weight"], P["decoder_embed.bias"])
    C = x.shape[-1]
    L = ids_restore.shape[1]
    mask_tokens = P["mask_token"].repeat(N, L - x.shape[1], 1)
    x_ = torch.cat([x, mask_tokens], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).expand(-1, -1, C))
    x = torch.cat((P["decoder_cls_token"].expand(N, -1, -1), x_), dim=1)
    hw = (cfg.hr_grid if high_res else cfg.grid)
    t_actual = L // (hw[1] * hw[2])
    pos = sep_pos_table(P["decoder_pos_embed_spatial"], P["decoder_pos_embed_temporal"], cfg, high_res, t_actual)
    pos = torch.cat([P["decoder_pos_embed_class"], pos], 1)
    x = x + pos
    for i in range(cfg.decoder_depth):
        x = block(x, P, f"decoder_blocks.{i}", cfg.decoder_num_heads, cfg.ln_eps,
                  final_residual=not (flash_compat and i == cfg.decoder_depth - 1))
    x = F.layer_norm(x, (C,), P["decoder_norm.weight"], P["decoder_norm.bias"], cfg.ln_eps)
    x = F.linear(x, P["decoder_pred.weight"], P["decoder_pred.bias"])
    return x[:, 1:, :]


def forward_loss(imgs, pred, mask, cfg: MAEConfig):
    """:613-667.  Returns (loss, frame_losses (N, T))."""
    T = imgs.shape[2]
    if T == 3:
        target = patchify(imgs, cfg)
    else:
        idx = torch.linspace(0, T - 1, cfg.pred_t_dim).long()
        target = patchify(torch.index_select(imgs, 2, idx), cfg)
    if cfg.norm_pix_loss:
        mean = target.mean(dim=-1, keepdim=True)
        var = target.var(dim=-1, keepdim=True)
        target = (target - mean) / (var + 1.0e-6) ** 0.5
    loss = ((pred - target) ** 2).mean(dim=-1)
    mask = mask.view(loss.shape)
    t = T // cfg.t_patch_size
    rl = loss.view(loss.shape[0], t, -1)
    rm = mask.view(mask.shape[0], t, -1)
    frame_losses = (rl * rm).sum(dim=2) / (rm.sum(dim=2) + 1e-6)
    return (loss * mask).sum() / mask.sum(), frame_losses


def forward(P, imgs, cfg: MAEConfig, mask_ratio=0.75, noise=None, frame_loss=False, flash_compat=False):
    """:669-680 -- returns (loss, pred, mask) (+ ids_restore for tests)."""
    high_res = imgs.shape[-2] == cfg.hr_grid[1] * cfg.patch_size
    latent, mask, ids_restore = forward_encoder(P, imgs, mask_ratio, noise, cfg, flash_compat)
    pred = forward_decoder(P, latent, ids_restore, cfg, high_res, flash_compat)
    loss, fl = forward_loss(imgs, pred, mask, cfg)
    if frame_loss:
        return (loss, fl), pred, mask, ids_restore
    return loss, pred, mask, ids_restore


def forward_backward(P, imgs, cfg: MAEConfig, mask_ratio=0.75, noise=None, flash_compat=False):
    """One oracle training-step's worth of math: returns loss, pred, mask, ids_restore, grads."""
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    loss, pred, mask, ids_restore = forward(Pg, imgs, cfg, mask_ratio, noise, flash_compat=flash_compat)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    return loss.detach(), pred.detach(), mask, ids_restore, grads


# --------------------------------------------------------------------------------------
# train utilities
# --------------------------------------------------------------------------------------
def grad_norm(grads) -> torch.Tensor:
    """misc.py:356-373 get_grad_norm_: 2-norm of the stack of per-tensor 2-norms."""
    gs = [g for g in grads if g is not None]
    if not gs:
        return torch.tensor(0.0)
    return torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in gs]), 2.0)


def weight_decay_groups(names_shapes, weight_decay=0.05, skip_list=(), bias_wd=False):
    """misc.py:678-696 add_weight_decay: returns (no_decay_names, decay_names)."""
    decay, no_decay = [], []
    for name, shape in names_shapes:
        if (not bias_wd) and len(shape) == 1 or name.endswith(".bias") or name in skip_list:
            no_decay.append(name)
        else:
            decay.append(name)
    return no_decay, decay


def cosine_lr(epoch: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """lr_sched.py:10-28."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def adamw_step(p, g, m, v, step: int, lr, beta1, beta2, eps, wd):
    """torch.optim.AdamW single-tensor update (what `_multi_tensor.AdamW` computes; main_pretrain :451)."""
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
