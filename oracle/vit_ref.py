"""CPU oracle, part 2: the spatio-temporal fine-tune ViT and the 2-D MAE (SURVEY R2 / BASELINE config 1).
TEST INFRASTRUCTURE ONLY (same rules as mae3d_ref.py: imported by tests/, smoke() and bench's cpu_baseline only).

  * ``vit_st_forward``  follows OCTCube/models_vit_st_flash_attn.py:181-258 (non-flash blocks = util/video_vit.Block):
    Conv3d patch embed -> cls + sep pos-embed (no masking) -> blocks -> mean over patch tokens (global_pool; the
    ``outcome = self.norm(x)`` it computes is unused, :247-249) or cls token -> dropout -> head.
  * ``mae2d_*``         follows OCTCube/models_mae.py:22-227.  Its PatchEmbed / Block / Attention are timm's
    (``timm==0.3.2`` asserted by OCTCube/main_pretrain.py:27; un-vendored third-party): Conv2d(k=s=p) + flatten + transpose,
    fused ``qkv`` Linear, pre-norm residual Block -- restated here from timm 0.3.2's published algorithm.
Parity status: PINNED by tests/golden/vit_st_small.npz and mae2d_small.npz (oracle/gen_golden.py runs the reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .mae3d_ref import block as st_block


# ----------------------------------------------------------------------------------------------------------------
# ST fine-tune ViT
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class ViTSTConfig:
    num_frames: int = 60
    t_patch_size: int = 3
    img_size: int = 256
    patch_size: int = 16
    in_chans: int = 1
    num_classes: int = 8
    embed_dim: int = 1024
    depth: int = 24
    num_heads: int = 16
    mlp_ratio: float = 4.0
    global_pool: bool = True
    ln_eps: float = 1e-6

    @property
    def grid(self):
        return (self.num_frames // self.t_patch_size, self.img_size // self.patch_size, self.img_size // self.patch_size)


def vit_st_param_shapes(cfg: ViTSTConfig) -> Dict[str, Tuple[int, ...]]:
    D = cfg.embed_dim
    T, h, w = cfg.grid
    s = {"patch_embed.proj.weight": (D, cfg.in_chans, cfg.t_patch_size, cfg.patch_size, cfg.patch_size),
         "patch_embed.proj.bias": (D,), "cls_token": (1, 1, D), "pos_embed_spatial": (1, h * w, D),
         "pos_embed_temporal": (1, T, D), "pos_embed_class": (1, 1, D)}
    hid = int(D * cfg.mlp_ratio)
    for i in range(cfg.depth):
        p = f"blocks.{i}"
        s[f"{p}.norm1.weight"] = (D,); s[f"{p}.norm1.bias"] = (D,)
        for n in ("q", "k", "v", "proj"):
            s[f"{p}.attn.{n}.weight"] = (D, D); s[f"{p}.attn.{n}.bias"] = (D,)
        s[f"{p}.norm2.weight"] = (D,); s[f"{p}.norm2.bias"] = (D,)
        s[f"{p}.mlp.fc1.weight"] = (hid, D); s[f"{p}.mlp.fc1.bias"] = (hid,)
        s[f"{p}.mlp.fc2.weight"] = (D, hid); s[f"{p}.mlp.fc2.bias"] = (D,)
    s["norm.weight"] = (D,); s["norm.bias"] = (D,)
    s["head.weight"] = (cfg.num_classes, D); s["head.bias"] = (cfg.num_classes,)
    return s


def init_from_shapes(shapes, seed, bias_std=0.02):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in shapes.items():
        if k.endswith(".weight") and len(shp) >= 2:
            fan_out, fan_in = shp[0], int(math.prod(shp[1:]))
            a = math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shp, generator=g) * 2 - 1) * a
        elif k.endswith(".weight"):
            t = torch.ones(shp) + bias_std * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            t = bias_std * torch.randn(shp, generator=g)
        else:
            t = 0.02 * torch.randn(shp, generator=g)
        out[k] = t.float()
    return out


def vit_st_forward(P, x, cfg: ViTSTConfig, flash_compat=False):
    """Eval-mode forward (dropout off): logits (N, num_classes) and the pooled embedding."""
    k = (cfg.t_patch_size, cfg.patch_size, cfg.patch_size)
    y = F.conv3d(x, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=k).flatten(3)
    y = torch.einsum("ncts->ntsc", y)
    N, T, L, C = y.shape
    y = y.reshape(N, T * L, C)
    y = torch.cat((P["cls_token"].expand(N, -1, -1), y), dim=1)
    T_, h, w = cfg.grid
    pos = P["pos_embed_spatial"].repeat(1, T_, 1) + torch.repeat_interleave(P["pos_embed_temporal"], h * w, dim=1)
    pos = torch.cat([P["pos_embed_class"].expand(pos.shape[0], -1, -1), pos], 1)
    y = y + pos
    for i in range(cfg.depth):
        y = st_block(y, P, f"blocks.{i}", cfg.num_heads, cfg.ln_eps, final_residual=not (flash_compat and i == cfg.depth - 1))
    emb = y[:, 1:, :].mean(dim=1) if cfg.global_pool else y[:, 0]
    return F.linear(emb, P["head.weight"], P["head.bias"]), emb


# ----------------------------------------------------------------------------------------------------------------
# 2-D MAE (timm blocks)
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class MAE2DConfig:
    img_size: int = 256
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    decoder_embed_dim: int = 512
    decoder_depth: int = 8
    decoder_num_heads: int = 16
    mlp_ratio: float = 4.0
    norm_pix_loss: bool = False
    ln_eps: float = 1e-6

    @property
    def grid(self):
        return self.img_size // self.patch_size

    @property
    def num_patches(self):
        return self.grid ** 2


def sincos_2d(embed_dim, grid_size, cls_token=False):
    """OCTCube/util/pos_embed.py:20-63 (w goes first in the meshgrid; half the channels per axis; sin then cos)."""
    def one_d(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float32)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    emb = np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def mae2d_param_shapes(cfg: MAE2DConfig):
    D, Dd, p, c = cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_size, cfg.in_chans
    L = cfg.num_patches
    s = {"cls_token": (1, 1, D), "pos_embed": (1, L + 1, D), "patch_embed.proj.weight": (D, c, p, p), "patch_embed.proj.bias": (D,)}

    def blk(prefix, dim):
        hid = int(dim * cfg.mlp_ratio)
        s[f"{prefix}.norm1.weight"] = (dim,); s[f"{prefix}.norm1.bias"] = (dim,)
        s[f"{prefix}.attn.qkv.weight"] = (3 * dim, dim); s[f"{prefix}.attn.qkv.bias"] = (3 * dim,)
        s[f"{prefix}.attn.proj.weight"] = (dim, dim); s[f"{prefix}.attn.proj.bias"] = (dim,)
        s[f"{prefix}.norm2.weight"] = (dim,); s[f"{prefix}.norm2.bias"] = (dim,)
        s[f"{prefix}.mlp.fc1.weight"] = (hid, dim); s[f"{prefix}.mlp.fc1.bias"] = (hid,)
        s[f"{prefix}.mlp.fc2.weight"] = (dim, hid); s[f"{prefix}.mlp.fc2.bias"] = (dim,)
    for i in range(cfg.depth):
        blk(f"blocks.{i}", D)
    s["norm.weight"] = (D,); s["norm.bias"] = (D,)
    s["decoder_embed.weight"] = (Dd, D); s["decoder_embed.bias"] = (Dd,)
    s["mask_token"] = (1, 1, Dd); s["decoder_pos_embed"] = (1, L + 1, Dd)
    for i in range(cfg.decoder_depth):
        blk(f"decoder_blocks.{i}", Dd)
    s["decoder_norm.weight"] = (Dd,); s["decoder_norm.bias"] = (Dd,)
    s["decoder_pred.weight"] = (p * p * c, Dd); s["decoder_pred.bias"] = (p * p * c,)
    return s


def mae2d_init(cfg: MAE2DConfig, seed=0, bias_std=0.02):
    P = init_from_shapes(mae2d_param_shapes(cfg), seed, bias_std)
    P["pos_embed"] = torch.from_numpy(sincos_2d(cfg.embed_dim, cfg.grid, True)).float().unsqueeze(0)
    P["decoder_pos_embed"] = torch.from_numpy(sincos_2d(cfg.decoder_embed_dim, cfg.grid, True)).float().unsqueeze(0)
    return P


def timm_block(x, P, prefix, num_heads, eps, final_residual=True):
    """timm 0.3.2 Block: x + attn(norm1(x)); x + mlp(norm2(x)); Attention with one fused qkv Linear.
    final_residual=False: the MLP branch alone -- what flash-attn 2.5.2's prenorm Block returns as ``hidden_states`` and the
    reference's flash models hand to their final norm (OCTCube/models_mae_flash_attn.py; restated, parity unpinned)."""
    B, N, C = x.shape
    hd = C // num_heads
    h = F.layer_norm(x, (C,), P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], eps)
    qkv = F.linear(h, P[f"{prefix}.attn.qkv.weight"], P[f"{prefix}.attn.qkv.bias"]).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(a, P[f"{prefix}.attn.proj.weight"], P[f"{prefix}.attn.proj.bias"])
    h = F.layer_norm(x, (C,), P[f"{prefix}.norm2.weight"], P[f"{prefix}.norm2.bias"], eps)
    h = F.gelu(F.linear(h, P[f"{prefix}.mlp.fc1.weight"], P[f"{prefix}.mlp.fc1.bias"]))
    y = F.linear(h, P[f"{prefix}.mlp.fc2.weight"], P[f"{prefix}.mlp.fc2.bias"])
    return x + y if final_residual else y


def mae2d_patchify(imgs, p):
    """OCTCube/models_mae.py:95-107: (N,3,H,W) -> (N, L, p*p*3) with per-token order (p, q, c)."""
    N, C, H, W = imgs.shape
    h = w = H // p
    x = imgs.reshape(N, C, h, p, w, p)
    x = torch.einsum("nchpwq->nhwpqc", x)
    return x.reshape(N, h * w, p * p * C)


def mae2d_forward(P, imgs, cfg: MAE2DConfig, mask_ratio=0.75, noise=None, flash_compat=False):
    """OCTCube/models_mae.py:151-227 -> (loss, pred, mask, ids_restore)."""
    p = cfg.patch_size
    x = F.conv2d(imgs, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)
    x = x + P["pos_embed"][:, 1:, :]
    N, L, D = x.shape
    len_keep = int(L * (1 - mask_ratio))
    if noise is None:
        noise = torch.rand(N, L)
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :len_keep]
    x = torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    mask = torch.ones(N, L); mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    cls = (P["cls_token"] + P["pos_embed"][:, :1, :]).expand(N, -1, -1)
    x = torch.cat((cls, x), dim=1)
    for i in range(cfg.depth):
        x = timm_block(x, P, f"blocks.{i}", cfg.num_heads, cfg.ln_eps, not (flash_compat and i == cfg.depth - 1))
    x = F.layer_norm(x, (D,), P["norm.weight"], P["norm.bias"], cfg.ln_eps)
    x = F.linear(x, P["decoder_embed.weight"], P["decoder_embed.bias"])
    Dd = x.shape[-1]
    mask_tokens = P["mask_token"].repeat(N, L + 1 - x.shape[1], 1)
    x_ = torch.cat([x[:, 1:, :], mask_tokens], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).expand(-1, -1, Dd))
    x = torch.cat([x[:, :1, :], x_], dim=1) + P["decoder_pos_embed"]
    for i in range(cfg.decoder_depth):
        x = timm_block(x, P, f"decoder_blocks.{i}", cfg.decoder_num_heads, cfg.ln_eps,
                       not (flash_compat and i == cfg.decoder_depth - 1))
    x = F.layer_norm(x, (Dd,), P["decoder_norm.weight"], P["decoder_norm.bias"], cfg.ln_eps)
    pred = F.linear(x, P["decoder_pred.weight"], P["decoder_pred.bias"])[:, 1:, :]
    target = mae2d_patchify(imgs, p)
    if cfg.norm_pix_loss:
        mean = target.mean(dim=-1, keepdim=True); var = target.var(dim=-1, keepdim=True)
        target = (target - mean) / (var + 1.0e-6) ** 0.5
    loss = ((pred - target) ** 2).mean(dim=-1)
    loss = (loss * mask).sum() / mask.sum()
    return loss, pred, mask, ids_restore


def mae2d_forward_backward(P, imgs, cfg, mask_ratio=0.75, noise=None):
    frozen = {"pos_embed", "decoder_pos_embed"}          # requires_grad=False in the reference (:37,:51)
    Pg = {k: (v.detach().clone().requires_grad_(k not in frozen)) for k, v in P.items()}
    loss, pred, mask, ids_restore = mae2d_forward(Pg, imgs, cfg, mask_ratio, noise)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    return loss.detach(), pred.detach(), mask, ids_restore, grads


# ----------------------------------------------------------------------------------------------------------------
# 2-D ViT (OCTCube/models_vit.py on timm 0.3.2's VisionTransformer; timm is un-vendored: PARITY UNPINNED beyond the Block /
# PatchEmbed semantics shared with the 2-D MAE above, which IS pinned by mae2d_small.npz)
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class ViT2DConfig:
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    num_classes: int = 512
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    global_pool: bool = True
    ln_eps: float = 1e-6

    @property
    def num_patches(self):
        return (self.img_size // self.patch_size) ** 2


def vit2d_param_shapes(cfg: ViT2DConfig):
    D, hid = cfg.embed_dim, int(cfg.embed_dim * cfg.mlp_ratio)
    s = {"cls_token": (1, 1, D), "pos_embed": (1, cfg.num_patches + 1, D),
         "patch_embed.proj.weight": (D, cfg.in_chans, cfg.patch_size, cfg.patch_size), "patch_embed.proj.bias": (D,)}
    for i in range(cfg.depth):
        p = f"blocks.{i}"
        s.update({f"{p}.norm1.weight": (D,), f"{p}.norm1.bias": (D,), f"{p}.attn.qkv.weight": (3 * D, D), f"{p}.attn.qkv.bias": (3 * D,),
                  f"{p}.attn.proj.weight": (D, D), f"{p}.attn.proj.bias": (D,), f"{p}.norm2.weight": (D,), f"{p}.norm2.bias": (D,),
                  f"{p}.mlp.fc1.weight": (hid, D), f"{p}.mlp.fc1.bias": (hid,), f"{p}.mlp.fc2.weight": (D, hid), f"{p}.mlp.fc2.bias": (D,)})
    nm = "fc_norm" if cfg.global_pool else "norm"
    s.update({f"{nm}.weight": (D,), f"{nm}.bias": (D,), "head.weight": (cfg.num_classes, D), "head.bias": (cfg.num_classes,)})
    return s


def vit2d_forward(P, x, cfg: ViT2DConfig):
    """models_vit.py:35-55 forward_features + head."""
    y = F.conv2d(x, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=cfg.patch_size).flatten(2).transpose(1, 2)
    y = torch.cat((P["cls_token"].expand(y.shape[0], -1, -1), y), dim=1) + P["pos_embed"]
    for i in range(cfg.depth):
        y = timm_block(y, P, f"blocks.{i}", cfg.num_heads, cfg.ln_eps)
    D = cfg.embed_dim
    if cfg.global_pool:
        out = F.layer_norm(y[:, 1:, :].mean(dim=1), (D,), P["fc_norm.weight"], P["fc_norm.bias"], cfg.ln_eps)
    else:
        out = F.layer_norm(y, (D,), P["norm.weight"], P["norm.bias"], cfg.ln_eps)[:, 0]
    return F.linear(out, P["head.weight"], P["head.bias"])


def clip_loss(image_features, enface_features, logit_scale):
    """open_clip/loss.py:181-230, world_size 1, plain labels (pinned by tests/golden/coem_loss.npz through octcubem_amd.coem)."""
    logits = logit_scale * image_features @ enface_features.T
    labels = torch.arange(logits.shape[0])
    return (F.cross_entropy(logits, labels) + F.cross_entropy(logits.T, labels)) / 2
