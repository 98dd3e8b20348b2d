"""2-D Vision Transformer for fine-tuning / the en-face (IR) tower: drop-in for the reference's ``OCTCube/models_vit.py``
``VisionTransformer`` -- timm 0.3.2's ``vision_transformer.VisionTransformer`` (PatchEmbed Conv2d k = s = 16, cls token,
learned ``pos_embed`` [1, 1 + L, D], fused-qkv Blocks, ``norm``, ``head``) plus the ``global_pool`` switch (:26-55): mean over
the patch tokens followed by ``fc_norm`` instead of ``norm`` + cls.  Same state_dict keys.  timm is not vendored by the
reference and absent here, so its part is restated (same blocks as models_mae_2d); GPU only."""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn

from . import ops
from .arena import get_arena
from .video_vit import TimmBlock as Block, TimmPatchEmbed as PatchEmbed, layer_norm
from ._autocast import autocast_invariant


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std)


@autocast_invariant
class VisionTransformer(nn.Module):
    """Vision Transformer with support for global average pooling"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=nn.LayerNorm,
                 global_pool=False, **kwargs):
        super().__init__()
        assert drop_rate == 0.0 and attn_drop_rate == 0.0
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop_path=dpr[i],
                                           norm_layer=norm_layer) for i in range(depth)])
        self.global_pool = global_pool
        if global_pool:
            self.fc_norm = norm_layer(embed_dim)         # the original ``norm`` is deleted in this mode (models_vit.py:29-33)
        else:
            self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=0.02)
        trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)
        self._ids = None

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def prepare(self):
        arena = get_arena(self, full_check=True)
        if torch.is_grad_enabled():
            arena.rebind_grads()
        arena.refresh_lp()
        return arena

    @property
    def arena(self):
        return get_arena(self, full_check=True)

    def invalidate_lp(self):
        """Call after writing weights behind PyTorch's version counters (``p.data.copy_()``, ``dist.broadcast(p.data)``, an EMA swap
        through ``.data``, a raw-pointer kernel): the next forward re-casts the whole 16-bit operand copy of the parameter arena
        (INTEGRATION.md section 1).  In-place writes that bump a version counter -- torch.optim optimizers, ``load_state_dict``,
        ``p.copy_()`` under ``no_grad`` -- are seen without it."""
        self.arena.invalidate_lp()

    def forward_features(self, x):
        self.prepare()
        x = x.float().contiguous()
        N, L = x.shape[0], self.patch_embed.num_patches
        tok = self.patch_embed.embed_tokens(x)
        if self._ids is None or self._ids.shape[0] != N or self._ids.device != x.device:
            object.__setattr__(self, "_ids", torch.arange(L, device=x.device, dtype=torch.int64).expand(N, L).contiguous())
        pe = self.pos_embed[0]
        x = ops.EncAssembleFn.apply(tok, pe[1:], self.cls_token, pe[:1].view(1, 1, -1), self._ids, self._ids)      # fp32 [N, 1 + L, D]; identity permutation: deterministic table gradient
        for blk in self.blocks:
            x = blk(x)
        if self.global_pool:
            x = x[:, 1:, :].mean(dim=1)
            return layer_norm(self.fc_norm, x.contiguous()).float()
        return layer_norm(self.norm, x)[:, 0].float()

    def forward(self, x):
        x = self.forward_features(x)
        if isinstance(self.head, nn.Identity):
            return x
        if self.head.out_features % 8 == 0:
            arena = get_arena(self)
            return ops.LinearFn.apply(x, arena.lp_view(self.head.weight), arena.f32_view(self.head.bias),
                                      lambda: arena.grad_view(self.head.weight), lambda: arena.grad_view(self.head.bias), True,
                                      self.head.weight, self.head.bias)
        return torch.nn.functional.linear(x, self.head.weight, self.head.bias)     # odd class counts: see models_vit_st


def vit_base_patch16(**kwargs):
    return VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_large_patch16(**kwargs):
    return VisionTransformer(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
