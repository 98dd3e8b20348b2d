"""Spatio-temporal ViT for fine-tuning / inference: drop-in for the reference's
``OCTCube/models_vit_st_flash_attn.py`` ``VisionTransformer`` (constructor :51-178, forward :181-258) with the non-flash
(standard residual) semantics.  No masking: all T*H*W + 1 tokens go through every block (5121 tokens for 60x256x256).
Same state_dict keys; ``forward(x, hidden_states=False, return_embeddings=False)``.  GPU only."""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn

from . import ops, video_vit
from .arena import get_arena
from .video_vit import Attention, Block, PatchEmbed
from ._autocast import autocast_invariant


@autocast_invariant
class VisionTransformer(nn.Module):
    """Vision Transformer with support for global average pooling"""

    def __init__(self, num_frames, t_patch_size, img_size=256, patch_size=16, in_chans=1, num_classes=400, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, no_qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=nn.LayerNorm, dropout=0.5, sep_pos_embed=False, cls_embed=False,
                 global_pool=False, use_flash_attn=False, flash_compat=False, **kwargs):
        super().__init__()
        # flash_compat: the last block returns its MLP branch only, as the reference's flash path does
        # (OCTCube/models_vit_st_flash_attn.py:230-234 keeps ``x`` and drops ``residual``); needed for the released weights
        self.flash_compat = bool(flash_compat)
        if not (sep_pos_embed and cls_embed):
            raise NotImplementedError("built for sep_pos_embed=True, cls_embed=True (how every reference script calls it)")
        self.global_pool = global_pool
        self.sep_pos_embed = sep_pos_embed
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, num_frames, t_patch_size)
        input_size = self.patch_embed.input_size
        self.input_size = input_size
        self.cls_embed = cls_embed
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed_spatial = nn.Parameter(torch.zeros(1, input_size[1] * input_size[2], embed_dim))
        self.pos_embed_temporal = nn.Parameter(torch.zeros(1, input_size[0], embed_dim))
        self.pos_embed_class = nn.Parameter(torch.zeros(1, 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        # attention is always the fused gfx950 kernel; use_flash_attn=True builds the create_block blocks (keys blocks.i.mixer.*)
        # and runs the ``x, residual = blk(x, residual)`` loop of models_vit_st_flash_attn.py:123-141,230-234
        self.use_flash_attn = bool(use_flash_attn)
        if self.use_flash_attn:
            self.blocks = nn.ModuleList([
                video_vit.create_block(embed_dim, num_heads, mlp_ratio, not no_qkv_bias, drop_rate, attn_drop_rate,
                                       drop_path1=dpr[i - 1] if i > 0 else 0.0, drop_path2=dpr[i], norm_layer=norm_layer,
                                       act_layer=nn.GELU, use_flash_attn=True, fused_bias_fc=False, fused_mlp=False,
                                       fused_dropout_add_ln=False, layer_idx=i, n_layer=depth, last_layer_subset=False)
                for i in range(depth)])
        else:
            self.blocks = nn.ModuleList([
                Block(embed_dim, num_heads, mlp_ratio, qkv_bias=not no_qkv_bias, qk_scale=None, norm_layer=norm_layer, drop_path=dpr[i],
                      attn_func=partial(Attention, input_size=self.patch_embed.input_size)) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.dropout = nn.Dropout(dropout)
        self.head = nn.Linear(embed_dim, num_classes)
        torch.nn.init.normal_(self.head.weight, std=0.02)
        self._ids = None

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"cls_token", "pos_embed", "pos_embed_spatial", "pos_embed_temporal", "pos_embed_class"}

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

    def forward(self, x, hidden_states=False, return_embeddings=False):
        arena = self.prepare()
        x = x.float().contiguous()
        N = x.shape[0]
        T, h, w = self.input_size
        L = T * h * w
        tok = self.patch_embed.embed_tokens(x)                                         # bf16 [N*L, C], all tokens
        pos = (self.pos_embed_spatial.repeat(1, T, 1) + torch.repeat_interleave(self.pos_embed_temporal, h * w, dim=1)).view(L, -1)
        if self._ids is None or self._ids.shape[0] != N or self._ids.device != x.device:
            object.__setattr__(self, "_ids", torch.arange(L, device=x.device, dtype=torch.int64).expand(N, L).contiguous())
        # every token kept, in order: the identity is its own inverse permutation, which selects the deterministic gather for the
        # positional table's gradient (octmae_scatter_add_rows) instead of ATen's atomic index_add_
        x = ops.EncAssembleFn.apply(tok, pos, self.cls_token, self.pos_embed_class, self._ids, self._ids)   # fp32 [N, 1+L, C]
        hidden_states_list = []
        residual = None
        for i, blk in enumerate(self.blocks):
            if self.use_flash_attn:
                x, residual = blk(x, residual)
            else:
                x = blk(x, final_residual=not (self.flash_compat and i == len(self.blocks) - 1))
            hidden_states_list.append(x)
        if hidden_states:
            return hidden_states_list
        if self.global_pool:
            x = x[:, 1:, :].mean(dim=1)      # global pool without cls token; the reference's norm(x) here is computed and unused
        else:
            x = x[:, 0]
        embedding = x
        x = self.dropout(x)
        if self.head.out_features % 8 == 0:
            hw, hb = arena.lp_view(self.head.weight), arena.f32_view(self.head.bias)
            x = ops.LinearFn.apply(x, hw, hb, lambda: arena.grad_view(self.head.weight), lambda: arena.grad_view(self.head.bias), True,
                                   self.head.weight, self.head.bias)
        else:
            # [N, C] x [C, num_classes] with a class count that is not a whole 16-byte chunk of bf16: a few hundred kFLOP, done
            # in fp32 by ATen on the device (its weight gradient lands in the same arena view through autograd)
            x = torch.nn.functional.linear(x.float(), self.head.weight, self.head.bias)
        if return_embeddings:
            return x, embedding
        return x

    def load_state_dict_to_backbone(self, state_dict, strict=False, filter_keys=()):
        """Accepts flash-layout keys (mixer.Wqkv / mixer.out_proj) as well as the native attn.q/k/v/proj layout."""
        from .checkpoint import to_flash_layout, to_native_layout
        sd = to_native_layout(state_dict)
        if self.use_flash_attn:
            sd = to_flash_layout(sd)
        sd = {k: v for k, v in sd.items() if not any(f in k for f in filter_keys)}
        return super().load_state_dict(sd, strict=strict)


def vit_base_patch16(**kwargs):
    return VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_large_patch16(**kwargs):
    return VisionTransformer(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def flash_attn_vit_large_patch16(**kwargs):
    kwargs.setdefault("use_flash_attn", True)
    return vit_large_patch16(**kwargs)
