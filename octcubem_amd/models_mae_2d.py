"""2-D masked auto-encoder: drop-in for the reference's ``OCTCube/models_mae.py`` ``MaskedAutoencoderViT``
(BASELINE config 1; timm 0.3.2 ``PatchEmbed`` / ``Block`` semantics, fixed 2-D sin-cos positional embeddings, the encoder's
cls token travels through the decoder).  Same state_dict keys (``blocks.i.attn.qkv`` fused), same
``model(imgs, mask_ratio) -> (loss, pred, mask)`` contract; ``noise`` may be injected for parity runs.  GPU only."""
from __future__ import annotations

from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .arena import get_arena
from . import video_vit
from .video_vit import TimmBlock as Block, TimmPatchEmbed as PatchEmbed, layer_norm
from ._autocast import autocast_invariant


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """2-D sine-cosine table (same construction as OCTCube/util/pos_embed.py:20-63: w first, sin | cos halves)."""
    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float32) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    g = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(g, g), axis=0).reshape([2, 1, grid_size, grid_size])
    emb = np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


@autocast_invariant
class MaskedAutoencoderViT(nn.Module):
    """Masked Autoencoder with VisionTransformer backbone"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                 decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0, norm_layer=nn.LayerNorm, norm_pix_loss=False,
                 use_flash_attn=False, flash_compat=False, no_qkv_bias=False, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, input_size=None, **kwargs):
        """``use_flash_attn=True`` builds the twin of OCTCube/models_mae_flash_attn.py (:70-176): blocks from the
        ``create_block`` factory (state_dict keys ``blocks.i.mixer.Wqkv / out_proj``), the ``x, residual = blk(x, residual)``
        loop and hence the dropped final residual of the flash path; ``input_size`` is that file's
        name for ``img_size``.  ``flash_compat=True`` keeps the timm layout and only drops the last block's residual.
        Attention itself is always the gfx950 kernel."""
        super().__init__()
        if input_size is not None:
            img_size = input_size
        self.in_chans = in_chans
        self.use_flash_attn = bool(use_flash_attn)
        self.flash_compat = bool(flash_compat)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.patch_embed.input_size = self.patch_embed.img_size          # attribute name of the flash file's PatchEmbed (:48-68)
        self.input_size = self.patch_embed.img_size
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)

        def stack(dim, heads, n):
            if not self.use_flash_attn:
                return nn.ModuleList([Block(dim, heads, mlp_ratio, qkv_bias=True, qk_scale=None, norm_layer=norm_layer)
                                      for _ in range(n)])
            dpr = [x.item() for x in torch.linspace(0, drop_path_rate, n)]
            return nn.ModuleList([
                video_vit.create_block(dim, heads, mlp_ratio, not no_qkv_bias, drop_rate, attn_drop_rate,
                                       drop_path1=dpr[i - 1] if i > 0 else 0.0, drop_path2=dpr[i], norm_layer=norm_layer,
                                       act_layer=nn.GELU, use_flash_attn=True, fused_bias_fc=False, fused_mlp=False,
                                       fused_dropout_add_ln=False, layer_idx=i, n_layer=n, last_layer_subset=False)
                for i in range(n)])

        self.blocks = stack(embed_dim, num_heads, depth)
        self.norm = norm_layer(embed_dim)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = stack(decoder_embed_dim, decoder_num_heads, decoder_depth)
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        self.initialize_weights()

    def initialize_weights(self):
        g = int(self.patch_embed.num_patches ** 0.5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.pos_embed.shape[-1], g, True)).float().unsqueeze(0))
        self.decoder_pos_embed.data.copy_(
            torch.from_numpy(get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], g, True)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.cls_token, std=0.02)
        torch.nn.init.normal_(self.mask_token, std=0.02)
        self.apply(self._init_weights)
        if getattr(self, "_arena", None) is not None:      # re-initialised after binding: the line above wrote through ``.data``
            self._arena.invalidate_lp()

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

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

    def patchify(self, imgs):
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        h = w = imgs.shape[2] // p
        x = imgs.reshape(shape=(imgs.shape[0], self.in_chans, h, p, w, p))
        x = torch.einsum("nchpwq->nhwpqc", x)
        return x.reshape(shape=(imgs.shape[0], h * w, p ** 2 * self.in_chans))

    def unpatchify(self, x):
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(shape=(x.shape[0], h, w, p, p, self.in_chans))
        x = torch.einsum("nhwpqc->nchpwq", x)
        return x.reshape(shape=(x.shape[0], self.in_chans, h * p, h * p))

    def _linear(self, lin, x, out_f32=False):
        arena = get_arena(self)
        return ops.LinearFn.apply(x, arena.lp_view(lin.weight), arena.f32_view(lin.bias), lambda: arena.grad_view(lin.weight),
                                  lambda: arena.grad_view(lin.bias), out_f32, lin.weight, lin.bias)

    def _run_blocks(self, blocks, x):
        if self.use_flash_attn:                       # OCTCube/models_mae_flash_attn.py: residual = None; x, residual = blk(x, residual)
            residual = None
            for blk in blocks:
                x, residual = blk(x, residual)
            return x
        for i, blk in enumerate(blocks):
            x = blk(x, final_residual=not (self.flash_compat and i == len(blocks) - 1))
        return x

    def forward_encoder(self, x, mask_ratio, noise=None):
        N = x.shape[0]
        L = self.patch_embed.num_patches
        len_keep = int(L * (1 - mask_ratio))
        if noise is None:
            noise = torch.rand(N, L, device=x.device)
        mask, ids_restore, ids_keep = ops.random_masking_ids(noise.to(device=x.device, dtype=torch.float32).contiguous(), len_keep)
        tok = self.patch_embed.embed_tokens(x, ids_keep)                               # kept patches only (output-identical)
        pe = self.pos_embed[0]
        xs = ops.EncAssembleFn.apply(tok, pe[1:], self.cls_token, pe[:1].view(1, 1, -1), ids_keep, ids_restore)   # fp32 [N, 1+keep, D]
        xs = self._run_blocks(self.blocks, xs)
        xs = layer_norm(self.norm, xs)
        self._ids_keep = ids_keep
        return xs, mask, ids_restore

    def forward_decoder(self, x, ids_restore, ids_keep=None):
        N = x.shape[0]
        if ids_keep is None:
            ids_keep = getattr(self, "_ids_keep", None)
            if ids_keep is None or ids_keep.shape[0] != N:
                ids_keep = torch.argsort(ids_restore, dim=1)[:, :x.shape[1] - 1]
        emb = self._linear(self.decoder_embed, x.reshape(-1, x.shape[-1]))             # bf16 [N*(1+keep), Dd], cls row included
        dpe = self.decoder_pos_embed[0]
        xd = ops.DecAssembleFn.apply(emb, self.mask_token, dpe[1:], None, dpe[:1].view(1, 1, -1), ids_restore, ids_keep)
        xd = self._run_blocks(self.decoder_blocks, xd)
        xd = layer_norm(self.decoder_norm, xd)
        pred_full = self._linear(self.decoder_pred, xd, out_f32=True)
        self._pred_full = pred_full
        return pred_full[:, 1:, :]

    def forward_loss(self, imgs, pred, mask, return_frame_loss=False):
        N, C, H, W = imgs.shape
        pred_full = getattr(self, "_pred_full", None)
        if pred_full is None or pred_full.shape[0] != N or pred.data_ptr() != pred_full[:, 1:, :].data_ptr():
            pred_full = torch.cat([torch.zeros_like(pred[:, :1, :]), pred], dim=1).float().contiguous()
        loss_tok = ops.PatchMSEFn.apply(pred_full, imgs.float().contiguous().view(N, C, 1, H, W), None, 1,
                                        self.patch_embed.patch_size[0], self.norm_pix_loss)
        frame_loss = loss_tok.mean(dim=-1)
        loss = (loss_tok * mask).sum() / mask.sum()
        if return_frame_loss:
            return loss, frame_loss
        return loss

    def forward(self, imgs, mask_ratio=0.75, return_frame_loss=False, noise=None):
        self.prepare()
        imgs = imgs.float().contiguous()
        latent, mask, ids_restore = self.forward_encoder(imgs, mask_ratio, noise)
        pred = self.forward_decoder(latent, ids_restore, self._ids_keep)
        loss = self.forward_loss(imgs, pred, mask, return_frame_loss=return_frame_loss)
        self._ids_restore = ids_restore
        if return_frame_loss:
            loss, frame_loss = loss
            return loss, pred, mask, frame_loss
        return loss, pred, mask


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512, decoder_depth=8,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_base_patch16_dec512d8b(**kwargs):
    """ViT-B encoder (BASELINE config 1; the reference file ships only the ViT-L factory, OCTCube/models_mae.py:231-241)."""
    return MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=8,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


mae_vit_large_patch16 = mae_vit_large_patch16_dec512d8b
mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b


def flash_attn_mae_vit_large_patch16(**kwargs):
    """The factory of OCTCube/models_mae_flash_attn.py (its class defaults to use_flash_attn=True)."""
    kwargs.setdefault("use_flash_attn", True)
    return mae_vit_large_patch16_dec512d8b(**kwargs)
