"""MI355X-native 3-D masked auto-encoder: drop-in for the reference's
``Pre-training/models_mae_joint_res_flash_attn.py`` ``MaskedAutoencoderViT`` (constructor :32-62, forward :669-680)
with the NON-flash (standard pre-norm residual) semantics BASELINE.json pins parity to.

Same constructor keywords, same attribute / parameter names (state_dict keys), same
``model(imgs, mask_ratio, frame_loss) -> (loss, pred, mask)`` contract.  What differs is how it is computed:
  * masking indices first (they depend only on the noise), then ONLY the kept 25 % of the patches are gathered
    and embedded (the reference embeds all 5120 tokens and gathers afterwards -- output-identical);
  * every dense contraction runs on bf16 MFMA with fp32 accumulation, attention is flash-style and never
    materialises (B, H, N, N); LayerNorm / softmax statistics and the residual stream stay fp32;
  * patchify + masked MSE is one fused kernel over the raw volume.
GPU only.  ``noise`` can be injected for parity runs (the reference draws torch.rand inside random_masking, :350).
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, video_vit
from .arena import get_arena
from .video_vit import layer_norm
from ._autocast import autocast_invariant


@autocast_invariant
class MaskedAutoencoderViT(nn.Module):
    """Masked Autoencoder with VisionTransformer backbone"""

    def __init__(self, input_size=256, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, mlp_ratio=4.0, norm_layer=nn.LayerNorm, norm_pix_loss=False, num_frames=16,
                 t_patch_size=4, patch_embed=video_vit.PatchEmbed, no_qkv_bias=False, sep_pos_embed=False,
                 trunc_init=False, cls_embed=False, pred_t_dim=8, high_res_input_size=512, use_flash_attn=False,
                 flash_compat=False, **kwargs):
        super().__init__()
        # flash_compat=True reproduces what the reference computes when it is built with use_flash_attn=True (how the
        # released OCTCube.pth was trained): the last encoder / decoder block hands only its MLP branch to the final norm,
        # the residual stream is dropped.  Default: standard pre-norm residual semantics.
        self.flash_compat = bool(flash_compat)
        if not (sep_pos_embed and cls_embed):
            raise NotImplementedError("the hot path is built for sep_pos_embed=True, cls_embed=True "
                                      "(the reference driver's defaults, main_pretrain…:248-249,283-288)")
        assert drop_rate == 0.0 and attn_drop_rate == 0.0
        self.trunc_init = trunc_init
        self.sep_pos_embed = sep_pos_embed
        self.cls_embed = cls_embed
        self.pred_t_dim = pred_t_dim
        self.in_chans = in_chans
        self.t_pred_patch_size = t_patch_size * pred_t_dim // num_frames
        if isinstance(input_size, int):
            input_size = (input_size, input_size)
        self.patch_embed = patch_embed(input_size, patch_size, in_chans, embed_dim, num_frames, t_patch_size)
        num_patches = self.patch_embed.num_patches
        input_size = self.patch_embed.input_size
        self.input_size = input_size
        self.high_res_patch_embed = video_vit.PatchEmbed(high_res_input_size, patch_size, in_chans, embed_dim, num_frames,
                                                         t_patch_size)
        self.high_res_input_size = self.high_res_patch_embed.input_size

        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.decoder_cls_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.pos_embed_spatial = nn.Parameter(torch.zeros(1, self.high_res_input_size[1] * self.high_res_input_size[2], embed_dim))
        self.pos_embed_temporal = nn.Parameter(torch.zeros(1, input_size[0], embed_dim))
        self.pos_embed_class = nn.Parameter(torch.zeros(1, 1, embed_dim))

        # Attention is ALWAYS the fused gfx950 kernel.  use_flash_attn=True builds what the reference builds with it (:122-152,
        # :191-220): blocks from the create_block factory (state_dict keys blocks.i.mixer.*), the ``x, residual = blk(x,
        # residual)`` loop, and therefore the flash path's semantics -- only ``x`` reaches the final norm.
        self.use_flash_attn = bool(use_flash_attn)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        if self.use_flash_attn:
            self.blocks = nn.ModuleList([
                video_vit.create_block(embed_dim, num_heads, mlp_ratio, not no_qkv_bias, drop_rate, attn_drop_rate,
                                       drop_path1=dpr[i - 1] if i > 0 else 0.0, drop_path2=dpr[i], norm_layer=norm_layer,
                                       act_layer=nn.GELU, use_flash_attn=True, fused_bias_fc=False, fused_mlp=False,
                                       fused_dropout_add_ln=False, layer_idx=i, n_layer=depth, last_layer_subset=False)
                for i in range(depth)])
        else:
            self.blocks = nn.ModuleList([
                video_vit.Block(embed_dim, num_heads, mlp_ratio, qkv_bias=not no_qkv_bias, qk_scale=None, norm_layer=norm_layer,
                                drop_path=dpr[i]) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed_spatial = nn.Parameter(
            torch.zeros(1, self.high_res_input_size[1] * self.high_res_input_size[2], decoder_embed_dim))
        self.decoder_pos_embed_temporal = nn.Parameter(torch.zeros(1, input_size[0], decoder_embed_dim))
        self.decoder_pos_embed_class = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        if self.use_flash_attn:
            ddpr = [x.item() for x in torch.linspace(0, drop_path_rate, decoder_depth)]
            self.decoder_blocks = nn.ModuleList([
                video_vit.create_block(decoder_embed_dim, decoder_num_heads, mlp_ratio, not no_qkv_bias, drop_rate, attn_drop_rate,
                                       drop_path1=ddpr[i - 1] if i > 0 else 0.0, drop_path2=ddpr[i], norm_layer=norm_layer,
                                       act_layer=nn.GELU, use_flash_attn=True, fused_bias_fc=False, fused_mlp=False,
                                       fused_dropout_add_ln=False, layer_idx=i, n_layer=decoder_depth, last_layer_subset=False)
                for i in range(decoder_depth)])
        else:
            self.decoder_blocks = nn.ModuleList([
                video_vit.Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=not no_qkv_bias, qk_scale=None,
                                norm_layer=norm_layer) for i in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, self.t_pred_patch_size * patch_size ** 2 * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        self._lin_views = None
        self.initialize_weights()

    # ------------------------------------------------------------------ init (reference :249-287)
    def initialize_weights(self):
        for t in (self.cls_token, self.pos_embed_spatial, self.pos_embed_temporal, self.decoder_pos_embed_spatial,
                  self.decoder_pos_embed_temporal, self.pos_embed_class, self.decoder_pos_embed_class):
            torch.nn.init.trunc_normal_(t, std=0.02)
        w = self.patch_embed.proj.weight.data
        if self.trunc_init:
            torch.nn.init.trunc_normal_(w)
            torch.nn.init.trunc_normal_(self.mask_token, std=0.02)
        else:
            torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
            torch.nn.init.normal_(self.mask_token, std=0.02)
        self.apply(self._init_weights)
        if getattr(self, "_arena", None) is not None:      # re-initialised after binding: the line above wrote through ``.data``
            self._arena.invalidate_lp()

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            if self.trunc_init:
                nn.init.trunc_normal_(m.weight, std=0.02)
            else:
                torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ helpers
    def _arena_views(self):
        arena = get_arena(self, full_check=True)
        if self._lin_views is None or self._lin_views[0] is not arena:
            de, dp = self.decoder_embed, self.decoder_pred
            v = {}
            for name, lin in (("decoder_embed", de), ("decoder_pred", dp)):
                v[name] = (arena.lp_view(lin.weight), arena.f32_view(lin.bias),
                           (lambda l=lin: arena.grad_view(l.weight)), (lambda l=lin: arena.grad_view(l.bias)))
            object.__setattr__(self, "_lin_views", (arena, v))
        return self._lin_views

    def prepare(self):
        """Validate/bind the parameter arena, re-attach .grad views and refresh the bf16 operand copy.
        Called at the top of every forward; cheap (one cast kernel over the arena)."""
        arena, _ = self._arena_views()
        if torch.is_grad_enabled():
            arena.rebind_grads()
        arena.refresh_lp()
        return arena

    @property
    def arena(self):
        return self._arena_views()[0]

    def invalidate_lp(self):
        """Call after writing weights behind PyTorch's version counters (``p.data.copy_()``, ``dist.broadcast(p.data)``, an EMA swap
        through ``.data``, a raw-pointer kernel): the next forward re-casts the whole 16-bit operand copy of the parameter arena
        (INTEGRATION.md section 1).  In-place writes that bump a version counter -- torch.optim optimizers, ``load_state_dict``,
        ``p.copy_()`` under ``no_grad`` -- are seen without it."""
        self.arena.invalidate_lp()

    def _linear(self, name, x, out_f32=False):
        _, v = self._arena_views()
        w_lp, b32, gw, gb = v[name]
        lin = getattr(self, name)
        return ops.LinearFn.apply(x, w_lp, b32, gw, gb, out_f32, lin.weight, lin.bias)

    def patchify(self, imgs, high_res=False):
        """(N, C, T, H, W) -> (N, L, u*p*p*C); plain tensor reshuffle for visualisers (reference :289-314).
        The loss does NOT go through this: it reads the volume inside the fused MSE kernel."""
        N, _, T, H, W = imgs.shape
        p = (self.high_res_patch_embed if high_res else self.patch_embed).patch_size[0]
        u = self.t_pred_patch_size
        assert W % p == 0 and H % p == 0 and T % u == 0
        h, w, t = H // p, W // p, T // u
        x = imgs.reshape(shape=(N, self.in_chans, t, u, h, p, w, p))
        x = torch.einsum("nctuhpwq->nthwupqc", x)
        x = x.reshape(shape=(N, t * h * w, p ** 2 * u * self.in_chans))
        info = (N, T, H, W, p, u, t, h, w)
        if high_res:
            self.patch_info_high_res = info
        else:
            self.patch_info = info
        return x

    def unpatchify(self, x, high_res=False, actual_t_dim=None):
        N, T, H, W, p, u, t, h, w = self.patch_info_high_res if high_res else self.patch_info
        if actual_t_dim is not None:
            T = actual_t_dim
        x = x.reshape(shape=(N, t, h, w, u, p, p, self.in_chans))
        x = torch.einsum("nthwupqc->nctuhpwq", x)
        return x.reshape(shape=(N, self.in_chans, T, H, W))

    def _is_high_res(self, imgs):
        return imgs.shape[-2] == self.high_res_input_size[1] * self.high_res_patch_embed.patch_size[0]

    def random_masking_ids(self, N, L, mask_ratio, device, noise=None):
        """Index part of random_masking (:336-372): returns mask, ids_restore, ids_keep."""
        len_keep = int(L * (1 - mask_ratio))
        if noise is None:
            if mask_ratio > 0:
                noise = torch.rand(N, L, device=device)
            else:
                noise = torch.arange(L, device=device, dtype=torch.float32).expand(N, L).contiguous()
        noise = noise.to(device=device, dtype=torch.float32).contiguous()
        return ops.random_masking_ids(noise, len_keep)

    def random_masking(self, x, mask_ratio, pre_mask=None, noise=None):
        """Reference signature (:336): x [N, L, D] -> x_masked, mask, ids_restore, ids_keep."""
        assert pre_mask is None, "pre_mask is unreachable from forward() in the reference (:669-677)"
        N, L, D = x.shape
        mask, ids_restore, ids_keep = self.random_masking_ids(N, L, mask_ratio, x.device, noise)
        x_masked = torch.gather(x, dim=1, index=ids_keep.unsqueeze(-1).expand(-1, -1, D))
        return x_masked, mask, ids_restore, ids_keep

    def _interp_matrix(self, device):
        """Bicubic (align_corners=False) resampling of the hr_h x hr_w positional grid to h x w as ONE dense matrix
        [h*w, hr_h*hr_w]: F.interpolate is linear in its input, so the matrix is obtained once by pushing the identity
        basis through it.  (ATen's bicubic kernel takes 1.6 ms per call on a 1024-channel 32x32 grid -- 4 ms per
        micro-batch with its backward; a 256x1024x1024 matmul is microseconds.)"""
        m = getattr(self, "_interp_m", None)
        if m is None or m.device != device:
            _, hh, hw = self.high_res_input_size
            _, h, w = self.input_size
            eye = torch.eye(hh * hw, dtype=torch.float32, device=device).view(1, hh * hw, hh, hw)
            m = F.interpolate(eye, [h, w], mode="bicubic", align_corners=False).reshape(hh * hw, h * w).t().contiguous()
            object.__setattr__(self, "_interp_m", m)
        return m

    def _pos_table(self, spatial, temporal, high_res, t_actual):
        """(:415-441, :532-557) bicubic-resampled spatial table tiled over T + repeat-interleaved temporal table.
        Tiny and differentiable (the tables are trainable): kept in PyTorch.  Returns [T*h*w, C] fp32."""
        _, hh, hw = self.high_res_input_size
        _, h, w = self.input_size
        C = spatial.shape[-1]
        if not high_res:
            pe = (self._interp_matrix(spatial.device) @ spatial.view(hh * hw, C)).view(1, h * w, C)
            ph, pw = h, w
        else:
            pe, ph, pw = spatial, hh, hw
        if t_actual == 1:
            return pe.reshape(ph * pw, C)
        pe = pe.repeat(1, t_actual, 1) + torch.repeat_interleave(temporal[:, :t_actual], ph * pw, dim=1)
        return pe.reshape(-1, C)

    # ------------------------------------------------------------------ forward pieces
    def _run_blocks(self, blocks, x):
        if self.use_flash_attn:                       # :480-483 / :584-587
            residual = None
            for blk in blocks:
                x, residual = blk(x, residual)
            return x
        for i, blk in enumerate(blocks):
            x = blk(x, final_residual=not (self.flash_compat and i == len(blocks) - 1))
        return x

    def forward_encoder(self, x, mask_ratio, pre_mask=None, noise=None):
        assert pre_mask is None
        imgs = x
        high_res = self._is_high_res(imgs)
        pe_mod = self.high_res_patch_embed if high_res else self.patch_embed
        N, _, T, _, _ = imgs.shape
        t_actual = T // pe_mod.t_patch_size
        L = t_actual * pe_mod.input_size[1] * pe_mod.input_size[2]
        mask, ids_restore, ids_keep = self.random_masking_ids(N, L, mask_ratio, imgs.device, noise)
        tok = pe_mod.embed_tokens(imgs, ids_keep)                                    # bf16 [N*nkeep, D]
        pos = self._pos_table(self.pos_embed_spatial, self.pos_embed_temporal, high_res, t_actual)
        x = ops.EncAssembleFn.apply(tok, pos, self.cls_token, self.pos_embed_class, ids_keep, ids_restore)   # fp32 [N, 1+nkeep, D]
        x = self._run_blocks(self.blocks, x)
        x = layer_norm(self.norm, x)                                                  # bf16
        x = x[:, 1:, :]
        self._ids_keep = ids_keep
        return x, mask, ids_restore

    def forward_decoder(self, x, ids_restore, high_res=False, ids_keep=None):
        N = x.shape[0]
        if ids_keep is None:
            ids_keep = getattr(self, "_ids_keep", None)
            if ids_keep is None or ids_keep.shape[0] != N:
                ids_keep = torch.argsort(ids_restore, dim=1)[:, :x.shape[1]]
        L = ids_restore.shape[1]
        hw = self.high_res_input_size if high_res else self.input_size
        t_actual = L // (hw[1] * hw[2])
        emb = self._linear("decoder_embed", x.reshape(-1, x.shape[-1]))               # bf16 [N*nkeep, Dd]
        dpos = self._pos_table(self.decoder_pos_embed_spatial, self.decoder_pos_embed_temporal, high_res, t_actual)
        x = ops.DecAssembleFn.apply(emb, self.mask_token, dpos, self.decoder_cls_token, self.decoder_pos_embed_class,
                                    ids_restore, ids_keep)                            # fp32 [N, 1+L, Dd]
        x = self._run_blocks(self.decoder_blocks, x)
        x = layer_norm(self.decoder_norm, x)
        pred_full = self._linear("decoder_pred", x, out_f32=True)                      # fp32 [N, 1+L, PD]
        self._pred_full = pred_full
        return pred_full[:, 1:, :]

    def forward_encoder_decoder(self, imgs):
        latent, mask, ids_restore = self.forward_encoder(imgs, 0)
        return self.forward_decoder(latent, ids_restore)

    def forward_loss(self, imgs, pred, mask, frame_loss=False):
        """(:613-667) imgs [N,C,T,H,W]; pred [N, L, u*p*p*C]; mask [N, L] 0 keep / 1 remove."""
        T, H, W = imgs.shape[2:]
        pe_mod = self.high_res_patch_embed if self._is_high_res(imgs) else self.patch_embed
        p = pe_mod.patch_size[0]
        pred_full = getattr(self, "_pred_full", None)
        if pred_full is None or pred_full.shape[0] != pred.shape[0] or pred_full.shape[1] != pred.shape[1] + 1 or \
                pred.data_ptr() != pred_full[:, 1:, :].data_ptr():
            pred_full = torch.cat([torch.zeros_like(pred[:, :1, :]), pred], dim=1).float().contiguous()
        if T == 3:
            frame_idx = None
        else:
            fi = torch.linspace(0, T - 1, self.pred_t_dim).long()
            frame_idx = None if (self.pred_t_dim == T and bool((fi == torch.arange(T)).all())) else \
                fi.to(device=imgs.device, dtype=torch.int32)
        loss_tok = ops.PatchMSEFn.apply(pred_full, imgs.contiguous(), frame_idx, self.t_pred_patch_size, p, self.norm_pix_loss)
        mask = mask.view(loss_tok.shape)
        t = T // pe_mod.t_patch_size
        rl = loss_tok.view(loss_tok.shape[0], t, -1)
        rm = mask.view(mask.shape[0], t, -1)
        frame_losses = (rl * rm).sum(dim=2) / (rm.sum(dim=2) + 1e-6)
        loss = (loss_tok * mask).sum() / mask.sum()
        if frame_loss:
            return loss, frame_losses
        return loss

    def forward(self, imgs, mask_ratio=0.75, frame_loss=False, pre_mask=None, noise=None):
        self.prepare()
        high_res = self._is_high_res(imgs)
        imgs = imgs.float().contiguous()
        latent, mask, ids_restore = self.forward_encoder(imgs, mask_ratio, noise=noise)
        pred = self.forward_decoder(latent, ids_restore, high_res=high_res, ids_keep=self._ids_keep)
        loss = self.forward_loss(imgs, pred, mask, frame_loss=frame_loss)
        self._ids_restore = ids_restore
        return loss, pred, mask

    def forward_patch_embed(self, imgs):
        self.prepare()
        pe_mod = self.high_res_patch_embed if self._is_high_res(imgs) else self.patch_embed
        x = pe_mod(imgs.float().contiguous())
        N, T, L, C = x.shape
        return x.reshape(N, T * L, C)

    # ------------------------------------------------------------------ checkpoint key compatibility
    def load_state_dict_to_backbone(self, state_dict, strict=False, filter_keys=()):
        """Accepts both the non-flash (attn.q/k/v/proj) and the flash (mixer.Wqkv/out_proj) key layouts
        (reference remap rules :693-724, applied in reverse)."""
        from .checkpoint import to_flash_layout, to_native_layout
        sd = to_native_layout(state_dict)
        if self.use_flash_attn:
            sd = to_flash_layout(sd)
        sd = {k: v for k, v in sd.items() if not any(f in k for f in filter_keys)}
        return super().load_state_dict(sd, strict=strict)


def mae_vit_base_patch16(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_large_patch16(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def flash_attn_mae_vit_large_patch16(**kwargs):
    """Same architecture; in the reference this factory swaps in flash-attn blocks (:792-803)."""
    kwargs.setdefault("use_flash_attn", True)
    return mae_vit_large_patch16(**kwargs)


def mae_vit_huge_patch14(**kwargs):
    raise NotImplementedError("patch 14 / head_dim 80 is outside the built kernel set (head_dim 32 and 64)")


def octcube_vit_large_3dmae(**kwargs):
    """The BASELINE configuration: ViT-L encoder, 512x8x16 decoder, 60x256x256 volumes, t_patch 3
    (defaults of main_pretrain_oph_joint_2d512_flash_attn.py:231-235,283-288)."""
    cfg = dict(input_size=256, in_chans=1, num_frames=60, t_patch_size=3, pred_t_dim=60, sep_pos_embed=True, cls_embed=True,
               high_res_input_size=512, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16)
    cfg.update(kwargs)
    return mae_vit_large_patch16(**cfg)
