"""Drop-in building blocks: same constructor arguments, attributes, parameter names and forward signatures
as the reference's ``custom_util/video_vit.py`` (PatchEmbed :22-83, Attention :86-138, Block :141-184) and timm's
``Mlp`` -- computed by the hand-written gfx950 kernels behind liboctmae.so.

The nn.Conv3d / nn.Linear / nn.LayerNorm children are PARAMETER CONTAINERS only (they keep the reference's
state_dict keys and initialisers); their own forward is never used.  GPU only, no fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .arena import get_arena
from ._autocast import autocast_invariant


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


@autocast_invariant
class DropPath(nn.Module):
    """Stochastic depth per sample (timm DropPath): identity when drop_prob == 0 or in eval mode."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def sample(self, batch, device):
        """fp32 [batch] of 0 or 1 / keep_prob: the per-sample factor forward() multiplies in."""
        keep = 1 - self.drop_prob
        return (keep + torch.rand(batch, dtype=torch.float32, device=device)).floor_().div_(keep)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        return x * self.sample(x.shape[0], x.device).to(x.dtype).view((x.shape[0],) + (1,) * (x.ndim - 1))


@autocast_invariant
class PatchEmbed(nn.Module):
    """Image to Patch Embedding -- Conv3d(k = s = (t_patch, p, p)) evaluated as gather + MFMA GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, frames=32, t_patch_size=4):
        super().__init__()
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        assert img_size[1] % patch_size[1] == 0
        assert img_size[0] % patch_size[0] == 0
        assert frames % t_patch_size == 0
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0]) * (frames // t_patch_size)
        self.input_size = (frames // t_patch_size, img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.img_size = img_size
        self.patch_size = patch_size
        self.frames = frames
        self.t_patch_size = t_patch_size
        self.grid_size = img_size[0] // patch_size[0]
        self.t_grid_size = frames // t_patch_size
        kernel_size = [t_patch_size] + list(patch_size)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=kernel_size, stride=kernel_size)
        self._views = None

    def _v(self):
        arena = get_arena(self)
        if self._views is None or self._views[0] is not arena:
            w, b = self.proj.weight, self.proj.bias
            object.__setattr__(self, "_views", (arena, arena.lp_view(w, shape=(w.shape[0], -1)), arena.f32_view(b),
                                                lambda: arena.grad_view(w), lambda: arena.grad_view(b)))
        return self._views

    def embed_tokens(self, x, ids_keep=None):
        """Embeds only the tokens listed in ids_keep [B, nkeep] (all tokens when None) -> bf16 [B*nkeep, D]."""
        B, C, T, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        assert T % self.t_patch_size == 0
        _, w_lp, b32, gw, gb = self._v()
        L = (T // self.t_patch_size) * self.input_size[1] * self.input_size[2]
        nkeep = L if ids_keep is None else ids_keep.shape[1]
        return ops.PatchEmbedFn.apply(x.contiguous(), ids_keep, w_lp, b32, gw, gb, self.t_patch_size, self.patch_size[0], nkeep,
                                      self.proj.weight, self.proj.bias)

    def forward(self, x):
        B, C, T, H, W = x.shape
        tok = self.embed_tokens(x)
        return tok.view(B, T // self.t_patch_size, self.input_size[1] * self.input_size[2], -1)   # [N, T, H*W, C]


@autocast_invariant
class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, input_size=(4, 14, 14)):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if head_dim not in (32, 64):
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 32 and 64")
        if qk_scale is not None and abs(qk_scale - head_dim ** -0.5) > 1e-12:
            raise NotImplementedError("qk_scale other than head_dim**-0.5")
        self.scale = qk_scale or head_dim ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        assert attn_drop == 0.0  # do not use
        assert proj_drop == 0.0, "proj_drop is not supported on the fused path (the reference trains with drop=0)"
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.input_size = input_size
        assert input_size[1] == input_size[2]
        self._views = None

    def _v(self):
        arena = get_arena(self)
        if self._views is None or self._views[0] is not arena:
            q, k, v, pr = self.q, self.k, self.v, self.proj
            D = q.weight.shape[0]
            if not arena.fused_ok(q.weight, k.weight, v.weight):
                raise RuntimeError("q/k/v weights are not adjacent in the arena")
            has_b = q.bias is not None
            if has_b and not arena.fused_ok(q.bias, k.bias, v.bias):
                raise RuntimeError("q/k/v biases are not adjacent in the arena")
            wqkv = arena.lp_view(q.weight, v.weight, (3 * D, D))
            bqkv = arena.f32_view(q.bias, v.bias, (3 * D,)) if has_b else None
            wproj = arena.lp_view(pr.weight)
            bproj = arena.f32_view(pr.bias) if pr.bias is not None else None

            def grads():
                return (arena.grad_view(q.weight, v.weight, (3 * D, D)),
                        arena.grad_view(q.bias, v.bias, (3 * D,)) if has_b else None,
                        arena.grad_view(pr.weight), arena.grad_view(pr.bias) if pr.bias is not None else None)
            params = [q.weight, k.weight, v.weight, pr.weight] + ([q.bias, k.bias, v.bias] if has_b else []) + \
                     ([pr.bias] if pr.bias is not None else [])
            object.__setattr__(self, "_views", (arena, wqkv, bqkv, wproj, bproj, grads, tuple(params)))
        return self._views

    def forward(self, x, residual=None):
        """x: [B, N, C] (LayerNorm output).  With ``residual`` (fp32 [B,N,C]) returns residual + attn(x) in fp32."""
        _, wqkv, bqkv, wproj, bproj, grads, params = self._v()
        return ops.AttentionFn.apply(x, residual, wqkv, bqkv, wproj, bproj, grads, self.num_heads, *params)


@autocast_invariant
class TimmAttention(nn.Module):
    """timm 0.3.2 ``vision_transformer.Attention`` (one fused ``qkv`` Linear) -- the block the reference's 2-D models build
    from ``timm.models.vision_transformer.Block`` (OCTCube/models_mae.py:18,40-42).  Same kernels as ``Attention``."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if head_dim not in (32, 64):
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 32 and 64")
        if qk_scale is not None and abs(qk_scale - head_dim ** -0.5) > 1e-12:
            raise NotImplementedError("qk_scale other than head_dim**-0.5")
        assert attn_drop == 0.0 and proj_drop == 0.0
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self._views = None

    def _v(self):
        arena = get_arena(self)
        if self._views is None or self._views[0] is not arena:
            qkv, pr = self.qkv, self.proj
            has_b = qkv.bias is not None

            def grads():
                return (arena.grad_view(qkv.weight), arena.grad_view(qkv.bias) if has_b else None, arena.grad_view(pr.weight),
                        arena.grad_view(pr.bias) if pr.bias is not None else None)
            params = [qkv.weight, pr.weight] + ([qkv.bias] if has_b else []) + ([pr.bias] if pr.bias is not None else [])
            object.__setattr__(self, "_views", (arena, arena.lp_view(qkv.weight), arena.f32_view(qkv.bias) if has_b else None,
                                                arena.lp_view(pr.weight), arena.f32_view(pr.bias) if pr.bias is not None else None,
                                                grads, tuple(params)))
        return self._views

    def forward(self, x, residual=None):
        _, wqkv, bqkv, wproj, bproj, grads, params = self._v()
        return ops.AttentionFn.apply(x, residual, wqkv, bqkv, wproj, bproj, grads, self.num_heads, *params)


@autocast_invariant
class TimmPatchEmbed(nn.Module):
    """timm 0.3.2 ``PatchEmbed``: Conv2d(k = s = patch) + flatten(2).transpose(1, 2), as gather + MFMA GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self._views = None

    def _v(self):
        arena = get_arena(self)
        if self._views is None or self._views[0] is not arena:
            w, b = self.proj.weight, self.proj.bias
            object.__setattr__(self, "_views", (arena, arena.lp_view(w, shape=(w.shape[0], -1)), arena.f32_view(b),
                                                lambda: arena.grad_view(w), lambda: arena.grad_view(b)))
        return self._views

    def embed_tokens(self, x, ids_keep=None):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        _, w_lp, b32, gw, gb = self._v()
        nkeep = self.num_patches if ids_keep is None else ids_keep.shape[1]
        return ops.PatchEmbedFn.apply(x.contiguous().view(B, C, 1, H, W), ids_keep, w_lp, b32, gw, gb, 1, self.patch_size[0], nkeep,
                                      self.proj.weight, self.proj.bias)

    def forward(self, x):
        return self.embed_tokens(x).view(x.shape[0], self.num_patches, -1)


@autocast_invariant
class Mlp(nn.Module):
    """timm.models.vision_transformer.Mlp: fc1 -> act -> drop -> fc2 -> drop (act = exact-erf GELU)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU:
            raise NotImplementedError("only nn.GELU is fused into the fc1 epilogue")
        assert drop == 0.0, "dropout is not supported on the fused path (the reference trains with drop=0)"
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self._views = None

    def _v(self):
        arena = get_arena(self)
        if self._views is None or self._views[0] is not arena:
            f1, f2 = self.fc1, self.fc2

            def grads():
                return (arena.grad_view(f1.weight), arena.grad_view(f1.bias), arena.grad_view(f2.weight), arena.grad_view(f2.bias))
            object.__setattr__(self, "_views", (arena, arena.lp_view(f1.weight), arena.f32_view(f1.bias), arena.lp_view(f2.weight),
                                                arena.f32_view(f2.bias), grads, (f1.weight, f1.bias, f2.weight, f2.bias)))
        return self._views

    def forward(self, x, residual=None):
        _, w1, b1, w2, b2, grads, params = self._v()
        return ops.MlpFn.apply(x, residual, w1, b1, w2, b2, grads, *params)


def layer_norm(norm: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
    """fp32 [.., D] -> bf16 LayerNorm output using `norm`'s weight / bias / eps."""
    return ops.LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps)


@autocast_invariant
class Block(nn.Module):
    """Transformer Block with specified Attention function (pre-norm residual, video_vit.py:181-184).
    The residual stream is fp32; both residual adds are fused into GEMM epilogues."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, attn_func=Attention):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = attn_func(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self._views = None

    def _v(self):
        arena = get_arena(self)
        if getattr(self, "_views", None) is None or self._views[0] is not arena:
            a, m = self.attn, self.mlp
            _, wqkv, bqkv, wproj, bproj, agrads, aparams = a._v()
            _, w1, b1, w2, b2, mgrads, mparams = m._v()
            n1, n2 = self.norm1, self.norm2

            def grads():
                return (arena.grad_view(n1.weight), arena.grad_view(n1.bias), arena.grad_view(n2.weight), arena.grad_view(n2.bias)) + \
                    agrads() + mgrads()
            params = (n1.weight, n1.bias, n2.weight, n2.bias) + tuple(aparams) + tuple(mparams)
            object.__setattr__(self, "_views", (arena, (wqkv, bqkv, wproj, bproj, w1, b1, w2, b2), grads, params))
        return self._views

    def forward(self, x, final_residual=True, return_stream=False):
        """``final_residual=False`` (flash_compat, last block only): return the MLP branch without the residual stream;
        with ``return_stream`` also the stream it would have been added to, as a pair."""
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        fused = isinstance(self.attn, (Attention, TimmAttention, FlashMixer)) and isinstance(self.norm1, nn.LayerNorm) and \
            isinstance(self.norm2, nn.LayerNorm)
        no_drop = isinstance(self.drop_path, nn.Identity) or not self.training or self.drop_path.drop_prob == 0.0
        if fused:
            _, lp, grads, params = self._v()
            s1 = s2 = None
            if not no_drop:     # stochastic depth: one keep/drop draw per sample and branch, folded into the residual epilogues
                s1, s2 = self.drop_path.sample(x.shape[0], x.device), self.drop_path.sample(x.shape[0], x.device)
            out = ops.BlockFn.apply(x, self.attn.num_heads, self.norm1.eps, self.norm2.eps, lp, grads, s1, s2, final_residual,
                                    *params)
            if final_residual:
                return out
            return out if return_stream else out[0]
        if not final_residual:
            raise NotImplementedError("flash_compat needs the fused Block (Attention / TimmAttention + nn.LayerNorm)")
        if isinstance(self.drop_path, nn.Identity) or not self.training:
            x = self.attn(layer_norm(self.norm1, x), residual=x)
            x = self.mlp(layer_norm(self.norm2, x), residual=x)
        else:   # stochastic depth: the branch output has to exist on its own
            x = x + self.drop_path(self.attn(layer_norm(self.norm1, x)).float())
            x = x + self.drop_path(self.mlp(layer_norm(self.norm2, x)).float())
        return x


@autocast_invariant
class TimmBlock(Block):
    """timm 0.3.2 ``vision_transformer.Block`` signature (OCTCube/models_mae.py:40-42,54-56): fused-qkv attention."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__(dim, num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                         drop_path=drop_path, act_layer=act_layer, norm_layer=norm_layer, attn_func=TimmAttention)


@autocast_invariant
class FlashMixer(nn.Module):
    """Parameter layout of flash-attn's ``MHA`` (``Wqkv`` [3D, D] + ``out_proj``): what the reference's flash models hold
    under ``blocks.i.mixer`` (remap rules at models_mae_joint_res_flash_attn.py:693-724)."""

    def __init__(self, embed_dim, num_heads, qkv_proj_bias=True, out_proj_bias=True):
        super().__init__()
        head_dim = embed_dim // num_heads
        if head_dim not in (32, 64):
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 32 and 64")
        self.num_heads = num_heads
        self.Wqkv = nn.Linear(embed_dim, 3 * embed_dim, bias=qkv_proj_bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=out_proj_bias)
        self._views = None

    def _v(self):
        arena = get_arena(self)
        if self._views is None or self._views[0] is not arena:
            qkv, pr = self.Wqkv, self.out_proj
            has_b = qkv.bias is not None

            def grads():
                return (arena.grad_view(qkv.weight), arena.grad_view(qkv.bias) if has_b else None, arena.grad_view(pr.weight),
                        arena.grad_view(pr.bias) if pr.bias is not None else None)
            params = [qkv.weight, pr.weight] + ([qkv.bias] if has_b else []) + ([pr.bias] if pr.bias is not None else [])
            object.__setattr__(self, "_views", (arena, arena.lp_view(qkv.weight), arena.f32_view(qkv.bias) if has_b else None,
                                                arena.lp_view(pr.weight), arena.f32_view(pr.bias) if pr.bias is not None else None,
                                                grads, tuple(params)))
        return self._views

    def forward(self, x, residual=None):
        _, wqkv, bqkv, wproj, bproj, grads, params = self._v()
        return ops.AttentionFn.apply(x, residual, wqkv, bqkv, wproj, bproj, grads, self.num_heads, *params)


@autocast_invariant
class FlashBlock(Block):
    """The module ``flash_attn.models.vit.create_block`` returns, as the reference uses it: pre-norm, residual carried
    separately, ``forward(hidden_states, residual=None) -> (hidden_states, residual)`` with (flash-attn 2.5.2
    ``modules/block.py``, prenorm branch; restated -- the package is CUDA-only and absent here):

        residual = drop_path1(hidden_states) + residual          (just hidden_states for the first block)
        residual = residual + drop_path2(mixer(norm1(residual)))
        hidden_states = mlp(norm2(residual))

    so a chain of these carries the standard pre-norm stream in ``hidden_states + residual`` and whoever consumes only
    ``hidden_states`` after the last block (the reference does, models_mae_joint_res_flash_attn.py:480-489) drops the stream.
    Parameters: ``mixer.Wqkv``, ``mixer.out_proj``, ``norm1``, ``norm2``, ``mlp.fc1``, ``mlp.fc2``."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, drop_path1=0.0, drop_path2=0.0, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm):
        nn.Module.__init__(self)
        self.norm1 = norm_layer(dim)
        self.mixer = FlashMixer(dim, num_heads, qkv_proj_bias=qkv_bias)
        self.drop_path1 = DropPath(drop_path1)
        self.drop_path2 = DropPath(drop_path2)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=0.0)
        self._views = None

    @property
    def attn(self):             # Block._v() reads self.attn; not a registered child, so state_dict keys stay ``mixer.*``
        return self.mixer

    def forward(self, hidden_states, residual=None):
        x = hidden_states.float()
        if self.training and self.drop_path1.drop_prob > 0.0:
            x = self.drop_path1(x)
        if residual is not None:
            x = x + residual
        x = x.contiguous()
        _, lp, grads, params = self._v()
        s1 = None
        if self.training and self.drop_path2.drop_prob > 0.0:
            s1 = self.drop_path2.sample(x.shape[0], x.device)
        return ops.BlockFn.apply(x, self.mixer.num_heads, self.norm1.eps, self.norm2.eps, lp, grads, s1, None, False, *params)


def create_block(embed_dim, num_heads, mlp_ratio, qkv_bias, drop_rate, attn_drop_rate, drop_path1, drop_path2, norm_layer,
                 act_layer, use_flash_attn, fused_bias_fc, fused_mlp, fused_dropout_add_ln, layer_idx=None, n_layer=None,
                 last_layer_subset=False):
    """Signature of ``flash_attn.models.vit.create_block`` (call sites: models_mae_joint_res_flash_attn.py:131-149,200-218,
    OCTCube/models_mae_flash_attn.py:109-127, OCTCube/models_vit_st_flash_attn.py:123-141).  The fusion switches select CUDA
    kernels in flash-attn and are accepted and ignored: everything here is always fused."""
    if drop_rate != 0.0 or attn_drop_rate != 0.0:
        raise NotImplementedError("dropout inside the block is not built (every reference call site passes 0)")
    if last_layer_subset:
        raise NotImplementedError("last_layer_subset (cls-only last layer) is not built (every reference call site passes False)")
    return FlashBlock(embed_dim, num_heads, mlp_ratio, qkv_bias=qkv_bias, drop_path1=drop_path1, drop_path2=drop_path2,
                      act_layer=act_layer, norm_layer=norm_layer)
