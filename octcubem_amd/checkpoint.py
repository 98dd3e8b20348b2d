"""Checkpoint key / shape conversions needed to load the weights the reference ecosystem ships.

The models here keep the NON-flash parameter layout (``blocks.i.attn.{q,k,v,proj}``); the released OCTCube weights were
saved from flash-attn blocks (``blocks.i.mixer.{Wqkv,out_proj}``) and RETFound / ImageNet MAE weights from timm blocks
(``blocks.i.attn.qkv``) with a Conv2d patch embedding and a single ``pos_embed``.  Reference counterparts:
  read_in_q_k_v                 Pre-training/custom_util/misc.py:1303-1324  (timm fused qkv -> q / k / v)
  convert_patchembed_2Dto3D     Pre-training/custom_util/misc.py:1326-1329  (Conv2d weight -> Conv3d weight, T axis of 1)
  to_native_layout              models_mae_joint_res_flash_attn.py:682-775   (the flash remap, applied in reverse)
  to_flash_layout               the same rules forward (for exporting weights the reference's flash models can read)
  interpolate_pos_embed_2Dto3D  OCTCube/util/misc.py:1073-1115 (RETFound ``pos_embed`` -> ``pos_embed_spatial`` + class)
"""
from __future__ import annotations

import re
from collections import OrderedDict

import torch

from .pos_embed import _resize_square, interpolate_pos_embed, interpolate_temporal_pos_embed


def read_in_q_k_v(state_dict, num_hidden_layers, hidden_size, prefix=""):
    """Split timm's fused ``attn.qkv`` of every block into ``attn.q / attn.k / attn.v`` (in place)."""
    for i in range(num_hidden_layers):
        w = state_dict.pop(f"{prefix}blocks.{i}.attn.qkv.weight")
        b = state_dict.pop(f"{prefix}blocks.{i}.attn.qkv.bias")
        for j, n in enumerate("qkv"):
            state_dict[f"{prefix}blocks.{i}.attn.{n}.weight"] = w[j * hidden_size:(j + 1) * hidden_size, :]
            state_dict[f"{prefix}blocks.{i}.attn.{n}.bias"] = b[j * hidden_size:(j + 1) * hidden_size]


def convert_patchembed_2Dto3D(state_dict):
    """Conv2d patch-embedding weight [O, C, p, p] -> Conv3d weight [O, 1... ] by inserting the temporal axis (in place)."""
    state_dict["patch_embed.proj.weight"] = state_dict.pop("patch_embed.proj.weight").unsqueeze(1)


def to_native_layout(state_dict):
    """flash (``mixer.Wqkv`` / ``mixer.out_proj``) or timm (``attn.qkv``) block keys -> ``attn.q/k/v/proj``; other keys pass."""
    out = OrderedDict()
    for k, v in state_dict.items():
        k = k.replace(".mixer.out_proj.", ".attn.proj.")
        m = re.match(r"(.*blocks\.\d+)\.(mixer\.Wqkv|attn\.qkv)\.(weight|bias)$", k)
        if m:
            for j, n in enumerate("qkv"):
                out[f"{m.group(1)}.attn.{n}.{m.group(3)}"] = v.chunk(3, dim=0)[j].clone()
            continue
        out[k] = v
    return out


def to_flash_layout(state_dict):
    """``attn.q/k/v/proj`` -> ``mixer.Wqkv`` / ``mixer.out_proj`` (what the reference's flash models load)."""
    out = OrderedDict()
    for k, v in state_dict.items():
        m = re.match(r"(.*blocks\.\d+)\.attn\.([qkv])\.(weight|bias)$", k)
        if m:
            if m.group(2) == "q":
                pre, kind = m.group(1), m.group(3)
                out[f"{pre}.mixer.Wqkv.{kind}"] = torch.cat([state_dict[f"{pre}.attn.{n}.{kind}"] for n in "qkv"], dim=0)
            continue
        out[re.sub(r"(blocks\.\d+)\.attn\.proj\.", r"\1.mixer.out_proj.", k)] = v
    return out


def interpolate_pos_embed_2Dto3D(model, checkpoint_model):
    """A 2-D ViT's ``pos_embed`` [1, 1 + g*g, C] -> ``pos_embed_class`` + ``pos_embed_spatial`` resized to the model's grid."""
    if "pos_embed" not in checkpoint_model:
        return
    ck = checkpoint_model.pop("pos_embed")
    cls_tok, spatial = ck[:, :1], ck[:, 1:]
    orig_size = int(spatial.shape[-2] ** 0.5)
    new_size = int(model.pos_embed_spatial.shape[-2] ** 0.5)
    checkpoint_model["pos_embed_spatial"] = spatial if orig_size == new_size else _resize_square(spatial, orig_size, new_size)
    checkpoint_model["pos_embed_class"] = cls_tok


def _fit_sep_tables(model, sd, smaller_interpolate_type="interp"):
    """Resize separable tables to the shapes of the model's own parameters: ``[decoder_]pos_embed_spatial`` bicubically on
    its square grid (the 3-D MAE keeps the HIGH-RES grid there, the ViT its own -- the reference needs two functions for
    that, pos_embed.py:71-107 and custom_util/misc.py:1336-1420) and ``[decoder_]pos_embed_temporal`` as
    interpolate_temporal_pos_embed does."""
    own = model.state_dict()
    for name in ("pos_embed_spatial", "decoder_pos_embed_spatial"):
        if name in sd and name in own and sd[name].shape[-2] != own[name].shape[-2]:
            o, n = int(sd[name].shape[-2] ** 0.5), int(own[name].shape[-2] ** 0.5)
            print(f"Position interpolate {name}" + " from %dx%d to %dx%d" % (o, o, n, n))
            sd[name] = _resize_square(sd[name], o, n)
    for name in ("pos_embed_temporal", "decoder_pos_embed_temporal"):
        if name in sd and name in own and sd[name].shape[-2] != own[name].shape[-2]:
            old_t, new_t = sd[name].shape[-2], own[name].shape[-2]
            t = sd[name].permute(0, 2, 1)
            if old_t > new_t and smaller_interpolate_type == "crop":
                start = (old_t - new_t) // 2
                t = t[:, :, start:start + new_t]
            else:
                t = torch.nn.functional.interpolate(t, size=new_t, mode="linear", align_corners=False)
            sd[name] = t.permute(0, 2, 1)


def load_pretrained(model, checkpoint, strict=False, filter_keys=(), smaller_interpolate_type="interp"):
    """One call for the common cases: accepts {"model": sd} or a bare state dict in native / flash / timm layouts, a 2-D patch
    embedding, and positional tables of another grid.  Returns ``load_state_dict``'s (missing, unexpected)."""
    sd = checkpoint.get("model", checkpoint) if isinstance(checkpoint, dict) else checkpoint
    sd = to_native_layout(OrderedDict(sd))
    flash_model = any(".mixer.Wqkv." in k for k in model.state_dict())
    w = sd.get("patch_embed.proj.weight")
    if w is not None and w.dim() == 4 and model.patch_embed.proj.weight.dim() == 5:
        own_w = model.patch_embed.proj.weight                            # [O, in_chans, t_patch, p, p]
        if own_w.shape[1] == 1 and w.shape[1] == own_w.shape[2]:
            # The reference's RETFound / ImageNet initialisation (load_model_retfound[_flash_attn],
            # Pre-training/custom_util/misc.py:498-523): convert_patchembed_2Dto3D = unsqueeze(1), i.e. the 3 RGB kernels of
            # [O, 3, p, p] become the 3 temporal taps of the single-channel [O, 1, 3, p, p] kernel.
            convert_patchembed_2Dto3D(sd)
        else:                                                            # any other channel / tap count: channel sum, equal taps
            if w.shape[1] != own_w.shape[1]:
                sd["patch_embed.proj.weight"] = w.sum(dim=1, keepdim=True)
            convert_patchembed_2Dto3D(sd)
            t = own_w.shape[2]
            sd["patch_embed.proj.weight"] = sd["patch_embed.proj.weight"].repeat(1, 1, t, 1, 1) / t
        hr = getattr(model, "high_res_patch_embed", None)
        if hr is not None and "high_res_patch_embed.proj.weight" not in sd \
                and tuple(hr.proj.weight.shape) == tuple(sd["patch_embed.proj.weight"].shape):
            # the reference copies the inflated embedding into the 512^2 branch as well (:520-522)
            sd["high_res_patch_embed.proj.weight"] = sd["patch_embed.proj.weight"].clone()
            if "patch_embed.proj.bias" in sd:
                sd["high_res_patch_embed.proj.bias"] = sd["patch_embed.proj.bias"].clone()
    if "pos_embed" in sd and hasattr(model, "pos_embed_spatial"):
        interpolate_pos_embed_2Dto3D(model, sd)
    if hasattr(model, "pos_embed_spatial"):
        _fit_sep_tables(model, sd, smaller_interpolate_type)
    else:
        interpolate_pos_embed(model, sd)
    if flash_model:
        sd = to_flash_layout(sd)
    own = model.state_dict()
    for k in list(sd.keys()):
        if any(f in k for f in filter_keys) or (k in own and own[k].shape != sd[k].shape and not strict):
            if k in own and own[k].shape != sd[k].shape:
                print(f"Removing key {k} from pretrained checkpoint (shape {tuple(sd[k].shape)} vs {tuple(own[k].shape)})")
            del sd[k]
    return model.load_state_dict(sd, strict=strict)
