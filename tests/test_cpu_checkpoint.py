"""CPU: checkpoint-conversion helpers against outputs of the reference's own functions on seeded inputs
(oracle/gen_golden_ckpt.py -> tests/golden/ckpt_utils.npz).  Pure host-side tensor bookkeeping: results must be bit-exact
(same ATen interpolate calls), the key remaps must round-trip."""
import os
import types

import numpy as np
import pytest
import torch

from octcubem_amd import checkpoint as C
from octcubem_amd import pos_embed as PE


class _PEmb:
    def __init__(self, num_patches, frames, t_patch_size):
        self.num_patches, self.frames, self.t_patch_size = num_patches, frames, t_patch_size


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "ckpt_utils.npz"))


def T(a):
    return torch.from_numpy(np.array(a))


def test_interpolate_pos_embed_2d_and_spatial(z):
    m = types.SimpleNamespace(patch_embed=_PEmb(256, 1, 1), pos_embed=torch.zeros(1, 257, 32))
    ck = {"pos_embed": T(z["pe2d_in"])}
    PE.interpolate_pos_embed(m, ck)
    assert torch.equal(ck["pos_embed"], T(z["pe2d_out"]))
    m = types.SimpleNamespace(patch_embed=_PEmb(20 * 256, 60, 3), pos_embed_spatial=torch.zeros(1, 256, 32))
    ck = {"pos_embed_spatial": T(z["pes_in"]), "pos_embed_temporal": T(z["pet_in"])}
    PE.interpolate_pos_embed(m, ck)
    PE.interpolate_temporal_pos_embed(m, ck)
    assert torch.equal(ck["pos_embed_spatial"], T(z["pes_out"])) and torch.equal(ck["pos_embed_temporal"], T(z["pet_out_interp"]))
    same = {"pos_embed_spatial": torch.ones(1, 256, 32), "pos_embed_temporal": torch.ones(1, 20, 32)}
    PE.interpolate_pos_embed(m, same); PE.interpolate_temporal_pos_embed(m, same)        # matching grids: untouched
    assert same["pos_embed_spatial"].shape == (1, 256, 32) and same["pos_embed_temporal"].shape == (1, 20, 32)


def test_temporal_shrink_interp_and_crop(z):
    m = types.SimpleNamespace(patch_embed=_PEmb(20 * 256, 60, 3))
    for kind in ("interp", "crop"):
        ck = {"pos_embed_temporal": T(z["pet24_in"])}
        PE.interpolate_temporal_pos_embed(m, ck, smaller_interpolate_type=kind)
        assert torch.equal(ck["pos_embed_temporal"], T(z[f"pet24_out_{kind}"])), kind


def test_sincos_table(z):
    assert np.array_equal(PE.get_2d_sincos_pos_embed(48, 8, cls_token=True), z["sincos_8_48_cls"])
    from octcubem_amd import models_mae_2d
    assert np.array_equal(models_mae_2d.get_2d_sincos_pos_embed(48, 8, True), z["sincos_8_48_cls"])


def test_2d_to_3d_conversions(z):
    m3 = types.SimpleNamespace(pos_embed_spatial=torch.zeros(1, 256, 32))
    ck = {"pos_embed": T(z["pe2d3d_in"])}
    C.interpolate_pos_embed_2Dto3D(m3, ck)
    assert "pos_embed" not in ck
    assert torch.equal(ck["pos_embed_spatial"], T(z["pe2d3d_spatial"])) and torch.equal(ck["pos_embed_class"], T(z["pe2d3d_class"]))
    ck = {"patch_embed.proj.weight": T(z["conv2d_in"])}
    C.convert_patchembed_2Dto3D(ck)
    assert torch.equal(ck["patch_embed.proj.weight"], T(z["conv3d_out"]))


def test_read_in_q_k_v(z):
    sd = {}
    for i in range(2):
        sd[f"blocks.{i}.attn.qkv.weight"] = T(z[f"qkv_in/{i}/weight"]); sd[f"blocks.{i}.attn.qkv.bias"] = T(z[f"qkv_in/{i}/bias"])
    C.read_in_q_k_v(sd, 2, 8)
    ref = {k[len("qkv_out/"):]: T(z[k]) for k in z.files if k.startswith("qkv_out/")}
    assert set(sd) == set(ref)
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k


def test_flash_key_remap_matches_reference_and_round_trips(z):
    native = {k[len("native/"):]: T(z[k]) for k in z.files if k.startswith("native/")}
    flash_ref = {k[len("flash/"):]: T(z[k]) for k in z.files if k.startswith("flash/")}
    # the reference flattens a 4-D (Conv2d) patch embedding only; this one is 5-D and passes through
    mine = C.to_flash_layout(native)
    assert set(mine) == set(flash_ref)
    for k in flash_ref:
        assert torch.equal(mine[k], flash_ref[k]), k
    back = C.to_native_layout(mine)
    assert set(back) == set(native)
    for k in native:
        assert torch.equal(back[k], native[k]), k
    timm = {"blocks.0.attn.qkv.weight": torch.arange(24.0).view(12, 2), "blocks.0.attn.qkv.bias": torch.arange(12.0)}
    nat = C.to_native_layout(timm)
    assert torch.equal(nat["blocks.0.attn.k.weight"], timm["blocks.0.attn.qkv.weight"][4:8])
    assert torch.equal(nat["blocks.0.attn.v.bias"], timm["blocks.0.attn.qkv.bias"][8:])
