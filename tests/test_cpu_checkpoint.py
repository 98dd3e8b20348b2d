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


def test_retfound_rgb_2d_checkpoint_initialises_the_3d_mae_like_the_reference(golden_dir):
    """a 2-D RGB (timm-layout) checkpoint -- RETFound / ImageNet -- must initialise the 3-D MAE exactly as the
    reference's load_model_retfound[_flash_attn] does (Pre-training/custom_util/misc.py:489-533): the 3 RGB kernels become the
    3 temporal taps (unsqueeze(1)), the embedding is mirrored into high_res_patch_embed, pos_embed is split into class +
    bicubically resized spatial table, fused qkv is split.  Fixture: oracle/gen_golden_retfound.py (the reference's own
    functions on a seeded synthetic checkpoint, loaded into the reference model)."""
    import json
    from functools import partial
    from octcubem_amd import models_mae
    from oracle import mae3d_ref as O
    z = np.load(os.path.join(golden_dir, "retfound_init.npz"))
    cfg = O.MAEConfig(**json.loads(str(z["cfg"])))
    m = models_mae.MaskedAutoencoderViT(
        input_size=cfg.input_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depth=cfg.depth,
        num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
        decoder_num_heads=cfg.decoder_num_heads, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=cfg.num_frames,
        t_patch_size=cfg.t_patch_size, sep_pos_embed=True, cls_embed=True, pred_t_dim=cfg.pred_t_dim,
        high_res_input_size=cfg.high_res_input_size)
    m.load_state_dict(O.init_params(cfg, seed=int(z["param_seed"]), bias_std=float(z["param_bias_std"])), strict=True)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    # the synthetic checkpoint, re-drawn exactly as the generator drew it
    g = torch.Generator().manual_seed(int(z["ckpt_seed"]))
    shapes = {"cls_token": (1, 1, 128), "pos_embed": (1, 197, 128), "patch_embed.proj.weight": (128, 3, 16, 16),
              "patch_embed.proj.bias": (128,), "norm.weight": (128,), "norm.bias": (128,), "decoder_embed.weight": (64, 128),
              "decoder_embed.bias": (64,), "mask_token": (1, 1, 64)}
    ck = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    for pre, n, d in (("blocks", 2, 128), ("decoder_blocks", 2, 64)):
        for i in range(n):
            ck[f"{pre}.{i}.attn.qkv.weight"] = torch.randn(3 * d, d, generator=g)
            ck[f"{pre}.{i}.attn.qkv.bias"] = torch.randn(3 * d, generator=g)
            ck[f"{pre}.{i}.attn.proj.weight"] = torch.randn(d, d, generator=g)
            ck[f"{pre}.{i}.attn.proj.bias"] = torch.randn(d, generator=g)
            for nm in ("norm1", "norm2"):
                ck[f"{pre}.{i}.{nm}.weight"] = torch.randn(d, generator=g); ck[f"{pre}.{i}.{nm}.bias"] = torch.randn(d, generator=g)
            ck[f"{pre}.{i}.mlp.fc1.weight"] = torch.randn(4 * d, d, generator=g); ck[f"{pre}.{i}.mlp.fc1.bias"] = torch.randn(4 * d, generator=g)
            ck[f"{pre}.{i}.mlp.fc2.weight"] = torch.randn(d, 4 * d, generator=g); ck[f"{pre}.{i}.mlp.fc2.bias"] = torch.randn(d, generator=g)
    assert list(ck.keys()) == json.loads(str(z["ckpt_keys"]))
    missing, unexpected = C.load_pretrained(m, {"model": ck})
    assert sorted(missing) == json.loads(str(z["missing"])) and sorted(unexpected) == json.loads(str(z["unexpected"]))
    after = m.state_dict()
    changed = sorted(k for k in after if not torch.equal(after[k], before[k]))
    assert changed == json.loads(str(z["changed"]))
    sums = np.array([[float(after[k].double().sum()), float((after[k].double() ** 2).sum())] for k in changed])
    assert np.allclose(sums, z["after_sums"], rtol=1e-6, atol=1e-6)
    for key in z.files:
        if key.startswith("after/"):
            k = key[6:]
            v = after[k].flatten()
            v = v if v.numel() <= 8192 else v[::13]
            assert torch.equal(v, torch.from_numpy(z[key])), k            # same ATen calls on the same numbers: bit-exact
    # the three RGB kernels are the three temporal taps, and the 512^2 branch starts from the same embedding
    assert torch.equal(after["patch_embed.proj.weight"][:, 0], ck["patch_embed.proj.weight"])
    assert torch.equal(after["high_res_patch_embed.proj.weight"], after["patch_embed.proj.weight"])
