#!/usr/bin/env python3
"""Golden vectors for the checkpoint-conversion helpers: runs the REAL reference functions on seeded random
state dicts and stores the results (build container only).   python oracle/gen_golden_ckpt.py -> tests/golden/ckpt_utils.npz

  OCTCube/util/pos_embed.py      interpolate_pos_embed (pos_embed and pos_embed_spatial branches), interpolate_temporal_pos_embed
                                 (interp and crop), get_2d_sincos_pos_embed
  OCTCube/util/misc.py           interpolate_pos_embed_2Dto3D (the 14x14 RETFound table), convert_patchembed_2Dto3D
  Pre-training/custom_util/misc  read_in_q_k_v
  Pre-training/models_mae_joint_res_flash_attn.load_state_dict_to_backbone: the native -> flash key remap, captured by calling it
                                 on a stand-in whose nn.Module.load_state_dict records what it is handed
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


class _PE:
    def __init__(self, num_patches, frames, t_patch_size):
        self.num_patches, self.frames, self.t_patch_size = num_patches, frames, t_patch_size


def main():
    from oracle.gen_golden import install_shims
    install_shims()
    OC = "/root/reference/OCTCube"
    sys.path.insert(0, OC); os.chdir(OC)
    import util.pos_embed as rpe
    import util.misc as rmisc
    g = torch.Generator().manual_seed(41)
    save = {"seed": 41}
    # --- 2-D table with cls token: 14x14 -> 16x16
    m = types.SimpleNamespace(patch_embed=_PE(256, 1, 1), pos_embed=torch.zeros(1, 257, 32))
    ck = {"pos_embed": torch.randn(1, 197, 32, generator=g)}
    save["pe2d_in"] = ck["pos_embed"].numpy().copy()
    rpe.interpolate_pos_embed(m, ck)
    save["pe2d_out"] = ck["pos_embed"].numpy()
    # --- spatial table of the 3-D model: 32x32 -> 16x16 (T grid 20)
    m = types.SimpleNamespace(patch_embed=_PE(20 * 256, 60, 3), pos_embed_spatial=torch.zeros(1, 256, 32))
    ck = {"pos_embed_spatial": torch.randn(1, 1024, 32, generator=g), "pos_embed_temporal": torch.randn(1, 16, 32, generator=g)}
    save["pes_in"] = ck["pos_embed_spatial"].numpy().copy(); save["pet_in"] = ck["pos_embed_temporal"].numpy().copy()
    rpe.interpolate_pos_embed(m, ck)
    rpe.interpolate_temporal_pos_embed(m, ck)
    save["pes_out"] = ck["pos_embed_spatial"].numpy(); save["pet_out_interp"] = ck["pos_embed_temporal"].numpy()
    # --- temporal shrink: interp vs crop (24 -> 20)
    for kind in ("interp", "crop"):
        ck = {"pos_embed_temporal": torch.from_numpy(np.arange(24 * 4, dtype=np.float32).reshape(1, 24, 4) ** 1.1)}
        save["pet24_in"] = ck["pos_embed_temporal"].numpy().copy()
        rpe.interpolate_temporal_pos_embed(m, ck, smaller_interpolate_type=kind)
        save[f"pet24_out_{kind}"] = ck["pos_embed_temporal"].numpy()
    save["sincos_8_48_cls"] = rpe.get_2d_sincos_pos_embed(48, 8, cls_token=True)
    # --- RETFound-style 2-D pos_embed -> 3-D spatial + class (util/misc.py hard-codes the 1 + 196 split)
    m3 = types.SimpleNamespace(patch_embed=_PE(20 * 256, 60, 3), pred_t_dim=60, t_pred_patch_size=3, pos_embed_spatial=torch.zeros(1, 256, 32))
    ck = {"pos_embed": torch.randn(1, 197, 32, generator=g)}
    save["pe2d3d_in"] = ck["pos_embed"].numpy().copy()
    rmisc.interpolate_pos_embed_2Dto3D(m3, ck)
    save["pe2d3d_spatial"] = ck["pos_embed_spatial"].numpy(); save["pe2d3d_class"] = ck["pos_embed_class"].numpy()
    ck = {"patch_embed.proj.weight": torch.randn(8, 1, 16, 16, generator=g)}
    save["conv2d_in"] = ck["patch_embed.proj.weight"].numpy().copy()
    rmisc.convert_patchembed_2Dto3D(ck)
    save["conv3d_out"] = ck["patch_embed.proj.weight"].numpy()
    # --- read_in_q_k_v
    for m_ in [k for k in list(sys.modules) if k == "util" or k.startswith("util.")]:
        del sys.modules[m_]
    PT = "/root/reference/Pre-training"
    sys.path.insert(0, PT); os.chdir(PT)
    import custom_util.misc as pmisc
    sd = {}
    for i in range(2):
        sd[f"blocks.{i}.attn.qkv.weight"] = torch.randn(24, 8, generator=g)
        sd[f"blocks.{i}.attn.qkv.bias"] = torch.randn(24, generator=g)
        save[f"qkv_in/{i}/weight"] = sd[f"blocks.{i}.attn.qkv.weight"].numpy().copy()
        save[f"qkv_in/{i}/bias"] = sd[f"blocks.{i}.attn.qkv.bias"].numpy().copy()
    pmisc.read_in_q_k_v(sd, 2, 8)
    for k, v in sd.items():
        save[f"qkv_out/{k}"] = v.numpy()
    # --- native -> flash remap, as the reference's loader performs it before handing over to nn.Module.load_state_dict
    import models_mae_joint_res_flash_attn as ref
    captured = {}

    stub = ref.MaskedAutoencoderViT.__new__(ref.MaskedAutoencoderViT)      # no __init__: only .blocks / .decoder_blocks are read
    torch.nn.Module.__init__(stub)
    object.__setattr__(stub, "blocks", [None, None]); object.__setattr__(stub, "decoder_blocks", [None])
    orig_load = torch.nn.Module.load_state_dict
    torch.nn.Module.load_state_dict = lambda self, sd_, strict=False: (captured.update(sd_), ([], []))[1]
    native = {"cls_token": torch.randn(1, 1, 8, generator=g), "patch_embed.proj.weight": torch.randn(8, 1, 3, 4, 4, generator=g)}
    for pre, n in (("blocks", 2), ("decoder_blocks", 1)):
        for i in range(n):
            for nm in ("q", "k", "v", "proj"):
                native[f"{pre}.{i}.attn.{nm}.weight"] = torch.randn(8, 8, generator=g)
                native[f"{pre}.{i}.attn.{nm}.bias"] = torch.randn(8, generator=g)
            native[f"{pre}.{i}.norm1.weight"] = torch.randn(8, generator=g)
            native[f"{pre}.{i}.mlp.fc1.weight"] = torch.randn(16, 8, generator=g)
    for k, v in native.items():
        save[f"native/{k}"] = v.numpy().copy()
    stub.load_state_dict_to_backbone(dict(native))
    torch.nn.Module.load_state_dict = orig_load
    for k, v in captured.items():
        save[f"flash/{k}"] = v.numpy()
    out = os.path.join(ROOT, "tests", "golden", "ckpt_utils.npz")
    np.savez_compressed(out, **save)
    print("wrote", out, os.path.getsize(out), "bytes;", len(captured), "flash keys:", sorted(captured)[:6])


if __name__ == "__main__":
    main()
