#!/usr/bin/env python3
"""Golden fixture for RETFound / ImageNet (2-D, RGB, timm layout) initialisation of the 3-D MAE (build container only).

    python oracle/gen_golden_retfound.py  ->  tests/golden/retfound_init.npz

Runs the REAL reference functions in the order load_model_retfound / load_model_retfound_flash_attn apply them
(Pre-training/custom_util/misc.py:489-533): read_in_q_k_v (encoder and decoder; the reference hard-codes the ViT-L sizes, here
the reduced model's), interpolate_pos_embed_2Dto3D(high_res_patch_embed=True), convert_patchembed_2Dto3D (unsqueeze(1): the 3 RGB
kernels become the 3 temporal taps), the copy of patch_embed.proj.* into high_res_patch_embed.proj.*, then the reference
model's own load_state_dict(strict=False).  Stored: the synthetic 2-D checkpoint and the reference model's state afterwards."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    from gen_golden import install_shims, build_reference, REF
    install_shims()
    sys.path.insert(0, REF); os.chdir(REF)
    from oracle import mae3d_ref as O
    import custom_util.misc as pmisc
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=2,
                      decoder_num_heads=2, num_frames=12, t_patch_size=3, pred_t_dim=12, high_res_input_size=128)
    model = build_reference(cfg)
    model.load_state_dict(O.init_params(cfg, seed=3, bias_std=0.02), strict=True)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(77)
    ck = {"cls_token": torch.randn(1, 1, 128, generator=g), "pos_embed": torch.randn(1, 197, 128, generator=g),
          "patch_embed.proj.weight": torch.randn(128, 3, 16, 16, generator=g), "patch_embed.proj.bias": torch.randn(128, generator=g),
          "norm.weight": torch.randn(128, generator=g), "norm.bias": torch.randn(128, generator=g),
          "decoder_embed.weight": torch.randn(64, 128, generator=g), "decoder_embed.bias": torch.randn(64, generator=g),
          "mask_token": torch.randn(1, 1, 64, generator=g)}
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
    save = {"ckpt_seed": 77, "ckpt_keys": __import__("json").dumps(list(ck.keys()))}     # the test re-draws the checkpoint
    sd = dict(ck)
    pmisc.read_in_q_k_v(sd, num_hidden_layers=2, hidden_size=128)
    pmisc.read_in_q_k_v(sd, num_hidden_layers=2, hidden_size=64, prefix="decoder_")
    pmisc.interpolate_pos_embed_2Dto3D(model, sd, high_res_patch_embed=True)
    pmisc.convert_patchembed_2Dto3D(sd)
    sd["high_res_patch_embed.proj.weight"] = sd["patch_embed.proj.weight"]
    sd["high_res_patch_embed.proj.bias"] = sd["patch_embed.proj.bias"]
    sd.pop("pos_embed", None)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    after = model.state_dict()
    changed = sorted(k for k in after if not torch.equal(after[k], before[k]))
    # every changed tensor: (sum, sum of squares); selected ones element-wise (large ones strided)
    save["changed"] = __import__("json").dumps(changed)
    save["after_sums"] = np.array([[float(after[k].double().sum()), float((after[k].double() ** 2).sum())] for k in changed])
    for k in ("patch_embed.proj.weight", "high_res_patch_embed.proj.weight", "high_res_patch_embed.proj.bias", "pos_embed_spatial",
              "pos_embed_class", "blocks.1.attn.q.weight", "blocks.1.attn.k.bias", "blocks.0.attn.v.weight",
              "decoder_blocks.1.attn.k.weight", "decoder_blocks.0.attn.q.bias", "blocks.1.mlp.fc1.weight", "cls_token"):
        v = after[k].flatten()
        save[f"after/{k}"] = (v if v.numel() <= 8192 else v[::13]).numpy().copy()
    import json
    save["missing"] = json.dumps(sorted(missing)); save["unexpected"] = json.dumps(sorted(unexpected))
    save["cfg"] = json.dumps(cfg.__dict__); save["param_seed"] = 3; save["param_bias_std"] = 0.02
    out = os.path.join(ROOT, "tests", "golden", "retfound_init.npz")
    np.savez_compressed(out, **save)
    print("wrote", out, os.path.getsize(out), "bytes; changed", len(changed), "missing", sorted(missing)[:8], "unexpected", sorted(unexpected))


if __name__ == "__main__":
    main()
