#!/usr/bin/env python3
"""Golden vectors for the 2-D ViT tower of the COEM model, build container only.

    python oracle/gen_golden_vit2d.py  ->  tests/golden/vit2d_small.npz

Runs the reference's OCTCube/models_vit.py ``VisionTransformer`` (its own forward_features: cls concat, pos_embed, blocks,
global average pool + fc_norm or norm + cls) on top of a RESTATED timm 0.3.2 ``VisionTransformer`` base class (timm is not
vendored by the reference and not installed here: gen_golden.install_shims; pinned ``timm==0.3.2`` by OCTCube/main_pretrain.py:27).
What is pinned: the reference subclass's arithmetic and parameter naming; the base class's Block / PatchEmbed / head are the
restatement (same as the 2-D MAE fixture)."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)


def main():
    from gen_golden import install_shims
    install_shims()
    OC = "/root/reference/OCTCube"
    sys.path.insert(0, OC); os.chdir(OC)
    from functools import partial
    from oracle import vit_ref as V
    import models_vit as ref
    save = {}
    x = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(8))
    tgt = torch.tensor([2, 0, 5])
    save["x"] = x.numpy(); save["target"] = tgt.numpy()
    for gp in (True, False):
        cfg = V.ViT2DConfig(img_size=64, patch_size=16, in_chans=3, num_classes=16, embed_dim=128, depth=2, num_heads=2, global_pool=gp)
        m = ref.VisionTransformer(global_pool=gp, img_size=64, patch_size=16, in_chans=3, num_classes=16, embed_dim=128, depth=2,
                                  num_heads=2, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6))
        P = V.init_from_shapes(V.vit2d_param_shapes(cfg), seed=61)
        assert set(m.state_dict()) == set(P), (set(m.state_dict()) ^ set(P))
        m.load_state_dict(P, strict=True)
        m.eval()
        out = m(x)
        loss = torch.nn.functional.cross_entropy(out, tgt)
        m.zero_grad(); loss.backward()
        tag = "gp1" if gp else "gp0"
        save[f"{tag}/cfg"] = json.dumps(cfg.__dict__); save[f"{tag}/out"] = out.detach().numpy(); save[f"{tag}/loss"] = loss.detach().numpy()
        for k, p in m.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            save[f"{tag}/gnorm/{k}"] = float(g.double().norm())
            save[f"{tag}/grad/{k}"] = g.numpy() if g.numel() <= 4096 else g.flatten()[::11].numpy()
    save["param_seed"] = 61
    outp = os.path.join(ROOT, "tests", "golden", "vit2d_small.npz")
    np.savez_compressed(outp, **save)
    print("wrote", outp, os.path.getsize(outp), "bytes")


if __name__ == "__main__":
    main()
