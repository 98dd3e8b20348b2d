#!/usr/bin/env python3
"""Full-size pins for BASELINE config 5 (retinal-COEM OCTCube-IR: 3-D OCT tower + 2-D en-face tower, contrastive), from the
reference's OWN classes, build container only (see gen_golden.py for the import shims):

    python oracle/gen_golden_coem_full.py  ->  tests/golden/coem_l_pins.npz

The shipped model config (retinal-COEM/src/open_clip/model_configs/vit_large_patch16_retFound-vit_large_patch16_OCTCube.json:
ViT-L ST tower on 60x256x256 volumes, ViT-L 2-D tower on 3x224x224 en-face images, embed_dim 512) is instantiated with
  * tower A: OCTCube/models_vit_st_flash_attn.py ``vit_large_patch16(num_classes=512, ...)`` -- the class the config's
    "ViT_ST_nodrop" tower is a dropout-free copy of (open_clip/models_vit_st_flash_attn_nodrop.py), use_flash_attn=False (flash-attn
    is CUDA-only and absent: the non-flash semantics are the pinned ones), drop_path 0, eval mode;
  * tower B: OCTCube/models_vit.py ``VisionTransformer`` on the restated timm 0.3.2 base (as gen_golden_vit2d.py), global_pool;
  * CustomTextCLIP.forward (open_clip/model.py:670-682) = L2-normalised tower outputs + logit_scale.exp() -- three lines,
    restated here (model.py imports a dozen un-vendored packages);
  * the loss: retinal-COEM/src/open_clip/loss.py ``ClipLoss`` itself (loaded as a single file).
B = 2 pairs.  The ST tower's (2,16,5121,5121) fp32 score tensors are kept out of the autograd graph by running each of the
reference's own Blocks under torch.utils.checkpoint (numerics unchanged), as gen_golden_fullsize.py does.
Only numbers are stored."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)

NSAMP = 2048
SAMPLED_A = ["blocks.0.attn.q.weight", "blocks.0.mlp.fc1.weight", "blocks.11.attn.v.weight", "blocks.23.attn.k.weight",
             "blocks.23.mlp.fc2.weight", "pos_embed_spatial", "patch_embed.proj.weight", "head.weight", "head.bias"]
SAMPLED_B = ["blocks.0.attn.qkv.weight", "blocks.12.mlp.fc1.weight", "blocks.23.attn.proj.weight", "pos_embed", "cls_token",
             "patch_embed.proj.weight", "head.weight", "fc_norm.weight"]


def main():
    from functools import partial
    from torch.utils.checkpoint import checkpoint
    from gen_golden import install_shims
    from gen_golden_coem import load_ref
    install_shims()
    from oracle import vit_ref as V
    OC = "/root/reference/OCTCube"
    sys.path.insert(0, OC); os.chdir(OC)
    import models_vit_st_flash_attn as ref_st
    import models_vit as ref_2d
    torch.set_num_threads(os.cpu_count() or 8)
    cA = V.ViTSTConfig(num_frames=60, t_patch_size=3, img_size=256, patch_size=16, in_chans=1, num_classes=512, embed_dim=1024,
                       depth=24, num_heads=16, global_pool=True)
    cB = V.ViT2DConfig(img_size=224, patch_size=16, in_chans=3, num_classes=512, embed_dim=1024, depth=24, num_heads=16, global_pool=True)
    mA = ref_st.vit_large_patch16(num_frames=60, t_patch_size=3, img_size=256, in_chans=1, num_classes=512, global_pool=True,
                                  sep_pos_embed=True, cls_embed=True, drop_path_rate=0.0)
    mB = ref_2d.VisionTransformer(global_pool=True, img_size=224, patch_size=16, in_chans=3, num_classes=512, embed_dim=1024, depth=24,
                                  num_heads=16, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    PA = V.init_from_shapes(V.vit_st_param_shapes(cA), seed=71)
    PB = V.init_from_shapes(V.vit2d_param_shapes(cB), seed=72)
    assert set(mB.state_dict()) == set(PB)
    mA.load_state_dict(PA, strict=True); mB.load_state_dict(PB, strict=True)
    mA.eval(); mB.eval()
    for blk in mA.blocks:
        blk.forward = (lambda f: (lambda x: checkpoint(f, x, use_reentrant=False)))(blk.forward)
    vol = torch.rand(2, 1, 60, 256, 256, generator=torch.Generator().manual_seed(0))
    ir = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    ls = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
    t0 = time.time()
    fa_raw = mA(vol)
    fb_raw = mB(ir)
    fa = torch.nn.functional.normalize(fa_raw, dim=-1)            # CustomTextCLIP.forward, model.py:680-682
    fb = torch.nn.functional.normalize(fb_raw, dim=-1)
    loss = load_ref().ClipLoss()(fa, fb, ls.exp())
    print(f"forward {time.time() - t0:.0f} s, loss {float(loss):.6f}")
    loss.backward()
    print(f"forward + backward {time.time() - t0:.0f} s")
    save = {"loss": float(loss), "feat_a": fa.detach().numpy(), "feat_b": fb.detach().numpy(), "raw_a": fa_raw.detach().numpy(),
            "raw_b": fb_raw.detach().numpy(), "logit_scale_grad": float(ls.grad), "seed_a": 71, "seed_b": 72, "vol_seed": 0, "ir_seed": 1,
            "cfg_a": json.dumps(cA.__dict__), "cfg_b": json.dumps(cB.__dict__)}
    for tag, m, sampled in (("a", mA, SAMPLED_A), ("b", mB, SAMPLED_B)):
        grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach() for k, p in m.named_parameters()}
        names = sorted(grads)
        save[f"grad_names_{tag}"] = json.dumps(names)
        save[f"grad_norms_{tag}"] = np.array([float(grads[k].double().norm()) for k in names])
        save[f"tower_grad_norm_{tag}"] = float(torch.norm(torch.stack([grads[k].double().norm() for k in names])))
        for k in sampled:
            g = grads[k].flatten()
            step = max(1, g.numel() // NSAMP)
            save[f"gsample_{tag}/{k}"] = g[::step][:NSAMP].numpy().copy()
            save[f"gstep_{tag}/{k}"] = step
        print(f"  tower {tag}: grad norm {save[f'tower_grad_norm_{tag}']:.6g}")
    outp = os.path.join(ROOT, "tests", "golden", "coem_l_pins.npz")
    np.savez_compressed(outp, **save)
    print("wrote", outp, os.path.getsize(outp), "bytes")


if __name__ == "__main__":
    main()
