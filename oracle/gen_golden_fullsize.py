#!/usr/bin/env python3
"""Full-size pins from the REAL reference (build container only; see gen_golden.py for the import shims).

    python oracle/gen_golden_fullsize.py [--only mae|st]

  tests/golden/vitl_bwd_pins.npz   ViT-L 3-D MAE (BASELINE config 2: (1,1,60,256,256), mask 0.75, decoder 512x8x16),
        the same weights / volume / noise seed as vitl_pins.npz, now WITH the backward of
        Pre-training/models_mae_joint_res_flash_attn.py:669-680 (non-flash): loss, global gradient norm
        (get_grad_norm_ form), the 2-norm of EVERY parameter's gradient, strided samples of selected gradient tensors.
  tests/golden/vit_st_l_pins.npz   ViT-L spatio-temporal fine-tune model (BASELINE config 4:
        OCTCube/models_vit_st_flash_attn.py:181-258, use_flash_attn=False, (1,1,60,256,256) -> 8 logits, N = 5121 tokens,
        16 heads x 64): logits, pooled embedding, cross-entropy loss, and the same gradient pins.  The non-flash attention
        materialises a (1,16,5121,5121) fp32 score tensor per layer (1.7 GB, twice with its softmax): 24 layers of saved
        activations do not fit this container's 64 GB, so each of the reference's OWN Block modules is run under
        torch.utils.checkpoint (re-executes the same module in backward; numerics unchanged).
Only numbers are stored (no reference source text).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from gen_golden import install_shims, build_reference, run_reference, tie_free, REF   # noqa: E402

# gradient tensors sampled element-wise (every tensor gets its norm pinned)
MAE_SAMPLED = ["decoder_blocks.7.attn.q.weight", "decoder_blocks.7.attn.k.weight", "decoder_blocks.0.attn.v.weight",
               "decoder_blocks.3.mlp.fc2.weight", "decoder_pred.weight", "decoder_embed.weight", "mask_token",
               "decoder_pos_embed_spatial", "blocks.0.mlp.fc1.weight", "blocks.0.attn.q.weight", "blocks.11.attn.proj.weight",
               "blocks.23.attn.k.weight", "blocks.23.norm1.weight", "blocks.12.mlp.fc1.bias", "pos_embed_spatial",
               "pos_embed_temporal", "cls_token", "patch_embed.proj.weight", "patch_embed.proj.bias", "norm.weight"]
ST_SAMPLED = ["blocks.0.attn.q.weight", "blocks.0.attn.k.weight", "blocks.0.mlp.fc1.weight", "blocks.11.attn.v.weight",
              "blocks.11.attn.proj.weight", "blocks.23.attn.q.weight", "blocks.23.mlp.fc2.weight", "blocks.23.norm2.weight",
              "pos_embed_spatial", "pos_embed_temporal", "pos_embed_class", "cls_token", "patch_embed.proj.weight",
              "patch_embed.proj.bias", "head.weight", "head.bias", "blocks.5.mlp.fc1.bias"]
NSAMP = 4096


def grad_pins(named_grads, sampled):
    out = {}
    names = sorted(named_grads)
    out["grad_names"] = json.dumps(names)
    out["grad_norms"] = np.array([float(named_grads[k].double().norm()) for k in names])
    out["global_grad_norm"] = float(torch.norm(torch.stack([named_grads[k].double().norm() for k in names])))
    for k in sampled:
        g = named_grads[k].flatten()
        step = max(1, g.numel() // NSAMP)
        out[f"gsample/{k}"] = g[::step][:NSAMP].numpy().copy()
        out[f"gstep/{k}"] = step
    return out


def gen_mae(out_dir):
    from oracle import mae3d_ref as O
    sys.path.insert(0, REF)
    os.chdir(REF)
    cfgL = O.VIT_L
    modelL = build_reference(cfgL)
    PL = O.init_params(cfgL, seed=0, bias_std=0.0)
    modelL.load_state_dict(PL, strict=True)
    modelL.train()
    imgsL = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(0))
    old = np.load(os.path.join(out_dir, "vitl_pins.npz"))
    seedL = int(old["noise_seed"])
    torch.manual_seed(seedL)
    assert tie_free(torch.rand(1, 5120))
    t0 = time.time()
    lossL, predL, maskL = run_reference(modelL, imgsL, seedL, 0.75)
    modelL.zero_grad()
    lossL.backward()
    print(f"ViT-L MAE fwd+bwd: {time.time() - t0:.0f} s, loss {float(lossL):.6f} (forward pin {float(old['loss']):.6f})")
    assert abs(float(lossL) - float(old["loss"])) <= 1e-6 * abs(float(old["loss"]))
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach() for k, p in modelL.named_parameters()}
    unused = sorted(k for k, p in modelL.named_parameters() if p.grad is None)
    save = {"loss": float(lossL), "noise_seed": seedL, "mask_ratio": 0.75, "param_seed": 0, "param_bias_std": 0.0,
            "img_seed": 0, "unused": json.dumps(unused)}
    save.update(grad_pins(grads, MAE_SAMPLED))
    np.savez_compressed(os.path.join(out_dir, "vitl_bwd_pins.npz"), **save)
    print("  global grad norm", save["global_grad_norm"], "unused", unused)


def gen_st(out_dir):
    from oracle import vit_ref as V
    from functools import partial
    from torch.utils.checkpoint import checkpoint
    OC = "/root/reference/OCTCube"
    for m_ in [k for k in list(sys.modules) if k == "util" or k.startswith("util.")]:
        del sys.modules[m_]
    sys.path.insert(0, OC)
    os.chdir(OC)
    import models_vit_st_flash_attn as ref_st
    cfg = V.ViTSTConfig(num_frames=60, t_patch_size=3, img_size=256, patch_size=16, in_chans=1, num_classes=8, embed_dim=1024,
                        depth=24, num_heads=16, global_pool=True)
    m = ref_st.vit_large_patch16(num_frames=60, t_patch_size=3, img_size=256, in_chans=1, num_classes=8, global_pool=True,
                                 sep_pos_embed=True, cls_embed=True, drop_path_rate=0.0)
    P = V.init_from_shapes(V.vit_st_param_shapes(cfg), seed=41)
    m.load_state_dict(P, strict=True)
    m.eval()                                    # dropout (p = 0.5 before the head) off: parity run
    for blk in m.blocks:                         # the reference's own Block.forward, re-executed in backward
        orig = blk.forward
        blk.forward = (lambda f: (lambda x: checkpoint(f, x, use_reentrant=False)))(orig)
    x = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(0))
    tgt = torch.tensor([3])
    t0 = time.time()
    logits, emb = m(x, return_embeddings=True)
    loss = torch.nn.functional.cross_entropy(logits, tgt)
    print(f"ViT-L ST forward: {time.time() - t0:.0f} s, logits {logits.detach().numpy().round(4)}, loss {float(loss):.6f}")
    m.zero_grad()
    loss.backward()
    print(f"ViT-L ST fwd+bwd: {time.time() - t0:.0f} s")
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach() for k, p in m.named_parameters()}
    unused = sorted(k for k, p in m.named_parameters() if p.grad is None)
    save = {"loss": float(loss), "logits": logits.detach().numpy(), "embedding": emb.detach().numpy(), "target": tgt.numpy(),
            "param_seed": 41, "param_bias_std": 0.02, "img_seed": 0, "cfg": json.dumps(cfg.__dict__), "unused": json.dumps(unused)}
    save.update(grad_pins(grads, ST_SAMPLED))
    np.savez_compressed(os.path.join(out_dir, "vit_st_l_pins.npz"), **save)
    print("  global grad norm", save["global_grad_norm"], "unused", unused)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["mae", "st"], default=None)
    args = ap.parse_args()
    install_shims()
    out_dir = os.path.join(ROOT, "tests", "golden")
    torch.set_num_threads(os.cpu_count() or 8)
    if args.only in (None, "mae"):
        gen_mae(out_dir)
    if args.only in (None, "st"):
        gen_st(out_dir)


if __name__ == "__main__":
    main()
