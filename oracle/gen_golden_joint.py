#!/usr/bin/env python3
"""Golden trajectory for the joint 3-D + 2-D/512 pre-training loop, produced by the REAL reference loop
(build container only).   python oracle/gen_golden_joint.py -> tests/golden/joint_small.npz

Runs Pre-training/engine_pretrain.train_one_epoch_joint (its own lr_sched, misc.get_mask, NativeScalerWithGradNormCount) around
the reference MaskedAutoencoderViT (non-flash blocks) reduced to width 64 / 1 block, decoder 32 / 1 block, 6 frames at 256x256
(get_mask hard-codes the 16x16 spatial grid) plus (B,1,3,512,512) "2-D" samples through high_res_patch_embed; 4 iterations,
accum_iter 2, clip_grad 1.0, torch AdamW(betas 0.9/0.95) as main_pretrain_oph_joint_2d512_flash_attn.py:441-455 builds it.
Masking noise comes from the global torch RNG (manual_seed(seed) once, then rand(N, L) per forward in call order); tests
replay exactly that sequence.  Volumes are regenerated from ``data_seed``.
"""
import json
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
PT = "/root/reference/Pre-training"


def tie_free(n):
    s, _ = torch.sort(n, dim=1)
    return bool((s[:, 1:] != s[:, :-1]).all())


def main():
    from oracle.gen_golden import install_shims, build_reference
    from oracle import mae3d_ref as O
    install_shims()
    torch.cuda.synchronize = lambda *a, **k: None
    sys.path.insert(0, PT); os.chdir(PT)
    import engine_pretrain as ref_engine
    import custom_util.misc as ref_misc

    cfg = O.MAEConfig(input_size=256, in_chans=1, embed_dim=64, depth=1, num_heads=1, decoder_embed_dim=32, decoder_depth=1,
                      decoder_num_heads=1, num_frames=6, t_patch_size=3, pred_t_dim=6, high_res_input_size=512)
    model = build_reference(cfg)
    P0 = O.init_params(cfg, seed=17, bias_std=0.02)
    model.load_state_dict(P0, strict=True)

    class Wrap(nn.Module):          # the loop calls model(...) and model.module.forward_patch_embed(...) (DDP-wrapped)
        def __init__(self, m):
            super().__init__(); self.module = m

        def forward(self, *a, **k):
            return self.module(*a, **k)
    wrapped = Wrap(model)
    n_iter, B3, B2 = 4, 2, 2
    gd = torch.Generator().manual_seed(19)
    vols = torch.rand(n_iter, B3, 1, 6, 256, 256, generator=gd)
    imgs2d = torch.rand(n_iter, B2, 1, 3, 512, 512, generator=gd)
    L3, L2 = 2 * 256, 1024
    for seed in range(500, 600):
        torch.manual_seed(seed)
        ok = all(tie_free(torch.rand(B3, L3)) and tie_free(torch.rand(B2, L2)) for _ in range(n_iter))
        if ok:
            break
    frames = [[[f"v{it}_{j}_f{nf}" for j in range(B3)] for nf in range(6)] for it in range(n_iter)]
    loader3d = [(vols[it], ([f"vol{it}_{j}" for j in range(B3)], {"frames": frames[it]})) for it in range(n_iter)]
    loader2d = [(imgs2d[it], None) for it in range(n_iter)]
    img_dict = {f: {} for it in range(n_iter) for nf in range(6) for f in frames[it][nf]}
    groups = ref_misc.add_weight_decay(model, 0.05)
    opt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.95))
    scaler = ref_misc.NativeScalerWithGradNormCount(fp32=True)
    norms = []

    def rec_scaler(loss, optimizer, **kw):
        n = scaler(loss, optimizer, **kw)
        norms.append(-1.0 if n is None else float(n))
        return n

    class Args:
        accum_iter = 2; lr = 1e-3; min_lr = 1e-6; warmup_epochs = 1; epochs = 4; mask_ratio = 0.75; batch_size_2d = B2
        clip_grad = 1.0; num_checkpoint_del = 0; repeat_aug = 1; resume = ""; resume_type = ""
    torch.manual_seed(seed)
    stats = ref_engine.train_one_epoch_joint(wrapped, loader3d, opt, torch.device("cpu"), 1, rec_scaler, loader2d, img_dict, 0.8,
                                             log_writer=None, args=Args, fp32=True)
    save = {"param_seed": 17, "cfg": json.dumps(cfg.__dict__), "data_seed": 19, "noise_seed": seed, "norms": np.array(norms),
            "vol_checksum": float(vols.double().sum()), "img2d_checksum": float(imgs2d.double().sum()),
            "stats": json.dumps({k: float(v) for k, v in stats.items() if k in ("loss", "loss_2d", "loss_all", "lr", "mask_ratio", "mask_ratio_2d")}),
            "frame_dict": json.dumps(img_dict), "param_checksum": np.array([float(v.double().sum()) for v in P0.values()]).sum()}
    for k, v in model.state_dict().items():
        save[f"final/{k}"] = v.numpy() if v.numel() <= 8192 else v.flatten()[::7].numpy()
    out = os.path.join(ROOT, "tests", "golden", "joint_small.npz")
    np.savez_compressed(out, **save)
    print("joint loop stats", stats, "norms", norms, "seed", seed)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
