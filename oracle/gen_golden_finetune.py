#!/usr/bin/env python3
"""Golden vectors for the ST fine-tune path, made by running the REAL reference loop (build container only).

    python oracle/gen_golden_finetune.py            # writes tests/golden/finetune_small.npz

What runs from the reference's own source files (/root/reference/OCTCube):
  models_vit_st_flash_attn.VisionTransformer (non-flash blocks)   engine_finetune.train_one_epoch
  util/lr_decay.param_groups_lrd + get_layer_id_for_vit            util/lr_sched.adjust_learning_rate
  util/misc.NativeScalerWithGradNormCount (clip_grad_norm_ branch)  torch.optim.AdamW(param_groups, lr) as main_finetune.py:303
on a reduced-width model (128 wide, 2 blocks, 12x64x64 volumes), 6 iterations of batch 2, accum_iter 2, clip 1.0,
CrossEntropyLoss (main_finetune.py:312), dropout / drop_path 0 so that the trajectory is deterministic.  The volumes are
not stored: tests regenerate them from ``data_seed`` (torch CPU generator) and check ``x_checksum``.

Shims on top of gen_golden.install_shims(): ``timm.data.Mixup`` and ``timm.utils.accuracy`` (imported, never called
here), ``pycm`` (absent), and ``torch.cuda.synchronize`` (the loop calls it every iteration; no device here).
"""
import json
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OC = "/root/reference/OCTCube"


def main():
    from oracle.gen_golden import install_shims
    from oracle import vit_ref as V
    install_shims()
    td = types.ModuleType("timm.data"); tu = types.ModuleType("timm.utils"); pycm = types.ModuleType("pycm")
    td.Mixup = type("Mixup", (), {})
    tu.accuracy = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not used"))
    pycm.__all__ = []
    sys.modules.update({"timm.data": td, "timm.utils": tu, "pycm": pycm})
    sys.modules["timm"].data = td; sys.modules["timm"].utils = tu
    torch.cuda.synchronize = lambda *a, **k: None
    sys.path.insert(0, OC)
    os.chdir(OC)
    import engine_finetune as ref_engine
    import models_vit_st_flash_attn as ref_st
    import util.lr_decay as lrd
    import util.misc as ref_misc

    cfg = V.ViTSTConfig(num_frames=12, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=8, embed_dim=128,
                        depth=2, num_heads=2, global_pool=True)
    model = ref_st.VisionTransformer(num_frames=12, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=8,
                                     embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                     sep_pos_embed=True, cls_embed=True, global_pool=True, drop_path_rate=0.0, dropout=0.0)
    P0 = V.init_from_shapes(V.vit_st_param_shapes(cfg), seed=23)
    model.load_state_dict(P0, strict=True)

    groups = lrd.param_groups_lrd(model, 0.05, no_weight_decay_list=model.no_weight_decay(), layer_decay=0.75)
    id2name = {id(p): n for n, p in model.named_parameters()}
    group_desc = [{"lr_scale": g["lr_scale"], "weight_decay": g["weight_decay"], "params": [id2name[id(p)] for p in g["params"]]}
                  for g in groups]
    layer_ids = {n: lrd.get_layer_id_for_vit(n, len(model.blocks) + 1) for n, _ in model.named_parameters()}
    opt = torch.optim.AdamW(groups, lr=2e-4)
    scaler = ref_misc.NativeScalerWithGradNormCount()

    g = torch.Generator().manual_seed(9)
    n_iter = 6
    xs = torch.rand(n_iter, 2, 1, 12, 64, 64, generator=g)
    ts = torch.randint(0, 8, (n_iter, 2), generator=g)
    loader = [(xs[i], ts[i]) for i in range(n_iter)]

    class Args:
        accum_iter = 2; variable_joint = False; patient_dataset_type = "volume"; task_mode = "binary_cls"; not_print_logits = True
        lr = 2e-4; min_lr = 1e-6; warmup_epochs = 1; epochs = 4; resume = ""; resume_type = ""
    crit = nn.CrossEntropyLoss()
    losses, norms, lrs = [], [], []

    def rec_crit(o, t):
        l = crit(o, t)
        losses.append(float(l))
        return l

    def rec_scaler(loss, optimizer, **kw):
        n = scaler(loss, optimizer, **kw)
        norms.append(-1.0 if n is None else float(n))
        lrs.append([gr["lr"] for gr in optimizer.param_groups])
        return n
    # isinstance(criterion, BCEWithLogitsLoss / FocalLoss2d) checks in the loop are False for a plain function
    stats = []
    for epoch in range(2):      # epoch 0 is inside the warm-up, epoch 1 on the cosine
        stats.append(ref_engine.train_one_epoch(model, rec_crit, loader, opt, torch.device("cpu"), epoch, rec_scaler, 1.0, None, None, Args))
    save = {"param_seed": 23, "cfg": json.dumps(cfg.__dict__), "data_seed": 9, "x_checksum": float(xs.double().sum()), "target": ts.numpy(),
            "losses": np.array(losses), "norms": np.array(norms), "lrs": np.array(lrs), "groups": json.dumps(group_desc),
            "layer_ids": json.dumps(layer_ids), "epoch_loss": np.array([s["loss"] for s in stats]),
            "param_checksum": np.array([float(v.double().sum()) for v in P0.values()]).sum()}
    for k, v in model.state_dict().items():
        save[f"final/{k}"] = v.numpy() if v.numel() <= 8192 else v.flatten()[::7].numpy()
    out = os.path.join(ROOT, "tests", "golden", "finetune_small.npz")
    np.savez_compressed(out, **save)
    print("fine-tune trajectory: losses", np.round(losses, 4), "norms", np.round(norms, 4), "groups", len(group_desc))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
