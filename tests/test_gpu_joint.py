"""GPU: engine_pretrain.train_one_epoch_joint (3-D volumes + 2-D/512 B-scan triplets, summed loss, per-frame loss feedback,
accumulation, clip, fused AdamW) against the epoch the REAL reference loop produced on the same seeded data and masking noise
(tests/golden/joint_small.npz).  Tolerances: epoch-mean losses 2e-3 relative, gradient norms 3e-2, per-frame losses 1e-2,
final parameters as in test_gpu_finetune.py."""
import json
import os
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from octcubem_amd import engine_pretrain, misc, models_mae
    from octcubem_amd import optim as foptim
from oracle import mae3d_ref as O
from tests.test_oracle_joint_golden import load_joint

DEV = "cuda"


def test_joint_epoch_matches_reference_trajectory(golden_dir):
    z, cfg, P0, vols, imgs2d, n3, n2, frames = load_joint(golden_dir)
    m = models_mae.MaskedAutoencoderViT(
        input_size=cfg.input_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depth=cfg.depth,
        num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
        decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio, norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps),
        num_frames=cfg.num_frames, t_patch_size=cfg.t_patch_size, sep_pos_embed=True, cls_embed=True, pred_t_dim=cfg.pred_t_dim,
        high_res_input_size=cfg.high_res_input_size)
    m.load_state_dict(P0, strict=True)
    m = m.to(DEV)
    opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
    scaler = misc.NativeScalerWithGradNormCount(fp32=True)
    norms = []

    def rec_scaler(loss, optimizer, **kw):
        n = scaler(loss, optimizer, **kw)
        norms.append(-1.0 if n is None else float(n))
        return n
    queue = [t for pair in zip(n3, n2) for t in pair]               # the order the reference consumed the RNG in
    noise_fn = lambda s: queue.pop(0).to(DEV)
    loader3d = [(vols[it], ([f"vol{it}_{j}" for j in range(2)], {"frames": frames[it]})) for it in range(4)]
    loader2d = [(imgs2d[it], None) for it in range(4)]
    table = {f: {} for it in range(4) for nf in range(6) for f in frames[it][nf]}

    class Args:
        accum_iter = 2; lr = 1e-3; min_lr = 1e-6; warmup_epochs = 1; epochs = 4; mask_ratio = 0.75; clip_grad = 1.0; repeat_aug = 1
    stats = engine_pretrain.train_one_epoch_joint(m, loader3d, opt, torch.device(DEV), 1, rec_scaler, loader2d, table, 0.8, args=Args,
                                                  noise_fn=noise_fn)
    ref = json.loads(str(z["stats"]))
    for k in ("loss", "loss_2d", "loss_all"):
        assert abs(stats[k] - ref[k]) <= 2e-3 * ref[k], (k, stats[k], ref[k])
    assert abs(stats["lr"] - ref["lr"]) < 1e-12 and stats["mask_ratio_2d"] == 0.8
    np.testing.assert_allclose(norms, z["norms"], rtol=3e-2)
    ref_tab = json.loads(str(z["frame_dict"]))
    assert set(table) == set(ref_tab)
    for k, e in ref_tab.items():
        assert abs(table[k]["mse_loss"] - e["mse_loss"]) <= 1e-2 * abs(e["mse_loss"]) and table[k]["hardness"] == table[k]["mse_loss"], k
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    max_step = 1e-3 * 2
    for k in z.files:
        if not k.startswith("final/"):
            continue
        n = k[len("final/"):]
        sub = (lambda t: t if t.numel() <= 8192 else t.flatten()[::7])
        refp, mine, init = torch.from_numpy(z[k]).flatten(), sub(sd[n]).flatten(), sub(P0[n]).flatten()
        assert float((mine - refp).abs().max()) <= 2.5 * max_step, n
        du_ref, du = (refp - init).double(), (mine - init).double()
        if n.endswith("attn.k.bias") or float(du_ref.norm()) < 1e-9:
            continue
        cos = float((du * du_ref).sum() / (du.norm() * du_ref.norm() + 1e-30))
        assert cos >= 0.97, (n, cos)
