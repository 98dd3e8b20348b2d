"""CPU restatement of the joint 3-D + 2-D/512 pre-training iteration.  TEST INFRASTRUCTURE ONLY.
Pinned against tests/golden/joint_small.npz (oracle/gen_golden_joint.py ran the reference's own loop) by
tests/test_oracle_joint_golden.py.

Follows Pre-training/engine_pretrain.py:29-204 (two forwards, summed loss, accumulation, clip, step, per-frame loss
feedback :133-146), custom_util/misc.py add_weight_decay / NativeScalerWithGradNormCount, lr_sched.py, and
main_pretrain_oph_joint_2d512_flash_attn.py:53-67 (K / mask-ratio-2d schedules), :441-455 (AdamW betas 0.9/0.95)."""
from __future__ import annotations

import torch

from . import mae3d_ref as O
from .finetune_ref import clip_coef


def K_scheduler(epoch, K_max=0.7, K_min=0.3, all_epoch=100, warmup_epochs=10, epoch_offset=0):
    n = epoch - epoch_offset
    return K_max if n <= warmup_epochs else K_max - (n - warmup_epochs) * (K_max - K_min) / (all_epoch - warmup_epochs - epoch_offset)


def mask_ratio_2d_scheduler(epoch, mask_ratio_max=0.85, mask_ratio_min=0.75, all_epoch=100, warmup_epochs=10, epoch_offset=0):
    n = epoch - epoch_offset
    if n <= warmup_epochs:
        return mask_ratio_min
    return mask_ratio_min + (n - warmup_epochs) * (mask_ratio_max - mask_ratio_min) / (all_epoch - warmup_epochs - epoch_offset)


def record_frame_losses(frame_loss, frames, table, cube_size=3):
    """frames[nf][j] = name of frame nf of volume j; frame_loss [N, T/t_patch]."""
    for j, vol in enumerate(frame_loss.tolist()):
        names = [frames[nf][j] for nf in range(len(frames))]
        for k, v in enumerate(vol):
            for fr in range(cube_size):
                table[names[k * cube_size + fr]] = {"mse_loss": v, "hardness": v}
        table[names[-1]] = {"mse_loss": vol[-1], "hardness": vol[-1]}


def joint_epoch(P0, cfg: O.MAEConfig, vols, imgs2d, noises3d, noises2d, frames, epoch, lr, min_lr, warmup_epochs, epochs, accum_iter,
                mask_ratio, mask_ratio_2d, clip_grad, weight_decay=0.05, betas=(0.9, 0.95), eps=1e-8):
    """vols [n_iter, B, 1, T, H, W], imgs2d [n_iter, B2, 1, 3, 2H, 2W], noises per forward.  Returns dict(loss, loss_2d (lists),
    norms, params, frame_table)."""
    P = {k: v.clone() for k, v in P0.items()}
    names = list(P)
    no_decay, decay = O.weight_decay_groups([(n, tuple(P[n].shape)) for n in names], weight_decay)
    wd = {n: 0.0 for n in no_decay}
    wd.update({n: weight_decay for n in decay})
    m = {n: torch.zeros_like(P[n]) for n in names}
    v = {n: torch.zeros_like(P[n]) for n in names}
    acc = {n: torch.zeros_like(P[n]) for n in names}
    touched = set()
    n_iter = vols.shape[0]
    out = {"loss": [], "loss_2d": [], "norms": [], "frame_table": {}}
    step, cur_lr = 0, 0.0
    for it in range(n_iter):
        if it % accum_iter == 0:
            cur_lr = O.cosine_lr(it / n_iter + epoch, lr, min_lr, warmup_epochs, epochs)
        Pg = {k: t.detach().clone().requires_grad_(True) for k, t in P.items()}
        (l3, fl), _, _, _ = O.forward(Pg, vols[it], cfg, mask_ratio, noises3d[it], frame_loss=True)
        l2, _, _, _ = O.forward(Pg, imgs2d[it], cfg, mask_ratio_2d, noises2d[it])
        out["loss"].append(float(l3)); out["loss_2d"].append(float(l2))
        record_frame_losses(fl.detach(), frames[it], out["frame_table"])
        grads = torch.autograd.grad((l3 + l2) / accum_iter, [Pg[n] for n in names], allow_unused=True)
        for n, g in zip(names, grads):
            if g is not None:
                acc[n] += g
                touched.add(n)
        if (it + 1) % accum_iter == 0:
            total = float(O.grad_norm([acc[n] for n in names if n in touched]))
            c = clip_coef(total, clip_grad) if clip_grad is not None else 1.0
            step += 1
            for n in names:
                if n in touched:
                    P[n], m[n], v[n] = O.adamw_step(P[n], acc[n] * c, m[n], v[n], step, cur_lr, betas[0], betas[1], eps, wd[n])
                acc[n].zero_()
            out["norms"].append(total)
        else:
            out["norms"].append(-1.0)
    out["params"] = P
    return out
