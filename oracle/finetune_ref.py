"""CPU restatement of the ST fine-tune step.  TEST INFRASTRUCTURE ONLY: imported by tests/, never by the
product package.  Pinned against tests/golden/finetune_small.npz, which oracle/gen_golden_finetune.py produced by running the
reference's own loop (tests/test_oracle_finetune_golden.py).

Follows, function by function:
  get_layer_id_for_vit / param_groups_lrd     OCTCube/util/lr_decay.py:10-76
  adjust_learning_rate                         OCTCube/util/lr_sched.py:9-26 (fresh-start branch)
  train_one_epoch                              OCTCube/engine_finetune.py:386-482 (accumulation, clip, step order)
  NativeScalerWithGradNormCount.__call__       OCTCube/util/misc.py:260-282 (clip_grad_norm_ when clip_grad is given)
  torch.optim.AdamW(param_groups, lr)          OCTCube/main_finetune.py:303 (betas 0.9/0.999, eps 1e-8)
  label smoothing / soft-target CE             timm.loss (un-vendored; timm 0.3.2): mean_i[(1-s)*nll_i + s*mean_c(-logp_ic)],
                                               mean_i[sum_c -t_ic logp_ic]
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

from . import mae3d_ref as O
from . import vit_ref as V


def get_layer_id_for_vit(name: str, num_layers: int) -> int:
    """lr_decay.py:62-76."""
    if name in ("cls_token", "pos_embed"):
        return 0
    if name.startswith("patch_embed"):
        return 0
    if name.startswith("blocks"):
        return int(name.split(".")[1]) + 1
    return num_layers


def param_groups_lrd(names_ndim: Sequence, depth: int, weight_decay=0.05, no_weight_decay_list=(), layer_decay=0.75) -> List[dict]:
    """lr_decay.py:10-59 on (name, ndim) pairs in named_parameters() order; groups in first-seen order."""
    num_layers = depth + 1
    scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    groups: Dict[str, dict] = {}
    for n, ndim in names_ndim:
        if ndim == 1 or n in no_weight_decay_list:
            tag, wd = "no_decay", 0.0
        else:
            tag, wd = "decay", weight_decay
        lid = get_layer_id_for_vit(n, num_layers)
        g = groups.setdefault("layer_%d_%s" % (lid, tag), {"lr_scale": scales[lid], "weight_decay": wd, "params": []})
        g["params"].append(n)
    return list(groups.values())


NO_WEIGHT_DECAY = ("cls_token", "pos_embed", "pos_embed_spatial", "pos_embed_temporal", "pos_embed_class")   # models_vit_st…:177-178


def label_smoothing_ce(logits, target, smoothing=0.1):
    logp = F.log_softmax(logits, dim=-1)
    nll = -logp.gather(-1, target.unsqueeze(1)).squeeze(1)
    return ((1.0 - smoothing) * nll + smoothing * (-logp.mean(dim=-1))).mean()


def soft_target_ce(logits, target):
    return torch.sum(-target * F.log_softmax(logits, dim=-1), dim=-1).mean()


def clip_coef(total_norm: float, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1."""
    return min(1.0, max_norm / (total_norm + 1e-6))


def finetune_trajectory(P0, cfg: V.ViTSTConfig, xs, ts, lr, min_lr, warmup_epochs, epochs, n_epochs, accum_iter, max_norm,
                        weight_decay=0.05, layer_decay=0.75, betas=(0.9, 0.999), eps=1e-8, criterion=None):
    """xs [n_iter, B, ...], ts [n_iter, B].  Returns dict(losses, norms (-1 where no step), lrs [iter, group], params)."""
    criterion = criterion or F.cross_entropy
    P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    names = list(P.keys())
    groups = param_groups_lrd([(n, P[n].dim()) for n in names], cfg.depth, weight_decay, NO_WEIGHT_DECAY, layer_decay)
    m = {n: torch.zeros_like(P[n]) for n in names}
    v = {n: torch.zeros_like(P[n]) for n in names}
    acc = {n: torch.zeros_like(P[n]) for n in names}
    step = 0
    used = set()
    n_iter = xs.shape[0]
    losses, norms, lrs = [], [], []
    cur_lr = [0.0] * len(groups)
    for epoch in range(n_epochs):
        for it in range(n_iter):
            if it % accum_iter == 0:
                base = O.cosine_lr(it / n_iter + epoch, lr, min_lr, warmup_epochs, epochs)
                cur_lr = [base * g["lr_scale"] for g in groups]
            logits, _ = V.vit_st_forward(P, xs[it], cfg)
            loss = criterion(logits, ts[it])
            losses.append(float(loss))
            gr = torch.autograd.grad(loss / accum_iter, [P[n] for n in names], allow_unused=True)
            for n, g_ in zip(names, gr):
                if g_ is not None:          # `norm.*` feeds the computed-but-unused outcome (models_vit_st…:247-249): grad None
                    acc[n] += g_
                    used.add(n)
            if (it + 1) % accum_iter == 0:
                total = float(O.grad_norm([acc[n] for n in names if n in used]))
                c = clip_coef(total, max_norm) if max_norm is not None else 1.0
                step += 1
                with torch.no_grad():
                    for gi, g in enumerate(groups):
                        for n in g["params"]:
                            if n not in used:   # torch.optim skips parameters whose .grad is None (no decay either)
                                continue
                            p_new, m[n], v[n] = O.adamw_step(P[n].detach(), acc[n] * c, m[n], v[n], step, cur_lr[gi], betas[0], betas[1],
                                                             eps, g["weight_decay"])
                            P[n] = p_new.requires_grad_(True)
                            acc[n].zero_()
                norms.append(total)
            else:
                norms.append(-1.0)
            lrs.append(list(cur_lr))
    return {"losses": losses, "norms": norms, "lrs": lrs, "params": {n: P[n].detach() for n in names}, "groups": groups}
