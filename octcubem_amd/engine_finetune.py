"""Fine-tuning loop and evaluation with the reference's signatures (OCTCube/engine_finetune.py:386-482 ``train_one_epoch``,
:498 ``evaluate``): per-iteration LR schedule (every ``accum_iter``), H2D, float targets for BCE-type criteria, optional
mixup callable, forward, non-finite guard, ``loss_scaler(loss / accum_iter, clip_grad=max_norm, update_grad=...)``,
zero_grad on step boundaries, min/max group LR logging, scalar loss all-reduce.

Not reproduced: the per-iteration ``torch.cuda.synchronize()`` and logits printing (pipeline stalls with no numerical effect),
the 2-D/3-D ``variable_joint`` and SLIViT reshapes (models outside SURVEY §8), and the CSV / confusion-matrix reporting of
the reference's ``evaluate`` -- this one returns loss, top-1 accuracy and the gathered logits / targets for the caller's
metric code.

Provenance, stated once: this file is a RESTATEMENT of the reference's host loop, written to be call-compatible with it -- same
function signatures, same order of operations per iteration, same ``MetricLogger`` keys -- because it is the caller SURVEY section 8
(R14 / N1, N3) requires and the reference's drivers import it by name.  It holds no kernel logic; everything it calls (models, scaler,
optimizer, schedules) is this package's own."""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch

from . import lr_sched, misc


def train_one_epoch(model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer,
                    device: torch.device, epoch: int, loss_scaler, max_norm: float = 0, mixup_fn=None, log_writer=None, args=None):
    model.train(True)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch)
    print_freq = 20
    accum_iter = args.accum_iter
    optimizer.zero_grad()
    n_iter = len(data_loader)
    float_targets = isinstance(criterion, torch.nn.BCEWithLogitsLoss) or getattr(args, "task_mode", "") == "regression"
    for data_iter_step, (samples, targets) in enumerate(metric_logger.log_every(misc.prefetched(data_loader, device, args, only=(0, 1)), print_freq, header)):
        if data_iter_step % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer, data_iter_step / n_iter + epoch, args)
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        if float_targets:
            targets = targets.float()
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)
        outputs = model(samples)
        loss = criterion(outputs, targets)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            return None
        loss = loss / accum_iter
        loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=False,
                    update_grad=(data_iter_step + 1) % accum_iter == 0)
        if (data_iter_step + 1) % accum_iter == 0:
            optimizer.zero_grad()
        metric_logger.update(loss=loss_value)
        min_lr, max_lr = 10.0, 0.0
        for group in optimizer.param_groups:
            min_lr = min(min_lr, group["lr"])
            max_lr = max(max_lr, group["lr"])
        metric_logger.update(lr=max_lr)
        loss_value_reduce = misc.all_reduce_mean(loss_value)
        if log_writer is not None and (data_iter_step + 1) % accum_iter == 0:
            epoch_1000x = int((data_iter_step / n_iter + epoch) * 1000)
            log_writer.add_scalar("loss", loss_value_reduce, epoch_1000x)
            log_writer.add_scalar("lr", max_lr, epoch_1000x)
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def evaluate(data_loader: Iterable, model: torch.nn.Module, device: torch.device, criterion: Optional[torch.nn.Module] = None):
    """Eval-mode pass: ``{"loss", "acc1", "logits" [n, C], "targets" [n]}`` (logits / targets on the host, fp32)."""
    criterion = criterion or torch.nn.CrossEntropyLoss()
    model.eval()
    logits_all, targets_all, loss_sum, n = [], [], 0.0, 0
    for samples, targets in misc.prefetched(data_loader, device, None, only=(0, 1)):
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        out = model(samples).float()
        t = targets.float() if isinstance(criterion, torch.nn.BCEWithLogitsLoss) else targets
        loss_sum += float(criterion(out, t)) * samples.shape[0]
        n += samples.shape[0]
        logits_all.append(out.cpu())
        targets_all.append(targets.cpu())
    logits = torch.cat(logits_all)
    tg = torch.cat(targets_all)
    acc1 = float((logits.argmax(-1) == (tg if tg.dim() == 1 else tg.argmax(-1))).float().mean()) if n else float("nan")
    return {"loss": loss_sum / max(n, 1), "acc1": acc1, "logits": logits, "targets": tg}
