"""Pre-training inner loop with the reference's signature and per-iteration order of operations
(OCTCube/engine_pretrain.py:31-91; Pre-training/engine_pretrain.py:29-204 is the same loop plus the 2-D branch):
per-iteration LR schedule -> H2D -> forward -> non-finite guard -> loss_scaler(backward / [all-reduce] / clip / step)
-> zero_grad -> logging.  The dead ``get_mask`` work of the 3-D engine (its result is dropped by forward) and the
per-iteration full-device synchronise are not reproduced; a single ``loss.item()`` per iteration remains (it is the
reference's non-finite guard)."""
from __future__ import annotations

import math
import sys
from typing import Iterable

import torch

from . import lr_sched, misc


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer, device: torch.device,
                    epoch: int, loss_scaler, log_writer=None, args=None, noise_fn=None):
    model.train(True)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch)
    print_freq = 20
    accum_iter = args.accum_iter
    optimizer.zero_grad()
    n_iter = len(data_loader)
    for data_iter_step, batch in enumerate(metric_logger.log_every(data_loader, print_freq, header)):
        samples = batch[0] if isinstance(batch, (tuple, list)) else batch
        if data_iter_step % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer, data_iter_step / n_iter + epoch, args)
        samples = samples.to(device, non_blocking=True)
        if samples.dim() == 6:
            b, r, c, t, h, w = samples.shape
            samples = samples.reshape(b * r, c, t, h, w)
        noise = noise_fn(samples) if noise_fn is not None else None
        loss, _, _ = model(samples, mask_ratio=args.mask_ratio, noise=noise) if noise is not None else \
            model(samples, mask_ratio=args.mask_ratio)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        loss = loss / accum_iter
        loss_scaler(loss, optimizer, parameters=model.parameters(), update_grad=(data_iter_step + 1) % accum_iter == 0,
                    clip_grad=getattr(args, "clip_grad", None))
        if (data_iter_step + 1) % accum_iter == 0:
            optimizer.zero_grad()
        metric_logger.update(loss=loss_value)
        lr = optimizer.param_groups[0]["lr"]
        metric_logger.update(lr=lr)
        loss_value_reduce = misc.all_reduce_mean(loss_value)
        if log_writer is not None and (data_iter_step + 1) % accum_iter == 0:
            epoch_1000x = int((data_iter_step / n_iter + epoch) * 1000)
            log_writer.add_scalar("train_loss", loss_value_reduce, epoch_1000x)
            log_writer.add_scalar("lr", lr, epoch_1000x)
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
