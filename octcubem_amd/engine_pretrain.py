"""Pre-training inner loops with the reference's signatures: ``train_one_epoch`` (3-D volumes only) and
``train_one_epoch_joint`` (3-D volumes + 2-D/512 B-scans, per-frame loss feedback for self-paced sampling) plus the two
epoch schedules of the joint recipe.

``train_one_epoch``: pre-training inner loop with the reference's signature and per-iteration order of operations
(OCTCube/engine_pretrain.py:31-91; Pre-training/engine_pretrain.py:29-204 is the same loop plus the 2-D branch):
per-iteration LR schedule -> H2D -> forward -> non-finite guard -> loss_scaler(backward / [all-reduce] / clip / step)
-> zero_grad -> logging.  The dead ``get_mask`` work of the 3-D engine (its result is dropped by forward) and the
per-iteration full-device synchronise are not reproduced; a single ``loss.item()`` per iteration remains (it is the
reference's non-finite guard).

Provenance, stated once: this file is a RESTATEMENT of the reference's host loop, written to be call-compatible with it -- same
function signatures, same order of operations per iteration, same ``MetricLogger`` keys -- because it is the caller SURVEY section 8
(R14 / N1, N3) requires and the reference's drivers import it by name.  It holds no kernel logic; everything it calls (models, scaler,
optimizer, schedules) is this package's own."""
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
    for data_iter_step, batch in enumerate(metric_logger.log_every(misc.prefetched(data_loader, device, args), print_freq, header)):
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


def K_scheduler(epoch, K_max=0.7, K_min=0.3, all_epoch=100, warmup_epochs=10, epoch_offset=0):
    """Self-paced-learning keep fraction K (main_pretrain_oph_joint_2d512_flash_attn.py:53-59): K_max through the warm-up, then
    linearly down to K_min at ``all_epoch``."""
    num_epochs = epoch - epoch_offset
    if num_epochs <= warmup_epochs:
        return K_max
    return K_max - (num_epochs - warmup_epochs) * (K_max - K_min) / (all_epoch - warmup_epochs - epoch_offset)


def mask_ratio_2d_scheduler(epoch, mask_ratio_max=0.85, mask_ratio_min=0.75, all_epoch=100, warmup_epochs=10, epoch_offset=0):
    """Mask ratio of the 2-D branch (:61-67): mask_ratio_min through the warm-up, then linearly up to mask_ratio_max."""
    num_epochs = epoch - epoch_offset
    if num_epochs <= warmup_epochs:
        return mask_ratio_min
    return mask_ratio_min + (num_epochs - warmup_epochs) * (mask_ratio_max - mask_ratio_min) / (all_epoch - warmup_epochs - epoch_offset)


def record_frame_losses(frame_loss, data_dict, dataset_2d_all_image_dict, cube_size=3):
    """Per-frame loss feedback of the joint loop (Pre-training/engine_pretrain.py:133-146): every temporal patch group's masked
    MSE is written to the ``mse_loss`` / ``hardness`` fields of the ``cube_size`` frames it covers (the 2-D dataset samples
    by hardness).  ``frame_loss``: [N, T / t_patch]; ``data_dict['frames'][nf][j]`` = name of frame nf of volume j.
    One device->host copy instead of one ``.item()`` per frame."""
    fl = frame_loss.detach().float().cpu().tolist()
    n_frames = len(data_dict["frames"])
    for j, vol in enumerate(fl):
        names = [data_dict["frames"][nf][j] for nf in range(n_frames)]
        for k, v in enumerate(vol):
            for fr in range(cube_size):
                e = dataset_2d_all_image_dict[names[k * cube_size + fr]]
                e["mse_loss"] = v
                e["hardness"] = v
        e = dataset_2d_all_image_dict[names[-1]]
        e["mse_loss"] = vol[-1]
        e["hardness"] = vol[-1]


def train_one_epoch_joint(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer, device: torch.device,
                          epoch: int, loss_scaler, data_loader_2d, dataset_2d_all_image_dict, mask_ratio_2d, log_writer=None,
                          args=None, fp32=False, fp16=False, noise_fn=None):
    """Joint 3-D + 2-D/512 pre-training epoch (Pre-training/engine_pretrain.py:29-204): per iteration one volume batch
    (``frame_loss=True``) and one batch of ``(B, C, 3, 512, 512)`` B-scan triplets through ``high_res_patch_embed``, the two
    losses summed before a single backward; the 2-D loader restarts when exhausted.  ``data_loader`` yields
    ``(samples, (img_names, data_dict))``.  Same omissions as ``train_one_epoch`` (the dropped ``get_mask`` result, the
    per-iteration device synchronise); ``fp32`` / ``fp16`` are accepted and ignored (bf16 operands, fp32 everything else).
    Both forwards feed the same parameters, so a parameter reports its gradient once per use: the data-parallel reducer learns
    the number of reports per parameter in the first exchanged backward (run without overlap) and overlaps from then on."""
    model.train(True)
    net = getattr(model, "module", model)
    reducer = getattr(loss_scaler, "reducer", None)
    if reducer is not None:
        reducer.multi_use = True
    metric_logger = misc.MetricLogger(delimiter="  ")
    for name in ("lr", "mask_ratio", "mask_ratio_2d"):
        metric_logger.add_meter(name, misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    metric_logger.add_meter("loss_2d", misc.SmoothedValue(window_size=1))
    metric_logger.add_meter("loss_all", misc.SmoothedValue(window_size=1))
    header = "Epoch: [{}]".format(epoch)
    print_freq = 20
    accum_iter = args.accum_iter
    optimizer.zero_grad()
    n_iter = len(data_loader)
    secondary_iter = iter(data_loader_2d)
    for data_iter_step, (samples, data_info) in enumerate(metric_logger.log_every(misc.prefetched(data_loader, device, args), print_freq, header)):
        if data_iter_step % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer, data_iter_step / n_iter + epoch, args)
        data_dict = data_info[1]
        try:
            secondary_data = next(secondary_iter)
        except StopIteration:
            secondary_iter = iter(data_loader_2d)
            secondary_data = next(secondary_iter)
        sample_2d = secondary_data[0].to(device, non_blocking=True)
        samples = samples.to(device, non_blocking=True)
        if samples.dim() == 6:
            b, r, c, t, h, w = samples.shape
            samples = samples.reshape(b * r, c, t, h, w)
        kw3 = {"noise": noise_fn(samples)} if noise_fn is not None else {}
        (loss, frame_loss), _, _ = net(samples, mask_ratio=args.mask_ratio, frame_loss=True, **kw3)
        kw2 = {"noise": noise_fn(sample_2d)} if noise_fn is not None else {}
        loss_2d, _, _ = net(sample_2d, mask_ratio=mask_ratio_2d, **kw2)
        loss_value, loss_2d_value = loss.item(), loss_2d.item()
        record_frame_losses(frame_loss, data_dict, dataset_2d_all_image_dict)
        loss = loss + loss_2d
        loss_all_value = loss_value + loss_2d_value
        if not math.isfinite(loss_value):
            raise Exception("Loss is {}, stopping training".format(loss_value))
        loss = loss / accum_iter
        loss_scaler(loss, optimizer, parameters=net.parameters(), update_grad=(data_iter_step + 1) % accum_iter == 0,
                    clip_grad=getattr(args, "clip_grad", None))
        if (data_iter_step + 1) % accum_iter == 0:
            optimizer.zero_grad()
        metric_logger.update(loss=loss_value, mask_ratio=args.mask_ratio, loss_2d=loss_2d_value, loss_all=loss_all_value,
                             mask_ratio_2d=mask_ratio_2d)
        lr = optimizer.param_groups[0]["lr"]
        metric_logger.update(lr=lr)
        loss_value_reduce = misc.all_reduce_mean(loss_value)
        if log_writer is not None and (data_iter_step + 1) % accum_iter == 0:
            epoch_1000x = int((data_iter_step / n_iter + epoch) * 1000 * getattr(args, "repeat_aug", 1))
            log_writer.add_scalar("train_loss", loss_value_reduce, epoch_1000x)
            log_writer.add_scalar("lr", lr, epoch_1000x)
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
