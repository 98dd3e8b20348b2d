"""Per-iteration learning-rate schedule: linear warm-up then half-cycle cosine
(same contract as the reference's custom_util/lr_sched.py:10-28: reads args.lr / min_lr / warmup_epochs / epochs,
honours an optional per-group ``lr_scale``, returns the un-scaled lr)."""
import math


def adjust_learning_rate(optimizer, epoch, args):
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        progress = (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * (1.0 + math.cos(math.pi * progress))
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr
