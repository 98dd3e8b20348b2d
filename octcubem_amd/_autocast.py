"""Autocast invariance of the drop-in modules.

The reference's engines call their models under ``torch.cuda.amp.autocast()`` (Pre-training/engine_pretrain.py:110,255,
OCTCube/engine_pretrain.py:57, OCTCube/engine_finetune.py:432,576, retinal-COEM/src/training/train_retclip.py:125).  The modules here
choose their own precisions (HIP kernels behind the C ABI: 16-bit MFMA operands, fp32 everything else -- DESIGN.md section 3), so an
enclosing autocast context must change NOTHING: the few ATen islands on the path (the bicubic resampling matrix of the positional
table, a class head that is not a multiple of 8 wide, the contrastive logits, softmax / cross-entropy of the losses) would otherwise be
down-cast by it, and a half-precision table would then reach a kernel that checks for fp32.  ``autocast_invariant`` wraps every public
forward-like method of a class (``forward``, ``forward_*``, ``encode_*``) in ``torch.autocast(device, enabled=False)`` when, and only
when, autocast is on: outputs and gradients are bit-identical with and without the context (tests/test_gpu_autocast.py).
"""
from __future__ import annotations

import functools

import torch


def _autocast_on() -> bool:
    try:
        return torch.is_autocast_enabled("cuda") or torch.is_autocast_enabled("cpu")
    except TypeError:      # a PyTorch whose is_autocast_enabled takes no device argument
        return torch.is_autocast_enabled() or torch.is_autocast_cpu_enabled()


def no_autocast(fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if not _autocast_on():
            return fn(*args, **kwargs)
        with torch.autocast("cuda", enabled=False), torch.autocast("cpu", enabled=False):
            return fn(*args, **kwargs)
    wrapper._octmae_no_autocast = True
    return wrapper


def autocast_invariant(cls):
    """Class decorator: ``forward``, ``forward_*`` and ``encode_*`` defined BY this class run with autocast switched off."""
    for name, fn in list(vars(cls).items()):
        if (name == "forward" or name.startswith(("forward_", "encode_"))) and callable(fn) and not getattr(fn, "_octmae_no_autocast", False):
            setattr(cls, name, no_autocast(fn))
    return cls
