"""Fused multi-tensor AdamW + gradient-norm kernels behind the torch.optim.Optimizer interface.

Replaces ``torch.optim._multi_tensor.AdamW(param_groups, lr, betas=(0.9, 0.95))`` as built by the reference
driver (main_pretrain_oph_joint_2d512_flash_attn.py:441-455): same param_groups / lr / lr_scale contract, so
``lr_sched.adjust_learning_rate`` and ``misc.add_weight_decay`` work unchanged.  One kernel launch per group
(16 B read + 12 B written per parameter), gradient clip / unscale folded in as a device-side scale.
"""
from __future__ import annotations

import os
import struct
from typing import List, Optional

import torch

from ._lib import call
from .ops import _stream


class _MultiTensorTable:
    """Device-side tables for a list of (p, g, m, v) tensors: octmae_mt_* calling convention.  ``lps``: per tensor the device
    address of the 16-bit operand copy of p (0: none) -- the optional table octmae_mt_adamw_fused writes through."""

    def __init__(self, ps: List[torch.Tensor], gs, ms, vs, lps=None):
        from ._lib import load
        chunk = load().octmae_mt_chunk_elems()
        dev = ps[0].device
        raw = bytearray()
        ct, co = [], []
        for i, (p, g, m, v) in enumerate(zip(ps, gs, ms, vs)):
            n = p.numel()
            raw += struct.pack("<QQQQq", p.data_ptr(), g.data_ptr() if g is not None else 0,
                               m.data_ptr() if m is not None else 0, v.data_ptr() if v is not None else 0, n)
            for off in range(0, n, chunk):
                ct.append(i); co.append(off)
        self.n_tensors = len(ps)
        self.n_chunks = len(ct)
        self.table = torch.frombuffer(raw, dtype=torch.uint8).clone().to(dev)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=dev)
        self.chunk_off = torch.tensor(co, dtype=torch.int64, device=dev)
        self.lp_table = None
        if lps is not None and any(lps):
            self.lp_table = torch.tensor([int(a) for a in lps], dtype=torch.int64, device=dev)
        # every pointer the table holds is part of its identity: a loaded optimizer state replaces exp_avg / exp_avg_sq
        self.key = tuple((p.data_ptr(), g.data_ptr() if g is not None else 0, m.data_ptr() if m is not None else 0,
                          v.data_ptr() if v is not None else 0) for p, g, m, v in zip(ps, gs, ms, vs))


def grad_norm_and_coef(params, max_norm: Optional[float], cache: dict):
    """get_grad_norm_ (misc.py:356-373) / clip_grad_norm_ statistics in two launches.
    Returns (total_norm, clip_coef) as 0-dim device tensors; clip_coef == 1 when max_norm is None."""
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return torch.tensor(0.0), None
    key = tuple((p.data_ptr(), p.grad.data_ptr(), 0, 0) for p in ps)
    tab = cache.get("norm")
    if tab is None or tab.key != key:
        tab = _MultiTensorTable([p.data for p in ps], [p.grad for p in ps], [None] * len(ps), [None] * len(ps))
        cache["norm"] = tab
    dev = ps[0].device
    sumsq = torch.zeros(tab.n_tensors, dtype=torch.float32, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    call("octmae_mt_sumsq", tab.table.data_ptr(), tab.chunk_tensor.data_ptr(), tab.chunk_off.data_ptr(), tab.n_chunks,
         sumsq.data_ptr(), _stream())
    call("octmae_mt_finish_norm", sumsq.data_ptr(), tab.n_tensors, float(max_norm) if max_norm is not None else 0.0,
         out.data_ptr(), out.data_ptr() + 4, _stream())
    return out[0], out[1]


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._tables = {}
        self._grad_scale: Optional[torch.Tensor] = None   # device scalar multiplied into every gradient
        # write_mirror: the AdamW kernel also writes the 16-bit operand copy (arena.ParamArena.lp) of every parameter it updates, so
        # the next forward needs no cast pass over the arena (OCTMAE_ADAMW_MIRROR=0: same-box A/B; the arena then re-casts)
        self.write_mirror = os.environ.get("OCTMAE_ADAMW_MIRROR", "1") != "0"

    def load_state_dict(self, state_dict):
        """The moment buffers are replaced: drop the device pointer tables (they are rebuilt on the next step)."""
        super().load_state_dict(state_dict)
        self._tables = {}

    def set_grad_scale(self, scale: Optional[torch.Tensor]):
        self._grad_scale = scale

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are views of one arena: they are zeroed in place, never detached -- and when the gradients of this
        optimizer are views of ONE flat buffer (the model's gradient arena), with one fill per contiguous RUN of them instead
        of one launch per parameter (~500 per step for ViT-L; a single run when this optimizer holds every trainable
        parameter).  Only bytes this optimizer owns are touched: two gradients join a run only when the second starts where
        the first ends or at the next arena alignment boundary, so a slice that belongs to a parameter of another optimizer
        (or to one that is accumulating for a later step) is never inside a fill."""
        grads = [p.grad for group in self.param_groups for p in group["params"] if p.grad is not None]
        if not grads:
            return
        base = grads[0]._base
        if base is None or base.dim() != 1 or not all(g._base is base and g.is_contiguous() for g in grads):
            for g in grads:
                g.zero_()
            return
        from .arena import ALIGN
        spans = sorted((g.storage_offset() - base.storage_offset(), g.numel()) for g in grads)
        start, end = spans[0][0], spans[0][0] + spans[0][1]
        for o, n in spans[1:]:
            if o == end or o == (end + ALIGN - 1) // ALIGN * ALIGN:
                end = o + n
            else:
                base[start:end].zero_()
                start, end = o, o + n
        base[start:end].zero_()

    def grads_key(self):
        """Identity of the set of gradients a step() will consume (misc.NativeScalerWithGradNormCount compares it with the set it is
        asked to take the norm of before it lets the optimizer produce that norm itself)."""
        return frozenset(id(p) for g in self.param_groups for p in g["params"] if p.grad is not None)

    @torch.no_grad()
    def step(self, closure=None, want_norm: bool = False):
        """want_norm: also return the 2-norm of the (raw, un-scaled) gradients this step consumed, accumulated by the AdamW kernels
        themselves (one read of the gradients for norm and update) -> (loss, norm) instead of loss."""
        from . import arena as _arena
        loss = closure() if closure is not None else None
        sumsq = None
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            if "_step" not in group:
                # a state loaded from torch.optim.AdamW carries the count per parameter ("step", int or tensor), not per
                # group: continue from it so that the bias correction does not restart
                steps = [self.state[p].get("step", 0) for p in ps]
                group["_step"] = int(max(float(s_) for s_ in steps)) if steps else 0
            group["_step"] += 1
            for p in ps:
                self.state[p]["step"] = group["_step"]
            key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                         self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
            tab = self._tables.get(gi)
            if tab is None or tab.key != key or tab.mirror != self.write_mirror:
                owners = [_arena.arena_of(p) for p in ps]
                lps = [(a.lp_ptr(p) if (a is not None and self.write_mirror) else 0) for a, p in zip(owners, ps)]
                tab = _MultiTensorTable([p.data for p in ps], [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps],
                                        [self.state[p]["exp_avg_sq"] for p in ps], lps)
                tab.mirror = self.write_mirror
                # arenas whose parameters this group updates WITHOUT writing their operand copy: they must re-cast
                tab.stale_arenas = [a for a in {id(a): a for a, lp in zip(owners, lps) if a is not None and lp == 0}.values()]
                self._tables[gi] = tab
            for a in tab.stale_arenas:
                a.invalidate_lp()
            if want_norm and sumsq is None:
                n_max = max(len([p for p in g["params"] if p.grad is not None]) for g in self.param_groups)
                sumsq = torch.zeros((len(self.param_groups), max(n_max, 1)), dtype=torch.float32, device=ps[0].device)
            b1, b2 = group["betas"]
            gs = self._grad_scale
            call("octmae_mt_adamw_fused", tab.table.data_ptr(), tab.chunk_tensor.data_ptr(), tab.chunk_off.data_ptr(), tab.n_chunks,
                 gs.data_ptr() if gs is not None else None, tab.lp_table.data_ptr() if tab.lp_table is not None else None,
                 sumsq[gi].data_ptr() if sumsq is not None else None, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                 float(group["weight_decay"]), int(group["_step"]), _stream())
        if not want_norm:
            return loss
        if sumsq is None:
            return loss, torch.tensor(0.0)
        out = torch.empty(2, dtype=torch.float32, device=sumsq.device)
        call("octmae_mt_finish_norm", sumsq.data_ptr(), sumsq.numel(), 0.0, out.data_ptr(), out.data_ptr() + 4, _stream())
        return loss, out[0]
