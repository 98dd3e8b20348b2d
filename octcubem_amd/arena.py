"""Flat parameter / gradient arenas.

All parameters of a model live in ONE fp32 buffer (master weights), mirrored by one bf16 buffer (the MFMA
operands, refreshed by a single cast kernel per forward) and one fp32 gradient buffer that every
``param.grad`` is a view of.  Consequences on MI355X:
  * q|k|v weights (and biases) of an Attention are adjacent, so the fused [3D, D] Wqkv GEMM operand and its
    gradient exist without any copy while the state_dict keeps the reference's separate ``attn.q/k/v`` keys
    (Pre-training/custom_util/video_vit.py:103-105);
  * the data-parallel reducer all-reduces contiguous slices of the gradient arena over RCCL (no bucket copies);
  * optimizer and grad-norm kernels see few, large, 16-byte-aligned tensors.
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import ops

ALIGN = 64  # elements; keeps every tensor 256-byte (fp32) / 128-byte (bf16) aligned


def _ordered(named: List[Tuple[str, nn.Parameter]]) -> List[Tuple[str, nn.Parameter]]:
    """Registration order, except that q/k/v weights (then biases) of an attention become adjacent."""
    by_name = dict(named)
    out, seen = [], set()
    for name, p in named:
        if name in seen:
            continue
        m = re.match(r"(.*\.)q\.(weight|bias)$", name)
        if m and all(f"{m.group(1)}{x}.weight" in by_name for x in "qkv"):
            pre = m.group(1)
            for kind in ("weight", "bias"):
                for x in "qkv":
                    k = f"{pre}{x}.{kind}"
                    if k in by_name and k not in seen:
                        out.append((k, by_name[k])); seen.add(k)
            continue
        out.append((name, p)); seen.add(name)
    return out


class ParamArena:
    def __init__(self, root: nn.Module):
        named = [(n, p) for n, p in root.named_parameters()]
        if not named:
            raise RuntimeError("module has no parameters")
        dev = named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("octcubem_amd modules run on the GPU only: move the model to a 'cuda' device first "
                               "(there is no CPU fallback)")
        self.entries: List[Tuple[str, nn.Parameter, int, int]] = []
        off = 0
        prev_fused = False
        for name, p in _ordered(named):
            fused_member = bool(re.search(r"\.(k|v)\.(weight|bias)$", name)) and prev_fused
            if not fused_member:
                off = (off + ALIGN - 1) // ALIGN * ALIGN
            self.entries.append((name, p, off, p.numel()))
            off += p.numel()
            prev_fused = bool(re.search(r"\.(q|k)\.(weight|bias)$", name))
        self.total = (off + ALIGN - 1) // ALIGN * ALIGN
        self.device = dev
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.lp = torch.zeros(self.total, dtype=ops.BF16, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.offset: Dict[int, int] = {}
        with torch.no_grad():
            for name, p, o, n in self.entries:
                if p.dtype != torch.float32:
                    raise RuntimeError(f"{name}: master weights must be fp32")
                view = self.flat[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                g = self.grad[o:o + n].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g if p.requires_grad else None
                self.offset[id(p)] = o
        self.refresh_lp()

    # ------------------------------------------------------------------
    def valid(self) -> bool:
        base = self.flat.data_ptr()
        for _, p, o, _ in self.entries:
            if p.data_ptr() != base + 4 * o:
                return False
        return True

    def owns(self, p: nn.Parameter) -> bool:
        o = self.offset.get(id(p))
        return o is not None and p.data_ptr() == self.flat.data_ptr() + 4 * o

    def rebind_grads(self):
        """Re-attach param.grad to the arena (after zero_grad(set_to_none=True) by a foreign optimizer)."""
        gbase = self.grad.data_ptr()
        for _, p, o, n in self.entries:
            if not p.requires_grad:
                continue
            if p.grad is None or p.grad.data_ptr() != gbase + 4 * o:
                g = self.grad[o:o + n].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                else:
                    g.zero_()
                p.grad = g

    def refresh_lp(self):
        ops.cast_bf16_into(self.flat, self.lp)

    def zero_grad(self):
        self.grad.zero_()

    # ------------------------------------------------------------------ views
    def _span(self, first: nn.Parameter, last: nn.Parameter) -> Tuple[int, int]:
        o0 = self.offset[id(first)]
        o1 = self.offset[id(last)] + last.numel()
        return o0, o1

    def lp_view(self, first: nn.Parameter, last: nn.Parameter = None, shape=None) -> torch.Tensor:
        last = first if last is None else last
        o0, o1 = self._span(first, last)
        return self.lp[o0:o1].view(shape if shape is not None else first.shape)

    def f32_view(self, first: nn.Parameter, last: nn.Parameter = None, shape=None) -> torch.Tensor:
        last = first if last is None else last
        o0, o1 = self._span(first, last)
        return self.flat[o0:o1].view(shape if shape is not None else first.shape)

    def grad_view(self, first: nn.Parameter, last: nn.Parameter = None, shape=None) -> torch.Tensor:
        last = first if last is None else last
        o0, o1 = self._span(first, last)
        return self.grad[o0:o1].view(shape if shape is not None else first.shape)

    def fused_ok(self, a: nn.Parameter, b: nn.Parameter, c: nn.Parameter) -> bool:
        oa, ob, oc = self.offset[id(a)], self.offset[id(b)], self.offset[id(c)]
        return ob == oa + a.numel() and oc == ob + b.numel()


def get_arena(module: nn.Module, full_check: bool = False) -> ParamArena:
    """Arena of `module`: the one its root model bound, or (stand-alone use of a building block) its own.
    Sub-modules only verify their first parameter (cheap); the root model passes full_check=True once per forward."""
    arena = getattr(module, "_arena", None)
    if arena is not None:
        if full_check:
            ok = arena.valid()
        else:
            p = next(module.parameters(), None)
            ok = p is None or arena.owns(p)
        if ok:
            return arena
    return bind_arena(module)


def bind_arena(root: nn.Module) -> ParamArena:
    arena = ParamArena(root)
    for m in root.modules():
        object.__setattr__(m, "_arena", arena)
        if hasattr(m, "_views"):
            object.__setattr__(m, "_views", None)
    return arena
