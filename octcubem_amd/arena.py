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

import os
import re
import weakref
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops

ALIGN = 64  # elements; keeps every tensor 256-byte (fp32) / 128-byte (bf16) aligned
# OCTMAE_ALWAYS_REFRESH_LP=1: cast the whole arena before every forward, as until round 4 (the escape hatch for code that edits
# weights behind PyTorch's back -- through ``.data`` -- which no version counter sees)
ALWAYS_REFRESH_LP = os.environ.get("OCTMAE_ALWAYS_REFRESH_LP", "0") == "1"
# The guard behind the skipped cast: every CHECK_LP_EVERY-th skipped refresh compares a sample of the operand copy with
# the master weights on the device (a few elements of EVERY parameter + every 4099th element of the arena: one gather, one host
# read) and, on a mismatch, warns and re-casts.  OCTMAE_CHECK_LP=1: on every refresh (the test suites); OCTMAE_CHECK_LP_EVERY=0: never.
CHECK_LP_EVERY = 1 if os.environ.get("OCTMAE_CHECK_LP", "0") == "1" else int(os.environ.get("OCTMAE_CHECK_LP_EVERY", "64"))

_OWNER: Dict[int, tuple] = {}      # id(param) -> (weakref(param), weakref(arena)): which arena a parameter lives in


def arena_of(p) -> Optional["ParamArena"]:
    """The arena that owns ``p`` right now (its data is that arena's view), or None."""
    ent = _OWNER.get(id(p))
    if ent is None or ent[0]() is not p:
        return None
    ar = ent[1]()
    return ar if ar is not None and ar.owns(p) else None


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
                _OWNER[id(p)] = (weakref.ref(p), weakref.ref(self))
        if len(_OWNER) > 4096:           # entries of collected parameters
            for k in [k for k, (r, a) in _OWNER.items() if r() is None or a() is None]:
                del _OWNER[k]
        self._lp_state = None
        self._lp_skips = 0
        self._lp_sample = None
        self.root = lambda: None          # weakref to the module that bound this arena (bind_arena)
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

    def _versions(self) -> int:
        # in-place writes to a parameter (optimizers of torch.optim, load_state_dict, p.copy_ / p.add_ under no_grad, the
        # logit-scale clamp of the contrastive step) bump ITS version counter -- the arena's own tensor does not see them
        return sum(p._version for _, p, _, _ in self.entries)

    def lp_matches(self) -> bool:
        """Does the operand copy still equal the cast of the master weights -- on a SAMPLE: first / middle / last element of every
        parameter and every 4099th element of the arena?  One device gather + one host read (~0.1 ms); not callable while a HIP graph
        is being captured."""
        if self._lp_sample is None or self._lp_sample.device != self.flat.device:
            idx = []
            for _, _, o, n in self.entries:
                idx += [o, o + n // 2, o + n - 1]
            idx = torch.tensor(idx, dtype=torch.int64)
            idx = torch.cat([idx, torch.arange(0, self.flat.numel(), 4099, dtype=torch.int64)])
            self._lp_sample = idx.to(self.flat.device)
        i = self._lp_sample
        return bool(torch.equal(self.flat[i].to(self.lp.dtype), self.lp[i]))

    def refresh_lp(self, force: bool = False):
        """Bring the 16-bit operand copy up to date with the fp32 master weights: one cast pass over the arena (6 B per parameter)
        -- skipped when nothing has written the parameters since the last pass except FusedAdamW, whose kernel writes the copy of
        what it updates itself (octmae_mt_adamw_fused).  "Nothing" = the sum of the parameters' PyTorch version counters.
        Writes that no version counter sees (``p.data.copy_()``, ``dist.broadcast(p.data)``, EMA swaps through ``.data``, raw-pointer
        kernels, anything written to ``arena.flat``) MUST be followed by ``invalidate_lp()`` -- on the model: ``model.invalidate_lp()``
        -- (INTEGRATION.md section 1); the sampled guard below catches a forgotten call within CHECK_LP_EVERY forwards and says so."""
        vs = self._versions()
        if not (force or ALWAYS_REFRESH_LP) and self._lp_state == vs:
            self._lp_skips += 1
            if CHECK_LP_EVERY > 0 and self._lp_skips % CHECK_LP_EVERY == 0 and self.flat.is_cuda \
                    and not torch.cuda.is_current_stream_capturing() and not self.lp_matches():
                import warnings
                warnings.warn("octcubem_amd: the 16-bit operand copy of the parameter arena no longer matches the fp32 master weights: "
                              "something wrote parameters behind PyTorch's version counters (p.data.copy_(), dist.broadcast(p.data), an "
                              "EMA swap through .data, a raw-pointer kernel).  Re-casting now; call model.invalidate_lp() after such "
                              "writes -- up to OCTMAE_CHECK_LP_EVERY forwards have run on stale operands.", RuntimeWarning, stacklevel=3)
                ops.cast_bf16_into(self.flat, self.lp)
            return
        ops.cast_bf16_into(self.flat, self.lp)
        self._lp_state = vs

    def invalidate_lp(self):
        """The next forward re-casts the whole arena (call after writing weights through ``.data`` or raw pointers)."""
        self._lp_state = None

    def lp_ptr(self, p: nn.Parameter) -> int:
        return self.lp.data_ptr() + self.lp.element_size() * self.offset[id(p)]

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
            # A building block used on its own (a Block / create_block() inside somebody else's model: seams 1 and 2 of SURVEY 8b)
            # is the root of its own small arena and nobody calls prepare() for it: its operand copy follows the master weights
            # here (a dozen version counters; until round 4 it was cast once, at binding, and went stale with the first optimizer
            # step).  Blocks inside this package's models are refreshed once per forward by the model's prepare().
            if not full_check and arena.root() is module:
                arena.refresh_lp()
            return arena
    return bind_arena(module)


def bind_arena(root: nn.Module) -> ParamArena:
    arena = ParamArena(root)
    arena.root = weakref.ref(root)
    for m in root.modules():
        object.__setattr__(m, "_arena", arena)
        if hasattr(m, "_views"):
            object.__setattr__(m, "_views", None)
    return arena
