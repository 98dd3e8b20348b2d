"""Data-parallel gradient exchange for the 3-D MAE step: replaces ``torch.nn.parallel.DistributedDataParallel``
(wrap site main_pretrain_oph_joint_2d512_flash_attn.py:434-439) with a reducer designed around one flat fp32
gradient arena and RCCL over xGMI.

  * one process per GPU (torchrun), ``torch.distributed`` backend "nccl" == RCCL on ROCm;
  * gradients already live in one contiguous arena (arena.ParamArena), so a "bucket" is just a slice:
    no flatten/unflatten copies;
  * the arena is cut into a few large contiguous chunks (default 8, ~166 MB each for ViT-L): xGMI is
    point-to-point (7 links x ~153 GB/s), ring collectives are per-link bound, and large messages amortise the
    per-collective latency better than DDP's 25 MB buckets;
  * a chunk is all-reduced on a dedicated communication stream as soon as every parameter in it has reported its
    gradient (ops.notify_grad_ready, fired by the backward Functions; autograd hooks for the few PyTorch-side
    parameters), i.e. overlapped with the rest of backward;
  * with gradient accumulation the exchange happens only on the LAST micro-step (the reference all-reduces on every
    micro-step, engine_pretrain.py:163-170 -- identical result, 1/accum_iter the traffic);
  * parameters that receive no gradient (high_res_patch_embed on 256x256 input) are simply zeros in
    the arena -- no find_unused_parameters machinery is needed.
Works on CPU with the gloo backend for tests (then the "stream" is the caller's thread).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import ops


class FlatGradReducer:
    def __init__(self, model, n_chunks: int = 8, process_group=None, average: bool = True, overlap: bool = True,
                 force: bool = False):
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        self.n_chunks = n_chunks
        self.overlap = overlap      # False: exchange everything in finish() (needed when a parameter is used twice per backward)
        self.force = force          # run the exchange even with a single rank (exercises the RCCL path on one GPU)
        self._arena = None
        self._sync = True
        self._pending: List = []
        self._hooks = []
        self._comm_stream = None

    # ------------------------------------------------------------------ layout
    def _layout(self):
        arena = self.model.arena if hasattr(self.model, "arena") else None
        if arena is None:
            raise RuntimeError("FlatGradReducer needs a model with a flat gradient arena")
        if arena is self._arena:
            return
        self._arena = arena
        total = arena.total
        per = (total + self.n_chunks - 1) // self.n_chunks
        per = (per + 1023) // 1024 * 1024
        self.bounds = [(s, min(s + per, total)) for s in range(0, total, per)]
        self.chunk_of = {}
        self.remaining_init = [0] * len(self.bounds)
        for name, p, o, n in arena.entries:
            if not p.requires_grad:
                continue
            cs = sorted({self._chunk_index(o), self._chunk_index(o + n - 1)})
            cs = list(range(cs[0], cs[-1] + 1))
            self.chunk_of[id(p)] = cs
            for c in cs:
                self.remaining_init[c] += 1
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for name, p, o, n in arena.entries:
            if p.requires_grad:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._autograd_hook))
        if arena.grad.is_cuda and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=arena.grad.device)

    def _chunk_index(self, off: int) -> int:
        per = self.bounds[0][1] - self.bounds[0][0]
        return min(off // per, len(self.bounds) - 1)

    # ------------------------------------------------------------------ per-step protocol
    def begin_backward(self, sync: bool = True):
        """Call before loss.backward().  sync=False (accumulation micro-step): no exchange."""
        self._layout()
        self._sync = sync and (self.world > 1 or self.force)
        self._remaining = list(self.remaining_init)
        self._seen = set()
        self._launched = [False] * len(self.bounds)
        self._pending = []
        ops.set_grad_ready_callback(self._on_ready if (self._sync and self.overlap) else None)

    def _autograd_hook(self, p):
        if self._sync and self.overlap:
            self._on_ready([p])

    def _on_ready(self, params):
        for p in params:
            k = id(p)
            if k in self._seen or k not in self.chunk_of:
                continue
            self._seen.add(k)
            for c in self.chunk_of[k]:
                self._remaining[c] -= 1
                if self._remaining[c] == 0:
                    self._launch(c)

    def _launch(self, c: int):
        if self._launched[c]:
            return
        self._launched[c] = True
        s, e = self.bounds[c]
        buf = self._arena.grad[s:e]
        if buf.is_cuda:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                if self.average:
                    buf.mul_(1.0 / self.world)
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append(work)
        else:
            if self.average:
                buf.mul_(1.0 / self.world)
            self._pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Call after backward, before the optimizer: launches whatever has not been exchanged yet (parameters that
        received no gradient this step) and makes the compute stream wait for the communication stream."""
        ops.set_grad_ready_callback(None)
        if not self._sync:
            return
        for c in range(len(self.bounds)):
            self._launch(c)
        for w in self._pending:
            w.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._pending = []

    def broadcast_parameters(self, src: int = 0):
        """DDP's constructor broadcast: make every rank start from rank `src`'s weights (one collective)."""
        self._layout()
        if self.world > 1 or self.force:
            dist.broadcast(self._arena.flat, src=src, group=self.group)
