"""Data-parallel gradient exchange for the 3-D MAE step: replaces ``torch.nn.parallel.DistributedDataParallel``
(wrap site main_pretrain_oph_joint_2d512_flash_attn.py:434-439) with a reducer designed around one flat fp32
gradient arena and RCCL over xGMI.

  * one process per GPU (torchrun); on the GPU the collectives go through the library's own RCCL communicator
    (``comm.NativeComm`` -> ``octmae_comm_*`` in include/octmae.h), on its own stream; ``torch.distributed`` is used
    for CPU tests (gloo) and as an explicit, logged fall-back when no native communicator is given;
  * gradients already live in one contiguous arena (arena.ParamArena), so a "bucket" is just a slice:
    no flatten/unflatten copies, and the mean is RCCL's own ``ncclAvg`` (no pre-scaling pass over the arena);
  * the arena is cut into a few large contiguous chunks (~8, ~166 MB each for ViT-L): xGMI is point-to-point
    (7 links x ~153 GB/s), ring collectives are per-link bound, and large messages amortise the per-collective
    latency better than DDP's 25 MB buckets;
  * a chunk is all-reduced as soon as every parameter in it has reported its gradient (ops.notify_grad_ready, fired by
    the backward Functions; autograd hooks for the few PyTorch-side parameters), i.e. overlapped with the rest of backward;
  * chunks follow READINESS, not just offsets: parameters that reported no gradient on ANY rank in the first exchanged
    backward (``high_res_patch_embed`` in the 3-D-only step) are cut out into chunks of their own, so they no
    longer hold back the ~166 MB slice they sit in.  The set is agreed across ranks once (one MAX all-reduce of a
    per-parameter mask at the end of that first step -- the ranks' collectives must have the same sizes) and then FROZEN:
    from the second exchanged step on the "cold" chunks (zeros in the arena) are launched in begin_backward(), under the
    whole backward.  A cold parameter that receives a gradient later is a change of the model's control flow after the
    layout was learned: the early exchange may race with the kernel that wrote it, so the reducer raises instead of
    exchanging something wrong (``relearn()`` starts the learning step over);
  * with gradient accumulation the exchange happens only on the LAST micro-step (the reference all-reduces on every
    micro-step, engine_pretrain.py:163-170 -- identical result, 1/accum_iter the traffic).
Works on CPU with the gloo backend for tests (then the "stream" is the caller's thread).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import ops


class FlatGradReducer:
    def __init__(self, model, n_chunks: int = 8, process_group=None, average: bool = True, overlap: bool = True,
                 force: bool = False, comm=None, allow_torch_nccl: bool = False):
        self.model = model
        self.group = process_group
        self.comm = comm            # comm.NativeComm (GPU) or None (torch.distributed)
        # The transport is fixed HERE, by the caller, and never changes behind its back: a NativeComm -> octmae_comm_* (RCCL behind
        # the C ABI); none -> torch.distributed (gloo for the CPU tests and the one-GPU diagnostic).  GPU gradients over
        # torch.distributed's own NCCL group -- a second RCCL backend on the data path -- only when asked for by name.
        self.allow_torch_nccl = allow_torch_nccl
        if comm is not None:
            self.world = comm.world
        else:
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
        self._cold = frozenset()    # ids of parameters that reported nothing, on any rank, in the learning step
        self._frozen = False        # True once the cold set has been agreed across ranks (after the first exchanged backward)
        self._seen = set()
        # A parameter used by SEVERAL forwards of one backward (the joint 3-D + 2-D/512 loop) reports once per use, and its
        # chunk may only go out after the last report.  multi_use = True: the first exchanged backward runs without overlap
        # and counts the reports per parameter; later backwards expect exactly those counts (one more is an error, raised).
        self.multi_use = False
        self._expected = None       # id(param) -> reports per backward, learned
        self._reports = {}
        # Launch ORDER.  Every rank must enqueue the chunks' collectives in the same order (RCCL / gloo match collectives by
        # sequence, not by buffer): the order of the first step on the frozen layout is compared across ranks once (one small
        # MAX all-reduce, host-blocking, in that step's finish()) and from then on every launch is checked locally against it --
        # a rank whose backward reports gradients in another order RAISES before it enqueues a mismatched collective, instead
        # of hanging the node or averaging the wrong slices.
        self._order_learned = None  # tuple of chunk indices in launch order, agreed across ranks (frozen layout only)
        self._order_now: List[int] = []
        self.stats = {"launched_in_backward": 0, "launched_in_finish": 0, "bytes_in_finish": 0, "bytes_total": 0,
                      "last_launch_bytes": 0}
        # Measurement (bench.py): with timing = True every exchanged step leaves device events behind -- backward begin, per chunk
        # "gradients final on the compute stream" / "collective complete on the communication stream", and the two ends of
        # finish()'s wait on the compute stream -- resolved by timing_summary().  What the optimizer WAITED for the exchange
        # (the compute stream idle between the end of backward and the completion of the last collective) is the exposed
        # communication of the step; DDP reports the same quantity through its logging data.  Off: no events, no cost.
        self.timing = False
        self._tl_steps: List = []

    # ------------------------------------------------------------------ layout
    def _layout(self, rebuild: bool = False):
        arena = self.model.arena if hasattr(self.model, "arena") else None
        if arena is None:
            raise RuntimeError("FlatGradReducer needs a model with a flat gradient arena")
        if arena is self._arena and not rebuild:
            return
        new_arena = arena is not self._arena
        self._arena = arena
        total = arena.total
        per = (total + self.n_chunks - 1) // self.n_chunks
        per = (per + 1023) // 1024 * 1024
        # Cut points: every `per` elements, plus the borders of each run of cold parameters (rounded outwards to the entries'
        # own offsets, which are 64-element aligned), so that a cold run is a chunk of its own.
        ents = [(o, n, p) for _, p, o, n in arena.entries if p.requires_grad]
        cuts = {0, total}
        cuts.update(range(per, total, per))
        for i, (o, n, p) in enumerate(ents):
            cold = id(p) in self._cold
            prev_cold = i > 0 and id(ents[i - 1][2]) in self._cold
            if cold != prev_cold and i > 0:
                cuts.add(o)
        cuts = sorted(cuts)
        self.bounds = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        starts = [a for a, _ in self.bounds]
        import bisect
        self._chunk_index = lambda off: max(0, bisect.bisect_right(starts, off) - 1)
        self.chunk_of = {}
        self.remaining_init = [0] * len(self.bounds)
        self.cold_chunk = [True] * len(self.bounds)
        for o, n, p in ents:
            c0, c1 = self._chunk_index(o), self._chunk_index(o + n - 1)
            cs = list(range(c0, c1 + 1))
            self.chunk_of[id(p)] = cs
            for c in cs:
                self.remaining_init[c] += 1
                if id(p) not in self._cold:
                    self.cold_chunk[c] = False
        for c in range(len(self.bounds)):
            if self.remaining_init[c] == 0:
                self.cold_chunk[c] = True       # padding only
        if new_arena:
            for h in self._hooks:
                h.remove()
            self._hooks = []
            for name, p, o, n in arena.entries:
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._autograd_hook))
        if arena.grad.is_cuda and self.comm is None and self.world > 1 and not self.allow_torch_nccl \
                and dist.get_backend(self.group) == "nccl":
            raise RuntimeError("FlatGradReducer: GPU gradients and no NativeComm -- pass comm=NativeComm.from_env() (the octmae_comm_* "
                               "RCCL path) or allow_torch_nccl=True to exchange through torch.distributed's NCCL group on purpose")
        if arena.grad.is_cuda and self.comm is None and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=arena.grad.device)

    # ------------------------------------------------------------------ per-step protocol
    def begin_backward(self, sync: bool = True):
        """Call before loss.backward().  sync=False (accumulation micro-step): no exchange."""
        self._layout()
        self._sync = sync and (self.world > 1 or self.force)
        self._remaining = list(self.remaining_init)
        self._seen = set()
        self._launched = [False] * len(self.bounds)
        self._pending = []
        self._in_backward = True
        self._reports = {}
        self._order_now = []
        # The LEARNING step (layout not frozen yet) launches nothing during backward: its chunks go out in finish(), in index order --
        # the same order on every rank whatever order their gradients arrived in.  (Until round 4 it launched by readiness and the
        # order was only compared afterwards: a rank that diverged in that one step could hang its peers before the check ran.)
        self._overlap_now = self._sync and self.overlap and self._frozen and not (self.multi_use and self._expected is None)
        self._tl = None
        if self.timing and self._sync:
            self._tl = {"b0": self._event_now(), "chunks": [], "host_wait_ms": 0.0}
        ops.add_grad_ready_callback(self, self._on_ready if self._overlap_now else self._on_ready_note)
        if self._overlap_now and self._frozen:
            # chunks made only of parameters that never report (zeros in the arena) are exchanged NOW, under the whole backward
            for c in range(len(self.bounds)):
                if self.cold_chunk[c]:
                    self._launch(c)

    def _autograd_hook(self, p):
        if getattr(self, "_overlap_now", False):
            self._on_ready([p])
        else:
            self._on_ready_note([p])

    def _on_ready_note(self, params):          # no overlap: only remember who reported, and how often
        for p in params:
            self._seen.add(id(p))
            self._reports[id(p)] = self._reports.get(id(p), 0) + 1

    def _on_ready(self, params):
        for p in params:
            k = id(p)
            if k not in self.chunk_of:
                continue
            n = self._reports.get(k, 0) + 1
            self._reports[k] = n
            exp = 1 if not self.multi_use else self._expected.get(k, 1)
            if n < exp:
                continue
            if n > exp:
                if self.multi_use:
                    raise RuntimeError("FlatGradReducer: a parameter reported more gradient contributions in this backward than "
                                       "in the one its schedule was learned from; its slice may already have been exchanged")
                continue                       # single-use mode: repeated reports of one Function are harmless duplicates
            self._seen.add(k)
            for c in self.chunk_of[k]:
                self._remaining[c] -= 1
                if self._launched[c]:
                    raise RuntimeError("FlatGradReducer: a parameter that reported no gradient on any rank in the learning step "
                                       "received one now, after its chunk was exchanged at begin_backward(); the model's "
                                       "control flow changed -- call reducer.relearn() before such a step")
                if self._remaining[c] == 0:
                    self._launch(c)

    # ------------------------------------------------------------------ measurement
    def _event_now(self, stream=None):
        """A timing event recorded now on `stream` (default: the current stream); None on the CPU."""
        if self._arena is None or not self._arena.grad.is_cuda:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream) if stream is not None else ev.record()
        return ev

    def timing_summary(self, last: int = 1):
        """Resolve the events of the exchanged steps recorded so far (synchronises the device).  Returns
        {"steps": n, "exposed_ms_per_step": mean wait of the compute stream in finish(), "exposed_ms_max": ...,
         "backward_ms_per_step": begin_backward() -> finish() on the compute stream,
         "timeline": for each of the `last` steps the chunks in launch order:
                     {chunk, mb, cold, in_finish, ready_ms, done_ms} relative to begin_backward(), plus finish_begin_ms / finish_end_ms}."""
        steps = self._tl_steps
        if not steps:
            return None
        if self._arena.grad.is_cuda:
            torch.cuda.synchronize(self._arena.grad.device)
        exposed, bwd, lines = [], [], []
        for st in steps:
            if st["b0"] is None:           # CPU transport: host wall-clock of the waits
                exposed.append(st["host_wait_ms"])
                continue
            exposed.append(st["f0"].elapsed_time(st["f1"]))
            bwd.append(st["b0"].elapsed_time(st["f0"]))
        for st in steps[-last:]:
            if st["b0"] is None:
                continue
            lines.append({"finish_begin_ms": round(st["b0"].elapsed_time(st["f0"]), 3), "finish_end_ms": round(st["b0"].elapsed_time(st["f1"]), 3),
                          "chunks": [{"chunk": c, "mb": round(nb / 1e6, 1), "cold": bool(self.cold_chunk[c]) if c < len(self.cold_chunk) else None,
                                      "in_finish": fin, "ready_ms": round(st["b0"].elapsed_time(r), 3), "done_ms": round(st["b0"].elapsed_time(d), 3)}
                                     for c, nb, fin, r, d in st["chunks"]]})
        return {"steps": len(steps), "exposed_ms_per_step": sum(exposed) / len(exposed), "exposed_ms_max": max(exposed),
                "backward_ms_per_step": (sum(bwd) / len(bwd)) if bwd else None, "timeline": lines}

    def _launch(self, c: int):
        if self._launched[c]:
            return
        if self._order_learned is not None:
            i = len(self._order_now)
            if i >= len(self._order_learned) or self._order_learned[i] != c:
                exp = self._order_learned[i] if i < len(self._order_learned) else None
                # This rank stops BEFORE it enqueues the mismatched collective.  Its peers have enqueued theirs and wait for it: they
                # end through the collective timeout of their transport (gloo: the group's timeout; RCCL: NCCL_TIMEOUT / the
                # launcher's watchdog) -- torchrun then tears the job down because this rank exited with an error.
                raise RuntimeError(f"FlatGradReducer: gradient chunk {c} became ready as launch #{i} of this backward, but the order "
                                   f"agreed across ranks has chunk {exp} there -- this rank's control flow diverged from the step the "
                                   "schedule was learned from (collectives are matched by sequence: exchanging now would hang or "
                                   "average the wrong slices).  Call reducer.relearn() on EVERY rank before a step whose backward "
                                   "differs")
        self._order_now.append(c)
        self._launched[c] = True
        s, e = self.bounds[c]
        buf = self._arena.grad[s:e]
        nbytes = 4 * (e - s)
        self.stats["bytes_total"] += nbytes
        self.stats["last_launch_bytes"] = nbytes        # after finish(): the chunk that went out last (its exchange is the exposed tail)
        if self._in_backward:
            self.stats["launched_in_backward"] += 1
        else:
            self.stats["launched_in_finish"] += 1
            self.stats["bytes_in_finish"] += nbytes
        tl = getattr(self, "_tl", None)
        ready = self._event_now() if (tl is not None and buf.is_cuda) else None
        if self.comm is not None:
            from . import comm as C
            self.comm.all_reduce_async(buf, C.AVG if self.average else C.SUM)
            if ready is not None:
                tl["chunks"].append((c, nbytes, not self._in_backward, ready, self._event_now(self.comm.torch_stream())))
        elif buf.is_cuda:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM, group=self.group,
                                       async_op=True)
            self._pending.append(work)
            if ready is not None:
                tl["chunks"].append((c, nbytes, not self._in_backward, ready, self._event_now(self._comm_stream)))
        else:
            if self.average:
                buf.mul_(1.0 / self.world)      # gloo has no AVG
            self._pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Call after backward, before the optimizer: launches whatever has not been exchanged yet (parameters that
        received no gradient this step) and makes the compute stream wait for the communication stream."""
        ops.remove_grad_ready_callback(self)
        self._in_backward = False
        # outside a begin_backward() / finish() window the autograd hooks only note (an evaluation or reference backward must
        # neither launch collectives nor trip the multi-use count of the step that has just ended)
        self._overlap_now = False
        if not self._sync:
            self._reports = {}
            return
        tl = getattr(self, "_tl", None)
        if tl is not None:
            import time
            tl["f0"] = self._event_now()
            t_host = time.perf_counter()
        for c in range(len(self.bounds)):
            self._launch(c)
        if self.comm is not None:
            self.comm.wait()
        for w in self._pending:
            w.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._pending = []
        if tl is not None:
            tl["host_wait_ms"] = 1e3 * (time.perf_counter() - t_host)
            tl["f1"] = self._event_now()
            self._tl_steps.append(tl)
            if len(self._tl_steps) > 64:
                del self._tl_steps[0]
            self._tl = None
        if self.multi_use and self._expected is None:
            self._expected = dict(self._reports)
        self._reports = {}
        # learning step: whoever reported on no rank is "cold"; agreed across ranks once, then frozen
        if self.overlap and not self._frozen:
            self._agree_order(learn=False)      # the learning step's own launches must already have matched
            self._cold = self._agree_cold()
            self._frozen = True
            self._layout(rebuild=True)
        elif self._order_learned is None and (not self.overlap or self._frozen):
            self._agree_order(learn=True)       # first step on the final layout: agree, then check locally ever after

    def relearn(self):
        """Forget the readiness layout: the next exchanged backward is a learning step again (no early launches)."""
        self._cold = frozenset()
        self._frozen = False
        self._order_learned = None
        if self._arena is not None:
            self._layout(rebuild=True)

    def _allreduce_max_host(self, v: torch.Tensor) -> torch.Tensor:
        """MAX over ranks of a small fp32 host vector through the reducer's own transport (host-blocking)."""
        if not (self.world > 1 or self.force):
            return v
        if self.comm is not None:
            from . import comm as C
            d = v.to(self._arena.grad.device)
            self.comm.all_reduce_async(d, C.MAX)
            self.comm.wait()
            return d.cpu()
        if dist.is_initialized():
            d = v.to(self._arena.grad.device) if self._arena.grad.is_cuda and dist.get_backend(self.group) == "nccl" else v.clone()
            dist.all_reduce(d, op=dist.ReduceOp.MAX, group=self.group)
            return d.cpu()
        return v

    def _agree_order(self, learn: bool):
        """Compare this backward's launch order across ranks (max(v) == -max(-v) element-wise); learn=True also makes it the
        order every later backward is checked against locally."""
        order = list(self._order_now)
        v = torch.tensor([float(len(order))] + [float(c) for c in order], dtype=torch.float32)
        n = len(self.bounds) + 1
        v = torch.cat([v, torch.full((max(0, n - v.numel()),), -1.0)])[:n]
        m = self._allreduce_max_host(torch.cat([v, -v]))
        if not torch.equal(m[:n], -m[n:]):
            raise RuntimeError("FlatGradReducer: the ranks launched their gradient chunks in different orders in this backward "
                               f"(this rank: {order}); the exchanged gradients of this step are not trustworthy -- every rank must "
                               "run the same control flow")
        if learn:
            self._order_learned = tuple(order)

    def _agree_cold(self):
        """ids of the parameters no rank saw a gradient for in the backward just finished (one small MAX all-reduce; the
        only host-blocking exchange of the reducer, once per learning step)."""
        keys = [id(p) for _, p, _, _ in self._arena.entries if p.requires_grad and id(p) in self.chunk_of]
        mask = self._allreduce_max_host(torch.tensor([1.0 if k in self._seen else 0.0 for k in keys], dtype=torch.float32))
        return frozenset(k for k, v in zip(keys, mask.tolist()) if v == 0.0)

    @property
    def transport(self) -> str:
        if self.comm is not None:
            return "octmae_comm"
        return f"torch.distributed/{dist.get_backend(self.group)}" if dist.is_initialized() else "none (single process)"

    def exposed_bytes_last_step(self) -> int:
        """Bytes whose all-reduce could only start in finish() (nothing of backward left to hide it), summed over all steps
        so far; divide by the number of exchanged steps."""
        return self.stats["bytes_in_finish"]

    def broadcast_parameters(self, src: int = 0):
        """DDP's constructor broadcast: make every rank start from rank `src`'s weights (one collective)."""
        self._layout()
        if self.world > 1 or self.force:
            if self.comm is not None:
                self.comm.broadcast_async(self._arena.flat, src)
                self.comm.wait()
            else:
                dist.broadcast(self._arena.flat, src=src, group=self.group)
            # the arena was written as ONE tensor: no parameter's version counter moved, so say it (the next forward re-casts)
            if hasattr(self._arena, "invalidate_lp"):
                self._arena.invalidate_lp()
