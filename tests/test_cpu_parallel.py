"""world_size-2 gloo tests (CPU) of the data-parallel path: FlatGradReducer all-reduces slices of a flat gradient arena as
parameters report their gradients, averages over ranks, handles parameters that never receive a gradient, skips the
exchange on accumulation micro-steps, and NativeScalerWithGradNormCount drives it in the right order.
The arena here is a small stand-in with the attributes the reducer reads (the real arena lives in HBM)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeArena:
    def __init__(self, shapes, seed):
        g = torch.Generator().manual_seed(seed)
        self.entries = []
        off = 0
        self.params = []
        for i, s in enumerate(shapes):
            n = int(torch.tensor(s).prod())
            self.entries.append([f"p{i}", None, off, n])
            off += (n + 63) // 64 * 64
        self.total = off
        self.flat = torch.randn(self.total, generator=g)
        self.grad = torch.zeros(self.total)
        for e, s in zip(self.entries, shapes):
            p = torch.nn.Parameter(self.flat[e[2]:e[2] + e[3]].view(s))
            p.grad = self.grad[e[2]:e[2] + e[3]].view(s)
            e[1] = p
            self.params.append(p)
        self.entries = [tuple(e) for e in self.entries]


class FakeModel:
    def __init__(self, arena):
        self.arena = arena


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from octcubem_amd import ops
    from octcubem_amd.parallel import FlatGradReducer
    shapes = [(40, 30), (30,), (1, 1, 16), (500,), (64, 64), (7,)]
    arena = FakeArena(shapes, seed=100 + rank)
    model = FakeModel(arena)
    red = FlatGradReducer(model, n_chunks=3)
    red.broadcast_parameters(0)
    flat0 = arena.flat.clone()
    # --- step 1: accumulation micro-step (no exchange), then a sync step; param 5 never gets a gradient
    gen = torch.Generator().manual_seed(7 + rank)
    local = [torch.randn(s, generator=gen) for s in shapes]
    red.begin_backward(sync=False)
    for p, g in zip(arena.params[:5], local[:5]):
        p.grad.add_(g); ops.notify_grad_ready([p])
    red.finish()
    after_micro = arena.grad.clone()
    red.begin_backward(sync=True)
    for p, g in zip(reversed(arena.params[:5]), reversed(local[:5])):      # backward order
        p.grad.add_(g); ops.notify_grad_ready([p])
    red.finish()
    out = {"rank": rank, "flat0": flat0, "after_micro": after_micro, "grad": arena.grad.clone(), "local": local,
           "offsets": [(e[2], e[3]) for e in arena.entries]}
    # --- autograd-hook route (PyTorch-side parameters): backward through real autograd
    arena.grad.zero_()
    red.begin_backward(sync=True)
    loss = sum((p * (rank + 1)).sum() for p in arena.params[:3])
    loss.backward()
    red.finish()
    out["grad_hook"] = arena.grad.clone()
    # --- frozen readiness layout: parameter 5 reported on no rank in the learning step, so its chunk goes out at
    # begin_backward(); a gradient for it after that is a change of control flow and must fail loudly
    out["frozen"] = bool(red._frozen and id(arena.params[5]) in red._cold)
    arena.grad.zero_()
    b0, f0 = red.stats["launched_in_backward"], red.stats["launched_in_finish"]
    red.begin_backward(sync=True)
    out["early_launches"] = red.stats["launched_in_backward"] - b0
    try:
        arena.params[5].grad.add_(1.0); ops.notify_grad_ready([arena.params[5]])
        out["late_raises"] = False
    except RuntimeError:
        out["late_raises"] = True
    arena.params[5].grad.zero_()
    for p in reversed(arena.params[:5]):
        p.grad.add_(float(rank + 1)); ops.notify_grad_ready([p])
    red.finish()
    out["finish_launches"] = red.stats["launched_in_finish"] - f0
    out["grad_frozen"] = arena.grad.clone()
    # by value (numpy), not as shared-memory handles: a handle needs this process alive when the parent unpickles it
    out = {k: (v.numpy() if isinstance(v, torch.Tensor) else [t.numpy() for t in v] if k == "local" else v) for k, v in out.items()}
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_grad_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=100) for _ in range(2)], key=lambda o: o["rank"])
    outs = [{k: (torch.from_numpy(v) if hasattr(v, "dtype") else [torch.from_numpy(t) for t in v] if k == "local" else v)
             for k, v in o.items()} for o in outs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    a, b = outs
    assert torch.equal(a["flat0"], b["flat0"])                             # broadcast made the weights identical
    for o in outs:                                                          # micro-step: purely local
        for (off, n), g in zip(o["offsets"][:5], o["local"][:5]):
            assert torch.equal(o["after_micro"][off:off + n], g.flatten())
    assert torch.allclose(a["grad"], b["grad"], atol=0, rtol=0)             # all ranks hold the same reduced gradient
    for i, (off, n) in enumerate(a["offsets"]):
        exp = torch.zeros(n) if i == 5 else (2 * a["local"][i] + 2 * b["local"][i]).flatten() / 2
        assert torch.allclose(a["grad"][off:off + n], exp, atol=1e-6), i
    assert torch.equal(a["grad_hook"], b["grad_hook"])
    for o in outs:
        assert o["frozen"] and o["early_launches"] >= 1 and o["late_raises"] and o["finish_launches"] == 0, \
            {k: o[k] for k in ("frozen", "early_launches", "late_raises", "finish_launches")}
    for i, (off, n) in enumerate(a["offsets"]):
        exp = torch.zeros(n) if i == 5 else torch.full((n,), 1.5)
        assert torch.allclose(a["grad_frozen"][off:off + n], exp) and torch.equal(a["grad_frozen"], b["grad_frozen"]), i
    for i, (off, n) in enumerate(a["offsets"]):
        exp = torch.full((n,), 1.5) if i < 3 else torch.zeros(n)
        assert torch.allclose(a["grad_hook"][off:off + n], exp), i


def _worker_two_reducers(rank, world, port, q):
    """Two reducers (one per tower, as coem.train_step drives them) listening during ONE backward, over three steps: the second
    subscription must not silence the first (with a single global callback tower A's parameters never reported, were
    all marked cold in the learning step and were exchanged at begin_backward() -- before backward wrote them -- ever after)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from octcubem_amd import ops
    from octcubem_amd.parallel import FlatGradReducer
    shapes_a = [(33, 10), (100,), (64, 8), (50,)]
    shapes_b = [(20, 20), (300,), (16,)]
    A, B = FakeArena(shapes_a, seed=1), FakeArena(shapes_b, seed=2)
    reds = [FlatGradReducer(FakeModel(A), n_chunks=2), FlatGradReducer(FakeModel(B), n_chunks=2)]
    res = {"rank": rank, "ok": True, "why": ""}
    for step in range(3):
        A.grad.zero_(); B.grad.zero_()
        b0 = [r.stats["launched_in_backward"] for r in reds]
        for r in reds:
            r.begin_backward(sync=True)
        # backward order: tower B's parameters first, then tower A's (interleaved notifications through ONE notify entry point)
        val = float((rank + 1) * (step + 1))
        for p in list(reversed(B.params)) + list(reversed(A.params)):
            p.grad.add_(val); ops.notify_grad_ready([p])
        for r in reds:
            r.finish()
        exp = 1.5 * (step + 1)
        for name, ar, shapes in (("A", A, shapes_a), ("B", B, shapes_b)):
            for (nm, p, off, n) in ar.entries:
                if not torch.allclose(ar.grad[off:off + n], torch.full((n,), exp)):
                    res["ok"] = False; res["why"] += f" step{step}:{name}.{nm}"
        for i, r in enumerate(reds):
            if r._cold:
                res["ok"] = False; res["why"] += f" step{step}:reducer{i} has cold parameters {len(r._cold)}"
            if step > 0 and r.stats["launched_in_backward"] - b0[i] != len(r.bounds):
                res["ok"] = False; res["why"] += f" step{step}:reducer{i} launched {r.stats['launched_in_backward'] - b0[i]} of {len(r.bounds)} in backward"
    # a backward outside a begin/finish window (evaluation, reference gradient) must not launch or raise
    before = [r.stats["bytes_total"] for r in reds]
    loss = sum((p * 2).sum() for p in A.params[:2])
    loss.backward()
    ops.notify_grad_ready([A.params[0]])
    if [r.stats["bytes_total"] for r in reds] != before:
        res["ok"] = False; res["why"] += " exchange outside a window"
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_reducers_share_one_backward_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_reducers, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=100) for _ in range(2)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for o in outs:
        assert o["ok"], o["why"]
