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
        self.lp_invalidations = 0

    def invalidate_lp(self):
        """arena.ParamArena's contract: whoever writes the flat weight buffer as ONE tensor (no parameter's version counter moves)
        says so, and the next forward re-casts the 16-bit operand copy."""
        self.lp_invalidations += 1


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
    assert arena.lp_invalidations == 1, "broadcast_parameters must invalidate the operand copy (it writes the arena as one tensor)"
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
    # --- frozen readiness layout: parameter 5 reported on no rank in the learning step, so its chunk goes out at
    # begin_backward(); a gradient for it after that is a change of control flow and must fail loudly
    out["frozen"] = bool(red._frozen and id(arena.params[5]) in red._cold)
    arena.grad.zero_()
    b0, f0 = red.stats["launched_in_backward"], red.stats["launched_in_finish"]
    red.timing = True                      # the measurement bench.py switches on for its timed steps (host clock on the CPU)
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
    red.timing = False
    out["timing"] = red.timing_summary()
    # --- autograd-hook route (PyTorch-side parameters): backward through real autograd.  Only three parameters take part: a
    # different control flow from the steps above, so every rank calls relearn() first (without it the launch-order check
    # of the frozen layout raises, see test_control_flow_divergence_raises_instead_of_hanging)
    red.relearn()
    arena.grad.zero_()
    red.begin_backward(sync=True)
    loss = sum((p * (rank + 1)).sum() for p in arena.params[:3])
    loss.backward()
    red.finish()
    out["grad_hook"] = arena.grad.clone()
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
        # one exchanged step was recorded with timing on: its wait (host clock over gloo) is reported, non-negative and finite
        t = o["timing"]
        assert t["steps"] == 1 and 0.0 <= t["exposed_ms_per_step"] < 1e4 and t["exposed_ms_max"] >= t["exposed_ms_per_step"], t
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


# ------------------------------------------------------------------------------------------------------------------------
# world sizes 4 and 8: what an 8-rank launch runs, except RCCL itself
# ------------------------------------------------------------------------------------------------------------------------
def _run_children(target, world, extra=(), timeout_s=150):
    """Start `world` spawn-children, collect one result each; whatever happens, no child outlives this function (children
    are ended by their own PIDs)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    outs = []
    try:
        import queue as _q
        import time as _t
        deadline = _t.time() + timeout_s
        while len(outs) < world and _t.time() < deadline:
            try:
                outs.append(q.get(timeout=1.0))
            except _q.Empty:
                if all(not p.is_alive() for p in procs) and q.empty():
                    break
        for p in procs:
            p.join(timeout=20)
    finally:
        alive = [p for p in procs if p.is_alive()]
        for p in alive:
            p.terminate()
        for p in alive:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    return sorted(outs, key=lambda o: o["rank"]), [p.exitcode for p in procs], bool(alive)


def _worker_world(rank, world, port, q):
    from datetime import timedelta
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
    from octcubem_amd import ops
    from octcubem_amd.parallel import FlatGradReducer
    shapes = [(24, 20), (130,), (40, 30), (30,), (1, 1, 16), (500,), (64, 64), (7,)]
    arena = FakeArena(shapes, seed=100 + rank)
    red = FlatGradReducer(FakeModel(arena), n_chunks=4)
    red.broadcast_parameters(0)
    res = {"rank": rank, "ok": True, "why": "", "flat_sum": float(arena.flat.double().sum())}

    def fail(msg):
        res["ok"] = False; res["why"] += " | " + msg

    def expect(step, vals):
        """vals[i]: expected (mean over ranks) constant in parameter i's gradient slice."""
        for i, (nm, p, off, n) in enumerate(arena.entries):
            if not torch.allclose(arena.grad[off:off + n], torch.full((n,), float(vals[i])), atol=1e-6):
                fail(f"{step}: p{i} holds {float(arena.grad[off])} expected {vals[i]}")

    mean_rank = sum(range(1, world + 1)) / world
    # ---- step 1 (learning): p7 never reports; p0 reports ONLY on the last rank, as the very last notification, so the
    # launch order is the same everywhere; the agreed cold set must hold p7 and must NOT hold p0 (MAX over ranks)
    arena.grad.zero_()
    red.begin_backward(sync=True)
    for i in range(6, 0, -1):
        arena.params[i].grad.add_(float(rank + 1)); ops.notify_grad_ready([arena.params[i]])
    if rank == world - 1:
        arena.params[0].grad.add_(float(world)); ops.notify_grad_ready([arena.params[0]])
    red.finish()
    expect("learning", [1.0] + [mean_rank] * 6 + [0.0])
    cold = {i for i, p in enumerate(arena.params) if id(p) in red._cold}
    if cold != {7} or not red._frozen:
        fail(f"cold set {cold}, frozen {red._frozen}")
    # ---- step 2: frozen layout; the cold chunk goes out at begin_backward(); the launch order is agreed in finish()
    arena.grad.zero_()
    b0 = red.stats["launched_in_backward"]
    red.begin_backward(sync=True)
    early = red.stats["launched_in_backward"] - b0
    for i in range(6, -1, -1):
        arena.params[i].grad.add_(float(rank + 1)); ops.notify_grad_ready([arena.params[i]])
    red.finish()
    expect("frozen", [mean_rank] * 7 + [0.0])
    if early < 1 or red._order_learned is None or len(red._order_learned) != len(red.bounds):
        fail(f"early {early}, order {red._order_learned}, chunks {len(red.bounds)}")
    # ---- step 3: accumulation -- the micro-step exchanges nothing (gradients stay local), the last one exchanges the sum
    arena.grad.zero_()
    bt = red.stats["bytes_total"]
    red.begin_backward(sync=False)
    for i in range(6, -1, -1):
        arena.params[i].grad.add_(float(rank + 1)); ops.notify_grad_ready([arena.params[i]])
    red.finish()
    if red.stats["bytes_total"] != bt:
        fail("an accumulation micro-step exchanged")
    for i, (nm, p, off, n) in enumerate(arena.entries[:7]):
        if not torch.equal(arena.grad[off:off + n], torch.full((n,), float(rank + 1))):
            fail(f"micro-step: p{i} not local")
    red.begin_backward(sync=True)
    for i in range(6, -1, -1):
        arena.params[i].grad.add_(float(rank + 1)); ops.notify_grad_ready([arena.params[i]])
    red.finish()
    expect("accumulated", [2 * mean_rank] * 7 + [0.0])
    # ---- a gradient for the cold parameter on the frozen layout raises (its chunk went out at begin_backward()) ...
    arena.grad.zero_()
    red.begin_backward(sync=True)
    try:
        # (only the notification: the cold chunk's asynchronous all-reduce went out at begin_backward() and may still be in flight --
        # writing its buffer here raced with the collective, one failure in four runs: ADVICE r05)
        ops.notify_grad_ready([arena.params[7]])
        fail("late gradient for a cold parameter did not raise")
    except RuntimeError:
        pass
    for i in range(6, -1, -1):
        arena.params[i].grad.add_(float(rank + 1)); ops.notify_grad_ready([arena.params[i]])
    red.finish()
    expect("after the refused gradient", [mean_rank] * 7 + [0.0])
    # ---- ... and relearn() (on every rank) makes the changed control flow legal: p7 now reports, first of all
    red.relearn()
    for step in ("relearn/learning", "relearn/frozen", "relearn/checked"):
        arena.grad.zero_()
        red.begin_backward(sync=True)
        for i in range(7, -1, -1):
            arena.params[i].grad.add_(float(rank + 1)); ops.notify_grad_ready([arena.params[i]])
        red.finish()
        expect(step, [mean_rank] * 8)
    if red._cold or red._order_learned is None:
        fail(f"after relearn: cold {len(red._cold)}, order {red._order_learned}")
    res["grad_sum"] = float(arena.grad.double().sum())
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [4, 8])
def test_flat_grad_reducer_gloo_world_4_and_8(world):
    outs, codes, killed = _run_children(_worker_world, world)
    assert not killed and len(outs) == world and all(c == 0 for c in codes), (len(outs), codes, killed)
    for o in outs:
        assert o["ok"], (o["rank"], o["why"])
    assert len({o["flat_sum"] for o in outs}) == 1           # the broadcast made the weights identical
    assert len({o["grad_sum"] for o in outs}) == 1           # every rank holds the same exchanged gradient


def _worker_diverge(rank, world, port, q):
    """Frozen layout, launch order agreed; then rank 1 runs its backward in another order.  It must raise BEFORE it enqueues a
    collective the others would not match; the others may fail (peer gone / time-out) but nobody may hang."""
    from datetime import timedelta
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=15))
    from octcubem_amd import ops
    from octcubem_amd.parallel import FlatGradReducer
    shapes = [(24, 20), (130,), (40, 30), (30,), (500,), (64, 64)]
    arena = FakeArena(shapes, seed=3)
    red = FlatGradReducer(FakeModel(arena), n_chunks=3)
    for _ in range(2):                      # learning step + the step that agrees on the order
        arena.grad.zero_()
        red.begin_backward(sync=True)
        for p in reversed(arena.params):
            p.grad.add_(1.0); ops.notify_grad_ready([p])
        red.finish()
    res = {"rank": rank, "learned": red._order_learned is not None, "raised": None, "peer_error": None}
    dist.barrier()
    arena.grad.zero_()
    try:
        red.begin_backward(sync=True)
        order = list(arena.params) if rank == 1 else list(reversed(arena.params))      # rank 1: forward order
        for p in order:
            p.grad.add_(1.0); ops.notify_grad_ready([p])
        red.finish()
    except RuntimeError as e:
        if "control flow diverged" in str(e):
            res["raised"] = str(e)[:80]
        else:
            res["peer_error"] = type(e).__name__ + ": " + str(e)[:120]
    except Exception as e:                  # gloo reports a vanished peer in several ways
        res["peer_error"] = type(e).__name__ + ": " + str(e)[:120]
    q.put(res)
    q.close(); q.join_thread()              # the feeder thread must have written the result before the process ends
    # no barrier, no destroy: the group is broken by design; the process just ends
    os._exit(0)


@pytest.mark.timeout(180)
def test_control_flow_divergence_raises_instead_of_hanging():
    outs, codes, killed = _run_children(_worker_diverge, 4, timeout_s=90)
    by_rank = {o["rank"]: o for o in outs}
    assert 1 in by_rank and by_rank[1]["learned"] and by_rank[1]["raised"], by_rank.get(1)
    assert not killed, "a rank was still blocked in a collective when the time limit ended"
    for r, o in by_rank.items():
        if r != 1:
            assert o["raised"] is None          # the others ran the agreed order; whatever they got came from the broken group


# ---- bench.py's control plane at world 8 with a gloo-backed stand-in for the RCCL communicator -------------------
# RCCL with N > 1 has never run anywhere (no multi-GPU node was available to the builder).  What CAN run without one is everything
# around the collective calls: comm.NativeComm.from_store's generation-keyed hand-off of the unique id through torch.distributed's
# store, bench.bring_up_native_comm (watchdog thread, "every rank counted in" gate, exit status 3 on EVERY rank when one fails),
# bench.agree_any / max_over_ranks (the micro-batch fallback agreement, the step time and exposed-communication MAX).


def _fake_comm_class():
    from octcubem_amd import comm as ocomm

    class GlooComm(ocomm.NativeComm):
        """comm.NativeComm with the three C-ABI touch points replaced: the unique id is 128 random bytes, "ncclCommInitRank" is a gloo
        barrier (collective: a rank that never joins blocks its peers, as RCCL's does), collectives run on CPU tensors."""
        created = []

        def __init__(self, id_bytes, rank, world, device):
            assert len(id_bytes) == ocomm.ID_BYTES
            self.id_bytes, self.rank, self.world, self.device = bytes(id_bytes), int(rank), int(world), int(device)
            self._h, self._keep = None, []
            dist.barrier()

        @staticmethod
        def unique_id():
            return os.urandom(ocomm.ID_BYTES)

        @staticmethod
        def _set_device(device):
            pass

        def all_reduce_scalar(self, value, op=ocomm.AVG):
            t = torch.tensor([value], dtype=torch.float64)
            dist.all_reduce(t, op={ocomm.SUM: dist.ReduceOp.SUM, ocomm.AVG: dist.ReduceOp.SUM, ocomm.MAX: dist.ReduceOp.MAX}[op])
            return float(t.item()) / (self.world if op == ocomm.AVG else 1)

        def barrier(self):
            dist.barrier()

        def destroy(self):
            pass
    return GlooComm


def _bench_control_worker(rank, world, port, fail_rank, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from octcubem_amd import comm as ocomm
    assert bench.MAX_OP == ocomm.MAX
    Comm = _fake_comm_class()
    store = dist.distributed_c10d._get_default_store()
    comm = bench.bring_up_native_comm(store, rank, world, 0, Comm, fail_rank=fail_rank, poll=0.1, timeout=60.0)
    res = {"rank": rank, "world_seen": int(round(comm.all_reduce_scalar(1.0, ocomm.SUM)))}
    # every rank holds the id rank 0 published (MAX and MIN of a digest agree with the rank's own)
    digest = float(int.from_bytes(comm.id_bytes[:6], "little"))
    res["same_id"] = bench.max_over_ranks(comm, digest) == digest and -bench.max_over_ranks(comm, -digest) == digest
    res["agree_none"] = bench.agree_any(comm, False)
    res["agree_one"] = bench.agree_any(comm, rank == 3)               # one rank ran out of memory: everybody falls back
    res["max"] = bench.max_over_ranks(comm, 1.5 * rank)               # step time / exposed-communication MAX
    # a second communicator on the same store (generation 1) must not read the first one's id
    comm2 = Comm.from_store(store, rank, world, 0)
    res["new_generation_new_id"] = comm2.id_bytes != comm.id_bytes
    d2 = float(int.from_bytes(comm2.id_bytes[:6], "little"))
    res["same_id2"] = bench.max_over_ranks(comm2, d2) == d2
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_bench_control_plane_with_a_fake_communicator_world8():
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_control_worker, args=(r, world, port, None, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r["rank"] for r in res) == list(range(world))
    for r in res:
        assert r["world_seen"] == world and r["same_id"] and r["same_id2"] and r["new_generation_new_id"], r
        assert r["agree_none"] is False and r["agree_one"] is True and r["max"] == 1.5 * (world - 1), r


@pytest.mark.timeout(240)
def test_every_rank_exits_3_when_one_rank_cannot_create_its_communicator_world8():
    """Rank 5's creation fails; the other seven sit inside the collective creation (the stand-in's barrier) -- only the watchdog thread
    can end them.  Every process must exit with status 3, promptly."""
    import time
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_control_worker, args=(r, world, port, 5, q)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert [p.exitcode for p in procs] == [3] * world, [p.exitcode for p in procs]
    assert time.time() - t0 < 110
    assert q.empty()
