"""GPU tests of the RCCL layer behind the C ABI (octmae_comm_*, include/octmae.h) and of FlatGradReducer on top of it.

One GPU is enough for the first group: a ONE-RANK communicator goes through the same ncclCommInitRank / ncclAllReduce /
stream-and-event path as an 8-rank one (the collective itself degenerates to a copy), so it checks the bootstrap, the stream
ordering (`after_stream` -> communication stream -> octmae_comm_wait) and the reducer's slice bookkeeping.
The second group needs >= 2 GPUs on the box and is skipped otherwise: a fresh torchrun child (started at COLLECTION time,
before this process touches the GPU) runs tests/dp_worker.py on 2 ranks and compares the exchanged gradients with the mean of
the two ranks' local gradients computed in one process.
"""
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- 2-rank child, launched before anything in this process initialises the GPU (device_count() does not) -------------------
_CHILD = None
_CHILD_OUT = None
if os.environ.get("OCTMAE_SKIP_2GPU_TEST") is None and torch.cuda.device_count() >= 2:
    _CHILD_OUT = tempfile.mkdtemp(prefix="octmae_dp_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OCTMAE_DP_OUT=_CHILD_OUT)
    _CHILD = subprocess.Popen(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29617", os.path.join(ROOT, "tests", "dp_worker.py")],
        cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)

# ---- 2 ranks on ONE GPU, exchange over gloo (every box): the world-size-2 run of the GPU training path -----------------------
_CHILD_GLOO = None
_CHILD_GLOO_OUT = None
if os.environ.get("OCTMAE_SKIP_2RANK_GLOO_TEST") is None and torch.cuda.device_count() >= 1:
    _CHILD_GLOO_OUT = tempfile.mkdtemp(prefix="octmae_dp_gloo_")
    env = dict(os.environ, OCTMAE_DP_OUT=_CHILD_GLOO_OUT, OCTMAE_DP_BACKEND="gloo")
    _CHILD_GLOO = subprocess.Popen(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29619", os.path.join(ROOT, "tests", "dp_worker.py")],
        cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)

# ---- bench.py itself with FOUR ranks on the one GPU (gloo): the multi-rank control flow of the script the driver launches on 8 ----
_CHILD_BENCH4 = None
if os.environ.get("OCTMAE_SKIP_BENCH4_TEST") is None and torch.cuda.device_count() >= 1:
    _CHILD_BENCH4 = subprocess.Popen(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
         "--master-port", "29623", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--gloo-one-gpu", "--same-data",
         "--global-batch", "16", "--micro-batch", "4", "--steps", "2", "--warmup", "2"],
        cwd=ROOT, env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE)

# ---- 2 ranks, native backend, rank 1 fails to create its communicator: every rank must leave with status 3 -------------------------
_CHILD_FAIL = None
if os.environ.get("OCTMAE_SKIP_COMM_FAIL_TEST") is None and torch.cuda.device_count() >= 1:
    _CHILD_FAIL = subprocess.Popen(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29627", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
        cwd=ROOT, env=dict(os.environ, OCTMAE_BENCH_FORCE_LOCAL_RANK="0", OCTMAE_BENCH_FAIL_COMM_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0"),
        stdout=subprocess.PIPE, stderr=subprocess.PIPE)

if torch.cuda.is_available():
    from octcubem_amd import comm as ocomm, models_mae, misc, optim as foptim
    from octcubem_amd.parallel import FlatGradReducer
from oracle import mae3d_ref as O

DEV = "cuda"


@pytest.fixture(scope="module")
def comm1():
    assert ocomm.available(), "librccl could not be loaded"
    c = ocomm.NativeComm(ocomm.NativeComm.unique_id(), 0, 1, 0)
    yield c
    c.destroy()


def test_one_rank_collectives_are_identities(comm1):
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(1 << 20, device=DEV, generator=g)
    for op in (ocomm.SUM, ocomm.AVG, ocomm.MAX):
        y = x.clone()
        comm1.all_reduce_async(y, op)
        comm1.wait()
        torch.cuda.synchronize()
        assert torch.equal(y, x)
    y = x.clone()
    comm1.broadcast_async(y, 0)
    comm1.wait()
    out = torch.empty_like(x)
    comm1.all_gather_async(x, out)
    rs = torch.empty_like(x)
    comm1.reduce_scatter_async(x, rs, ocomm.SUM)
    comm1.wait()
    torch.cuda.synchronize()
    assert torch.equal(y, x) and torch.equal(out, x) and torch.equal(rs, x)
    xb = x.to(torch.bfloat16)
    yb = xb.clone()
    comm1.all_reduce_async(yb, ocomm.SUM)
    comm1.wait()
    torch.cuda.synchronize()
    assert torch.equal(yb, xb)
    assert comm1.all_reduce_scalar(3.25, ocomm.MAX) == 3.25
    comm1.barrier()


def test_collective_is_ordered_behind_the_callers_stream_and_wait_orders_the_consumer(comm1):
    """The all-reduce must see what the compute stream wrote BEFORE the call, and work enqueued on the compute stream AFTER
    wait() must see the collective's result -- with a long-running producer in front so that a missing dependency shows."""
    n = 1 << 24
    a = torch.zeros(n, device=DEV)
    big = torch.randn(4096, 4096, device=DEV)
    for it in range(5):
        a.zero_()
        for _ in range(20):                       # ~ms of queued work in front of the producer
            big = big @ big
            big = big / big.norm()
        a.add_(float(it + 1))                     # producer on the compute stream
        comm1.all_reduce_async(a, ocomm.SUM)      # must run after the add
        comm1.wait()
        b = a * 2.0                               # consumer on the compute stream: must run after the collective
        torch.cuda.synchronize()
        assert float(b.min()) == 2.0 * (it + 1) and float(b.max()) == 2.0 * (it + 1)


def _small_model(seed=7):
    from functools import partial
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=2,
                      decoder_num_heads=2, num_frames=12, t_patch_size=3, pred_t_dim=12, high_res_input_size=128)
    P = O.init_params(cfg, seed=seed, bias_std=0.02)
    m = models_mae.MaskedAutoencoderViT(
        input_size=64, patch_size=16, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=2,
        decoder_num_heads=2, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=12, t_patch_size=3, sep_pos_embed=True,
        cls_embed=True, pred_t_dim=12, high_res_input_size=128)
    m.load_state_dict(P, strict=True)
    return cfg, m.to(DEV)


def test_reducer_one_rank_equals_local_gradient_and_isolates_cold_parameters(comm1):
    cfg, m = _small_model()
    m.train()
    imgs = torch.rand(4, 1, 12, 64, 64, generator=torch.Generator().manual_seed(0)).to(DEV)
    noise = torch.rand(4, cfg.num_patches, generator=torch.Generator().manual_seed(1)).to(DEV)
    m.prepare()
    # local gradient, no reducer (the weight-gradient GEMMs use fp32 atomics for split-K: compare with a tolerance)
    loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
    loss.backward()
    torch.cuda.synchronize()
    local = m.arena.grad.clone()
    m.arena.zero_grad()
    red = FlatGradReducer(m, n_chunks=4, force=True, comm=comm1)
    red.broadcast_parameters(0)
    scaler = misc.NativeScalerWithGradNormCount(fp32=True, reducer=red)
    opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=0.0, betas=(0.9, 0.95))      # lr 0: the step leaves the weights alone
    for step in range(3):
        red.timing = step > 0                 # the measurement bench.py switches on for its timed steps (HIP events on both streams)
        opt.zero_grad()
        loss2, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        scaler(loss2, opt, parameters=list(m.parameters()))
        torch.cuda.synchronize()
        assert float((m.arena.grad - local).abs().max()) <= 1e-5 * float(local.abs().max()) + 1e-7, step
        if step == 0:
            assert red._cold and red._frozen, "high_res_patch_embed never reports a gradient on 64x64 input"
            after_learning = dict(red.stats)
    # exposed-communication fields: two frozen-layout steps were recorded; every chunk of the last one has a ready (compute stream)
    # and a done (communication stream) time after begin_backward(), done >= ready, the cold chunk at the very start
    ts = red.timing_summary(last=1)
    assert ts["steps"] == 2 and 0.0 <= ts["exposed_ms_per_step"] <= ts["exposed_ms_max"] < 1e3 and ts["backward_ms_per_step"] > 0.0
    line = ts["timeline"][0]
    assert len(line["chunks"]) == len(red.bounds) and line["finish_end_ms"] >= line["finish_begin_ms"] > 0.0
    for ch in line["chunks"]:
        assert ch["done_ms"] >= ch["ready_ms"] >= 0.0 and not ch["in_finish"], ch
    assert any(ch["cold"] and ch["ready_ms"] < 0.5 * line["finish_begin_ms"] for ch in line["chunks"])
    names = {id(p): n for n, p in m.named_parameters()}
    cold_names = sorted(names[k] for k in red._cold)
    assert cold_names == ["high_res_patch_embed.proj.bias", "high_res_patch_embed.proj.weight"], cold_names
    # after the learning step the cold parameters sit in chunks of their own, exchanged at begin_backward(): finish() has
    # nothing left to launch, i.e. no byte of the exchange starts after backward has ended
    assert red.stats["launched_in_finish"] == after_learning["launched_in_finish"]
    assert red.stats["bytes_in_finish"] == after_learning["bytes_in_finish"]
    cold_bytes = sum(4 * (e - s) for (s, e), c in zip(red.bounds, red.cold_chunk) if c)
    hot_in_finish = [c for c in range(len(red.bounds)) if not red.cold_chunk[c] and red._remaining[c] > 0]
    assert not hot_in_finish
    assert 0 < cold_bytes < 4 * m.arena.total // 4
    assert red.stats["launched_in_backward"] >= 2 * sum(1 for c in red.cold_chunk if not c)


def test_reducer_skips_the_exchange_on_accumulation_micro_steps(comm1):
    cfg, m = _small_model()
    m.train()
    m.prepare()
    red = FlatGradReducer(m, n_chunks=4, force=True, comm=comm1)
    scaler = misc.NativeScalerWithGradNormCount(fp32=True, reducer=red)
    opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=0.0, betas=(0.9, 0.95))
    imgs = torch.rand(2, 1, 12, 64, 64, generator=torch.Generator().manual_seed(3)).to(DEV)
    opt.zero_grad()
    before = dict(red.stats)
    loss, _, _ = m(imgs, mask_ratio=0.75)
    assert scaler(loss / 2, opt, parameters=list(m.parameters()), update_grad=False) is None
    assert red.stats == before
    loss, _, _ = m(imgs, mask_ratio=0.75)
    norm = scaler(loss / 2, opt, parameters=list(m.parameters()), update_grad=True)
    torch.cuda.synchronize()
    assert red.stats["bytes_total"] - before["bytes_total"] == 4 * m.arena.total
    assert torch.isfinite(norm)


@pytest.mark.skipif(_CHILD is None, reason="needs >= 2 GPUs on the box")
def test_two_rank_gradients_equal_the_mean_of_the_local_ones():
    out, _ = _CHILD.communicate(timeout=900)
    assert _CHILD.returncode == 0, out.decode(errors="replace")[-4000:]
    res = json.load(open(os.path.join(_CHILD_OUT, "result.json")))
    assert res["world"] == 2 and res["backend"].startswith("octmae_comm")
    assert res["params_equal_after_broadcast"]
    assert res["max_rel_err"] <= 1e-5, res
    assert res["ranks_agree"]


@pytest.mark.skipif(_CHILD_GLOO is None, reason="needs a GPU")
def test_two_ranks_on_one_gpu_gradients_equal_the_mean_of_the_local_ones():
    """World size 2 ON HARDWARE with the box's one GPU: two processes share it, each runs the HIP training path on its own
    volumes, FlatGradReducer exchanges the gradient arena chunk by chunk from inside backward (torch.distributed gloo group on
    the device tensors: RCCL refuses two ranks on one device).  Three steps -- the learning step, then two with the frozen
    readiness layout (cold chunks exchanged at begin_backward) -- and after each the exchanged arena must equal the mean of the
    two ranks' local gradients, recomputed by every rank alone; different initial weights per rank check the broadcast."""
    out, _ = _CHILD_GLOO.communicate(timeout=900)
    assert _CHILD_GLOO.returncode == 0, out.decode(errors="replace")[-4000:]
    res = json.load(open(os.path.join(_CHILD_GLOO_OUT, "result.json")))
    assert res["world"] == 2 and res["backend"].startswith("gloo")
    assert res["params_equal_after_broadcast"]
    assert res["max_rel_err"] <= 1e-5, res
    assert res["ranks_agree"]
    assert res["cold"] == ["high_res_patch_embed.proj.bias", "high_res_patch_embed.proj.weight"], res["cold"]
    assert res["reducer"]["launched_in_backward"] > 0


@pytest.mark.skipif(_CHILD_BENCH4 is None, reason="needs a GPU")
def test_bench_script_with_four_ranks_prints_one_line_and_ranks_agree():
    """bench.py as the driver launches it (torch.distributed.run, one process per rank), world size 4, every rank on GPU 0 and
    the exchange over gloo (--gloo-one-gpu; RCCL refuses several ranks per device): ViT-L, global batch 16 as 4 volumes per rank.
    Exercises, on hardware, everything of an N-rank run except RCCL itself: rendezvous, the reducer's learning step, the cold-set
    and launch-order agreements, two frozen-layout steps, the max-over-ranks timing and rank 0's single JSON line.  With
    --same-data every rank holds the same weights, volumes and masking noise, so the ranks' losses must be BIT-equal after the
    exchanged optimizer steps."""
    out, err = _CHILD_BENCH4.communicate(timeout=1500)
    assert _CHILD_BENCH4.returncode == 0, err.decode(errors="replace")[-4000:]
    lines = [l for l in out.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                                  # ONE line on stdout (banners and logs go to stderr)
    res = json.loads(lines[0])
    assert res["n_gpus"] == 4 and res["config"]["global_batch"] == 16 and res["config"]["micro_batch_per_gpu"] == 4
    assert res["unit"] == "volumes/s" and res["value"] > 0 and res["scaling"] == "strong"
    c = res["comm"]
    assert c["backend"].startswith("torch.distributed gloo") and c["ranks_seen"] == 4 and c["agreed_micro_batch"] == 4
    lo, hi = c["last_loss_min_max_over_ranks"]
    assert lo == hi == res["loss"], (lo, hi, res["loss"])
    red = c["reducer"]
    assert red["transport"] == "torch.distributed/gloo" and red["launched_in_backward"] > 0
    assert any(red["cold_chunk"])                                  # high_res_patch_embed: exchanged at begin_backward()


def test_every_rank_exits_3_when_one_rank_cannot_create_the_communicator():
    """bench.py --gpus 2 with the native backend: rank 1's communicator creation fails (simulated), rank 0 is then blocked inside the
    collective ncclCommInitRank waiting for it -- its watchdog thread must see the failure key in the control-plane store and end the
    process with status 3 (only the failing rank used to exit)."""
    if _CHILD_FAIL is None:
        pytest.skip("no GPU at collection time")
    try:
        out, err = _CHILD_FAIL.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        _CHILD_FAIL.kill()
        raise AssertionError("the ranks did not end: a rank is still blocked in the communicator creation")
    text = err.decode(errors="replace")
    assert _CHILD_FAIL.returncode != 0
    assert "simulated communicator failure" in text
    # (rank 0 ends through its own watchdog -- "a peer could not create ..." -- unless the launcher's SIGTERM, sent because rank 1
    # exited with an error, reaches it first: either way nobody is left blocked, which is what the time limit above checks)
