"""CPU: octcubem_amd.coem.ClipLoss / gather_features against the reference's own ClipLoss (oracle/gen_golden_coem.py ->
tests/golden/coem_loss.npz): world size 1 (plain and corrected labels) and world size 2 over gloo for every
(local_loss, gather_with_grad) combination -- loss, feature gradients and the temperature gradient of each rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def feats(seed, n, d=32, dup=False):
    g = torch.Generator().manual_seed(seed)
    a = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)
    b = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)
    if dup:
        b[2] = b[0]
    return a, b


def run(loss_mod, a, b):
    a = a.clone().requires_grad_(True); b = b.clone().requires_grad_(True)
    ls = torch.tensor(np.log(1 / 0.07), dtype=torch.float32, requires_grad=True)
    loss = loss_mod(a, b, ls.exp())
    loss.backward()
    return loss.detach().numpy(), a.grad.numpy(), b.grad.numpy(), ls.grad.numpy()


def check(z, tag, got):
    for k, v in zip(("loss", "ga", "gb", "gls"), got):
        np.testing.assert_allclose(v, z[f"{tag}/{k}"], rtol=2e-5, atol=1e-7, err_msg=f"{tag}/{k}")


def test_clip_loss_world1(golden_dir):
    from octcubem_amd.coem import ClipLoss
    z = np.load(os.path.join(golden_dir, "coem_loss.npz"))
    check(z, "w1/plain", run(ClipLoss(), *feats(7, 6)))
    check(z, "w1/corrected", run(ClipLoss(correct_label=1), *feats(7, 6, dup=True)))
    cached = ClipLoss(cache_labels=True)
    check(z, "w1/plain", run(cached, *feats(7, 6)))
    check(z, "w1/plain", run(cached, *feats(7, 6)))             # second call takes the cached labels


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from octcubem_amd.coem import ClipLoss
    out = {}
    for ll in (False, True):
        for gg in (False, True):
            got = run(ClipLoss(local_loss=ll, gather_with_grad=gg, rank=rank, world_size=world), *feats(100 + rank, 3))
            for k, v in zip(("loss", "ga", "gb", "gls"), got):
                out[f"w2/ll{int(ll)}_gg{int(gg)}/r{rank}/{k}"] = np.array(v)
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_clip_loss_gloo_world2(golden_dir):
    z = np.load(os.path.join(golden_dir, "coem_loss.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        got.update(q.get(timeout=150))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == 32
    for k, v in got.items():
        np.testing.assert_allclose(v, z[k], rtol=2e-5, atol=1e-7, err_msg=k)
