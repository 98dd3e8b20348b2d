#!/usr/bin/env python3
"""Golden vectors for the COEM contrastive loss, from the REAL reference (build container only).
    python oracle/gen_golden_coem.py -> tests/golden/coem_loss.npz
Runs retinal-COEM/src/open_clip/loss.py ``ClipLoss`` (loaded as a single file; the open_clip package __init__ pulls in
un-vendored dependencies) on seeded L2-normalised features:
  * world_size 1: plain labels and ``correct_label`` (two samples sharing identical en-face features);
  * world_size 2 over gloo, 2 spawned processes, every (local_loss, gather_with_grad) combination: per-rank loss and the
    gradients of that rank's features and of logit_scale.
"""
import importlib.util
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LOSS = "/root/reference/retinal-COEM/src/open_clip/loss.py"


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_clip_loss", LOSS)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def feats(seed, n, d=32, dup=False):
    g = torch.Generator().manual_seed(seed)
    a = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)
    b = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)
    if dup:
        b[2] = b[0]
    return a, b


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = load_ref()
    out = {}
    for local_loss in (False, True):
        for gwg in (False, True):
            a, b = feats(100 + rank, 3)
            a.requires_grad_(True); b.requires_grad_(True)
            ls = torch.tensor(np.log(1 / 0.07), dtype=torch.float32, requires_grad=True)
            loss = ref.ClipLoss(local_loss=local_loss, gather_with_grad=gwg, rank=rank, world_size=world)(a, b, ls.exp())
            loss.backward()
            tag = f"w2/ll{int(local_loss)}_gg{int(gwg)}/r{rank}"
            out[tag + "/loss"] = loss.detach().numpy(); out[tag + "/ga"] = a.grad.numpy(); out[tag + "/gb"] = b.grad.numpy()
            out[tag + "/gls"] = ls.grad.numpy()
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ref = load_ref()
    save = {}
    for name, dup, cl in (("plain", False, 0), ("corrected", True, 1)):
        a, b = feats(7, 6, dup=dup)
        a.requires_grad_(True); b.requires_grad_(True)
        ls = torch.tensor(np.log(1 / 0.07), dtype=torch.float32, requires_grad=True)
        loss = ref.ClipLoss(correct_label=cl)(a, b, ls.exp())
        loss.backward()
        save[f"w1/{name}/loss"] = loss.detach().numpy(); save[f"w1/{name}/ga"] = a.grad.numpy(); save[f"w1/{name}/gb"] = b.grad.numpy()
        save[f"w1/{name}/gls"] = ls.grad.numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for _ in range(2):
        save.update(q.get(timeout=120))
    for p in procs:
        p.join(timeout=60)
    out = os.path.join(ROOT, "tests", "golden", "coem_loss.npz")
    np.savez_compressed(out, **save)
    print("wrote", out, os.path.getsize(out), "bytes;", len(save), "entries; plain loss", float(save["w1/plain/loss"]))


if __name__ == "__main__":
    main()
