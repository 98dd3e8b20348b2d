#!/usr/bin/env python3
"""Golden vectors for the COEM contrastive loss, from the REAL reference (build container only).
    python oracle/gen_golden_coem.py -> tests/golden/coem_loss.npz
Runs retinal-COEM/src/open_clip/loss.py ``ClipLoss`` (loaded as a single file; the open_clip package __init__ pulls in
un-vendored dependencies) on seeded L2-normalised features:
  * world_size 1: plain labels and ``correct_label`` (two samples sharing identical en-face features);
  * world_size 2 over gloo, 2 spawned processes, every (local_loss, gather_with_grad) combination: per-rank loss and the
    gradients of that rank's features and of logit_scale;
  * ``ThreeModalityClipLoss`` (loss.py:230-385) the same way: world 1 with all / some / no samples of a modality present, and
    world 2 for local_loss = False with and without gather_with_grad (with local_loss = True the reference builds its labels from
    the GLOBAL sample count and its cross-entropy raises on the local logits: not a usable mode).
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


def feats3(seed, n, d=32):
    g = torch.Generator().manual_seed(seed)
    f = [torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1) for _ in range(3)]
    w1 = (torch.rand(n, generator=g) > 0.3).float(); w2 = (torch.rand(n, generator=g) > 0.4).float()
    return f[0], f[1], f[2], w1, w2


def run3(ref, rank, world, local_loss, gwg, seed, n, w_override=None):
    a, b, c, w1, w2 = feats3(seed, n)
    if w_override is not None:
        w1, w2 = w_override(w1, w2)
    for t in (a, b, c):
        t.requires_grad_(True)
    ls = [torch.tensor(np.log(1 / t_), dtype=torch.float32, requires_grad=True) for t_ in (0.07, 0.05, 0.1)]
    loss = ref.ThreeModalityClipLoss(local_loss=local_loss, gather_with_grad=gwg, rank=rank, world_size=world)(
        a, b, c, ls[0].exp(), ls[1].exp(), ls[2].exp(), w1, w2)
    loss.backward()
    z = lambda t: (t.grad if t.grad is not None else torch.zeros_like(t)).numpy()
    return {"loss": loss.detach().numpy(), "ga": z(a), "gb": z(b), "gc": z(c), "gls": np.array([float(z(l)) for l in ls]),
            "w1": w1.numpy(), "w2": w2.numpy()}


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
    for local_loss, gwg in ((False, False), (False, True)):
        r = run3(ref, rank, world, local_loss, gwg, 300 + rank, 4)
        for k, v in r.items():
            out[f"w2m3/ll{int(local_loss)}_gg{int(gwg)}/r{rank}/{k}"] = v
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
    cases = {"all": lambda w1, w2: (torch.ones_like(w1), torch.ones_like(w2)), "some": None,
             "none2": lambda w1, w2: (w1, torch.zeros_like(w2))}
    for name, ov in cases.items():
        r = run3(ref, 0, 1, False, False, 55, 7, ov)
        for k, v in r.items():
            save[f"w1m3/{name}/{k}"] = v
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
