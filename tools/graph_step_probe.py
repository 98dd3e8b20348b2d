"""Probe: capture forward + backward of the ViT-L 3-D MAE step in a HIP graph (torch.cuda.CUDAGraph over the ctypes-launched kernels)
and replay it: same loss / gradients as the eager step?  how much faster at 1, 2, 4, 8 volumes (the eager step is host-bound there:
~240 launches need ~23 ms to enqueue)?   python tools/graph_step_probe.py [batches...]"""
import faulthandler
import sys
import time

import torch

faulthandler.enable()

sys.path.insert(0, ".")
from octcubem_amd import models_mae, misc, ops, optim as foptim  # noqa: E402

dev = torch.device("cuda", 0)
batches = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
torch.manual_seed(0)
model = models_mae.octcube_vit_large_3dmae().to(dev).train()
model.prepare()
opt = foptim.FusedAdamW(misc.add_weight_decay(model, 0.05), lr=1e-4, betas=(0.9, 0.95))
scaler = misc.NativeScalerWithGradNormCount()
params = list(model.parameters())


def eager_fb(x, noise):
    loss, _, _ = model(x, mask_ratio=0.75, noise=noise)
    loss.backward()
    return loss


for B in batches:
    x = torch.rand(B, 1, 60, 256, 256, device=dev)
    noise = torch.rand(B, 5120, device=dev)
    # capture FIRST (the AccumulateGrad nodes of the PyTorch-side parameters are then created on the capture stream), eager reference after
    sx, sn = x.clone(), noise.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            opt.zero_grad()
            eager_fb(sx, sn)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad()
    try:
        with torch.cuda.graph(g):
            sl = eager_fb(sx, sn)
    except Exception as e:
        print(f"B={B}: capture FAILED: {type(e).__name__}: {e}")
        continue
    opt.zero_grad()
    g.replay()
    torch.cuda.synchronize()
    l_g, g_g = sl.detach().clone(), model.arena.grad.clone()
    opt.zero_grad()
    l_e = eager_fb(x, noise).detach().clone()
    g_e = model.arena.grad.clone()
    torch.cuda.synchronize()
    same_loss = torch.equal(l_g, l_e)
    dg = float((g_g.double() - g_e.double()).norm() / g_e.double().norm())
    # timing: eager step vs replay + eager optimizer
    def step_eager():
        opt.zero_grad()
        loss, _, _ = model(x, mask_ratio=0.75, noise=noise)
        scaler(loss, opt, parameters=params)

    def step_graph():
        opt.zero_grad()
        g.replay()
        opt.step()

    res = {}
    for name, fn in (("eager", step_eager), ("graph", step_graph)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B}: loss bit-equal {same_loss}, gradient rel diff {dg:.2e}; eager {res['eager']:.2f} ms/step ({B / res['eager'] * 1e3:.1f} vol/s), "
          f"graph replay + eager AdamW {res['graph']:.2f} ms/step ({B / res['graph'] * 1e3:.1f} vol/s)", flush=True)
    del g
