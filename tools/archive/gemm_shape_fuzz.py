"""Random shapes through every GEMM entry the step uses -- forward (bf16 / fp32 / GELU / residual), dgrad (plain, x GELU' with column
sums), weight gradient (single, with bias gradient, paired) -- against fp64: tile edges, ragged k-tiles, split choices, small-tile
fallbacks.   python tools/gemm_shape_fuzz.py [cases]"""
import random
import sys

import torch

from octcubem_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda")
rng = random.Random(3)


def rel(a, b):
    a = a.detach().double().flatten(); b = b.detach().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def dims():
    pick = lambda: rng.choice([8, 16, 64, 120, 248, 256, 264, 504, 512, 520, 768, 1024, 1032, 8 * rng.randrange(1, 260)])
    return rng.choice([1, 7, 63, 64, 65, 255, 256, 257, 300, 511, 513, 1000, rng.randrange(1, 3000)]), pick(), pick()


worst = {}
for i in range(cases):
    M, N, K = dims()
    g = torch.Generator(device=dev).manual_seed(i)
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g)
    dy = torch.randn(M, N, device=dev, generator=g).bfloat16()
    pre = torch.randn(M, K, device=dev, generator=g).bfloat16()
    ref = x.double() @ w.double().t() + b.double()
    e = {}
    e["fwd_bf16"] = rel(ops.linear_fwd(x, w, b, "bf16"), ref)
    if N % 4 == 0:
        e["fwd_f32"] = rel(ops.linear_fwd(x, w, b, "f32"), ref)
        p2, a2 = ops.linear_fwd(x, w, b, "gelu")
        e["fwd_gelu"] = max(rel(p2, ref), rel(a2, torch.nn.functional.gelu(ref)))
        e["fwd_resid"] = rel(ops.linear_fwd(x, w, b, "resid", res=res), ref + res.double())
    dref = dy.double() @ w.double()
    e["dgrad"] = rel(ops.linear_dgrad(dy, w), dref)
    xg = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(xg).backward(dref)
    cs = torch.zeros(K, device=dev)
    dxg = ops.linear_dgrad(dy, w, pre=pre, colsum=cs)
    e["dgrad_dgelu"] = rel(dxg, xg.grad)
    e["dgelu_colsum"] = float((cs.double() - dxg.double().sum(0)).norm() / (dxg.double().sum(0).norm() + 1e-6 * dxg.double().norm() + 1e-30))
    gw0 = torch.randn(N, K, device=dev, generator=g); gb0 = torch.randn(N, device=dev, generator=g)
    gw, gb = gw0.clone(), gb0.clone()
    ops.linear_wgrad_accum(dy, x, gw, gb)
    wref = gw0.double() + dy.double().t() @ x.double()
    e["wgrad"] = rel(gw, wref); e["wgrad_bias"] = rel(gb, gb0.double() + dy.double().sum(0))
    # pair: this problem and its transpose-shaped sibling over the same rows
    gwa, gwb = gw0.clone(), torch.zeros(K, N, device=dev)
    ops.linear_wgrad_accum_pair((dy, x, gwa, None), (x, dy, gwb, None))
    e["pair"] = max(rel(gwa, wref), rel(gwb, x.double().t() @ dy.double()))
    tol = {"fwd_bf16": 4e-3, "fwd_f32": 1e-5, "fwd_gelu": 6e-3, "fwd_resid": 4e-3, "dgrad": 4e-3, "dgrad_dgelu": 6e-3, "dgelu_colsum": 2e-4, "wgrad": 1e-5,
           "wgrad_bias": 1e-4, "pair": 1e-5}
    for kk, vv in e.items():
        if vv > worst.get(kk, (0,))[0]:
            worst[kk] = (vv, (M, N, K))
        assert vv < tol[kk], (kk, vv, (M, N, K))
print(f"{cases} random shapes: worst relative errors against fp64 (at M, N, K):")
for kk, vv in sorted(worst.items()):
    print(f"  {kk:13s} {vv[0]:.2e}  {vv[1]}")
