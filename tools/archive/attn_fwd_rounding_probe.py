"""Diagnostic: which float64 + bf16-rounding model reproduces octmae_attn_fwd's output for short and long sequences."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops
D = torch.float64
def bf(x): return x.to(torch.float32).to(torch.bfloat16).to(D)
LOG2E = 1.4426950408889634
for HD, N in ((64, 46), (64, 13), (64, 300), (64, 64), (64, 65), (32, 181), (32, 46), (32, 300), (64, 1281), (32, 5121)):
    B, H = 2, 3
    g = torch.Generator().manual_seed(N)
    qkv = (torch.randn(B * N, 3 * H * HD, generator=g) * 1.2).to(torch.bfloat16)
    o, lse = ops.attn_fwd(qkv.cuda(), B, N, H, HD, HD ** -0.5)
    q, k, v = qkv.to(D).view(B, N, 3, H, HD).permute(2, 0, 3, 1, 4)
    sc2 = torch.tensor(HD ** -0.5, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)
    qs = (q.to(torch.float32) * sc2).to(torch.bfloat16).to(D)
    S2 = qs @ k.transpose(-1, -2)
    res = {}
    for name, mx in (("no max", None), ("row max", S2.amax(-1, keepdim=True))):
        Sx = S2 if mx is None else S2 - mx
        Pm = torch.exp2(Sx)
        for lname, l in (("l=sum P", Pm.sum(-1, keepdim=True)), ("l=sum bf(P)", bf(Pm).sum(-1, keepdim=True))):
            oo = bf((bf(Pm) @ v) / l).transpose(1, 2).reshape(B * N, H * HD)
            res[f"{name}, {lname}"] = float((o.double().cpu() - oo).norm() / oo.norm())
    print(f"HD={HD} N={N}: " + "; ".join(f"{k_}: {v_:.2e}" for k_, v_ in res.items()))
