"""Diagnostic: octmae_attn_bwd_fused / octmae_attn_bwd against the float64 + bf16-rounding model of oracle/bf16_points.py."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops
D = torch.float64
F = torch.float32
def bf(x): return x.to(F).to(torch.bfloat16).to(D)
LOG2E = 1.4426950408889634
def rel(a, b): return float((a.double().cpu() - b).norm() / (b.norm() + 1e-30))
for (B, H, N, HD) in ((3, 4, 46, 64), (2, 4, 46, 64), (3, 4, 181, 32), (2, 4, 197, 32), (2, 2, 300, 64), (3, 2, 300, 64), (3, 4, 600, 32), (1, 4, 46, 64)):
    g = torch.Generator().manual_seed(N + B)
    qkv = (torch.randn(B * N, 3 * H * HD, generator=g) * 1.2).to(torch.bfloat16)
    do = (torch.randn(B * N, H * HD, generator=g)).to(torch.bfloat16)
    scale = HD ** -0.5
    o, lse = ops.attn_fwd(qkv.cuda(), B, N, H, HD, scale)
    q, k, v = qkv.to(D).view(B, N, 3, H, HD).permute(2, 0, 3, 1, 4)
    dod = do.to(D).view(B, N, H, HD).transpose(1, 2)
    od = o.double().cpu().view(B, N, H, HD).transpose(1, 2)
    lsed = lse.double().cpu()                      # [B, H, N]
    sc2 = torch.tensor(scale, dtype=F) * torch.tensor(LOG2E, dtype=F)
    delta = (dod * od).sum(-1, keepdim=True).to(F).to(D)
    nl = (-(lsed.to(F) * torch.tensor(LOG2E, dtype=F))).to(D).view(B, H, N, 1)
    ks = (k.to(F) * sc2).to(torch.bfloat16).to(D)
    qs = (q.to(F) * sc2).to(torch.bfloat16).to(D)
    Pk = torch.exp2(q @ ks.transpose(-1, -2) + nl)
    dPk = dod @ v.transpose(-1, -2) - delta
    dSk = Pk * dPk
    dV = bf(Pk).transpose(-1, -2) @ dod
    dK = scale * (bf(dSk).transpose(-1, -2) @ q)
    dQf = scale * (bf(dSk) @ k)
    Pq = torch.exp2(qs @ k.transpose(-1, -2) + nl)
    dQp = scale * (bf(Pq * dPk) @ k)
    out = {}
    for fused in (True, False):
        d = ops.attn_bwd(qkv.cuda(), o, do.cuda(), lse, B, N, H, HD, scale, fused=fused).double().cpu().view(B, N, 3, H, HD).permute(2, 0, 3, 1, 4)
        dQ = dQf if fused else dQp
        out[fused] = (rel(d[0], bf(dQ)), rel(d[1], bf(dK)), rel(d[2], bf(dV)))
    print(f"B={B} H={H} N={N} HD={HD}: fused dq/dk/dv " + " ".join(f"{x:.2e}" for x in out[True]) + "   two-kernel " + " ".join(f"{x:.2e}" for x in out[False]))
