"""Diagnostic: the Block's forward / backward stage by stage (the ops BlockFn calls, fed with the ROUNDING-POINT MODEL's tensors at
every stage) against oracle/bf16_points.py -- each stage one rounding point deep, so a stage that disagrees is a kernel problem,
not tie-flip amplification."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops
from oracle import bf16_points as R
D = torch.float64; F = torch.float32; BF = torch.bfloat16
def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
def dev(t, dt=F): return t.to(dt).cuda().contiguous()
C, H, N, B = [int(v) for v in os.environ.get("SHAPE", "256,4,46,3").split(",")]
g = torch.Generator().manual_seed(C + N)
names = [("norm1.weight", (C,)), ("norm1.bias", (C,)), ("attn.q.weight", (C, C)), ("attn.q.bias", (C,)), ("attn.k.weight", (C, C)), ("attn.k.bias", (C,)),
         ("attn.v.weight", (C, C)), ("attn.v.bias", (C,)), ("attn.proj.weight", (C, C)), ("attn.proj.bias", (C,)), ("norm2.weight", (C,)), ("norm2.bias", (C,)),
         ("mlp.fc1.weight", (4 * C, C)), ("mlp.fc1.bias", (4 * C,)), ("mlp.fc2.weight", (C, 4 * C)), ("mlp.fc2.bias", (C,))]
P = {}
for n_, s_ in names:
    if len(s_) > 1:
        a = (6.0 / (s_[0] + s_[1])) ** 0.5; P[n_] = (torch.rand(s_, generator=g) * 2 - 1) * a
    else:
        P[n_] = torch.randn(s_, generator=g) * 0.05 + (1.0 if "norm" in n_ and n_.endswith("weight") else 0.0)
x = torch.randn(B, N, C, generator=g) * 1.5
dx3 = torch.randn(B, N, C, generator=g)
x3, S = R.block_forward(P, x, H, 1e-6)
dx, G = R.block_backward(S, dx3)
p = S["p"]; M = B * N
# ---- forward stages, each fed with the model's input
y1, mean1, rstd1 = ops.layernorm_fwd(dev(x.view(M, C)), dev(P["norm1.weight"]), dev(P["norm1.bias"]), 1e-6)
print("LN1 fwd", rel(y1, S["y1"].view(M, C)))
wqkv = dev(S["wqkv"], BF); bqkv = dev(torch.cat([P["attn.q.bias"], P["attn.k.bias"], P["attn.v.bias"]]))
qkv_m = torch.cat([t.transpose(1, 2).reshape(B, N, C) for t in (S["q"], S["k"], S["v"])], -1).view(M, 3 * C)
qkv = ops.linear_fwd(dev(S["y1"].view(M, C), BF), wqkv, bqkv, "bf16")
print("qkv GEMM", rel(qkv, qkv_m))
x2_m = x.double() + S["o2"] @ S["wp"].T + p["attn.proj.bias"]
x2 = ops.linear_fwd(dev(S["o2"].view(M, C), BF), dev(S["wp"], BF), dev(P["attn.proj.bias"]), "resid", res=dev(x.view(M, C)))
print("proj+resid", rel(x2, x2_m.view(M, C)))
y2, mean2, rstd2 = ops.layernorm_fwd(dev(x2_m.view(M, C)), dev(P["norm2.weight"]), dev(P["norm2.bias"]), 1e-6)
print("LN2 fwd", rel(y2, S["y2"].view(M, C)))
pre, act = ops.linear_fwd(dev(S["y2"].view(M, C), BF), dev(S["w1"], BF), dev(P["mlp.fc1.bias"]), "gelu")
print("fc1 pre / act", rel(pre, S["pre"].view(M, -1)), rel(act, S["act"].view(M, -1)))
# ---- backward stages
d3 = dx3.double(); d3b = R.bf(d3)
gw2 = torch.zeros(C, 4 * C, device="cuda")
ops.linear_wgrad_accum(dev(d3b.view(M, C), BF), dev(S["act"].view(M, -1), BF), gw2)
print("fc2 wgrad", rel(gw2, G["mlp.fc2.weight"]))
gb1 = torch.zeros(4 * C, device="cuda")
dpre = ops.linear_dgrad(dev(d3b.view(M, C), BF), dev(S["w2"], BF), pre=dev(S["pre"].view(M, -1), BF), colsum=gb1)
dpre_m = R.bf(R.bf(d3b @ S["w2"]) * R.dgelu(S["pre"]))
print("fc2 dgrad x gelu'", rel(dpre, dpre_m.view(M, -1)), " fc1 bias grad", rel(gb1, G["mlp.fc1.bias"]))
gw1 = torch.zeros(4 * C, C, device="cuda")
ops.linear_wgrad_accum(dev(dpre_m.view(M, -1), BF), dev(S["y2"].view(M, C), BF), gw1)
print("fc1 wgrad", rel(gw1, G["mlp.fc1.weight"]))
dy2_m = R.bf(dpre_m @ S["w1"])
dy2 = ops.linear_dgrad(dev(dpre_m.view(M, -1), BF), dev(S["w1"], BF))
print("fc1 dgrad", rel(dy2, dy2_m.view(M, C)))
gg2 = torch.zeros(C, device="cuda"); gb2n = torch.zeros(C, device="cuda"); gbproj = torch.zeros(C, device="cuda")
dx2, dx2b = ops.layernorm_bwd(dev(dy2_m.view(M, C), BF), dev(x2_m.view(M, C)), mean2, rstd2, dev(P["norm2.weight"]), gg2, gb2n, dres=dev(d3.view(M, C)), want_bf16=True, dxsum=gbproj)
dln2, gw_n2, gb_n2 = R.ln_bwd(dy2_m, S["xh2"], S["rs2"], p["norm2.weight"])
dx2_m = d3 + dln2
print("LN2 bwd dx2", rel(dx2, dx2_m.view(M, C)), " bf16 copy", rel(dx2b, R.bf(dx2_m).view(M, C)), " gamma", rel(gg2, gw_n2), " beta", rel(gb2n, gb_n2), " proj bias (dx2 colsum)", rel(gbproj, G["attn.proj.bias"]))
dx2b_m = R.bf(dx2_m)
gwp = torch.zeros(C, C, device="cuda")
ops.linear_wgrad_accum(dev(dx2b_m.view(M, C), BF), dev(S["o2"].view(M, C), BF), gwp)
print("proj wgrad", rel(gwp, G["attn.proj.weight"]))
do_m = R.bf(dx2b_m @ S["wp"])
do = ops.linear_dgrad(dev(dx2b_m.view(M, C), BF), dev(S["wp"], BF))
print("proj dgrad", rel(do, do_m.view(M, C)))
