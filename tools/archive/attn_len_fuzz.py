"""Random sequence lengths through the attention entry points: forward (optimistic + safe), fused and two-kernel backward, every form of
the one-wave kernels' block / tail logic (N mod 64, N mod 256 / 512, one key past a block, fewer keys than a block), against fp64.
python tools/attn_len_fuzz.py [cases] [max N]"""
import random
import sys

import torch

from octcubem_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 1700
dev = torch.device("cuda")
rng = random.Random(7)


def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


special = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 258, 288, 289, 511, 512, 513, 514, 544, 545, 575, 576, 577, 767, 768, 769,
           1023, 1024, 1025, 1026, 1056, 1057, 1281, 1535, 1536, 1537, 1538]
worst = {}
for i in range(cases):
    N = special[i] if i < len(special) else rng.randrange(1, nmax)
    for HD in (32, 64):
        B, H = rng.choice([1, 2]), rng.choice([1, 2, 3])
        g = torch.Generator(device=dev).manual_seed(N * 7 + HD)
        qkv = torch.randn(B * N, 3 * H * HD, device=dev, generator=g).bfloat16()
        do = torch.randn(B * N, H * HD, device=dev, generator=g).bfloat16()
        qd = qkv.double().requires_grad_(True)
        q, k, v = qd.view(B, N, 3, H, HD).permute(2, 0, 3, 1, 4)
        s = (q @ k.transpose(-2, -1)) * HD ** -0.5
        o_ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * N, H * HD)
        o_ref.backward(do.double())
        o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
        o2, lse2 = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5, optimistic=False)
        e = {"fwd": rel(o, o_ref), "fwd_safe": rel(o2, o_ref)}
        for fused in (True, False):
            dq = ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5, fused=fused)
            assert torch.isfinite(dq).all(), (N, HD, fused)
            e["bwd_fused" if fused else "bwd_pair"] = rel(dq, qd.grad)
        for kk, vv in e.items():
            if vv > worst.get((kk, HD), (0, 0))[0]:
                worst[(kk, HD)] = (vv, N)
        assert max(e.values()) < 1.5e-2, (N, HD, B, H, e)
print(f"{cases} lengths x 2 head dims: worst relative errors against fp64 (at N):")
for kk, vv in sorted(worst.items()):
    print(f"  {kk[0]:10s} hd {kk[1]}: {vv[0]:.2e}  (N = {vv[1]})")
