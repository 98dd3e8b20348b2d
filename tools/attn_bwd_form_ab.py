"""Same-process A/B of the two main kernels behind octmae_attn_bwd_fused (octmae_set_option "attn_bwd_hd32_form" /
"attn_bwd_hd64_form": 1 = one wave per SIMD, csrc/attn_bwd1w.hip / attn_bwd1w64.hip; 0 = two waves per SIMD, csrc/attn_bwd.hip)
at the decoder shape (HD=32, N=5121: default) or the encoder's (HD=64 N=1281); interleaved rounds on random data, medians.
B=<micro-batch> (default 32)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops

dev = "cuda"
B = int(os.environ.get("B", "32"))
HD = int(os.environ.get("HD", "32"))
H, N = 16, int(os.environ.get("N", "5121" if HD == 32 else "1281"))
KEY = f"attn_bwd_hd{HD}_form"
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B * N, 3 * H * HD, device=dev, generator=g).to(torch.bfloat16)
do = torch.randn(B * N, H * HD, device=dev, generator=g).to(torch.bfloat16)
scale = HD ** -0.5
o, lse = ops.attn_fwd(qkv, B, N, H, HD, scale)
res = {0: [], 1: []}
outs = {}
for r in range(int(os.environ.get("ROUNDS", "9"))):
    for form in (1, 0):
        ops.set_option(KEY, form)
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        d = ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, scale, fused=True)
        e.record(); torch.cuda.synchronize()
        res[form].append(s.elapsed_time(e))
        outs[form] = d
ops.set_option(KEY, 1)
unit = 2.0 * B * H * N * N * HD
for form in (1, 0):
    ms = statistics.median(res[form][1:]); mn = min(res[form][1:])
    print(f"B={B} H={H} N={N} HD={HD} form {form} ({'1 wave/SIMD' if form else '2 waves/SIMD'}): median {ms:8.3f} ms  min {mn:8.3f} ms  "
          f"algorithmic {4 * unit / ms / 1e9:7.1f} TF/s ({4 * unit / ms / 1e9 / 2500:.3f} of peak)", flush=True)
a, b = outs[1].double(), outs[0].double()
print("forms agree to rel-L2", float((a - b).norm() / b.norm()))
