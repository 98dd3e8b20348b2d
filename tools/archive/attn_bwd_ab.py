"""Same-process A/B of the attention backward forms (two-kernel vs fused) on the step's shapes; interleaved rounds, medians."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops

dev = "cuda"
B = int(os.environ.get("B", "32"))
shapes = [(B, 16, 5121, 32), (B, 16, 1281, 64), (max(B // 2, 1), 16, 5121, 64)]
for (b, H, N, HD) in shapes:
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(b * N, 3 * H * HD, device=dev, generator=g).to(torch.bfloat16)
    do = torch.randn(b * N, H * HD, device=dev, generator=g).to(torch.bfloat16)
    scale = HD ** -0.5
    o, lse = ops.attn_fwd(qkv, b, N, H, HD, scale)
    res = {"fused": [], "two": []}
    for r in range(7):
        for name, f in (("fused", True), ("two", False)):
            torch.cuda.synchronize()
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            d = ops.attn_bwd(qkv, o, do, lse, b, N, H, HD, scale, fused=f)
            e.record(); torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e))
    unit = 2.0 * b * H * N * N * HD
    for name in res:
        ms = statistics.median(res[name][1:])
        print(f"B={b} H={H} N={N} HD={HD} {name:6s}: {ms:8.3f} ms  algorithmic {4 * unit / ms / 1e9:7.1f} TF/s ({4 * unit / ms / 1e9 / 2500:.3f} of peak)", flush=True)
