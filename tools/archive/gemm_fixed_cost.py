"""Per-tile fixed cost of the 256-tile GEMM (prologue + epilogue) from the K dependence of the time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops
def t(f, iters=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
M, N = 40960, 3072
for two in (False, True):
    ops.FORCE_TWO_STAGE = two
    for mode in ("bf16", "gelu", "resid"):
        row = []
        for K in (64, 256, 1024, 4096):
            x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
            us = t(lambda: ops.linear_fwd(x, w, b, mode, res=res if mode == "resid" else None))
            row.append(f"K={K}: {us:7.1f} us")
        tiles = (M // 256) * (N // 256)
        print(("two-stage" if two else "phased   "), mode, " | ".join(row), f"| tiles {tiles} = {tiles / 256:.2f} rounds")
