"""Forward GEMM epilogue costs by difference, same process: plain bf16 / fp32 / GELU (two outputs) / fp32 residual at the ViT-L shapes."""
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
B = int(os.environ.get("B", "128"))
for (M, K, N) in ((B * 1281, 1024, 4096), (B * 1281, 1024, 1024), (B * 1281, 4096, 1024), (B * 5121, 512, 2048), (B * 5121, 512, 512), (B * 5121, 2048, 512)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
    fl = 2.0 * M * K * N
    r = {m: t(lambda: ops.linear_fwd(x, w, b, m, res=res if m == "resid" else None)) for m in ("bf16", "f32", "gelu", "resid")}
    print(f"[{M}x{K}]x[{N}x{K}]: " + "   ".join(f"{m} {v:7.1f} us ({fl / v / 1e6:5.0f} TF/s)" for m, v in r.items()), flush=True)
