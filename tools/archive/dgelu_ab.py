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
for (M, C, Hd) in ((64 * 1281, 1024, 4096), (64 * 5121, 512, 2048)):
    dy = torch.randn(M, C, device="cuda").to(torch.bfloat16); w = (torch.randn(C, Hd, device="cuda") * 0.02).to(torch.bfloat16)
    pre = torch.randn(M, Hd, device="cuda").to(torch.bfloat16); gb = torch.zeros(Hd, device="cuda")
    def sep():
        d = ops.linear_dgrad(dy, w, pre=pre); ops.colsum_accum(d, gb)
    def fused():
        ops.linear_dgrad(dy, w, pre=pre, colsum=gb)
    for rep in range(3):
        print(f"M={M} C={C}: separate {t(sep):8.1f} us   fused {t(fused):8.1f} us   gemm alone {t(lambda: ops.linear_dgrad(dy, w, pre=pre)):8.1f} us")
