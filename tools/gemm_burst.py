import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops
def t(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
N = 3072
for K in (64, 1024):
    for M in (256 * 21, 256 * 10, 256 * 5, 256 * 2):
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        print(f"K={K} tiles={(M // 256) * 12:4d}: bf16 {t(lambda: ops.linear_fwd(x, w, b, 'bf16')):6.1f} us   f32 {t(lambda: ops.linear_fwd(x, w, b, 'f32')):6.1f} us")
