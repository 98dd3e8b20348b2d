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
out = [os.path.basename(os.environ.get("OCTMAE_LIB", "default"))]
for (M, K, N, tag) in ((81984, 1024, 3072, "enc qkv"), (81984, 1024, 4096, "enc fc1"), (327744, 512, 2048, "dec fc1"), (81984, 4096, 1024, "enc fc2")):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); gw = torch.zeros(N, K, device="cuda")
    out.append(f"{tag}: dgrad {t(lambda: ops.linear_dgrad(dy, w)):6.1f} wgrad {t(lambda: ops.linear_wgrad_accum(dy, x, gw)):6.1f}")
print(" | ".join(out))
