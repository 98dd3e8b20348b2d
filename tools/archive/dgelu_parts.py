"""fc2 dgrad: plain bf16 epilogue vs x GELU' vs x GELU' + fc1 bias gradient, same process."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from octcubem_amd import ops
def t(f, iters=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (M, C, Hd) in ((128 * 1281, 1024, 4096), (128 * 5121, 512, 2048)):
    dy = torch.randn(M, C, device="cuda").to(torch.bfloat16); w = (torch.randn(C, Hd, device="cuda") * 0.02).to(torch.bfloat16)
    pre = torch.randn(M, Hd, device="cuda").to(torch.bfloat16); gb = torch.zeros(Hd, device="cuda")
    fl = 2.0 * M * C * Hd
    for rep in range(2):
        a = t(lambda: ops.linear_dgrad(dy, w)); b = t(lambda: ops.linear_dgrad(dy, w, pre=pre)); c = t(lambda: ops.linear_dgrad(dy, w, pre=pre, colsum=gb))
        print(f"M={M} C={C} Hd={Hd}: plain bf16 {a:8.1f} us ({fl/a/1e6:6.0f} TF/s)   x gelu' {b:8.1f} us ({fl/b/1e6:6.0f})   x gelu' + colsum {c:8.1f} us ({fl/c/1e6:6.0f})")
