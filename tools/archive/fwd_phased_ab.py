"""Forward GEMMs: two-stage vs phased main loop of the 256-tile kernel, same process, interleaved (micro-batch 128 shapes)."""
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
B = 128
for (M, K, N, mode) in ((B*1281,1024,3072,"bf16"),(B*1281,1024,4096,"gelu"),(B*1281,1024,1024,"resid"),(B*1281,4096,1024,"resid"),(B*5121,512,1536,"bf16"),(B*5121,512,2048,"gelu"),(B*5121,2048,512,"resid")):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
    fl = 2.0 * M * K * N
    out = []
    for rep in range(2):
        for ph in (False, True):
            ops.FORCE_PHASED = ph
            v = t(lambda: ops.linear_fwd(x, w, b, mode, res=res if mode == "resid" else None))
            out.append(f"{'phased' if ph else '2stage'} {v:7.1f} us ({fl/v/1e6:5.0f})")
    ops.FORCE_PHASED = False
    print(f"[{M}x{K}]x[{N}x{K}] {mode}: " + " | ".join(out), flush=True)
