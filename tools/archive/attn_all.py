"""Time every attention kernel on the two ViT-L shapes (variant builds are A/B-ed through OCTMAE_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops
def t(f, iters=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
out = [os.path.basename(os.environ.get("OCTMAE_LIB", "default"))]
for (B, N, H, HD) in ((32, 1281, 16, 64), (32, 5121, 16, 32)):
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(torch.bfloat16)
    sc = HD ** -0.5
    o, lse = ops.attn_fwd(qkv, B, N, H, HD, sc)
    out.append(f"hd{HD}: fwd {t(lambda: ops.attn_fwd(qkv, B, N, H, HD, sc)):7.1f}  bwd {t(lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, sc)):7.1f} us")
print(" | ".join(out))
