"""A few launches of the fused attention backward at the step's shapes (profiling target for rocprofv3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops
B = int(os.environ.get("B", "32"))
fused = os.environ.get("FUSED", "1") == "1"
if "FORM" in os.environ:      # 0: the round-2 two-waves-per-SIMD main kernels
    for hd in (32, 64):
        ops.set_option(f"attn_bwd_hd{hd}_form", int(os.environ["FORM"]))
for (N, H, HD) in ((5121, 16, 32), (1281, 16, 64)):
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(torch.bfloat16)
    o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
    for _ in range(3):
        ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5, fused=fused)
    torch.cuda.synchronize()
