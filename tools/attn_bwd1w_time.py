"""Times the fused attention backward (current octmae_set_option form) at the decoder shape (HD=32, default) or the encoder's
(HD=64); one line.  Used with
OCTMAE_LIB=<variant library> by tools/abl_attn_bwd1w.sh (timing-only ablation builds: their results are wrong by design)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops
B = int(os.environ.get("B", "32")); HD = int(os.environ.get("HD", "32")); H, N = 16, int(os.environ.get("N", "5121" if HD == 32 else "1281"))
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(torch.bfloat16)
if "FORM" in os.environ:
    ops.set_option(f"attn_bwd_hd{HD}_form", int(os.environ["FORM"]))
o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
ts = []
for r in range(7):
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5, fused=True); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
print(f"{os.path.basename(os.environ.get('OCTMAE_LIB', 'in-tree')):40s} B={B} HD={HD} N={N} median {statistics.median(ts[1:]):7.3f} ms  min {min(ts[1:]):7.3f} ms", flush=True)
