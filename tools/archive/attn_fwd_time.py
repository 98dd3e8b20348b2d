"""Times octmae_attn_fwd (the library OCTMAE_LIB selects) at the decoder shape (HD=32, default) or the encoder's (HD=64); one line,
plus the rel-L2 distance of O to a float64 reference on the same bf16 inputs for the first (batch, head)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops
B = int(os.environ.get("B", "128")); HD = int(os.environ.get("HD", "32")); H, N = 16, int(os.environ.get("N", "5121" if HD == 32 else "1281"))
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
ts = []
for r in range(9):
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
q = qkv.view(B, N, 3, H, HD)[0, :, :, 0].double()
ref = torch.softmax(q[:, 0] @ q[:, 1].T * HD ** -0.5, -1) @ q[:, 2]
got = o.view(B, N, H, HD)[0, :, 0].double()
err = float((got - ref).norm() / ref.norm())
ms = statistics.median(ts[1:])
print(f"{os.path.basename(os.environ.get('OCTMAE_LIB', 'in-tree')):32s} B={B} HD={HD} N={N} median {ms:7.3f} ms  min {min(ts[1:]):7.3f} ms  "
      f"{4.0 * B * H * N * N * HD / ms / 1e9:7.1f} TF/s  rel-L2 vs fp64 {err:.2e}", flush=True)
