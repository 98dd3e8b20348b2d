import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops
B, N, H, HD = 8, 5121, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 32
if HD == 64: N = 1281; B = 32
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(torch.bfloat16)
for _ in range(3):
    o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
    ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5)
torch.cuda.synchronize()
