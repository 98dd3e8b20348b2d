"""Times the three attention launches separately (forward, dQ, dK/dV) at the two ViT-L shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = _lib.load()
def t(f, iters=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
SHAPES = (("enc", 1281, 16, 64), ("dec", 5121, 16, 32))
if len(sys.argv) > 2:      # extra shapes: N:H:HD ...
    SHAPES = tuple((f"n{a.split(':')[0]}", *(int(v) for v in a.split(":"))) for a in sys.argv[2:])
for name, N, H, HD in SHAPES:
    qkv = torch.randn(B * N, 3 * H * HD, device="cuda").to(torch.bfloat16); do = torch.randn(B * N, H * HD, device="cuda").to(torch.bfloat16)
    o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
    rowc = torch.empty(2 * B * H * N, dtype=torch.float32, device="cuda"); dqkv = torch.empty_like(qkv)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.octmae_attn_bwd_rowconst(o.data_ptr(), do.data_ptr(), lse.data_ptr(), rowc.data_ptr(), B, N, H, HD, st) == 0
    unit = 2.0 * B * H * N * N * HD
    tf = t(lambda: ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5))
    tq = t(lambda: lib.octmae_attn_bwd_dq_rowconst(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), rowc.data_ptr(), dqkv.data_ptr(),
                                                   B, N, H, HD, HD ** -0.5, st))   # dQ + the row constants
    tk = t(lambda: lib.octmae_attn_bwd_dkv(qkv.data_ptr(), do.data_ptr(), rowc.data_ptr(), dqkv.data_ptr(), B, N, H, HD, HD ** -0.5, st))
    print(f"{name} hd{HD} N={N} B={B}: fwd {tf:8.1f} us {2*unit/tf/1e6:7.1f} TF | dq {tq:8.1f} us {3*unit/tq/1e6:7.1f} TF | dkv {tk:8.1f} us {4*unit/tk/1e6:7.1f} TF", flush=True)
