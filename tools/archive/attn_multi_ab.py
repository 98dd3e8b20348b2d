"""Same-process, interleaved A/B of attention entry points across SEVERAL builds of liboctmae (cdna_hip_programming.md rule 24:
perf deltas come from interleaved rounds in one process).  Usage (GPU box, repo root):
    python tools/attn_multi_ab.py [bwd|fwd] <HD> <B> <N> <rounds> <lib.so> [<lib.so> ...]      (the in-tree build is always arm 0)
Prints median / min per arm and, for bwd, whether each arm's dqkv equals arm 0's bit for bit (ablation builds will not)."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import _lib, ops

what, HD, B, N, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
paths = [_lib.LIB_PATH] + [os.path.abspath(p) for p in sys.argv[6:]]
H = 16
libs = []
for p in paths:
    lib = C.CDLL(p)
    for name, at in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = at; fn.restype = C.c_int
    libs.append(lib)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(torch.bfloat16)
if os.environ.get("ZERO") == "1":      # all-zero operands: the same instruction stream at minimal switching power (DVFS probe)
    qkv.zero_(); do.zero_()
scale = HD ** -0.5
o, lse = ops.attn_fwd(qkv, B, N, H, HD, scale)
kib = libs[0].octmae_attn_bwd_fused_ws_kib(B, N, H, HD)
ws = torch.empty((kib * 256,), dtype=torch.float32, device="cuda")
outs = [torch.empty_like(qkv) for _ in libs]
o2 = torch.empty_like(o); lse2 = torch.empty_like(lse); flag = torch.zeros(1, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def run(i):
    if what == "bwd":
        rc = libs[i].octmae_attn_bwd_fused(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), ws.data_ptr(), outs[i].data_ptr(),
                                           B, N, H, HD, scale, st)
    else:
        rc = libs[i].octmae_attn_fwd(qkv.data_ptr(), o2.data_ptr(), lse2.data_ptr(), flag.data_ptr(), B, N, H, HD, scale, st)
    assert rc == 0, rc


ts = [[] for _ in libs]
for r in range(R + 1):
    for i in range(len(libs)):
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); run(i); e.record(); torch.cuda.synchronize()
        if r > 0:
            ts[i].append(s.elapsed_time(e))
flop = (8.0 if what == "bwd" else 4.0) * B * H * N * N * HD
for i, p in enumerate(paths):
    med = statistics.median(ts[i])
    same = "" if what != "bwd" or i == 0 else ("  == arm 0" if torch.equal(outs[i], outs[0]) else "  != arm 0")
    print(f"{what} hd{HD} B{B} N{N} {os.path.basename(p):34s} median {med:8.3f} ms  min {min(ts[i]):8.3f} ms  {flop / med / 1e9:7.1f} TFLOP/s{same}", flush=True)
