"""Diagnostic (BWD_STAMP build via OCTMAE_LIB): where a wave of the fused attention backward spends its cycles per tile."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops, _lib
lib = _lib.load()
B = int(os.environ.get("B", "16"))
for (N, H, HD) in ((5121, 16, 32), (5121, 16, 64)):
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(torch.bfloat16)
    o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
    for _ in range(2):
        ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5, fused=True)
    torch.cuda.synchronize()
    buf = np.zeros(512 * 8 * 8, dtype=np.uint64)
    rc = lib.octmae_debug_bwd_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    assert rc == 0
    a = buf.reshape(512, 8, 8)[: min(B * H, 512), :, :6].astype(np.float64)
    ntiles = (N + 63) // 64
    per = a / ntiles
    names = ["finish+issue", "phase1 units", "lgkm+barrier A", "phase2 dQ", "wait_vm(t+1)", "barrier B"]
    print(f"HD={HD} N={N}: cycles per tile per wave (median over workgroups; waves 0-3 | 4-7), key block 0")
    for i, n in enumerate(names):
        lo = np.median(per[:, :4, i]); hi = np.median(per[:, 4:, i])
        print(f"   {n:16s} {lo:8.0f} {hi:8.0f}")
    print(f"   {'total':16s} {np.median(per[:, :4, :].sum(-1)):8.0f} {np.median(per[:, 4:, :].sum(-1)):8.0f}")
