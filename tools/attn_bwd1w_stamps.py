"""Diagnostic (OCTMAE_LIB=build_ab/liboctmae_stamp.so, `make -C octcubem_amd/csrc stamp`): cycles per interval of a sub-step of the
one-wave-per-SIMD attention backward, per wave, key block 1, averaged over the tiles (median over workgroups).  HD=32 (default,
N=5121) or HD=64 (N=1281); B=<micro-batch>."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops, _lib
lib = _lib.load()
B = int(os.environ.get("B", "32")); HD = int(os.environ.get("HD", "32")); H, N = 16, (5121 if HD == 32 else 1281)
NI = 10 if HD == 32 else 8
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(torch.bfloat16)
ops.set_option(f"attn_bwd_hd{HD}_form", 1)
o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
for _ in range(2):
    ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5, fused=True)
torch.cuda.synchronize()
buf = np.zeros(512 * 4 * 2 * NI, dtype=np.uint32)
fn = lib.octmae_debug_bwd1w_stamps if HD == 32 else lib.octmae_debug_bwd1w64_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
a = buf.reshape(512, 4, 2, NI)[: min(B * H, 512)].astype(np.float64) / ((N + 63) // 64 + 1)
names = ["g0 first half", "g0 second half", "g1 first half", "g1 second half", "wait + barrier", "loop overhead (to next g0)", "  of it: vmcnt wait", "-"] if HD == 64 else ["g0 first half", "g0 second half", "g1 first half", "g1 second half", "g2 first half", "g2 second half", "g3 first half",
         "g3 second half", "wait + barrier", "loop overhead (to next g0)"]
print("cycles per sub-step interval (median over workgroups), waves 0..3; sub-step 0 | sub-step 1")
for k, n in enumerate(names):
    print(f"  {n:28s} " + " ".join(f"{np.median(a[:, w, 0, k]):6.0f}" for w in range(4)) + "   |  " + " ".join(f"{np.median(a[:, w, 1, k]):6.0f}" for w in range(4)))
tot = a[..., :(10 if HD == 32 else 6)].sum(-1)
print(f"  {'total per sub-step':28s} " + " ".join(f"{np.median(tot[:, w, 0]):6.0f}" for w in range(4)) + "   |  " + " ".join(f"{np.median(tot[:, w, 1]):6.0f}" for w in range(4)))
if HD == 64:
    raw = buf.reshape(512, 4, 2, NI)[: min(B * H, 512)].astype(np.float64)
    core, real = np.median(raw[:, :, 0, 7]), np.median(raw[:, :, 1, 7])
    print(f"  whole kernel, per wave: {core:.0f} shader-clock ticks in {real:.0f} ticks of the 100 MHz clock = {real / 100:.1f} us -> {core / real * 100 / 1000:.3f} GHz")
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); s_ = torch.cuda.Event(enable_timing=True); e_ = torch.cuda.Event(enable_timing=True)
        s_.record(); ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5, fused=True); e_.record(); torch.cuda.synchronize()
        ts.append(s_.elapsed_time(e_))
    print(f"  this build's op time: {min(ts):.3f} ms")
