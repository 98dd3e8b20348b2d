"""Per-CU timeline of the two-stage 256-tile GEMM from a -DGEMM_TRACE build (build_ab/liboctmae_trace.so):
prologue / k-loop / epilogue per workgroup and the idle gap between consecutive workgroups on one CU.
    OCTMAE_LIB=build_ab/liboctmae_trace.so python tools/gemm_trace.py [mode] [M N K]"""
import ctypes, os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (81984, 3072, 1024)
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.octmae_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda") if mode == "resid" else None
for _ in range(3): ops.linear_fwd(x, w, b, mode, res=res)
nblk = ((M + 255) // 256) * ((N + 255) // 256)
tr = torch.zeros(nblk, 6, dtype=torch.int64, device="cuda")
assert raw.octmae_debug_set_gemm_trace(tr.data_ptr()) == 0
torch.cuda.synchronize()
ops.linear_fwd(x, w, b, mode, res=res)
torch.cuda.synchronize()
raw.octmae_debug_set_gemm_trace(None)
t = tr.cpu().numpy().astype(np.int64)
t0 = t[:, 0].min()
us = lambda v: v / 100.0     # 100 MHz
print(f"{mode} [{M}x{K}]x[{N}x{K}]: {nblk} workgroups, kernel span {us(t[:, 3].max() - t0):.1f} us")
pro, loop, epi = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2])
for name, v in (("prologue", pro), ("k-loop", loop), ("epilogue", epi)):
    print(f"  {name:9s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
cu = collections.defaultdict(list)
for i in range(nblk):
    cu[(int(t[i, 4]) >> 32, (int(t[i, 4]) >> 8) & 0xff)].append((t[i, 0], t[i, 3]))
gaps = []
for k, v in cu.items():
    v.sort()
    gaps += [us(v[i + 1][0] - v[i][1]) for i in range(len(v) - 1)]
gaps = np.array(gaps)
print(f"  CUs seen {len(cu)}, workgroups per CU {nblk / len(cu):.2f}")
print(f"  gap between workgroups on a CU: mean {gaps.mean():6.2f}  p10 {np.percentile(gaps, 10):6.2f}  p50 {np.percentile(gaps, 50):6.2f}  p90 {np.percentile(gaps, 90):6.2f} us")
first = np.array([v[0][0] for v in cu.values()]); last = np.array([v[-1][1] for v in cu.values()])
print(f"  first start spread {us(first.max() - first.min()):.2f} us, last end spread {us(last.max() - last.min()):.2f} us")
