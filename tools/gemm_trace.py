import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OCTMAE_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboctmae_gtrace.so")
from octcubem_amd import ops, _lib
M, N, K = 40960, 3072, 1024
x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16); b = torch.randn(N, device="cuda")
for mode in ("bf16", "gelu"):
    for _ in range(3): ops.linear_fwd(x, w, b, mode)
    torch.cuda.synchronize()
    lib = _lib.load(); buf = np.zeros((4096, 4), dtype=np.int64)
    lib.octmae_debug_gemm_trace.argtypes = [ctypes.c_void_p]; lib.octmae_debug_gemm_trace(buf.ctypes.data)
    nt = (M // 256) * (N // 256)
    tr = buf[:nt]
    t0 = tr[:, 0].min()
    pro = tr[:, 1] - tr[:, 0]; loop = tr[:, 2] - tr[:, 1]; epi = tr[:, 3] - tr[:, 2]
    print(mode, f"tiles {nt}: prologue {np.median(pro)} loop {np.median(loop)} ({np.median(loop) / 16:.0f}/k-tile) epilogue {np.median(epi)} cycles; kernel span {(tr[:, 3].max() - t0)} cycles")
    # per-CU sequence: sort block start times, look at gaps between consecutive blocks on the same "slot"
    starts = np.sort(tr[:, 0] - t0); ends = np.sort(tr[:, 3] - t0)
    print("   first 3 starts", starts[:3], "256th..258th start", starts[256:259], "first 3 ends", ends[:3])
