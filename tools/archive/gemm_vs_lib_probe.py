"""Profiling target: this repo's forward GEMM (plain bf16 epilogue) and the library GEMM PyTorch dispatches to, on the two big
encoder MLP shapes (micro-batch 128).  Run under rocprofv3 --pmc (tools/_gemm_vs_lib.sh) to compare effective clock
(GRBM_GUI_ACTIVE / 8 / duration) and MFMA-pipe occupancy of the two kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops
B = 128
for (M, K, N) in ((B * 1281, 1024, 4096), (B * 1281, 4096, 1024)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.zeros(N, device="cuda")
    for _ in range(4):
        y = ops.linear_fwd(x, w, b, "bf16")
        z = x @ w.t()
    torch.cuda.synchronize()
