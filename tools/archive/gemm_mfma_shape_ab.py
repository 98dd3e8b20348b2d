"""Forward and dgrad GEMMs of the ViT-L step (micro-batch B, default 128): 32x32x16 (gemm256p_kernel) vs 16x16x32
(gemm256q_kernel) MFMAs, same process, interleaved rounds (cdna_hip_programming.md rule 24), random data.
    python tools/gemm_mfma_shape_ab.py [B] [rounds]"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def t(f, iters=6):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


g = torch.Generator(device="cuda").manual_seed(0)
tot = {0: 0.0, 1: 0.0}
cases = []
for name, M, D in (("enc", B * 1281, 1024), ("dec", B * 5121, 512)):
    for lname, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
        mode = {"qkv": "bf16", "proj": "resid", "fc1": "gelu", "fc2": "resid"}[lname]
        cases.append((name, lname, M, N, K, mode))
for name, lname, M, N, K, mode in cases:
    x = (torch.randn(M, K, device="cuda", generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    dy = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda", generator=g) if mode == "resid" else None
    pre = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) if lname == "fc2" else None
    cs = torch.zeros(K, device="cuda") if lname == "fc2" else None
    fl = 2.0 * M * N * K
    for kind, fn in (("fwd  " + mode, lambda: ops.linear_fwd(x, w, b, mode, res=res)),
                     ("dgrad" + (" x gelu'" if pre is not None else ""), lambda: ops.linear_dgrad(dy, w, pre=pre, colsum=cs))):
        ts = {0: [], 1: []}
        for r in range(R):
            for m32 in (1, 0):
                ops.set_option("gemm_mfma16", 0 if m32 else 1)
                ts[m32].append(t(fn))
        ops.set_option("gemm_mfma16", 0)
        a, q = statistics.median(ts[1]), statistics.median(ts[0])
        # calls per step at global batch 256 = 2 micro-batches x layers
        calls = (24 if name == "enc" else 8) * 2
        tot[1] += a * calls; tot[0] += q * calls
        print(f"{name} {lname:4s} {kind:16s} [{M}x{K}]x[{N}x{K}]  32x32x16 {a:8.1f} us ({fl / a / 1e6:5.0f} TF)   16x16x32 {q:8.1f} us ({fl / q / 1e6:5.0f} TF)   ratio {q / a:.3f}", flush=True)
    del x, w, dy, res, pre
print(f"per 256-volume step (forward + dgrad GEMMs): 32x32x16 {tot[1] / 1e3:.1f} ms   16x16x32 {tot[0] / 1e3:.1f} ms   ratio {tot[0] / tot[1]:.3f}")
