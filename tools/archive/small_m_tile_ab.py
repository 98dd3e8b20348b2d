"""256-tile vs 128-tile GEMM kernels at SMALL row counts (1, 2, 4, 8 volumes of 1281 / 5121 tokens): where few 256 x 256 tiles leave most
of the 256 CUs idle, the 128-tile kernel's four times as many workgroups win.   python tools/small_m_tile_ab.py"""
import statistics

import torch

from octcubem_amd import ops

dev = torch.device("cuda")


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'rows':>7s} {'N':>5s} {'K':>5s} {'tiles256':>8s}   fwd bf16 256 / 128 tile (us)    fwd resid 256 / 128     dgrad 256 / 128")
for vols in (1, 2, 4, 8, 16):
    for tokens, shapes in ((1281, [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)]), (5121, [(512, 512), (1536, 512), (2048, 512), (512, 2048)])):
        M = vols * tokens
        for N, K in shapes:
            g = torch.Generator(device=dev).manual_seed(1)
            x = torch.randn(M, K, device=dev, generator=g).bfloat16(); w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
            b = torch.zeros(N, device=dev); res = torch.randn(M, N, device=dev, generator=g); dy = torch.randn(M, N, device=dev, generator=g).bfloat16()
            out = []
            for small in (False, True):
                ops.FORCE_SMALL_TILE = small
                out.append((statistics.median(timed(lambda: ops.linear_fwd(x, w, b, "bf16")) for _ in range(3)),
                            statistics.median(timed(lambda: ops.linear_fwd(x, w, b, "resid", res=res)) for _ in range(3)),
                            statistics.median(timed(lambda: ops.linear_dgrad(dy, w)) for _ in range(3))))
            ops.FORCE_SMALL_TILE = False
            t256 = ((M + 255) // 256) * ((N + 255) // 256)
            print(f"{M:7d} {N:5d} {K:5d} {t256:8d}   " + "    ".join(f"{a:8.1f} / {b_:8.1f} ({a / b_:4.2f}x)" for a, b_ in zip(out[0], out[1])), flush=True)
