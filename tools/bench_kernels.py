#!/usr/bin/env python3
"""Micro-benchmarks of the individual kernels at the BASELINE shapes (ViT-L 3-D MAE, micro-batch B volumes).

    python tools/bench_kernels.py [--batch 32] [--only attn|gemm|ln|all] [--iters 10]

Prints one line per kernel: average time (HIP events), algorithmic TFLOP/s or GB/s, fraction of the MI355X peak.
Used for A/B-ing kernel changes in one process and as the target of `rocprofv3 --pmc` runs."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def report(name, t, flops=None, nbytes=None):
    if flops:
        print(f"{name:44s} {t*1e6:9.1f} us  {flops/t/1e12:8.1f} TFLOP/s  {flops/t/2.5e15*100:5.1f}% of bf16 peak", flush=True)
    else:
        print(f"{name:44s} {t*1e6:9.1f} us  {nbytes/t/1e9:8.1f} GB/s     {nbytes/t/8e12*100:5.1f}% of HBM peak", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="all")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    B = a.batch
    g = torch.Generator(device=DEV).manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, device=DEV, generator=g) * scale).to(BF)

    if a.only in ("all", "attn"):
        for name, N, H, HD in (("enc", 1281, 16, 64), ("dec", 5121, 16, 32)):
            qkv = rnd(B * N, 3 * H * HD)
            do = rnd(B * N, H * HD)
            o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
            unit = 2.0 * B * H * N * N * HD
            report(f"attn_fwd  {name} N={N} hd={HD}", timeit(lambda: ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5), a.iters), 2 * unit)
            report(f"attn_bwd  {name} N={N} hd={HD} (delta+dq+dkv)", timeit(lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5), a.iters), 5 * unit)
    if a.only in ("all", "gemm"):
        for name, M, D in (("enc", B * 1281, 1024), ("dec", B * 5121, 512)):
            for lname, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
                x = rnd(M, K); w = rnd(N, K, scale=K ** -0.5); b = torch.zeros(N, device=DEV)
                dy = rnd(M, N); res = torch.zeros(M, N, device=DEV)
                fl = 2.0 * M * N * K
                mode = {"qkv": "bf16", "proj": "resid", "fc1": "gelu", "fc2": "resid"}[lname]
                report(f"gemm fwd   {name} {lname} [{M}x{K}]x[{N}x{K}] {mode}", timeit(lambda: ops.linear_fwd(x, w, b, mode, res=res), a.iters), fl)
                pre = rnd(M, K) if lname == "fc2" else None
                report(f"gemm dgrad {name} {lname}" + (" +dgelu" if pre is not None else ""), timeit(lambda: ops.linear_dgrad(dy, w, pre=pre), a.iters), fl)
                gw = torch.zeros(N, K, device=DEV)
                report(f"gemm wgrad {name} {lname}", timeit(lambda: ops.linear_wgrad_accum(dy, x, gw), a.iters), fl)
    if a.only in ("all", "ln"):
        for name, M, D in (("enc", B * 1281, 1024), ("dec", B * 5121, 512)):
            x = torch.randn(M, D, device=DEV, generator=g); gm = torch.ones(D, device=DEV); bt = torch.zeros(D, device=DEV)
            y, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-6)
            dy = rnd(M, D); dg = torch.zeros(D, device=DEV); db = torch.zeros(D, device=DEV)
            report(f"layernorm fwd {name} [{M}x{D}]", timeit(lambda: ops.layernorm_fwd(x, gm, bt, 1e-6), a.iters), nbytes=6.0 * M * D)
            report(f"layernorm bwd {name}", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, gm, dg, db), a.iters), nbytes=10.0 * M * D)
            report(f"colsum bf16 {name} [{M}x{4*D}]", timeit(lambda: ops.colsum_accum(rnd(1, 8).new_zeros(M, 4 * D) if False else big, torch.zeros(4 * D, device=DEV)), a.iters), nbytes=2.0 * M * 4 * D) if False else None
            big = rnd(M, D)
            out = torch.zeros(D, device=DEV)
            report(f"colsum bf16 {name} [{M}x{D}]", timeit(lambda: ops.colsum_accum(big, out), a.iters), nbytes=2.0 * M * D)
            report(f"cast f32->bf16 {name}", timeit(lambda: ops.cast_bf16(x), a.iters), nbytes=6.0 * M * D)


if __name__ == "__main__":
    main()
