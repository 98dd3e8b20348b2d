"""A kernel-selection option on / off for the forward / dgrad GEMM shapes of one ViT-L encoder / decoder Block at a given number of
volumes per micro-batch (same process, interleaved, random operands).
usage: python tools/gemm_option_ab.py <option, e.g. gemm_small> [volumes ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from octcubem_amd import ops  # noqa: E402

DEV = "cuda"


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    opt = sys.argv[1]
    vols = [int(a) for a in sys.argv[2:]] or [32, 128]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    for v in vols:
        print(f"--- {v} volumes per micro-batch")
        tot = {True: 0.0, False: 0.0}
        for name, Ntok, C in (("enc", 1281, 1024), ("dec", 5121, 512)):
            M = v * Ntok
            g = torch.Generator().manual_seed(0)
            x = torch.randn(M, C, generator=g).to(ops.BF16).to(DEV)
            x4 = torch.randn(M, 4 * C, generator=g).to(ops.BF16).to(DEV)
            x3 = torch.randn(M, 3 * C, generator=g).to(ops.BF16).to(DEV)
            res = torch.randn(M, C, generator=g).to(DEV)
            wq = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(ops.BF16).to(DEV)
            wp = (torch.randn(C, C, generator=g) * C ** -0.5).to(ops.BF16).to(DEV)
            w1 = (torch.randn(4 * C, C, generator=g) * C ** -0.5).to(ops.BF16).to(DEV)
            w2 = (torch.randn(C, 4 * C, generator=g) * C ** -0.5).to(ops.BF16).to(DEV)
            b = torch.zeros(4 * C, device=DEV)
            H, HD = 16, C // 16
            kinds = [
                ("fwd qkv", lambda: ops.linear_fwd(x, wq, b[:3 * C], "bf16"), 3 * C),
                ("fwd proj+res", lambda: ops.linear_fwd(x, wp, b[:C], "resid", res=res), C),
                ("fwd fc1+gelu", lambda: ops.linear_fwd(x, w1, b, "gelu"), 4 * C),
                ("fwd fc2+res", lambda: ops.linear_fwd(x4, w2, b[:C], "resid", res=res), C),
                ("dgrad fc2*gelu'", lambda: ops.linear_dgrad(x, w2, pre=x4, colsum=torch.zeros(4 * C, device=DEV)), 4 * C),
                ("dgrad fc1", lambda: ops.linear_dgrad(x4, w1), C),
                ("dgrad proj+delta", lambda: ops.linear_dgrad_delta(x, wp, x, H, HD), C),
                ("dgrad qkv", lambda: ops.linear_dgrad(x3, wq), C),
            ]
            for kn, fn, ncol in kinds:
                nt = -(-M // 256) * (ncol // 256)
                r = {}
                for rep in range(2):
                    for on in (True, False):
                        ops.set_option(opt, 1 if on else 0)
                        r.setdefault(on, []).append(bench(fn))
                ops.set_option(opt, 0)
                a, p_ = min(r[True]), min(r[False])
                tot[True] += a; tot[False] += p_
                print(f"{name} {kn:18s} tiles {nt:5d} = {nt / ncu:6.2f} rounds   shipped {p_:8.1f} us   {opt} {a:8.1f} us   {100 * (a / p_ - 1):+5.1f} %")
        print(f"sum of the 16 launches: shipped {tot[False]:.0f} us, {opt} {tot[True]:.0f} us ({100 * (tot[True] / tot[False] - 1):+.1f} %)")


if __name__ == "__main__":
    main()
