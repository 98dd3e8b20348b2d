"""Weight-gradient GEMMs of the ViT-L step with equal and with STAGGERED split-K slices (octmae_set_option "wgrad_stagger", csrc/gemm.hip
split_range): the slices of a split finish one after the other, so that the fp32-atomic epilogue of one overlaps the k-loops of the rest.
Same process, interleaved rounds, medians; checks every result against an fp32 matmul.   python tools/wgrad_stagger_ab.py [B] [v ...]"""
import statistics
import sys

import torch

from octcubem_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
VALUES = [int(v) for v in sys.argv[2:]] or [0, 15, 29, 45]
dev = torch.device("cuda")
SHAPES = [("enc qkv", 1281, 3072, 1024), ("enc proj", 1281, 1024, 1024), ("enc fc1", 1281, 4096, 1024), ("enc fc2", 1281, 1024, 4096),
          ("dec qkv", 5121, 1536, 512), ("dec proj", 5121, 512, 512), ("dec fc1", 5121, 2048, 512), ("dec fc2", 5121, 512, 2048)]
WEIGHT = {"enc": 24, "dec": 8}


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {v: 0.0 for v in VALUES}
for name, N, NO, NI in SHAPES:
    M = B * N
    g = torch.Generator(device=dev).manual_seed(5)
    dy = (torch.randn(M, NO, device=dev, generator=g) * 0.1).bfloat16()
    x = torch.randn(M, NI, device=dev, generator=g).bfloat16()
    gw = torch.zeros(NO, NI, device=dev)
    ref = None
    ts = {v: [] for v in VALUES}
    for v in VALUES:                                    # correctness of each setting first
        ops.set_option("wgrad_stagger", v)
        gw.zero_()
        ops.linear_wgrad_accum(dy, x, gw)
        if ref is None:
            ref = torch.zeros_like(gw)
            for c in range(0, M, 65536):
                ref += dy[c:c + 65536].float().T @ x[c:c + 65536].float()
        err = float((gw - ref).norm() / ref.norm())
        assert err < 2e-5, (name, v, err)
    for _ in range(5):
        for v in VALUES:
            ops.set_option("wgrad_stagger", v)
            ts[v].append(timed(lambda: ops.linear_wgrad_accum(dy, x, gw)))
    med = {v: statistics.median(ts[v]) for v in VALUES}
    for v in VALUES:
        tot[v] += med[v] * WEIGHT[name[:3]]
    fl = 2.0 * M * NO * NI
    print(f"B {B} {name:9s} " + "  ".join(f"v={v}: {med[v]:8.1f} us ({fl / med[v] / 1e6:6.0f} TF)" for v in VALUES) +
          "   ratio to v=0: " + " ".join(f"{med[v] / med[VALUES[0]]:.3f}" for v in VALUES[1:]), flush=True)
    del dy, x, gw, ref
ops.set_option("wgrad_stagger", 0)
print(f"B {B} all weight gradients of one backward: " + "  ".join(f"v={v}: {tot[v] / 1e3:.2f} ms" for v in VALUES))
