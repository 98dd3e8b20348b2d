"""Upper bound for GROUPED weight-gradient launches: the same flops as one launch with more output tiles and a smaller split
(e.g. fc1 + fc2 of a Block: 2 x 64 tiles x 4 slices  ->  128 tiles x 2 slices), timed with the existing kernel on a weight of the
combined shape.   python tools/wgrad_group_bound.py [B]"""
import statistics
import sys

import torch

from octcubem_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def wg(M, NO, NI):
    g = torch.Generator(device=dev).manual_seed(5)
    dy = (torch.randn(M, NO, device=dev, generator=g) * 0.1).bfloat16()
    x = torch.randn(M, NI, device=dev, generator=g).bfloat16()
    gw = torch.zeros(NO, NI, device=dev)
    return statistics.median(timed(lambda: ops.linear_wgrad_accum(dy, x, gw)) for _ in range(5))


for name, N, parts, combined in [
        ("enc fc1 + fc2", 1281, [(4096, 1024), (4096, 1024)], (8192, 1024)),
        ("enc qkv + proj", 1281, [(3072, 1024), (1024, 1024)], (4096, 1024)),
        ("enc all four", 1281, [(4096, 1024), (4096, 1024), (3072, 1024), (1024, 1024)], (12288, 1024)),
        ("dec fc1 + fc2", 5121, [(2048, 512), (2048, 512)], (4096, 512)),
        ("dec qkv + proj", 5121, [(1536, 512), (512, 512)], (2048, 512)),
        ("dec all four", 5121, [(2048, 512), (2048, 512), (1536, 512), (512, 512)], (6144, 512))]:
    M = B * N
    sep = sum(wg(M, a, b) for a, b in parts)
    comb = wg(M, *combined)
    print(f"B {B} {name:15s} separate {sep:8.1f} us   one launch of the combined shape {comb:8.1f} us   ratio {comb / sep:.3f}", flush=True)
