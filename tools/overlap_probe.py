"""Premise check for running the optimizer under the backward: does a bandwidth-bound pass (a LayerNorm forward
over 1.3 GB, ~ the AdamW pass) hide beside MFMA-bound GEMMs on another stream, or does the power cap hand its time back?
python tools/overlap_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M, C = 32 * 1281, 1024
x = torch.randn(M, C, device=dev, generator=g).to(ops.BF16)
w1 = (torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5).to(ops.BF16)
b = torch.zeros(4 * C, device=dev)
big = torch.randn(8 * 128 * 1281 // 4, 1024, device=dev, generator=g)        # 1.3 GB fp32: one pass ~ the optimizer's
gm, bt = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
side = torch.cuda.Stream()


def gemms(n=40):
    for _ in range(n):
        ops.linear_fwd(x, w1, b, "bf16")


def hbm_pass():
    ops.layernorm_fwd(big, gm, bt, 1e-6)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def both_sequential():
    gemms(); hbm_pass()


def both_concurrent():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hbm_pass()
    gemms()
    torch.cuda.current_stream().wait_stream(side)


tg, th = timed(gemms), timed(hbm_pass)
ts, tc = timed(both_sequential), timed(both_concurrent)
print(f"40 GEMMs (fc1 forward, 32 volumes) {tg:.3f} ms; bandwidth-bound pass over 1.3 GB {th:.3f} ms; one after the other {ts:.3f} ms; "
      f"the pass on a second stream beside the GEMMs {tc:.3f} ms  -> hidden: {100 * (ts - tc) / th:.0f} % of the pass")
