"""Energy per call of the kernel variants this library can choose between ("rank variants by joules per call").
The step is power-bound: a variant that saves cycles but not joules returns the cycles as a lower clock.  For each
variant one kernel runs back to back for a few seconds while a side thread reads the board's hwmon power / clock files:
    J/call = average board power x time per call;  dynamic J/call = (power - idle power) x time per call.
Random N(0, 1) operands (operand switching is 40 % of a GEMM's dynamic energy: zeros would rank by a power the step does not have).
python tools/joules_per_call.py [seconds per variant]      (GPU box; OCTMAE_LIB selects the build: bf16 or the half build)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops

import bench          # BoardSampler: amdgpu hwmon files in sysfs (no child process after the GPU is initialised)

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
ROWS = []


def measure(name, fn, flops=None):
    fn(); torch.cuda.synchronize()
    b = bench.BoardSampler(0, period=0.1); b.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < SECONDS:
        for _ in range(5):
            fn()
        torch.cuda.synchronize(); n += 5
    ms = (time.time() - t0) / n * 1e3
    st = b.stop() or {"power_w_avg": float("nan"), "sclk_mhz_avg": float("nan")}
    ROWS.append((name, ms, st["power_w_avg"], st["sclk_mhz_avg"], flops))


def main():
    time.sleep(1.0)
    b = bench.BoardSampler(0, period=0.1); b.start(); time.sleep(2.0)
    idle = (b.stop() or {"power_w_avg": float("nan")})["power_w_avg"]
    g = torch.Generator(device="cuda").manual_seed(0)
    B = 128
    LP = ops.BF16
    for (N, H, HD) in ((5121, 16, 32), (1281, 16, 64)):
        qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(LP)
        do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(LP)
        sc = HD ** -0.5
        o, lse = ops.attn_fwd(qkv, B, N, H, HD, sc)
        fl = 2.0 * B * H * N * N * HD
        if not ops.LP_IS_F16:
            measure(f"attn fwd hd{HD} optimistic (shipped)", lambda: ops.attn_fwd(qkv, B, N, H, HD, sc, optimistic=True), 2 * fl)
        measure(f"attn fwd hd{HD} online max", lambda: ops.attn_fwd(qkv, B, N, H, HD, sc, optimistic=False), 2 * fl)
        key = f"attn_bwd_hd{HD}_form"
        ops.set_option(key, 1)
        measure(f"attn bwd hd{HD} fused, one wave / SIMD (shipped)", lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, sc, fused=True), 4 * fl)
        ops.set_option(key, 0)
        measure(f"attn bwd hd{HD} fused, two waves / SIMD", lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, sc, fused=True), 4 * fl)
        ops.set_option(key, 1)
        measure(f"attn bwd hd{HD} dQ + dK/dV pair", lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, sc, fused=False), 4 * fl)
        del qkv, do, o, lse
    for vols in (128, 32):
        M, C = vols * 1281, 1024
        x = torch.randn(M, C, device="cuda", generator=g).to(LP)
        x4 = torch.randn(M, 4 * C, device="cuda", generator=g).to(LP)
        w1 = (torch.randn(4 * C, C, device="cuda", generator=g) * C ** -0.5).to(LP)
        w2 = (torch.randn(C, 4 * C, device="cuda", generator=g) * (4 * C) ** -0.5).to(LP)
        res = torch.randn(M, C, device="cuda", generator=g)
        bias = torch.zeros(4 * C, device="cuda")
        gw = torch.zeros(4 * C, C, device="cuda")
        fl = 2.0 * M * C * 4 * C
        measure(f"gemm fwd fc1 bf16 out, {vols} vol", lambda: ops.linear_fwd(x, w1, bias, "bf16"), fl)
        measure(f"gemm dgrad fc1, {vols} vol", lambda: ops.linear_dgrad(x4, w1), fl)
        measure(f"gemm wgrad fc1, {vols} vol", lambda: ops.linear_wgrad_accum(x4, x, gw), fl)
        for force, tag in ((-1, "256-tile (shipped at this size)"), (2, "small-launch kernel, 2-stage ring")):
            ops.FORCE_SMALL_LAUNCH = force
            measure(f"gemm fwd fc2 + residual, {vols} vol, {tag}", lambda: ops.linear_fwd(x4, w2, bias[:C], "resid", res=res), fl)
        ops.FORCE_SMALL_LAUNCH = 0
        del x, x4, w1, w2, res, gw
    xf = torch.randn(128 * 1281, 1024, device="cuda", generator=g); gm = torch.ones(1024, device="cuda"); bt = torch.zeros(1024, device="cuda")
    measure("layernorm fwd D 1024 (bandwidth-bound)", lambda: ops.layernorm_fwd(xf, gm, bt, 1e-6))
    print(f"# lib {os.environ.get('OCTMAE_LIB', 'liboctmae.so')}  operand type {LP}  idle board power {idle:.0f} W  ({SECONDS:.1f} s per variant, random operands)")
    print(f"{'variant':58s} {'ms/call':>9s} {'W':>7s} {'GHz':>6s} {'J/call':>8s} {'dyn J':>8s} {'pJ/flop (dyn)':>14s}")
    for name, ms, w, mhz, fl in ROWS:
        j, dj = w * ms * 1e-3, (w - idle) * ms * 1e-3
        print(f"{name:58s} {ms:9.3f} {w:7.0f} {mhz / 1e3:6.2f} {j:8.2f} {dj:8.2f} {(dj / fl * 1e12) if fl else float('nan'):14.2f}")


if __name__ == "__main__":
    main()
