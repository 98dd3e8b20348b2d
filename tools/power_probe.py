"""Board power / clock while ONE kernel kind runs back to back (amdgpu hwmon files sampled from a side thread): which kernels run at the
power cap, and at what clock.  python tools/power_probe.py   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops

import bench          # BoardSampler: amdgpu hwmon files in sysfs (no child process after the GPU is initialised)


def run_sampled(fn_loop):
    b = bench.BoardSampler(0, period=0.1); b.start()
    r = fn_loop()
    st = b.stop()
    return r, st


def loop(name, fn, seconds=4.0):
    fn(); torch.cuda.synchronize()

    def body():
        t0 = time.time(); n = 0
        while time.time() - t0 < seconds:
            for _ in range(5):
                fn()
            torch.cuda.synchronize(); n += 5
        return (time.time() - t0) / n * 1e3
    ms, st = run_sampled(body)
    st = st or {"power_w_avg": float("nan"), "power_w_max": float("nan"), "sclk_mhz_avg": float("nan"), "samples": 0}
    print(f"{name:34s} {ms:9.3f} ms/call   power avg {st['power_w_avg']:7.1f} W (max {st['power_w_max']:7.1f}, {st['samples']} samples)   "
          f"sclk avg {st['sclk_mhz_avg']:7.1f} MHz", flush=True)



def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    B = 128
    for zero in (False, True):
        tag = " [zeros]" if zero else ""
        for (N, H, HD) in ((5121, 16, 32), (1281, 16, 64)):
            qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
            do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(torch.bfloat16)
            if zero:
                qkv.zero_(); do.zero_()
            o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
            loop(f"attn_fwd hd{HD}{tag}", lambda: ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5))
            loop(f"attn_bwd hd{HD}{tag}", lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5))
            del qkv, do, o, lse
        M, K, Nn = B * 1281, 1024, 4096
        x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16); w = (torch.randn(Nn, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        dy = torch.randn(M, Nn, device="cuda", generator=g).to(torch.bfloat16); gw = torch.zeros(Nn, K, device="cuda"); b = torch.zeros(Nn, device="cuda")
        if zero:
            x.zero_(); w.zero_(); dy.zero_()
        loop(f"gemm fwd fc1 bf16{tag}", lambda: ops.linear_fwd(x, w, b, "bf16"))
        loop(f"gemm wgrad fc1{tag}", lambda: ops.linear_wgrad_accum(dy, x, gw))
        xf = torch.randn(M, K, device="cuda", generator=g); gm = torch.ones(K, device="cuda"); bt = torch.zeros(K, device="cuda")
        loop(f"layernorm fwd{tag}", lambda: ops.layernorm_fwd(xf, gm, bt, 1e-6))
        del x, w, dy, gw, xf
    time.sleep(1.0)
    _, st = run_sampled(lambda: time.sleep(2.0))
    print("idle", st)


if __name__ == "__main__":
    main()
