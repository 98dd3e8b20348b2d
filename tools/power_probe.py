"""Board power / clock while ONE kernel kind runs back to back (rocm-smi sampled from a side thread): which kernels run at the
power cap, and at what clock.  python tools/power_probe.py   (GPU box)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops

samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out))
        except Exception as e:
            samples.append((time.time(), "ERR " + repr(e)))
        time.sleep(0.2)


def summarize(name, t0, t1, per_call_ms):
    import json
    pw, sclk = [], []
    for ts, out in samples:
        if not (t0 + 0.5 <= ts <= t1):
            continue
        try:
            d = json.loads(out)
            c = d[sorted(d)[0]]
            for k, v in c.items():
                if "Power (W)" in k and "Max" not in k:
                    pw.append(float(v))
                if k.lower().startswith("sclk clock speed"):
                    sclk.append(float(str(v).strip("()").lower().replace("mhz", "")))
        except Exception:
            pass
    avg = lambda x: sum(x) / len(x) if x else float("nan")
    print(f"{name:34s} {per_call_ms:9.3f} ms/call   power avg {avg(pw):7.1f} W (max {max(pw) if pw else float('nan'):7.1f}, {len(pw)} samples)   sclk avg {avg(sclk):7.1f} MHz", flush=True)


def loop(name, fn, seconds=4.0):
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(5):
            fn()
        torch.cuda.synchronize(); n += 5
    t1 = time.time()
    summarize(name, t0, t1, (t1 - t0) / n * 1e3)


th = threading.Thread(target=sampler, daemon=True); th.start()
time.sleep(1.0)
try:
    print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout.strip()[-300:])
except Exception as e:
    print("rocm-smi --showmaxpower:", e)
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
t0 = time.time(); time.sleep(2.0)
summarize("idle", t0 - 0.5, time.time(), 0.0)
stop = True
print("first raw sample:", samples[0][1][:600] if samples else None)
