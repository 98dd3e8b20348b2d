"""Yardstick (NOT product code): what the flash attention PyTorch-ROCm itself dispatches to (F.scaled_dot_product_attention: AOTriton /
CK) reaches on the SAME shapes, box and data as this repo's kernels -- forward and forward + backward, bf16, random N(0,1) inputs.
cdna_hip_programming.md rule 10: a ceiling claim needs a known-good reference measured on the same hardware.
    python tools/sdpa_yardstick.py [B]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from octcubem_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = 16


def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


for N, HD in ((5121, 32), (1281, 64), (5121, 64)):
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * N, 3 * H * HD, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(B * N, H * HD, device="cuda", generator=g).to(torch.bfloat16)
    unit = 2.0 * B * H * N * N * HD
    # ---- this repo
    o, lse = ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5)
    t_f = timeit(lambda: ops.attn_fwd(qkv, B, N, H, HD, HD ** -0.5))
    t_b = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, HD, HD ** -0.5))
    # ---- the library path: q, k, v as [B, H, N, HD] views of the same packed tensor (strided, as a fused-qkv model would pass them)
    x = qkv.view(B, N, 3, H, HD)
    q, k, v = (x[:, :, i].transpose(1, 2).detach().requires_grad_(True) for i in range(3))
    dout = do.view(B, N, H, HD).transpose(1, 2)
    res = {}
    for name, ctx in (("default", None),):
        try:
            out = F.scaled_dot_product_attention(q, k, v)
            lf = timeit(lambda: F.scaled_dot_product_attention(q, k, v))

            def fb():
                oo = F.scaled_dot_product_attention(q, k, v)
                oo.backward(dout)
                q.grad = k.grad = v.grad = None
            lfb = timeit(fb)
            err = float((out.transpose(1, 2).reshape(B * N, H * HD).float() - o.float()).norm() / o.float().norm())
            res[name] = (lf, lfb - lf, err)
        except Exception as ex:      # a shape the library has no kernel for
            res[name] = repr(ex)[:120]
    print(f"N {N} hd {HD} B {B}: this repo  fwd {t_f:8.3f} ms ({2 * unit / t_f / 1e9:6.0f} TFLOP/s)   bwd {t_b:8.3f} ms ({4 * unit / t_b / 1e9:6.0f} TFLOP/s algorithmic)")
    for name, r in res.items():
        if isinstance(r, tuple):
            print(f"{'':22s} torch SDPA fwd {r[0]:8.3f} ms ({2 * unit / r[0] / 1e9:6.0f} TFLOP/s)   bwd {r[1]:8.3f} ms ({4 * unit / r[1] / 1e9:6.0f} TFLOP/s algorithmic)   "
                  f"|o - o_repo| / |o_repo| = {r[2]:.1e}")
        else:
            print(f"{'':22s} torch SDPA: {r}")
