"""tools/power_probe.py for the K = 512 decoder GEMMs and the fused-epilogue kinds: are they at the 1400 W cap too, or is their time
fixed cost per tile that a better schedule could hide?   python tools/power_probe_dec.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import ops

from power_probe import loop

g = torch.Generator(device="cuda").manual_seed(0)
B = 128
for zero in (False, True):
    tag = " [zeros]" if zero else ""
    for name, M, K, N in (("dec fc1 (K 512 -> 2048)", B * 5121, 512, 2048), ("dec qkv (K 512 -> 1536)", B * 5121, 512, 1536),
                          ("dec fc2 (K 2048 -> 512)", B * 5121, 2048, 512), ("enc fc1 (K 1024 -> 4096)", B * 1281, 1024, 4096)):
        x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        b = torch.zeros(N, device="cuda")
        if zero:
            x.zero_(); w.zero_()
        loop(f"fwd bf16 {name}{tag}", lambda: ops.linear_fwd(x, w, b, "bf16"))
        if "fc1" in name:
            loop(f"fwd gelu {name}{tag}", lambda: ops.linear_fwd(x, w, b, "gelu"))
            dy = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)      # fc2 dgrad: [M, K] x [K, N] -> [M, N], x GELU'(pre)
            w2 = (torch.randn(K, N, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
            pre = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
            if zero:
                dy.zero_(); w2.zero_(); pre.zero_()
            loop(f"dgrad x gelu' {name}{tag}", lambda: ops.linear_dgrad(dy, w2, pre=pre))
            del dy, w2, pre
        del x, w
