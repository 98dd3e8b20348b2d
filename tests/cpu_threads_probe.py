"""Why bench.py's cpu_baseline caps the oracle at 32 torch threads: ONE forward of the ViT-L 3-D MAE oracle (1 volume, fp32) at 32 threads
and at every host thread, timed once each (the all-core run is oversubscribed on the 256-thread hosts of the pool).
python tests/cpu_threads_probe.py   -> prints the two timings (kept in profiles/ per round)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mae3d_ref as O

P = O.init_params(O.VIT_L, seed=0)
imgs = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(0))
noise = torch.rand(1, 5120, generator=torch.Generator().manual_seed(1))
host = os.cpu_count() or 1
for threads in [32, 64, host]:
    if threads > host:
        continue
    torch.set_num_threads(threads)
    ts = []
    for it in range(2):
        t0 = time.time()
        with torch.no_grad():
            O.forward(P, imgs, O.VIT_L, 0.75, noise)
        ts.append(time.time() - t0)
    print(f"oracle forward, 1 volume, fp32, {threads:3d} of {host} host threads: warm-up {ts[0]:.1f} s, timed {ts[1]:.1f} s", flush=True)
