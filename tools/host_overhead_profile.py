"""Where the HOST time of a small-batch step goes (cProfile over a few steps at batch 1, where the step is launch-bound).
python tools/host_overhead_profile.py [batch] [steps]"""
import cProfile
import pstats
import sys
import time

import torch

from octcubem_amd import misc, models_mae, optim as foptim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
torch.manual_seed(0)
model = models_mae.octcube_vit_large_3dmae().to(dev).train()
opt = foptim.FusedAdamW(misc.add_weight_decay(model, 0.05), lr=1e-4, betas=(0.9, 0.95))
scaler = misc.NativeScalerWithGradNormCount(fp32=True)
params = list(model.parameters())
x = torch.rand(B, 1, 60, 256, 256, device=dev)


def step():
    opt.zero_grad()
    loss, _, _ = model(x, mask_ratio=0.75)
    scaler(loss, opt, parameters=params, update_grad=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_host = time.perf_counter() - t0            # host time to ENQUEUE the steps
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"batch {B}: {1e3 * t_all / steps:.2f} ms per step, of which the host needs {1e3 * t_host / steps:.2f} ms to enqueue it")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
