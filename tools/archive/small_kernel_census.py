"""Who launches the small ATen kernels of a training step (fills, copies, adds): a census by Python call site, from torch.profiler
with stacks.   python tools/small_kernel_census.py [micro-batch]"""
import collections
import sys

import torch
from torch.profiler import ProfilerActivity, profile

from octcubem_amd import misc, models_mae, optim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda")
torch.manual_seed(0)
model = models_mae.mae_vit_large_patch16(input_size=256, in_chans=1, num_frames=60, t_patch_size=3, pred_t_dim=60, sep_pos_embed=True,
                                         cls_embed=True, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16).to(dev)
groups = misc.add_weight_decay(model, 0.05) if hasattr(misc, "add_weight_decay") else model.parameters()
opt = optim.FusedAdamW(groups, lr=1e-4, betas=(0.9, 0.95))
scaler = misc.NativeScalerWithGradNormCount(fp32=True) if "fp32" in misc.NativeScalerWithGradNormCount.__init__.__code__.co_varnames else misc.NativeScalerWithGradNormCount()
params = [p for p in model.parameters() if p.requires_grad]
x = torch.rand(B, 1, 60, 256, 256, device=dev)


def step():
    opt.zero_grad()
    loss, _, _ = model(x, mask_ratio=0.75)
    scaler(loss, opt, parameters=params, update_grad=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
sites = collections.Counter()
names = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::add", "aten::zeros", "aten::clone", "aten::mul", "aten::sum")
for ev in prof.events():
    if ev.name in names and ev.device_type == torch.autograd.DeviceType.CPU:
        st = [f for f in (ev.stack or []) if "octcubem_amd" in f or "bench.py" in f or "tools/" in f]
        top = st[0] if st else ((ev.stack or ["<no python frame: autograd engine>"])[0])
        sites[(ev.name, top.strip()[-110:])] += 1
for (name, site), n in sites.most_common(40):
    print(f"{n:5d}  {name:12s} {site}")
