"""Which ATen ops (fills, adds, copies, sums ...) ONE optimizer step of the ViT-L 3-D MAE at one volume issues beside the library's own kernels,
with shapes -- the small launches of DESIGN.md section 5 (one volume per step).   python tools/aten_census.py"""
import sys, collections, torch
sys.path.insert(0, ".")
from octcubem_amd import models_mae, misc, optim as foptim
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
torch.manual_seed(0)
model = models_mae.octcube_vit_large_3dmae().to(dev).train()
opt = foptim.FusedAdamW(misc.add_weight_decay(model, 0.05), lr=1e-4, betas=(0.9, 0.95))
scaler = misc.NativeScalerWithGradNormCount(fp32=True)
params = list(model.parameters())
x = torch.rand(1, 1, 60, 256, 256, device=dev)
def step():
    opt.zero_grad()
    loss, _, _ = model(x, mask_ratio=0.75)
    scaler(loss, opt, parameters=params)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter(); sites = collections.defaultdict(collections.Counter)
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::add_", "aten::contiguous", "aten::clone", "aten::empty_like", "aten::to", "aten::_to_copy", "aten::cat", "aten::sum"):
        cnt[e.name] += 1
        st = [f for f in (e.stack or []) if "octcubem_amd" in f or "bench" in f]
        sites[e.name][(st[0] if st else "?")[-90:] + " " + str(e.input_shapes)[:60]] += 1
for k, v in cnt.most_common(): 
    print(k, v)
    for s_, c in sites[k].most_common(8): print("     ", c, s_)
