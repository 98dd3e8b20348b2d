"""The reference's SHIPPED pre-training recipe as one step (1 volume of 60 x 256 x 256 + 64 B-scans of 3 x 512 x 512 per GPU,
mask ratio 0.9 for both, Pre-training/scripts/run_chunks_pretraining_vitl_oph_joint_flash_attn.sh:25-66): the joint 3-D + 2-D/512 step of
engine_pretrain.train_one_epoch_joint -- two forwards, one backward, AdamW.   python tools/bench_joint.py [volumes] [images] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import misc, models_mae, optim as foptim

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1
I = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda")
torch.manual_seed(0)
model = models_mae.octcube_vit_large_3dmae().to(dev).train()
opt = foptim.FusedAdamW(misc.add_weight_decay(model, 0.05), lr=1e-5, betas=(0.9, 0.95))
scaler = misc.NativeScalerWithGradNormCount(fp32=True)
params = list(model.parameters())
x3 = torch.rand(V, 1, 60, 256, 256, device=dev)
x2 = torch.rand(I, 1, 3, 512, 512, device=dev)


def step():
    opt.zero_grad()
    (loss, frame_loss), _, _ = model(x3, mask_ratio=0.9, frame_loss=True)
    loss_2d, _, _ = model(x2, mask_ratio=0.9)
    scaler(loss + loss_2d, opt, parameters=params, update_grad=True)
    return loss, loss_2d


for _ in range(3):
    l3, l2 = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l3, l2 = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": f"joint 3-D + 2-D/512 pre-training step (the shipped recipe): {V} volume(s) + {I} B-scans, mask 0.9", "ms_per_step": dt * 1e3,
                  "volumes_per_s": V / dt, "images_per_s": I / dt, "loss_3d": float(l3.detach()), "loss_2d": float(l2.detach()),
                  "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}))
