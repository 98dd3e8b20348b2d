"""Fine-tune step throughput: ViT-L spatio-temporal classifier on 60x256x256 volumes, 5121 tokens through all
24 blocks, fwd + bwd + clip + layer-decay AdamW.   python tools/bench_finetune.py [--batch 16] [--drop-path 0.2] [--steps 3]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import models_vit_st, lr_decay, misc, losses
from octcubem_amd import optim as foptim

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--drop-path", type=float, default=0.2)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--classes", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
m = models_vit_st.vit_large_patch16(num_frames=60, t_patch_size=3, img_size=256, in_chans=1, num_classes=a.classes, sep_pos_embed=True,
                                    cls_embed=True, global_pool=True, drop_path_rate=a.drop_path, dropout=0.5).to(dev).train()
opt = foptim.FusedAdamW(lr_decay.param_groups_lrd(m, 0.05, m.no_weight_decay(), 0.65), lr=1e-4)
scaler = misc.NativeScalerWithGradNormCount()
crit = losses.LabelSmoothingCrossEntropy(0.1)
x = torch.rand(a.batch, 1, 60, 256, 256, device=dev)
t = torch.randint(0, a.classes, (a.batch,), device=dev)
params = list(m.parameters())
def step():
    opt.zero_grad()
    loss = crit(m(x), t)
    scaler(loss, opt, clip_grad=1.0, parameters=params, update_grad=True)
    return loss
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps): loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
gf = 5.7e3 * 3                                   # 5.7 TFLOP forward per volume
print(json.dumps({"workload": "ST ViT-L fine-tune step, 5121 tokens", "batch": a.batch, "drop_path": a.drop_path, "ms_per_step": dt * 1e3,
                  "volumes_per_s": a.batch / dt, "model_tflops": a.batch * gf / dt / 1e3, "loss": float(loss),
                  "max_mem_gb": torch.cuda.max_memory_allocated() / 2**30}))
