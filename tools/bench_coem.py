"""BASELINE config 5 (retinal-COEM OCTCube-IR) on one GPU: the shipped tower config (ViT-L ST tower on 60x256x256 volumes + ViT-L
2-D tower on 224x224 en-face images, embed 512), contrastive step = forward both towers, ClipLoss, backward, AdamW per tower,
temperature clamp (coem.train_step; retinal-COEM/src/training/train_retclip.py).  One JSON line: pairs/s.
    python tools/bench_coem.py [batch] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octcubem_amd import coem, optim as foptim
from tests.test_gpu_coem import SHIPPED_CFG

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = json.loads(json.dumps(SHIPPED_CFG))
cfg["vision_cfg"]["drop_path_rate"] = 0.2; cfg["text_cfg"]["drop_path_rate"] = 0.2        # the shipped values
torch.manual_seed(0)
model = coem.create_model_from_config(cfg).to("cuda").train()
opts = [foptim.FusedAdamW(model.visual.parameters(), lr=1e-4), foptim.FusedAdamW(model.text.parameters(), lr=1e-4),
        torch.optim.AdamW([model.logit_scale], lr=1e-4)]
g = torch.Generator(device="cuda").manual_seed(1)
vol = torch.rand(B, 1, 60, 256, 256, device="cuda", generator=g); ir = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
loss_fn = coem.ClipLoss()
for _ in range(2):
    coem.train_step(model, loss_fn, vol, ir, opts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = coem.train_step(model, loss_fn, vol, ir, opts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# model FLOP per pair: ST tower 5.7 TF forward + 2-D ViT-L at 197 tokens ~0.12 TF, x3 for forward + backward
print(json.dumps({"metric": "COEM contrastive step pairs/s (ViT-L ST tower 60x256x256 + ViT-L 2-D tower 224x224, embed 512)", "value": B / dt,
                  "unit": "pairs/s", "batch": B, "ms_per_step": 1e3 * dt, "loss": float(loss), "n_gpus": 1, "dtype": "bf16", "data": "synthetic",
                  "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}))
