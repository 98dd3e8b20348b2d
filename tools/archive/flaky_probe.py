import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import models_mae
DEV = "cuda"
def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
torch.manual_seed(0)
m = models_mae.octcube_vit_large_3dmae().to(DEV)
imgs = torch.rand(2, 1, 60, 256, 256, device=DEV)
noise = torch.rand(2, 5120, device=DEV)
ref = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    m.zero_grad()
    loss, pred, mask = m(imgs, mask_ratio=0.75, noise=noise)
    loss.backward()
    fin = all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None)
    hr = float(m.high_res_patch_embed.proj.weight.grad.abs().max())
    with torch.no_grad():
        l1, p1, _ = m(imgs[1:], mask_ratio=0.75, noise=noise[1:])
    r = rel(p1, pred[1:])
    gsum = float(sum(p.grad.double().abs().sum() for p in m.parameters() if p.grad is not None))
    if ref is None:
        ref = (float(loss), pred.detach().clone(), gsum)
    dp = float((pred.detach() - ref[1]).abs().max())
    print(f"iter {it}: loss {float(loss):.7f} finite {fin} hr_grad {hr} batch-indep rel {r:.2e} pred maxdiff vs iter0 {dp:.3e} gradsum rel {abs(gsum - ref[2]) / ref[2]:.2e}", flush=True)
