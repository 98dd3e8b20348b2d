"""Index-width screen for large micro-batches: the gradients of ONE micro-batch of B volumes must equal the accumulated gradients
of TWO micro-batches of B/2 on the same volumes and the same masking noise (fp32 accumulation order differs only in the weight
gradients' split-K atomics and row order: agreement to ~1e-5 relative; a 32-bit offset overflow would be off by O(1))."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import models_mae
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = models_mae.octcube_vit_large_3dmae().to(dev).train()
arena = model.prepare()
g = torch.Generator(device=dev).manual_seed(5)
x = torch.rand(B, 1, 60, 256, 256, device=dev, generator=g)
noise = torch.rand(B, 5120, device=dev, generator=g)

def grads(parts):
    model.prepare()
    for p in model.parameters():
        if p.grad is not None: p.grad.zero_()
    tot = 0.0
    for i in range(parts):
        sl = slice(i * B // parts, (i + 1) * B // parts)
        loss, _, _ = model(x[sl], mask_ratio=0.75, noise=noise[sl])
        (loss / parts).backward()
        tot += float(loss) / parts
    torch.cuda.synchronize()
    return tot, torch.cat([p.grad.flatten().double() for p in model.parameters() if p.grad is not None])

l1, g1 = grads(1)
l2, g2 = grads(2)
rel = float((g1 - g2).norm() / g2.norm())
print(f"B={B}: loss one-shot {l1:.6f}  two halves {l2:.6f}  |dg|/|g| = {rel:.3e}  max|dg| = {float((g1 - g2).abs().max()):.3e}  |g| = {float(g2.norm()):.4f}")
print("peak memory GB", torch.cuda.max_memory_allocated() / 2**30)
assert abs(l1 - l2) < 2e-5 * abs(l2) + 1e-7 and rel < 1e-3
print("OK")
