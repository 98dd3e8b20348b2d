"""The library GEMM PyTorch dispatches to (hipBLASLt / rocBLAS, plain bf16 output) on the ViT-L shapes: forward, dgrad, wgrad.
A yardstick for the hand-written kernels (profiles/r02_blas_yardstick.txt), not part of the product path."""
import torch, time
def t(f, iters=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
B=128
for (M,K,N) in ((B*1281,1024,3072),(B*1281,1024,4096),(B*1281,4096,1024),(B*5121,512,1536),(B*5121,512,2048)):
    x=torch.randn(M,K,device='cuda').bfloat16(); w=(torch.randn(N,K,device='cuda')*K**-0.5).bfloat16(); dy=torch.randn(M,N,device='cuda').bfloat16()
    fl=2.0*M*K*N
    a=t(lambda: x@w.t()); b=t(lambda: dy@w); c=t(lambda: dy.t()@x)
    print(f"[{M}x{K}]x[{N}x{K}]: fwd {a:7.1f} us {fl/a/1e6:6.0f} TF/s | dgrad {b:7.1f} us {fl/b/1e6:6.0f} | wgrad {c:7.1f} us {fl/c/1e6:6.0f}", flush=True)
