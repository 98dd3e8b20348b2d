import torch, sys
from octcubem_amd import ops
dev=torch.device("cuda")
for n in (331_600_000, 128*1281*1024):
    x=torch.randn(n,device=dev); y=torch.empty(n,dtype=torch.bfloat16,device=dev)
    ops.cast_bf16_into(x,y); torch.cuda.synchronize()
    ref = x.bfloat16()
    assert torch.equal(y, ref)
    ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.cast_bf16_into(x,y)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/10)
    t=sorted(ts)[2]
    print(f"n {n}: {t*1e3:.1f} us  {6*n/t/1e9:.2f} TB/s")
