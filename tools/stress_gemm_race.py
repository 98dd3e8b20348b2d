"""Race screen for the phased GEMM main loop: it must be BIT-identical to the two-stage loop (same MFMA order per accumulator)
on every launch, also while another stream hammers HBM/L2 to perturb the DMA timing."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops
torch.manual_seed(0)
side = torch.cuda.Stream()
junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
shapes = [(2562, 1024, 3072), (5121, 512, 2048), (4096, 4096, 1024), (1281, 1024, 1024), (3000, 768, 512), (2048, 256, 4096)]
bad = 0; n = 0; t0 = time.time()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
while time.time() - t0 < budget:
    for (M, K, N) in shapes:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); pre = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        with torch.cuda.stream(side):
            junk.mul_(1.0001)
        outs = {}
        for name, flags in (("two", (True, False)), ("pha", (False, True))):
            ops.FORCE_TWO_STAGE, ops.FORCE_PHASED = flags
            gw = torch.zeros(N, K, device="cuda")
            ops._gemm(dy, x, gw, N, K, M, N, K, K, 1, 1, ops.EPI_ACCUM, splitk=1)
            outs[name] = (ops.linear_fwd(x, w, None, "bf16"), ops.linear_dgrad(dy, w), ops.linear_dgrad(dy, w, pre=pre), gw)
        ops.FORCE_TWO_STAGE = ops.FORCE_PHASED = False
        if N >= 256 and K >= 256:
            # the paired weight-gradient launch without a split (no atomics): bit-identical to the single launches, twice over
            from octcubem_amd._lib import load
            gwa, gwb = torch.zeros(N, K, device="cuda"), torch.zeros(K, N, device="cuda")
            a = [dy.data_ptr(), x.data_ptr(), gwa.data_ptr(), 0, N, K, N, K, K, x.data_ptr(), dy.data_ptr(), gwb.data_ptr(), 0, K, N, K, N, N]
            rc = load().octmae_wgrad_accum_pair(*a, M, 1, torch.cuda.current_stream().cuda_stream)
            gwb1 = torch.zeros(K, N, device="cuda")
            ops._gemm(x, dy, gwb1, K, N, M, K, N, N, 1, 1, ops.EPI_ACCUM, splitk=1)
            for a_, b_ in ((gwa, outs["pha"][3]), (gwb, gwb1)):
                n += 1
                if rc != 0 or not torch.equal(a_, b_):
                    bad += 1
                    print("MISMATCH pair", (M, K, N), rc, flush=True)
        for a, b in zip(outs["two"], outs["pha"]):
            n += 1
            if not torch.equal(a, b):
                bad += 1
                print("MISMATCH", (M, K, N), float((a.float() - b.float()).abs().max()), flush=True)
torch.cuda.synchronize()
print(f"{n} comparisons, {bad} mismatches in {time.time() - t0:.0f} s")
