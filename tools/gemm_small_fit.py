#!/usr/bin/env python3
"""Fits / checks the cost model behind the small-launch GEMM kernel (csrc/gemm.hip plan128): every forward / dgrad Linear of the ViT-L
3-D MAE at B volumes per step, timed in one process through
   256   the kernels of rounds 1-5 (256-tile phased; ops.FORCE_SMALL_LAUNCH = -1)
   d4sS  gemm128d_kernel, 4-stage ring (one workgroup per CU), S-way k split
   d2sS  gemm128d_kernel, 2-stage ring (two workgroups per CU)
   auto  what the library picks by itself
and the weight-gradient pairs with the split rule of ops._splitk_for against the pre-round-6 one.

    python tools/gemm_small_fit.py [volumes ...]          (default 1 2 4 8)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octcubem_amd import ops  # noqa: E402

DEV = "cuda"
BF = ops.BF16


_side = None


def timeit(fn, iters=20):
    """Device time per call in us: `iters` calls captured in one HIP graph and replayed (no host time between the launches), on a
    stream whose split workspace exists before the capture."""
    global _side
    if _side is None:
        _side = torch.cuda.Stream()
    _side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(_side):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=_side):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    del g
    return s.elapsed_time(e) / (3 * iters) * 1e3      # us


def main():
    vols = [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]
    g = torch.Generator(device=DEV).manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, device=DEV, generator=g) * scale).to(BF)

    variants = [("256", -1, 1), ("d4s1", 4, 1), ("d4s2", 4, 2), ("d4s3", 4, 3), ("d4s4", 4, 4), ("d2s1", 2, 1), ("d2s2", 2, 2), ("auto", 0, 1)]
    print("shape".ljust(46) + "".join(n.rjust(8) for n, _, _ in variants) + "   plan (use, S, stages)")
    import ctypes
    lib = ops.load()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    tot = {n: 0.0 for n, _, _ in variants}
    for B in ([] if os.environ.get("FIT_SKIP_FWD") else vols):
        for name, M, D, nblk in (("enc", B * 1281, 1024, 24), ("dec", B * 5121, 512, 8)):
            for lname, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
                x = rnd(M, K); w = rnd(N, K, scale=K ** -0.5); b = torch.zeros(N, device=DEV)
                dy = rnd(M, N); res = torch.zeros(M, N, device=DEV)
                pre = rnd(M, K) if lname == "fc2" else None
                mode = {"qkv": "bf16", "proj": "resid", "fc1": "gelu", "fc2": "resid"}[lname]
                cs = torch.zeros(K, device=DEV)
                for kind, fn, dims in (("fwd", lambda: ops.linear_fwd(x, w, b, mode, res=res), (N, M, K)),
                                       ("dgrad", lambda: ops.linear_dgrad(dy, w, pre=pre, colsum=cs if pre is not None else None), (K, M, N))):
                    row = []
                    for vn, force, S in variants:
                        ops.FORCE_SMALL_LAUNCH, ops.FORCE_SPLITK = force, S
                        t = timeit(fn)
                        row.append(t)
                        tot[vn] += t * nblk
                    ops.FORCE_SMALL_LAUNCH, ops.FORCE_SPLITK = 0, 1
                    s_, st_ = ctypes.c_int(0), ctypes.c_int(0)
                    use = lib.octmae_gemm_small_plan(dims[0], dims[1], dims[2], ncu, 1, 1, ctypes.byref(s_), ctypes.byref(st_))
                    print(f"B={B} {name} {lname:4s} {kind:5s} [{dims[1]}x{dims[2]}]->{dims[0]}".ljust(46) + "".join(f"{t:8.1f}" for t in row) +
                          f"   {use} {s_.value} {st_.value}", flush=True)
    print("sum over the blocks of one step (us)".ljust(46) + "".join(f"{tot[n]:8.0f}" for n, _, _ in variants))

    # weight-gradient pairs: the split rule
    print("\nweight-gradient pairs (us): the library's choice; the 256-tile pair kernel with the split ops._splitk_for chooses, unsplit with "
          "each of its three epilogues, and with every split 1 .. 8 forced")
    for B in vols:
        for name, M, D in (("enc", B * 1281, 1024), ("dec", B * 5121, 512)):
            y1 = rnd(M, D); act = rnd(M, 4 * D); dpre = rnd(M, 4 * D); o = rnd(M, D); dqkv = rnd(M, 3 * D)
            gw2 = torch.zeros(D, 4 * D, device=DEV); gw1 = torch.zeros(4 * D, D, device=DEV)
            gwp = torch.zeros(D, D, device=DEV); gwq = torch.zeros(3 * D, D, device=DEV); gbq = torch.zeros(3 * D, device=DEV)
            for pname, first, second in (("fc2+fc1", (y1, act, gw2, None), (dpre, y1, gw1, None)), ("proj+qkv", (y1, o, gwp, None), (dqkv, y1, gwq, gbq))):
                tiles = sum(((dy.shape[1] + 255) // 256) * ((xx.shape[1] + 255) // 256) for dy, xx, _, _ in (first, second))
                chosen = ops._splitk_for(tiles, (M + 63) // 64, 256)
                orig = ops._splitk_for
                t_auto = timeit(lambda: ops.linear_wgrad_accum_pair(first, second))          # the library's choice (may be the 128-tile pair kernel)
                prev_small = ops.set_option("gemm_small", 0)                                    # everything below: the 256-tile pair kernel
                t_rule256 = timeit(lambda: ops.linear_wgrad_accum_pair(first, second))
                ops._splitk_for = lambda a, b_, c: 1
                t_s1 = []
                prev_opt = ops.set_option("wgrad_s1_atomic", 0)
                for form in (0, 1, 2):          # guarded read-modify-write / atomics / batched buffer read-modify-write
                    ops.set_option("wgrad_s1_atomic", form)
                    t_s1.append(timeit(lambda: ops.linear_wgrad_accum_pair(first, second)))
                ops.set_option("wgrad_s1_atomic", prev_opt)
                row = []
                for S in range(1, 9):
                    ops._splitk_for = lambda a, b_, c, S=S: S
                    row.append(timeit(lambda: ops.linear_wgrad_accum_pair(first, second)))
                ops._splitk_for = orig
                ops.set_option("gemm_small", prev_small)
                print(f"B={B} {name} {pname:9s} tiles {tiles:4d} ktiles {(M + 63) // 64:5d}  library: {t_auto:7.1f}  256-tile pair, rule S={chosen}: {t_rule256:7.1f}  S=1 rmw/atomic/batched: " + "/".join(f"{t:.1f}" for t in t_s1) + "   forced: " +
                      " ".join(f"{t:7.1f}" for t in row), flush=True)


if __name__ == "__main__":
    main()
