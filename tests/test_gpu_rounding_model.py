"""Kernel arithmetic error vs operand rounding, separated (VERDICT r01 item 1c / ADVICE: "feed identical bf16-rounded operands
to an fp64 reference so kernel error can be bounded at about 1e-3, separately from operand rounding").

oracle/bf16_points.py evaluates one transformer Block, forward and backward, in float64 with a bf16 rounding at exactly the
points where the HIP path rounds.  Every output of octcubem_amd's fused Block (ops.BlockFn: LayerNorm, GEMM epilogues, flash
attention forward, fused or two-kernel attention backward, GELU / GELU' epilogues, weight / bias gradients) must agree with it
to <= 1e-3 relative L2 -- the north star's bound -- while the SAME outputs differ from the plain fp32 oracle (no roundings)
by the 5e-3 ... 3e-2 that bf16 operands cost.  The measured numbers are printed (-s) and tabulated in DESIGN.md section 2."""
from functools import partial

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from octcubem_amd import ops, video_vit
from oracle import bf16_points as R
from oracle import mae3d_ref as O

DEV = "cuda"
# polynomial GELU (forward only) vs erf on the float64 model: 4.2e-5 worst (mlp.fc2.weight), it was 3.3e-3 with the polynomial GELU'
POLY_GELU_BOUND = 1e-4


def rel(a, b):
    a = torch.as_tensor(a).detach().double().flatten().cpu(); b = torch.as_tensor(b).detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("C,H,N,B", [(128, 2, 300, 2),      # head_dim 64: one 256-key block + a 44-key tail in the fused form
                                     (64, 2, 600, 2),       # head_dim 32: one 512-key block + an 88-key tail
                                     (128, 4, 197, 2)])     # head_dim 32, shorter than a key block
def test_block_matches_the_rounding_point_model_to_1e_3(C, H, N, B):
    g = torch.Generator().manual_seed(C + N)
    blk = video_vit.Block(C, H, 4.0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    P = {}
    with torch.no_grad():
        for n_, p_ in blk.named_parameters():
            if p_.dim() > 1:
                a = (6.0 / (p_.shape[0] + p_.shape[1])) ** 0.5
                p_.copy_((torch.rand(p_.shape, generator=g) * 2 - 1) * a)
            else:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.05 + (1.0 if "norm" in n_ and n_.endswith("weight") else 0.0))
            P[n_] = p_.detach().clone()
    blk = blk.to(DEV).train()
    x = torch.randn(B, N, C, generator=g) * 1.5
    dx3 = torch.randn(B, N, C, generator=g)
    hd = C // H
    results = {}
    for fused in (True, False):
        old = dict(ops.ATTN_BWD_FUSED)
        ops.ATTN_BWD_FUSED[hd] = fused
        try:
            for p_ in blk.parameters():
                if p_.grad is not None:
                    p_.grad.zero_()              # gradients are views of the flat arena and ACCUMULATE: clear, keep the binding
            xg = x.to(DEV).requires_grad_(True)
            out = blk(xg)
            out.backward(dx3.to(DEV))
            torch.cuda.synchronize()
        finally:
            ops.ATTN_BWD_FUSED.update(old)
        x3_r, dx_r, G = R.block_forward_backward(P, x, dx3, H, 1e-6, fused_bwd=fused)
        x3_e, dx_e, Ge = R.block_forward_backward(P, x, dx3, H, 1e-6, fused_bwd=fused, exact_gelu=True)
        errs = {"x3": rel(out, x3_r), "dx": rel(xg.grad, dx_r)}
        for n_, p_ in blk.named_parameters():
            if n_ == "attn.k.bias":
                # d/d(k bias) is ZERO in exact arithmetic (a constant added to every key shifts each softmax row by a
                # constant): both sides hold rounding noise only -- bounded against the q-bias gradient instead
                qn = float(G["attn.q.bias"].norm())
                assert float(p_.grad.double().norm()) <= 2e-2 * qn and float(G[n_].norm()) <= 2e-2 * qn
                continue
            errs["g:" + n_] = rel(p_.grad, G[n_])
        results[fused] = errs
        top = sorted(errs, key=errs.get, reverse=True)[:3]
        print(f"\n[rounding-point model] C={C} H={H} N={N} fused_bwd={fused}: x3 {errs['x3']:.2e}, dx {errs['dx']:.2e}, "
              "largest: " + ", ".join(f"{k} {errs[k]:.2e}" for k in top))
        assert errs["x3"] <= 1e-4, errs["x3"]
        # 1e-3 for the activations' gradient and every weight / bias gradient; the four LayerNorm vectors (sums over only
        # B*N rows of bf16-rounded dy, where a handful of tie flips show) get 2e-3
        bad = {k: v for k, v in errs.items() if v > (2e-3 if "norm" in k else 1e-3)}
        assert not bad, bad
        # what the fc1 epilogue's polynomial GELU (|Phi error| <= 1.4e-5) costs against the erf form, same roundings (GELU' IS the
        # erf form since round 3: the 3.3e-3 its polynomial used to cost on the q / k weight gradients is gone)
        poly = {"x3": rel(x3_r, x3_e), "dx": rel(dx_r, dx_e)}
        poly.update({"g:" + k: rel(G[k], Ge[k]) for k in G if k != "attn.k.bias"})
        wp = max(poly, key=poly.get)
        print(f"[polynomial GELU vs erf] x3 {poly['x3']:.2e}, dx {poly['dx']:.2e}, largest {wp} {poly[wp]:.2e}")
        assert max(poly.values()) <= POLY_GELU_BOUND, poly
    # the plain fp32 oracle (no rounding anywhere) on the same inputs: the distance the tests in test_gpu_model.py tolerate
    Pr = {f"blocks.0.{k}": v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.block(xr, Pr, "blocks.0", H, 1e-6)
    yr.backward(dx3)
    plain = {"x3": rel(out, yr), "dx": rel(xg.grad, xr.grad)}
    for n_, p_ in blk.named_parameters():
        if n_ != "attn.k.bias":
            plain["g:" + n_] = rel(p_.grad, Pr[f"blocks.0.{n_}"].grad)
    worst = max(plain, key=plain.get)
    print(f"[plain fp32 oracle]    C={C} H={H} N={N}: x3 {plain['x3']:.2e}, dx {plain['dx']:.2e}, worst gradient {worst} {plain[worst]:.2e}")
    assert max(plain.values()) <= 5e-2
