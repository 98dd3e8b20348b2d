"""CPU: the rounding-point model (oracle/bf16_points.py, oracle/bf16_points_mae.py) pinned on both sides, so that the evidence
chain  reference -> fp32/fp64 oracle -> rounding-point model -> HIP  has no link that only the HIP path itself validates
(VERDICT r03 "next" item 2):

  (a) with every rounding switched off (bf16_points.exact_arithmetic(): bf / f32 are the identity, GELU is the erf form) the
      model -- including its HAND-WRITTEN Block backward (attention with P, dS, delta, LSE and the exp2-domain pre-scaling; the
      fc2-dgrad x GELU'; LayerNorm backward) -- must equal float64 autograd over oracle/mae3d_ref.py, the restatement that
      tests/test_oracle_golden.py pins to the reference's own golden vectors (models_mae_joint_res_flash_attn.py:374-680), to
      <= 1e-10: loss, pred, every gradient tensor;
  (b) with the roundings ON it must land where bf16 operands are documented to land against the reference's golden vectors
      (DESIGN.md section 2): loss <= 1e-3, pred 5e-3 ... 1e-2, gradients <= 3e-2 -- i.e. it is the reference's computation with
      bf16 operand rounding and nothing else;
  (c) switching the roundings off changes nothing but the roundings: the two modes agree on masks and indices bit for bit.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import bf16_points as R
from oracle import bf16_points_mae as M
from oracle import mae3d_ref as O

D = torch.float64


def rel(a, b):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def _case(name, golden_dir):
    if name == "small":                      # the configuration of the reference golden mae3d_small.npz
        z = np.load(os.path.join(golden_dir, "mae3d_small.npz"))
        cfg = O.MAEConfig(**json.loads(str(z["cfg"])))
        P = O.init_params(cfg, seed=int(z["param_seed"]), bias_std=float(z["param_bias_std"]))
        return cfg, P, torch.from_numpy(z["imgs"]), torch.from_numpy(z["noise"]), float(z["mask_ratio"]), z
    # "mid": 3 + 2 Blocks, head_dim 64 and 32 -- the second configuration of tests/test_gpu_rounding_model.py
    cfg = O.MAEConfig(input_size=96, in_chans=1, embed_dim=256, depth=3, num_heads=4, decoder_embed_dim=128, decoder_depth=2,
                      decoder_num_heads=4, num_frames=15, t_patch_size=3, pred_t_dim=15, high_res_input_size=192)
    P = O.init_params(cfg, seed=3, bias_std=0.02)
    imgs = torch.rand(3, 1, 15, 96, 96, generator=torch.Generator().manual_seed(1))
    noise = torch.rand(3, cfg.num_patches, generator=torch.Generator().manual_seed(2))
    return cfg, P, imgs, noise, 0.75, None


@pytest.mark.parametrize("case", ["small", "mid"])
def test_rounding_model_without_roundings_is_the_fp64_oracle(case, golden_dir):
    cfg, P, imgs, noise, ratio, _ = _case(case, golden_dir)
    P64 = {k: v.to(D) for k, v in P.items()}
    loss_o, pred_o, mask_o, ids_o, Go = O.forward_backward(P64, imgs.to(D), cfg, ratio, noise)      # float64 autograd
    assert pred_o.dtype == D and loss_o.dtype == D
    with R.exact_arithmetic():
        loss_m, pred_m, mask_m, ids_m, Gm = M.forward_backward(P, imgs, cfg, ratio, noise)
    assert torch.equal(ids_m, ids_o) and torch.equal(mask_m.to(D), mask_o.to(D))
    assert abs(float(loss_m) - float(loss_o)) <= 1e-10 * abs(float(loss_o))
    assert rel(pred_m, pred_o) <= 1e-10
    total = float(torch.sqrt(sum(g.pow(2).sum() for g in Go.values())))
    worst = 0.0
    for k, g in Go.items():
        gn = float(g.norm())
        if gn < 1e-9 * total:      # unused (high_res_patch_embed) or mathematically zero (attn.k.bias): round-off on both sides
            assert float(Gm[k].norm()) <= 1e-9 * total, k
            continue
        e = rel(Gm[k], g)
        worst = max(worst, e)
        assert e <= 1e-10, (k, e)
    print(f"\n[rounding model, roundings off, {case}] loss {abs(float(loss_m) - float(loss_o)) / float(loss_o):.1e}, "
          f"pred {rel(pred_m, pred_o):.1e}, worst gradient {worst:.1e}")


def test_single_block_both_backward_forms_without_roundings():
    """bf16_points.block_forward / block_backward alone (head_dim 64 and 32; the fused and the two-kernel attention backward
    differ only in WHICH operand carries scale*log2e, so without roundings both are the exact gradient)."""
    g = torch.Generator().manual_seed(11)
    for C, H, N, B in ((128, 2, 70, 2), (64, 2, 130, 3)):
        shapes = O.param_shapes(O.MAEConfig(input_size=32, in_chans=1, embed_dim=C, depth=1, num_heads=H, decoder_embed_dim=C,
                                            decoder_depth=1, decoder_num_heads=H, num_frames=3, t_patch_size=3, pred_t_dim=3,
                                            high_res_input_size=64))
        P = {k[len("blocks.0."):]: (torch.randn(s, generator=g, dtype=D) * (0.08 if len(s) == 2 else 0.05) + (1.0 if "norm" in k and k.endswith("weight") else 0.0))
             for k, s in shapes.items() if k.startswith("blocks.0.")}
        x = torch.randn(B, N, C, generator=g, dtype=D); d3 = torch.randn(B, N, C, generator=g, dtype=D)
        Pg = {f"blocks.0.{k}": v.clone().requires_grad_(True) for k, v in P.items()}
        xg = x.clone().requires_grad_(True)
        y = O.block(xg, Pg, "blocks.0", H, 1e-6)
        y.backward(d3)
        for fused in (True, False):
            with R.exact_arithmetic():
                x3, dx, G = R.block_forward_backward(P, x, d3, H, fused_bwd=fused)
            assert rel(x3, y.detach()) <= 1e-12 and rel(dx, xg.grad) <= 1e-11
            for k in P:
                gref = Pg[f"blocks.0.{k}"].grad
                if k == "attn.k.bias":          # mathematically zero (softmax is invariant to a per-query shift)
                    assert float(G[k].abs().max()) <= 1e-12 * float(Pg["blocks.0.attn.q.bias"].grad.abs().max())
                    continue
                assert rel(G[k], gref) <= 1e-10, (k, fused, rel(G[k], gref))


def test_rounding_model_with_roundings_lands_where_bf16_operands_do(golden_dir):
    """Against the REFERENCE's golden vectors (tests/golden/mae3d_small.npz, written by oracle/gen_golden.py from the reference's
    own MaskedAutoencoderViT): the bounds are the documented cost of bf16 MFMA operands (measured on this
    configuration: loss 5e-5, pred 6.2e-3, worst gradient 2.0e-2), so a model that rounded anything else, or rounded wrongly,
    would not fit them -- and the lower bounds show that the roundings are really on."""
    cfg, P, imgs, noise, ratio, z = _case("small", golden_dir)
    loss, pred, mask, ids, G = M.forward_backward(P, imgs, cfg, ratio, noise)
    assert torch.equal(ids, torch.from_numpy(z["ids_restore"])) and torch.equal(mask, torch.from_numpy(z["mask"]))
    e_loss = abs(float(loss) - float(z["loss"])) / abs(float(z["loss"]))
    e_pred = rel(pred, z["pred"])
    assert e_loss <= 1e-3, e_loss
    assert 1e-3 <= e_pred <= 1e-2, e_pred
    total = float(np.sqrt(sum(float(z[f"gnorm/{k}"]) ** 2 for k in G)))
    errs = {}
    for k, g in G.items():
        gn = float(z[f"gnorm/{k}"])
        ref = torch.from_numpy(z[f"grad/{k}"])
        mine = (g if g.numel() <= 8192 else g.flatten()[::7]).reshape(ref.shape)
        if gn < 1e-3 * total:
            assert float((mine - ref.double()).norm()) <= 2e-3 * total, k
            continue
        errs[k] = rel(mine, ref)
        assert abs(float(g.norm()) - gn) <= 1.5e-2 * gn, k
    worst = max(errs, key=errs.get)
    med = sorted(errs.values())[len(errs) // 2]
    print(f"\n[rounding model vs reference golden, small] loss {e_loss:.2e}, pred {e_pred:.2e}, worst gradient {worst} "
          f"{errs[worst]:.2e}, median {med:.2e}")
    assert errs[worst] <= 3e-2 and 1e-3 <= med <= 1.5e-2, (worst, errs[worst], med)


def test_error_of_the_rounding_model_scales_with_the_operand_epsilon(golden_dir):
    """What tests/test_gpu_f16_parity.py measures on the GPU (the same kernels on bfloat16 and on half operands), predicted on the
    CPU: the model with 8, 11 and 14 significant bits at every rounding point (unbounded exponent = loss scaling), against float64
    autograd over the oracle, at the reference golden's configuration.  Typical quantities follow the epsilon (x 8 per 3 bits: pred,
    the median gradient tensor); a SINGLE tensor's error is one realisation of a rounding pattern and follows it only within a factor
    (the worst tensor, blocks.1.attn.k/q.weight: x 5.4 from 8 to 11 bits, x 10 from 11 to 14) -- which is why the GPU test asks
    >= 6 x of the former and >= 3.5 x of the latter.  Measured: pred 6.25e-3 / 7.07e-4 / 9.65e-5, median 5.86e-3 / 8.23e-4 /
    1.08e-4, worst 2.09e-2 / 3.86e-3 / 3.85e-4; the HIP half build lands at 6.86e-4, 8.03e-4 and 3.89e-3 (profiles/r05_f16_parity.json)."""
    cfg, P, imgs, noise, ratio, _ = _case("small", golden_dir)
    P64 = {k: v.to(D) for k, v in P.items()}
    loss_o, pred_o, _, _, Go = O.forward_backward(P64, imgs.to(D), cfg, ratio, noise)
    total = float(torch.sqrt(sum(g.pow(2).sum() for g in Go.values())))
    res = {}
    for bits in (8, 11, 14):
        with R.operand_bits(bits):
            loss_m, pred_m, _, _, Gm = M.forward_backward(P, imgs, cfg, ratio, noise)
        errs = sorted(rel(Gm[k], Go[k]) for k in Go if float(Go[k].norm()) >= 1e-3 * total)
        res[bits] = (rel(pred_m, pred_o), errs[len(errs) // 2], errs[-1], abs(float(loss_m) - float(loss_o)) / abs(float(loss_o)))
    with R.operand_type(torch.bfloat16):                 # 8 bits with bfloat16's own exponent range: the same numbers
        _, pred_b, _, _, _ = M.forward_backward(P, imgs, cfg, ratio, noise)
    assert abs(rel(pred_b, pred_o) - res[8][0]) <= 1e-6
    for lo, hi in ((8, 11), (11, 14)):
        assert 6.0 <= res[lo][0] / res[hi][0] <= 11.0, ("pred", lo, hi, res)
        assert 5.5 <= res[lo][1] / res[hi][1] <= 11.0, ("median gradient", lo, hi, res)
        assert 3.5 <= res[lo][2] / res[hi][2] <= 16.0, ("worst gradient tensor", lo, hi, res)
    # the north star's 1e-3 on pred / the median gradient tensor is where half operands land; bfloat16 cannot
    assert res[11][0] <= 1e-3 and res[11][1] <= 1.5e-3 and res[11][2] <= 5e-3 and res[11][3] <= 1e-4
    assert res[8][0] >= 4e-3 and res[8][2] >= 1e-2
