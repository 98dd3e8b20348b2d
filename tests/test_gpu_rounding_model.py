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
                                     (128, 4, 197, 2),      # head_dim 32, shorter than a key block
                                     (256, 4, 46, 3),       # head_dim 64, shorter than one 64-key tile (the "mid" encoder)
                                     (128, 4, 181, 3)])     # head_dim 32, the "mid" decoder
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
        # What two CORRECT evaluations of this block may differ by: the model against itself with its input perturbed by 1e-7
        # relative (fp32 epsilon).  A value that lands within that of a bf16 tie rounds the other way, the flip (2^-9 relative
        # on one element) feeds the next rounding point, and so on through the seven rounding points of a Block's backward:
        # for few rows (B * N = 138) the model's own gradients move by several 1e-3.  Bounds: 1e-4 (x3) / 1e-3 (gradients; 2e-3
        # for the four LayerNorm vectors), or 2 x that self-sensitivity (largest of four draws) where it is larger.
        sens = {}
        for seed in range(4):                    # the flips are few and discrete: take the largest of four draws
            xp = x.double() * (1.0 + 1e-7 * torch.randn(x.shape, generator=torch.Generator().manual_seed(99 + seed), dtype=torch.float64))
            x3_p, dx_p, Gp = R.block_forward_backward(P, xp, dx3, H, 1e-6, fused_bwd=fused)
            one = {"x3": rel(x3_p, x3_r), "dx": rel(dx_p, dx_r)}
            one.update({"g:" + k: rel(Gp[k], G[k]) for k in G if k != "attn.k.bias"})
            sens = {k: max(v, sens.get(k, 0.0)) for k, v in one.items()}
        assert errs["x3"] <= max(1e-4, 2.0 * sens["x3"]), (errs["x3"], sens["x3"])
        bad = {k: (v, sens[k]) for k, v in errs.items() if v > max(2e-3 if "norm" in k else 1e-3, 2.0 * sens[k])}
        assert not bad, bad
        ws = max(sens, key=sens.get)
        print(f"[model vs itself, input perturbed 1e-7] x3 {sens['x3']:.2e}, dx {sens['dx']:.2e}, largest {ws} {sens[ws]:.2e}")
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


def _mae_case(name, golden_dir):
    import json, os
    import numpy as np
    if name == "small":
        z = np.load(os.path.join(golden_dir, "mae3d_small.npz"))
        cfg = O.MAEConfig(**json.loads(str(z["cfg"])))
        P = O.init_params(cfg, seed=int(z["param_seed"]), bias_std=float(z["param_bias_std"]))
        return cfg, P, torch.from_numpy(z["imgs"]), torch.from_numpy(z["noise"]), float(z["mask_ratio"])
    cfg = O.MAEConfig(input_size=96, in_chans=1, embed_dim=256, depth=3, num_heads=4, decoder_embed_dim=128, decoder_depth=2,
                      decoder_num_heads=4, num_frames=15, t_patch_size=3, pred_t_dim=15, high_res_input_size=192)
    P = O.init_params(cfg, seed=3, bias_std=0.02)
    imgs = torch.rand(3, 1, 15, 96, 96, generator=torch.Generator().manual_seed(1))
    noise = torch.rand(3, cfg.num_patches, generator=torch.Generator().manual_seed(2))
    return cfg, P, imgs, noise, 0.75


@pytest.mark.parametrize("case", ["small", "mid"])
def test_whole_mae_step_matches_the_rounding_point_model_to_1e_3(case, golden_dir):
    """the 1e-3 evidence chain END TO END.  oracle/bf16_points_mae.py evaluates the whole 3-D MAE step
    (patch embedding, encoder assembly, every Block, decoder_embed, decoder assembly, decoder_pred, masked MSE and the backward of
    all of it; models_mae_joint_res_flash_attn.py:374-680) in float64 with bf16 roundings at the HIP path's rounding points.
    Asserted: the loss to <= 1e-3 (measured 4e-9 / 2e-5); pred and EVERY gradient tensor to <= 1e-3 (2e-3 for the LayerNorm
    vectors) OR 2 x the model's own sensitivity to an fp32-epsilon perturbation of its parameters, whichever is larger (see
    the comment at the assertions: a chain of bf16 roundings amplifies a 1e-7 difference to 4e-3 on pred, so no two correct
    implementations can agree better); tensors whose gradient is below 1e-3 of the global norm are bounded absolutely against
    the global norm.  Each kernel in isolation agrees with its rounding model to <= 2e-4 (test_attention_matches_its_rounding_model,
    test_block_matches_the_rounding_point_model_to_1e_3), and both differ from the plain fp32 oracle by the 6e-3 (pred) ...
    2e-2 (gradients) that bf16 operands cost."""
    from functools import partial
    from octcubem_amd import models_mae
    from oracle import bf16_points_mae as M
    from tests.conftest import parity
    cfg, P, imgs, noise, ratio = _mae_case(case, golden_dir)
    loss_r, pred_r, mask_r, ids_r, G = M.forward_backward(P, imgs, cfg, ratio, noise)
    m = models_mae.MaskedAutoencoderViT(
        input_size=cfg.input_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depth=cfg.depth,
        num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
        decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio, norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps),
        norm_pix_loss=cfg.norm_pix_loss, num_frames=cfg.num_frames, t_patch_size=cfg.t_patch_size, sep_pos_embed=True,
        cls_embed=True, pred_t_dim=cfg.pred_t_dim, high_res_input_size=cfg.high_res_input_size)
    m.load_state_dict(P, strict=True)
    m = m.to(DEV).train()
    loss, pred, mask = m(imgs.to(DEV), mask_ratio=ratio, noise=noise.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(mask.cpu(), mask_r) and torch.equal(m._ids_restore.cpu(), ids_r)
    e_loss = abs(float(loss) - float(loss_r)) / float(loss_r)
    e_pred = rel(pred, pred_r)
    total = float(torch.sqrt(sum(g.double().pow(2).sum() for g in G.values())))
    errs, small_abs = {}, {}
    for k, p_ in m.named_parameters():
        gr = G[k]
        gn = float(gr.norm())
        mine = p_.grad if p_.grad is not None else torch.zeros_like(p_)
        if gn < 1e-6 * total:        # unused (high_res_patch_embed) or mathematically zero (attn.k.bias): noise on both sides
            assert float(mine.double().norm()) <= 1e-4 * total, k
            continue
        if gn >= 1e-3 * total:
            errs[k] = rel(mine, gr)
        else:
            small_abs[k] = float((mine.double().cpu() - gr).norm()) / total
    worst = max(errs, key=errs.get)
    top = sorted(errs, key=errs.get, reverse=True)[:4]
    print(f"\n[whole-model rounding-point model, {case}] loss {e_loss:.2e}, pred {e_pred:.2e}, gradients: " +
          ", ".join(f"{k} {errs[k]:.2e}" for k in top) + f"; median {sorted(errs.values())[len(errs) // 2]:.2e}")
    # The model against ITSELF with every parameter perturbed by 1e-7 relative (fp32 epsilon), two draws: what two correct
    # evaluations of this bf16-operand pipeline may differ by.  A pre-rounding value within 1e-7 of a bf16 tie rounds the other
    # way; the flip (2^-9 on one element) reaches the next rounding point, ... -- through 2 x (depth + decoder_depth) x 7 rounding
    # points the pipeline is chaotic at the 4e-3 level: measured on the CPU, pred moves by 3.6e-3 ... 3.9e-3 and the gradients by
    # 1e-3 (median) ... 1.8e-2 (q / k weights of the last encoder block, whose dS = P (dP - delta) cancels).  The HIP path must
    # be no further from the model than 2 x that (largest of four draws; or 1e-3, whichever is larger) -- and is in fact CLOSER to it than the
    # perturbed model is.
    sens_pred, sens = 0.0, {k: 0.0 for k in errs}
    for seed in range(4):
        gq = torch.Generator().manual_seed(500 + seed)
        Pq = {k: v.double() * (1.0 + 1e-7 * torch.randn(v.shape, generator=gq, dtype=torch.float64)) for k, v in P.items()}
        _, pred_q, _, _, Gq = M.forward_backward(Pq, imgs, cfg, ratio, noise)
        sens_pred = max(sens_pred, rel(pred_q, pred_r))
        for k in errs:
            sens[k] = max(sens[k], rel(Gq[k], G[k]))
    ws = max(sens, key=sens.get)
    print(f"[model vs itself, parameters perturbed 1e-7, {case}] pred {sens_pred:.2e}, gradients: worst {ws} {sens[ws]:.2e}, "
          f"median {sorted(sens.values())[len(sens) // 2]:.2e}")
    # Every "2 x the model's own sensitivity" bound has an ABSOLUTE ceiling, and the sensitivities themselves are bounded at 2 x what
    # was measured (pred 3.6e-3 ... 3.9e-3, median gradient 9.6e-4 ... 1.8e-3, worst gradient tensor 1.2e-2 ... 1.7e-2): a model that
    # became more chaotic -- or a sensitivity probe that broke -- fails here instead of widening the bound.
    CEIL = 1.5e-2
    parity(f"rp_model/{case}/loss", e_loss, 1e-3)
    parity(f"rp_model/{case}/pred", e_pred, min(max(1e-3, 2.0 * sens_pred), CEIL))
    parity(f"rp_model/{case}/pred_self_sensitivity", sens_pred, 8e-3)
    parity(f"rp_model/{case}/worst_grad", errs[worst], min(max(2e-3, 2.0 * sens[worst]), CEIL))
    parity(f"rp_model/{case}/worst_grad_self_sensitivity", max(sens.values()), 3.5e-2)
    parity(f"rp_model/{case}/median_grad", sorted(errs.values())[len(errs) // 2], 2e-3)
    parity(f"rp_model/{case}/median_grad_self_sensitivity", sorted(sens.values())[len(sens) // 2], 4e-3)
    bad = {k: (v, sens[k]) for k, v in errs.items()
           if v > min(max(2e-3 if ("norm" in k and k.endswith(("weight", "bias"))) else 1e-3, 2.0 * sens[k]), CEIL)}
    assert not bad, bad
    assert all(v <= 1e-3 for v in small_abs.values()), small_abs
    # and what the same quantities are against the PLAIN fp32 oracle (no roundings): the distance test_gpu_model.py tolerates
    loss_o, pred_o, _, _, Go = O.forward_backward(P, imgs, cfg, ratio, noise)
    plain = {k: rel(p_.grad, Go[k]) for k, p_ in m.named_parameters() if k in errs}
    wp = max(plain, key=plain.get)
    print(f"[plain fp32 oracle, {case}] loss {abs(float(loss) - float(loss_o)) / float(loss_o):.2e}, pred {rel(pred, pred_o):.2e}, "
          f"worst gradient {wp} {plain[wp]:.2e}")


@pytest.mark.parametrize("B,H,N,HD", [(3, 4, 46, 64), (3, 4, 181, 32), (2, 2, 300, 64), (3, 4, 600, 32), (1, 2, 1281, 64), (1, 2, 1537, 32)])
def test_attention_matches_its_rounding_model(B, H, N, HD):
    """The attention kernels ALONE against their float64 + bf16-rounding model (the attention part of oracle/bf16_points.py) on
    identical inputs: forward (optimistic, no running max: P = exp2(bf(q scale log2e) . k), l = sum of the unrounded P, the PV
    product on bf(P)) and both backward forms (P and dS rounded to bf16 for their products, K pre-scaled in the kernels that keep
    the key on the lane).  <= 2e-4: one kernel, one rounding point deep, so tie flips stay rare -- the kernels are arithmetically
    exact up to accumulation order and the hardware exp2.  Covers sequences shorter than one key tile, the single-key tail and,
    at head_dim 32, the one-wave-per-SIMD main kernel."""
    F = torch.float32
    D_ = torch.float64
    LOG2E = 1.4426950408889634

    def bf(t):
        return t.to(F).to(torch.bfloat16).to(D_)

    g = torch.Generator().manual_seed(N + B)
    qkv = (torch.randn(B * N, 3 * H * HD, generator=g) * 1.2).to(torch.bfloat16)
    do = torch.randn(B * N, H * HD, generator=g).to(torch.bfloat16)
    scale = HD ** -0.5
    o, lse = ops.attn_fwd(qkv.to(DEV), B, N, H, HD, scale)
    q, k, v = qkv.to(D_).view(B, N, 3, H, HD).permute(2, 0, 3, 1, 4)
    sc2 = torch.tensor(scale, dtype=F) * torch.tensor(LOG2E, dtype=F)
    qs = (q.to(F) * sc2).to(torch.bfloat16).to(D_)
    Pm = torch.exp2(qs @ k.transpose(-1, -2))
    o_m = bf((bf(Pm) @ v) / Pm.sum(-1, keepdim=True)).transpose(1, 2).reshape(B * N, H * HD)
    assert rel(o, o_m) <= 2e-4, rel(o, o_m)
    dod = do.to(D_).view(B, N, H, HD).transpose(1, 2)
    od = o.double().cpu().view(B, N, H, HD).transpose(1, 2)
    delta = (dod * od).sum(-1, keepdim=True).to(F).to(D_)
    nl = (-(lse.cpu().to(F) * torch.tensor(LOG2E, dtype=F))).to(D_).view(B, H, N, 1)
    ks = (k.to(F) * sc2).to(torch.bfloat16).to(D_)
    Pk = torch.exp2(q @ ks.transpose(-1, -2) + nl)
    dPk = dod @ v.transpose(-1, -2) - delta
    dV = bf(bf(Pk).transpose(-1, -2) @ dod)
    dK = bf(scale * (bf(Pk * dPk).transpose(-1, -2) @ q))
    dQ = {True: bf(scale * (bf(Pk * dPk) @ k)), False: bf(scale * (bf(torch.exp2(qs @ k.transpose(-1, -2) + nl) * dPk) @ k))}
    for fused in (True, False):
        d = ops.attn_bwd(qkv.to(DEV), o, do.to(DEV), lse, B, N, H, HD, scale, fused=fused).double().cpu().view(B, N, 3, H, HD).permute(2, 0, 3, 1, 4)
        e = (rel(d[0], dQ[fused]), rel(d[1], dK), rel(d[2], dV))
        assert max(e) <= 2e-4, (fused, e)
