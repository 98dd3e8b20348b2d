"""GPU parity of the whole hot path: the HIP model (octcubem_amd.models_mae) against
  (1) the golden vectors produced by the REAL reference (tests/golden/mae3d_small*.npz), and
  (2) the CPU oracle on the same seeded inputs, up to the full ViT-L / 60x256x256 configuration.

Bounds (bf16 MFMA operands, fp32 accumulation / residual stream / statistics) are the errors MEASURED on MI355X in round 2
x ~1.5, each through tests.conftest.parity (recorded to gpurun_out/parity_measured.json; table in DESIGN.md section 2):
loss-level quantities 1e-4 ... 4e-4 (the north star's 1e-3 with margin), pred rel-L2 <= 9.7e-3, per-tensor gradient rel-L2
<= 3e-2 (median 1.2e-2), global gradient norm <= 1.5e-3; masking indices bit-exact.  Where the element-wise 5e-3 ... 2e-2 comes
from is shown in tests/test_gpu_rounding_model.py: the same computation with bf16 rounding at the HIP path's rounding points
reproduces the HIP results to <= 1e-3.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from octcubem_amd import models_mae, misc, ops, optim as foptim, lr_sched, engine_pretrain
from oracle import mae3d_ref as O
from tests.conftest import parity

DEV = "cuda"


def rel(a, b):
    a = torch.as_tensor(a).double().flatten().cpu(); b = torch.as_tensor(b).double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(cfg: O.MAEConfig, P=None, **extra):
    from functools import partial
    m = models_mae.MaskedAutoencoderViT(**extra, **dict(
        input_size=cfg.input_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depth=cfg.depth,
        num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
        decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio, norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps),
        norm_pix_loss=cfg.norm_pix_loss, num_frames=cfg.num_frames, t_patch_size=cfg.t_patch_size, sep_pos_embed=True,
        cls_embed=True, pred_t_dim=cfg.pred_t_dim, high_res_input_size=cfg.high_res_input_size))
    if P is not None:
        missing = m.load_state_dict(P, strict=True)
    return m.to(DEV)


def small(golden_dir):
    z = np.load(os.path.join(golden_dir, "mae3d_small.npz"))
    cfg = O.MAEConfig(**json.loads(str(z["cfg"])))
    P = O.init_params(cfg, seed=int(z["param_seed"]), bias_std=float(z["param_bias_std"]))
    return z, cfg, P


def test_state_dict_keys_match_reference_layout(golden_dir):
    z, cfg, P = small(golden_dir)
    m = build(cfg)
    assert set(m.state_dict().keys()) == set(O.param_shapes(cfg).keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(O.param_shapes(cfg)[k]), k
    full = models_mae.octcube_vit_large_3dmae()
    assert sum(p.numel() for p in full.parameters()) == 331_632_384


def test_small_model_vs_reference_golden(golden_dir):
    z, cfg, P = small(golden_dir)
    m = build(cfg, P)
    m.train()
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)
    (loss, fl), pred, mask = m(imgs, mask_ratio=float(z["mask_ratio"]), frame_loss=True, noise=noise)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(mask.cpu(), torch.from_numpy(z["mask"]))                       # bit-exact
    assert torch.equal(m._ids_restore.cpu(), torch.from_numpy(z["ids_restore"]))      # bit-exact
    parity("small/loss", abs(float(loss) - float(z["loss"])) / float(z["loss"]), 2e-4)          # measured 6.8e-5 (r02)
    parity("small/pred", rel(pred, z["pred"]), 9.5e-3)                                          # measured 6.2e-3: bf16 operand rounding, see test_gpu_rounding_model.py
    parity("small/frame_losses", rel(fl, z["frame_losses"]), 3e-4)                             # measured 1.9e-4
    worst_g = 0.0
    total_ref = float(np.sqrt(sum(float(z[k]) ** 2 for k in z.files if k.startswith("gnorm/"))))
    sq = 0.0
    for k, p in m.named_parameters():
        gn = float(z[f"gnorm/{k}"])
        g = p.grad
        if gn == 0.0:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        sq += float(g.double().pow(2).sum())
        if gn < 1e-6 * total_ref:
            # mathematically zero gradient (attn.k.bias: a per-row constant shift of the scores leaves softmax unchanged);
            # the reference holds fp32 rounding noise there, so only the magnitude is comparable
            assert float(g.double().norm()) <= 1e-4 * total_ref, k
            continue
        ref = torch.from_numpy(z[f"grad/{k}"])
        mine = g.cpu() if g.numel() <= 8192 else g.cpu().flatten()[::7]
        if gn >= 1e-3 * total_ref:
            worst_g = max(worst_g, rel(mine.reshape(ref.shape), ref))
        else:                                    # tensors far below the global norm: absolute, against the global norm
            assert float((mine.reshape(ref.shape).double() - ref.double()).norm()) <= 2e-3 * total_ref, k
    parity("small/worst_grad", worst_g, 3e-2)                                                   # measured 2.0e-2
    parity("small/grad_norm", abs(sq ** 0.5 - total_ref) / total_ref, 1e-3)                      # measured 3.3e-4


def test_small_model_variants_vs_reference_golden(golden_dir):
    z, cfg, P = small(golden_dir)
    v = np.load(os.path.join(golden_dir, "mae3d_small_variants.npz"))
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)
    m = build(O.MAEConfig(**{**cfg.__dict__, "norm_pix_loss": True}), P)
    with torch.no_grad():
        loss, pred, mask = m(imgs, mask_ratio=0.9, noise=noise)
    assert torch.equal(mask.cpu(), torch.from_numpy(v["mask_r90"]))
    parity("variants/loss_normpix_r90", abs(float(loss) - float(v["loss_normpix_r90"])) / float(v["loss_normpix_r90"]), 1e-4)   # measured 2.4e-5
    parity("variants/pred_r90", rel(pred, v["pred_r90"]), 8.5e-3)                                # measured 5.5e-3
    # high-res (2-D / 512-style) branch through high_res_patch_embed, un-interpolated spatial table, no temporal table
    m2 = build(cfg, P)
    with torch.no_grad():
        loss, pred, mask = m2(torch.from_numpy(v["imgs_hr"]).to(DEV), mask_ratio=0.75, noise=torch.from_numpy(v["noise_hr"]).to(DEV))
    assert torch.equal(mask.cpu(), torch.from_numpy(v["mask_hr"]))
    parity("variants/loss_hr", abs(float(loss) - float(v["loss_hr"])) / float(v["loss_hr"]), 3e-4)   # measured 1.7e-4
    parity("variants/pred_hr", rel(pred, v["pred_hr"]), 9.7e-3)                                  # measured 6.5e-3


def test_mid_model_vs_oracle_seeded():
    """A mid-size configuration the oracle finishes in seconds: heads of 64 and 32, N not a multiple of 64."""
    cfg = O.MAEConfig(input_size=96, in_chans=1, embed_dim=256, depth=3, num_heads=4, decoder_embed_dim=128, decoder_depth=2,
                      decoder_num_heads=4, num_frames=15, t_patch_size=3, pred_t_dim=15, high_res_input_size=192)
    P = O.init_params(cfg, seed=3, bias_std=0.02)
    imgs = torch.rand(3, 1, 15, 96, 96, generator=torch.Generator().manual_seed(1))
    noise = torch.rand(3, cfg.num_patches, generator=torch.Generator().manual_seed(2))
    loss_r, pred_r, mask_r, ids_r, grads_r = O.forward_backward(P, imgs, cfg, 0.75, noise)
    m = build(cfg, P)
    loss, pred, mask = m(imgs.to(DEV), mask_ratio=0.75, noise=noise.to(DEV))
    loss.backward()
    assert torch.equal(mask.cpu(), mask_r) and torch.equal(m._ids_restore.cpu(), ids_r)
    parity("mid/loss", abs(float(loss) - float(loss_r)) / float(loss_r), 2e-4)                    # measured 8.2e-5
    parity("mid/pred", rel(pred, pred_r), 9.7e-3)                                                 # measured 6.5e-3
    total = float(O.grad_norm(grads_r.values()))
    errs = {}
    for k, p in m.named_parameters():
        gr = grads_r[k]
        if float(gr.norm()) < 1e-6 * total:          # exactly-zero (unused) or mathematically-zero (attn.k.bias) gradients
            assert p.grad is None or float(p.grad.double().norm()) <= 1e-4 * total, k
        elif float(gr.norm()) < 1e-3 * total:         # far below the global norm: absolute, against the global norm
            assert float((p.grad.cpu().double() - gr.double()).norm()) <= 2e-3 * total, k
        else:
            errs[k] = rel(p.grad, gr)
    worst = max((v, k) for k, v in errs.items())
    parity("mid/worst_grad", worst[0], 3e-2)                                                      # measured 2.0e-2
    parity("mid/median_grad", sorted(errs.values())[len(errs) // 2], 1.2e-2)                      # measured 8.1e-3
    print("mid model: loss rel %.2e pred rel %.2e worst grad rel %.2e (%s) median %.2e" % (
        abs(float(loss) - float(loss_r)) / float(loss_r), rel(pred, pred_r), worst[0], worst[1], sorted(errs.values())[len(errs) // 2]))


def test_train_step_matches_oracle_adamw():
    """Two optimizer steps through NativeScalerWithGradNormCount + FusedAdamW + add_weight_decay + lr schedule, vs the
    oracle's forward/backward + AdamW restatement (grad-norm value, updated parameters)."""
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=1,
                      decoder_num_heads=2, num_frames=6, t_patch_size=3, pred_t_dim=6, high_res_input_size=128)
    P = O.init_params(cfg, seed=5, bias_std=0.02)
    m = build(cfg, P)
    groups = misc.add_weight_decay(m, 0.05)
    opt = foptim.FusedAdamW(groups, lr=1e-3, betas=(0.9, 0.95))
    scaler = misc.NativeScalerWithGradNormCount(fp32=True)

    class A: pass
    a = A(); a.lr = 1e-3; a.min_lr = 0.0; a.warmup_epochs = 1; a.epochs = 10
    nd, _ = O.weight_decay_groups([(k, tuple(v.shape)) for k, v in P.items()], 0.05)
    nd = set(nd)
    Pr = {k: v.clone() for k, v in P.items()}
    Mr = {k: torch.zeros_like(v) for k, v in P.items()}; Vr = {k: torch.zeros_like(v) for k, v in P.items()}
    Pt = None                                          # the oracle's OWN two-step trajectory (never re-seeded from the HIP model)
    for step in (1, 2):
        imgs = torch.rand(2, 1, 6, 64, 64, generator=torch.Generator().manual_seed(10 + step))
        noise = torch.rand(2, cfg.num_patches, generator=torch.Generator().manual_seed(20 + step))
        lr = lr_sched.adjust_learning_rate(opt, 0.5 * step, a)
        opt.zero_grad()
        p_before2 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()} if step == 2 else None
        loss, _, _ = m(imgs.to(DEV), mask_ratio=0.75, noise=noise.to(DEV))
        norm = scaler(loss, opt, parameters=m.parameters(), clip_grad=None)
        if step == 2:
            # Adam's first step moves every weight by ~lr * sign(g), so weights whose gradient sits below the bf16 noise floor
            # land on different sides in the two trajectories; the second step is therefore compared from the SAME point (the
            # HIP model's parameters after step 1, the oracle's own moments): loss / gradient-norm parity of a step, not the
            # sensitivity of a two-step trajectory to sign flips (which moved from 8.7e-4 to 2.1e-3 when the GELU fits changed)
            # -- and, beside it, the original two-step trajectory check: the oracle's own step-1 parameters (Pt), bound stated.
            loss_t, _, _, _, Gt = O.forward_backward(Pt, imgs, cfg, 0.75, noise)
            parity("train_step/loss2_trajectory", abs(float(loss) - float(loss_t)) / float(loss_t), 3e-4)     # measured 8.4e-5 (r04)
            parity("train_step/grad_norm2_trajectory",
                   abs(float(norm) - float(O.grad_norm(Gt.values()))) / float(O.grad_norm(Gt.values())), 3.5e-3)      # measured 2.1e-3
            # parameters after step 1: every element moved by ~lr (Adam), the two trajectories may differ by 2 lr where a
            # gradient below the bf16 noise floor changed sign, never by more
            for k in Pt:
                assert float((p_before2[k].double() - Pt[k].double()).abs().max()) <= 2.05 * lr1 + 1e-7, k
            Pr = {k: p_before2[k].clone() for k in Pr}
        loss_r, _, _, _, G = O.forward_backward(Pr, imgs, cfg, 0.75, noise)
        parity(f"train_step/loss{step}", abs(float(loss) - float(loss_r)) / float(loss_r), 2e-4)      # measured 4.8e-5 / 1.2e-4 (r03)
        parity(f"train_step/grad_norm{step}", abs(float(norm) - float(O.grad_norm(G.values()))) / float(O.grad_norm(G.values())), 1.5e-3)   # measured 1.0e-3
        assert abs(lr - O.cosine_lr(0.5 * step, 1e-3, 0.0, 1, 10)) < 1e-12
        for k in Pr:
            Pr[k], Mr[k], Vr[k] = O.adamw_step(Pr[k], G[k], Mr[k], Vr[k], step, lr, 0.9, 0.95, 1e-8, 0.0 if k in nd else 0.05)
        if step == 1:
            Pt, lr1 = {k: v.clone() for k, v in Pr.items()}, lr
    sd = m.state_dict()
    # Adam's first steps move every weight by ~lr * sign(g) whatever the gradient scale, so elements whose gradient is
    # below the bf16 noise floor may flip: compare the UPDATE VECTORS in aggregate (cosine), not element-wise.
    # (FusedAdamW itself is checked to 1e-6 on identical gradients in test_gpu_kernels.py.)
    dot = sum(float(((sd[k].cpu().double() - P[k].double()) * (Pr[k].double() - P[k].double())).sum()) for k in Pr)
    n1 = sum(float((sd[k].cpu().double() - P[k].double()).pow(2).sum()) for k in Pr) ** 0.5
    n2 = sum(float((Pr[k].double() - P[k].double()).pow(2).sum()) for k in Pr) ** 0.5
    parity("train_step/1-cos(update)", 1.0 - dot / (n1 * n2), 3e-3)                               # measured 1.4e-3


def test_zero_grad_touches_only_its_own_slices_of_the_gradient_arena():
    """FusedAdamW.zero_grad fills contiguous runs of the flat gradient arena; a slice that belongs to a parameter of ANOTHER
    optimizer (or to a frozen / foreign parameter in between) must keep its contents (opt1.step();
    opt1.zero_grad(); opt2.step() saw zero gradients)."""
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=1,
                      decoder_num_heads=2, num_frames=6, t_patch_size=3, pred_t_dim=6, high_res_input_size=128)
    m = build(cfg)
    m.prepare()                                                      # binds the arena: every param.grad becomes a view of it
    named = [(k, p) for k, p in m.named_parameters() if p.grad is not None]
    assert len(named) > 40
    for split in ("halves", "interleaved", "small_vs_rest"):
        if split == "halves":
            a = named[:len(named) // 2]; b = named[len(named) // 2:]
        elif split == "interleaved":
            a = named[0::2]; b = named[1::2]
        else:                                                        # the tiny tensors (biases, cls, mask token) vs the matrices
            a = [kp for kp in named if kp[1].numel() <= 128]; b = [kp for kp in named if kp[1].numel() > 128]
        o1 = foptim.FusedAdamW([p for _, p in a], lr=1e-3); o2 = foptim.FusedAdamW([p for _, p in b], lr=1e-3)
        m.arena.grad.fill_(3.0)
        o1.zero_grad()
        assert all(float(p.grad.abs().max()) == 0.0 for _, p in a), split
        assert all(float(p.grad.min()) == 3.0 for _, p in b), split
        o2.zero_grad()
        assert all(float(p.grad.abs().max()) == 0.0 for _, p in named), split
    o = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=1e-3)
    m.arena.grad.fill_(3.0)
    o.zero_grad()
    assert all(float(p.grad.abs().max()) == 0.0 for _, p in named)


def test_engine_train_one_epoch_runs_and_learns():
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=1,
                      decoder_num_heads=2, num_frames=6, t_patch_size=3, pred_t_dim=6, high_res_input_size=128)
    torch.manual_seed(0)
    m = build(cfg)
    opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
    scaler = misc.NativeScalerWithGradNormCount(fp32=True)

    class A: pass
    a = A(); a.lr = 2e-3; a.min_lr = 0.0; a.warmup_epochs = 0; a.epochs = 4; a.accum_iter = 2; a.mask_ratio = 0.75; a.clip_grad = 1.0
    g = torch.Generator().manual_seed(0)
    base = torch.rand(1, 1, 6, 64, 64, generator=g)
    data = [(base + 0.01 * torch.rand(2, 1, 6, 64, 64, generator=g), None) for _ in range(8)]
    first = engine_pretrain.train_one_epoch(m, data, opt, torch.device(DEV), 0, scaler, args=a)
    for ep in (1, 2, 3):
        last = engine_pretrain.train_one_epoch(m, data, opt, torch.device(DEV), ep, scaler, args=a)
    assert last["loss"] < first["loss"]


@pytest.mark.skipif(os.environ.get("OCTMAE_SKIP_VITL", "0") == "1", reason="full-size run disabled")
def test_vitl_full_size_forward_vs_reference_pins(golden_dir):
    """BASELINE configuration (ViT-L, 1x60x256x256, mask 0.75) against scalar pins captured from the real reference."""
    pins = np.load(os.path.join(golden_dir, "vitl_pins.npz"))
    P = O.init_params(O.VIT_L, seed=0)
    m = models_mae.octcube_vit_large_3dmae()
    m.load_state_dict(P, strict=True)
    m = m.to(DEV)
    imgs = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(0))
    torch.manual_seed(int(pins["noise_seed"]))
    noise = torch.rand(1, 5120)
    with torch.no_grad():
        loss, pred, mask = m(imgs.to(DEV), mask_ratio=0.75, noise=noise.to(DEV))
    assert torch.equal(m._ids_restore.cpu().int(), torch.from_numpy(pins["ids_restore"]))      # bit-exact at L = 5120
    assert float(mask.sum()) == 3840.0
    parity("vitl/loss", abs(float(loss) - float(pins["loss"])) / float(pins["loss"]), 4e-4)              # measured 2.1e-4
    samp = pred.flatten()[torch.from_numpy(pins["pred_idx"]).to(DEV)]
    parity("vitl/pred_samples", rel(samp, pins["pred_samples"]), 1.2e-2)                          # measured 8.0e-3
    parity("vitl/pred_l2", abs(float(pred.double().norm()) - float(pins["pred_l2"])) / float(pins["pred_l2"]), 3e-4)   # measured 1.3e-4


def test_full_size_properties_batch2():
    """Size-independent properties at BASELINE size: permutation round trip, mask count, finite loss and gradients,
    batch independence of the forward (row b of a batch == the same volume alone)."""
    torch.manual_seed(0)
    m = models_mae.octcube_vit_large_3dmae().to(DEV)
    imgs = torch.rand(2, 1, 60, 256, 256, device=DEV)
    noise = torch.rand(2, 5120, device=DEV)
    loss, pred, mask = m(imgs, mask_ratio=0.75, noise=noise)
    loss.backward()
    ir = m._ids_restore
    assert torch.equal(torch.sort(ir, dim=1).values, torch.arange(5120, device=DEV).expand(2, -1))
    assert float(mask.sum()) == 2 * 3840 and torch.equal(mask, (ir >= 1280).float())
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    assert float(m.high_res_patch_embed.proj.weight.grad.abs().max()) == 0.0
    # Row b of a batch == the same volume alone.  Exactly (1e-6) when no GEMM is k-split: every kernel then adds a row's products in
    # the same order whatever the number of rows.  The small-launch kernel's deterministic k split (round 6) is chosen from the TILE
    # COUNT, i.e. from the number of token rows: one volume alone splits the K = 3072 / 4096 reductions two ways, two volumes three ways
    # -- another order of the same fp32 additions, a 1-ulp flip of a 16-bit rounding here and there, and from there the chain's own
    # sensitivity (a 1e-7 perturbation of the parameters moves pred by 3.6-3.9e-3).  As with the reference's
    # cuBLAS heuristics, bit-equality across batch sizes is not a property of the fast path; it is one of the unsplit path.
    with torch.no_grad():
        l1, p1, _ = m(imgs[1:], mask_ratio=0.75, noise=noise[1:])
        prev, ops.SPLIT_WS = ops.SPLIT_WS, False
        try:
            _, p2u, _ = m(imgs, mask_ratio=0.75, noise=noise)
            _, p1u, _ = m(imgs[1:], mask_ratio=0.75, noise=noise[1:])
        finally:
            ops.SPLIT_WS = prev
    assert rel(p1u, p2u[1:]) <= 1e-6
    assert rel(p1, pred[1:].detach()) <= 1e-2 and rel(p1, p1u) <= 1e-2          # measured 3.7e-3


def test_operand_copy_guard_catches_a_write_behind_the_version_counters(golden_dir):
    """The arena re-casts its 16-bit operand copy only when a parameter's PyTorch version counter moved (arena.refresh_lp).  A write
    no counter sees -- ``p.data.mul_()`` here; ``dist.broadcast(p.data)`` or an EMA swap through ``.data`` in the wild -- leaves the
    MFMA operands stale WITHOUT an error; the documented remedy is model.invalidate_lp(), and the sampled guard must
    notice a forgotten call, warn and repair.  Also: after FusedAdamW steps the copy its kernel wrote equals the cast of the master
    weights exactly (the guard raises no false alarm in a training loop)."""
    import warnings
    from octcubem_amd import arena as A
    z, cfg, P = small(golden_dir)
    m = build(cfg, P).train()
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)
    opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
    scaler = misc.NativeScalerWithGradNormCount(fp32=True)
    prev, A.CHECK_LP_EVERY = A.CHECK_LP_EVERY, 1                    # check at every skipped refresh
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)          # three optimizer steps: never a (false) alarm
            for _ in range(3):
                opt.zero_grad()
                loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
                scaler(loss, opt, parameters=m.parameters())
            assert m.arena.lp_matches()
            with torch.no_grad():
                ref_loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        v0 = m.decoder_pred.weight._version
        m.decoder_pred.weight.data.mul_(2.0)                        # behind the version counter
        assert m.decoder_pred.weight._version == v0 and not m.arena.lp_matches()
        with pytest.warns(RuntimeWarning, match="operand copy"), torch.no_grad():
            m(imgs, mask_ratio=0.75, noise=noise)                   # the guard notices, warns, re-casts
        assert m.arena.lp_matches()
        with torch.no_grad():
            loss2, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        assert abs(float(loss2) - float(ref_loss)) > 1e-3 * float(ref_loss)          # the doubled weights are in use now
        m.decoder_pred.weight.data.mul_(0.5)                        # the documented way: say so
        m.invalidate_lp()
        with warnings.catch_warnings(), torch.no_grad():
            warnings.simplefilter("error", RuntimeWarning)
            loss3, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        assert float(loss3) == float(ref_loss)
    finally:
        A.CHECK_LP_EVERY = prev


def test_fused_block_backward_with_shared_activation():
    """BlockFn hands the bf16 copy / column sums of its input gradient to the upstream Block through a side channel that
    is only valid when autograd delivers that very tensor.  Here the first Block's output feeds BOTH the second Block and
    the loss, so autograd sums two contributions into a new tensor and the side channel must be ignored (work redone
    locally, nothing double counted).  Checked against the oracle's blocks."""
    from octcubem_amd import video_vit
    from functools import partial
    D, H, B, N = 128, 2, 2, 70
    g = torch.Generator().manual_seed(0)
    blocks = [video_vit.Block(D, H, 4.0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6)) for _ in range(2)]
    P = {}
    for i, blk in enumerate(blocks):
        for n_, p_ in blk.named_parameters():
            with torch.no_grad():
                p_.copy_(torch.randn(p_.shape, generator=g) * (0.05 if p_.dim() > 1 else 0.02) + (1.0 if "norm" in n_ and n_.endswith("weight") else 0.0))
            P[f"blocks.{i}.{n_}"] = p_.detach().clone()
    model = torch.nn.ModuleList(blocks).to(DEV)
    x = torch.randn(B, N, D, generator=g)
    w1 = torch.randn(B, N, D, generator=g); w2 = torch.randn(B, N, D, generator=g)
    xg = x.to(DEV).requires_grad_(True)
    h1 = model[0](xg)
    h2 = model[1](h1)
    loss = (h2 * w2.to(DEV)).sum() + (h1 * w1.to(DEV)).sum()          # h1 is used twice
    loss.backward()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    r1 = O.block(xr, Pr, "blocks.0", H, 1e-6)
    r2 = O.block(r1, Pr, "blocks.1", H, 1e-6)
    ((r2 * w2).sum() + (r1 * w1).sum()).backward()
    assert rel(xg.grad, xr.grad) <= 3e-2
    for i, blk in enumerate(model):
        for n_, p_ in blk.named_parameters():
            ref = Pr[f"blocks.{i}.{n_}"].grad
            if float(ref.norm()) < 1e-6 * float(xr.grad.norm()):
                continue
            assert rel(p_.grad, ref) <= 5e-2, (i, n_, rel(p_.grad, ref))


@pytest.mark.parametrize("attn_kind", ["qkv_separate", "timm_fused"])
def test_fused_block_stochastic_depth_vs_oracle(attn_kind):
    """Training-mode Blocks with drop_path > 0: the per-sample keep factors (0 or 1/keep) are folded into the two residual
    epilogues and into the bf16 gradient casts.  With the draws pinned, forward and every gradient must match the oracle's
    block evaluated with the same factors (timm DropPath semantics), including samples whose branch is dropped."""
    from octcubem_amd import video_vit
    from functools import partial
    D, H, B, N = 128, 2, 4, 70
    g = torch.Generator().manual_seed(5)
    cls = video_vit.Block if attn_kind == "qkv_separate" else video_vit.TimmBlock
    blocks = [cls(D, H, 4.0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), drop_path=0.25) for _ in range(2)]
    P = {}
    for i, blk in enumerate(blocks):
        for n_, p_ in blk.named_parameters():
            with torch.no_grad():
                p_.copy_(torch.randn(p_.shape, generator=g) * (0.05 if p_.dim() > 1 else 0.02) + (1.0 if "norm" in n_ and n_.endswith("weight") else 0.0))
            if ".qkv." in n_:
                for j, nm in enumerate("qkv"):
                    P[f"blocks.{i}.{n_.replace('.qkv.', '.' + nm + '.')}"] = p_.detach().chunk(3, 0)[j].clone()
            else:
                P[f"blocks.{i}.{n_}"] = p_.detach().clone()
    model = torch.nn.ModuleList(blocks).to(DEV).train()
    draws = [torch.tensor(v) / 0.75 for v in ([1., 0., 1., 1.], [1., 1., 0., 1.], [0., 1., 1., 1.], [1., 0., 1., 0.])]
    it = iter(draws)
    for blk in model:
        blk.drop_path.sample = lambda batch, device: next(it).to(device)
    x = torch.randn(B, N, D, generator=g)
    w = torch.randn(B, N, D, generator=g)
    xg = x.to(DEV).requires_grad_(True)
    out = model[1](model[0](xg))
    (out * w.to(DEV)).sum().backward()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    r = O.block(O.block(xr, Pr, "blocks.0", H, 1e-6, (draws[0], draws[1])), Pr, "blocks.1", H, 1e-6, (draws[2], draws[3]))
    (r * w).sum().backward()
    assert rel(out, r) <= 1e-2 and rel(xg.grad, xr.grad) <= 3e-2
    for i, blk in enumerate(model):
        for n_, p_ in blk.named_parameters():
            if ".qkv." in n_:
                ref = torch.cat([Pr[f"blocks.{i}.{n_.replace('.qkv.', '.' + nm + '.')}"].grad for nm in "qkv"], 0)
            else:
                ref = Pr[f"blocks.{i}.{n_}"].grad
            if float(ref.norm()) < 1e-6 * float(xr.grad.norm()) or n_.endswith("attn.k.bias"):
                continue
            assert rel(p_.grad, ref) <= 5e-2, (i, n_, rel(p_.grad, ref))
    # eval mode ignores drop_path entirely
    model.eval()
    with torch.no_grad():
        e = model[1](model[0](x.to(DEV)))
        re_ = O.block(O.block(x, P, "blocks.0", H, 1e-6), P, "blocks.1", H, 1e-6)
    assert rel(e, re_) <= 1e-2


def test_flash_compat_drops_the_final_residual_like_the_flash_path():
    """flash_compat=True: the last encoder / decoder block hands only its MLP branch to the final norm (what the reference
    computes when built with use_flash_attn=True).  flash-attn itself cannot run here, so this mode is
    pinned to the oracle's restatement of flash-attn 2.5.2's prenorm Block only (parity unpinned beyond that)."""
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=2,
                      decoder_num_heads=2, num_frames=6, t_patch_size=3, pred_t_dim=6, high_res_input_size=128)
    P = O.init_params(cfg, seed=11, bias_std=0.02)
    imgs = torch.rand(2, 1, 6, 64, 64, generator=torch.Generator().manual_seed(4))
    noise = torch.rand(2, cfg.num_patches, generator=torch.Generator().manual_seed(5))
    loss_r, pred_r, mask_r, ids_r, grads_r = O.forward_backward(P, imgs, cfg, 0.75, noise, flash_compat=True)
    loss_std, pred_std, _, _, _ = O.forward_backward(P, imgs, cfg, 0.75, noise)
    assert rel(pred_r, pred_std) > 5e-2                      # the two semantics really differ
    m = build(cfg, P, flash_compat=True)
    loss, pred, mask = m(imgs.to(DEV), mask_ratio=0.75, noise=noise.to(DEV))
    loss.backward()
    assert torch.equal(mask.cpu(), mask_r)
    assert abs(float(loss) - float(loss_r)) <= 2e-3 * float(loss_r) and rel(pred, pred_r) <= 1e-2
    total = float(O.grad_norm(grads_r.values()))
    for k, p in m.named_parameters():
        gr = grads_r[k]
        if float(gr.norm()) < 1e-6 * total:
            assert p.grad is None or float(p.grad.double().norm()) <= 1e-4 * total, k
        else:
            assert rel(p.grad, gr) <= 5e-2, (k, rel(p.grad, gr))


def test_load_pretrained_accepts_flash_layout_and_other_grids():
    """checkpoint.load_pretrained: a flash-layout checkpoint whose positional tables come from another grid (24 -> 32
    spatial on the high-res table, 4 -> 2 temporal) loads into the native layout and computes the same thing as the oracle
    fed the converted weights."""
    from octcubem_amd import checkpoint as CK
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=2,
                      decoder_num_heads=2, num_frames=6, t_patch_size=3, pred_t_dim=6, high_res_input_size=128)
    P = O.init_params(cfg, seed=12, bias_std=0.02)
    g = torch.Generator().manual_seed(8)
    ck = CK.to_flash_layout(dict(P))
    assert any(".mixer.Wqkv." in k for k in ck) and not any(".attn.q." in k for k in ck)
    ck["pos_embed_spatial"] = torch.randn(1, 36, 128, generator=g) * 0.02          # 6x6 grid in the checkpoint, model has 8x8
    ck["pos_embed_temporal"] = torch.randn(1, 4, 128, generator=g) * 0.02          # 4 temporal slots, model has 2
    m = build(cfg)
    missing, unexpected = CK.load_pretrained(m, {"model": ck}, strict=False)
    assert not unexpected and not [k for k in missing if "decoder_pos_embed" not in k]
    P2 = dict(P)
    P2["pos_embed_spatial"] = torch.nn.functional.interpolate(ck["pos_embed_spatial"].reshape(1, 6, 6, 128).permute(0, 3, 1, 2), size=(8, 8),
                                                               mode="bicubic", align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)
    P2["pos_embed_temporal"] = torch.nn.functional.interpolate(ck["pos_embed_temporal"].permute(0, 2, 1), size=2, mode="linear",
                                                                align_corners=False).permute(0, 2, 1)
    sd = m.state_dict()
    for k in P2:
        assert torch.allclose(sd[k].cpu(), P2[k], atol=0), k
    imgs = torch.rand(2, 1, 6, 64, 64, generator=g)
    noise = torch.rand(2, cfg.num_patches, generator=g)
    loss_r, pred_r, _, _ = O.forward(P2, imgs, cfg, 0.75, noise)
    loss, pred, _ = m(imgs.to(DEV), mask_ratio=0.75, noise=noise.to(DEV))
    assert abs(float(loss) - float(loss_r)) <= 2e-3 * float(loss_r) and rel(pred, pred_r) <= 1e-2


@pytest.mark.parametrize("train_drop", [False, True])
def test_flash_block_factory_seam(train_drop):
    """video_vit.create_block: the flash-attn factory signature and ``x, residual = blk(x, residual)`` loop idiom
    (models_mae_joint_res_flash_attn.py:131-149, 480-483).  hidden + residual after the loop is the standard pre-norm stream
    (oracle blocks, same weights re-keyed from mixer.Wqkv / mixer.out_proj); hidden alone is the flash_compat output.
    With stochastic depth the draws are pinned: drop_path1 of block i acts on the MLP branch of block i-1."""
    from octcubem_amd import video_vit
    from octcubem_amd import checkpoint as CK
    from functools import partial
    D, H, B, N = 128, 4, 3, 70
    g = torch.Generator().manual_seed(9)
    dpr = [0.25, 0.25] if train_drop else [0.0, 0.0]
    blocks = [video_vit.create_block(D, H, 4.0, True, 0.0, 0.0, drop_path1=dpr[i - 1] if i > 0 else 0.0, drop_path2=dpr[i],
                                     norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), act_layer=torch.nn.GELU, use_flash_attn=True,
                                     fused_bias_fc=False, fused_mlp=False, fused_dropout_add_ln=False, layer_idx=i, n_layer=2,
                                     last_layer_subset=False) for i in range(2)]
    sd = {}
    for i, blk in enumerate(blocks):
        for n_, p_ in blk.named_parameters():
            with torch.no_grad():
                p_.copy_(torch.randn(p_.shape, generator=g) * (0.05 if p_.dim() > 1 else 0.02) + (1.0 if "norm" in n_ and n_.endswith("weight") else 0.0))
            sd[f"blocks.{i}.{n_}"] = p_.detach().clone()
    P = CK.to_native_layout(sd)
    model = torch.nn.ModuleList(blocks).to(DEV).train(train_drop)
    draws = None
    if train_drop:      # block 0: drop_path2 on its attention branch; block 1: drop_path1 on block 0's MLP branch, drop_path2
        draws = {"b0_dp2": torch.tensor([1., 1., 0.]) / 0.75, "b1_dp1": torch.tensor([1., 0., 1.]) / 0.75,
                 "b1_dp2": torch.tensor([0., 1., 1.]) / 0.75}
        model[0].drop_path2.sample = lambda b, d: draws["b0_dp2"].to(d)
        model[1].drop_path1.sample = lambda b, d: draws["b1_dp1"].to(d)
        model[1].drop_path2.sample = lambda b, d: draws["b1_dp2"].to(d)
    x = torch.randn(B, N, D, generator=g)
    w1 = torch.randn(B, N, D, generator=g); w2 = torch.randn(B, N, D, generator=g)
    xg = x.to(DEV).requires_grad_(True)
    hidden, residual = xg, None
    for blk in model:
        hidden, residual = blk(hidden, residual)
    ((hidden * w1.to(DEV)).sum() + (residual * w2.to(DEV)).sum()).backward()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    one = torch.ones(B)
    s_b0 = (draws["b0_dp2"], draws["b1_dp1"]) if train_drop else None
    r1 = O.block(xr, Pr, "blocks.0", H, 1e-6, s_b0)
    # block 1, split by hand: stream before its MLP and the MLP branch
    Cc = D
    h = torch.nn.functional.layer_norm(r1, (Cc,), Pr["blocks.1.norm1.weight"], Pr["blocks.1.norm1.bias"], 1e-6)
    a = O.attention(h, Pr, "blocks.1.attn", H)
    res_ref = r1 + (a * draws["b1_dp2"].view(-1, 1, 1) if train_drop else a)
    h2 = torch.nn.functional.layer_norm(res_ref, (Cc,), Pr["blocks.1.norm2.weight"], Pr["blocks.1.norm2.bias"], 1e-6)
    hid_ref = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(h2, Pr["blocks.1.mlp.fc1.weight"], Pr["blocks.1.mlp.fc1.bias"])),
                                         Pr["blocks.1.mlp.fc2.weight"], Pr["blocks.1.mlp.fc2.bias"])
    ((hid_ref * w1).sum() + (res_ref * w2).sum()).backward()
    assert rel(hidden, hid_ref) <= 1e-2 and rel(residual, res_ref) <= 1e-2
    if not train_drop:
        std = O.block(O.block(x, P, "blocks.0", H, 1e-6), P, "blocks.1", H, 1e-6)
        assert rel(hidden + residual, std) <= 1e-2
    assert rel(xg.grad, xr.grad) <= 3e-2
    mine = CK.to_native_layout({f"blocks.{i}.{n_}": p_.grad.detach().cpu() for i, blk in enumerate(model) for n_, p_ in blk.named_parameters()})
    for k, gr in ((k, v.grad) for k, v in Pr.items()):
        if float(gr.norm()) < 1e-6 * float(xr.grad.norm()) or k.endswith("attn.k.bias"):
            continue
        assert rel(mine[k], gr) <= 5e-2, (k, rel(mine[k], gr))


def test_use_flash_attn_model_has_flash_keys_and_flash_semantics():
    """use_flash_attn=True builds create_block blocks: state_dict keys blocks.i.mixer.{Wqkv,out_proj}, and the output is the
    flash path's (final residual dropped) -- identical to the oracle's flash_compat restatement on the re-keyed weights."""
    from octcubem_amd import checkpoint as CK
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=2,
                      decoder_num_heads=2, num_frames=6, t_patch_size=3, pred_t_dim=6, high_res_input_size=128)
    P = O.init_params(cfg, seed=13, bias_std=0.02)
    m = build(cfg, None, use_flash_attn=True)
    keys = set(m.state_dict())
    assert "blocks.0.mixer.Wqkv.weight" in keys and "decoder_blocks.1.mixer.out_proj.bias" in keys and not any(".attn." in k for k in keys)
    assert keys == set(CK.to_flash_layout(P))
    missing, unexpected = m.load_state_dict_to_backbone(dict(P), strict=True)          # native-layout checkpoint into the flash model
    assert not missing and not unexpected
    imgs = torch.rand(2, 1, 6, 64, 64, generator=torch.Generator().manual_seed(4))
    noise = torch.rand(2, cfg.num_patches, generator=torch.Generator().manual_seed(5))
    loss_r, pred_r, mask_r, _, grads_r = O.forward_backward(P, imgs, cfg, 0.75, noise, flash_compat=True)
    loss, pred, mask = m(imgs.to(DEV), mask_ratio=0.75, noise=noise.to(DEV))
    loss.backward()
    assert torch.equal(mask.cpu(), mask_r)
    assert abs(float(loss) - float(loss_r)) <= 2e-3 * float(loss_r) and rel(pred, pred_r) <= 1e-2
    mine = CK.to_native_layout({k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None})
    total = float(O.grad_norm(grads_r.values()))
    for k, gr in grads_r.items():
        if float(gr.norm()) < 1e-6 * total or k.endswith("attn.k.bias"):
            continue
        assert rel(mine[k], gr) <= 5e-2, (k, rel(mine[k], gr))


def test_dynamic_loss_scale_state_machine(golden_dir):
    """NativeScalerWithGradNormCount(fp32=False, dynamic_loss_scale=True): the reference's fp16 GradScaler behaviour
    (custom_util/misc.py:311-344) on top of the bf16 path -- a power-of-two scale changes nothing (same update as the unscaled
    step, bit for bit: every product scales exactly), a non-finite gradient skips the step and halves the scale, growth_interval
    good steps double it, and the state round-trips through the "amp_scaler" checkpoint entry."""
    z, cfg, P = small(golden_dir)
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)

    def one(scaler, m, opt, loss_mul=1.0):
        opt.zero_grad()
        loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        return scaler(loss * loss_mul, opt, parameters=list(m.parameters()), clip_grad=1.0)

    ma, mb = build(cfg, P), build(cfg, P)
    oa = foptim.FusedAdamW(misc.add_weight_decay(ma, 0.05), lr=1e-3, betas=(0.9, 0.95))
    ob = foptim.FusedAdamW(misc.add_weight_decay(mb, 0.05), lr=1e-3, betas=(0.9, 0.95))
    sa = misc.NativeScalerWithGradNormCount(fp32=True)
    sb = misc.NativeScalerWithGradNormCount(fp32=False, dynamic_loss_scale=True, init_scale=65536.0, growth_interval=2)
    assert sb.get_scale() == 65536.0 and sa.get_scale() == 1.0
    na, nb = one(sa, ma, oa), one(sb, mb, ob)
    torch.cuda.synchronize()
    assert abs(float(na) - float(nb)) <= 1e-6 * float(na)
    for (k, pa), (_, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(pa, pb), k                               # the scaled step IS the unscaled step
    before = {k: v.clone() for k, v in mb.state_dict().items()}
    nn_ = one(sb, mb, ob, loss_mul=float("inf"))                    # non-finite gradients: skipped, scale halves, tracker resets
    assert not torch.isfinite(nn_) and sb.last_step_skipped and sb.get_scale() == 32768.0
    for k, v in mb.state_dict().items():
        assert torch.equal(v, before[k]), k
    one(sb, mb, ob); assert sb.get_scale() == 32768.0 and not sb.last_step_skipped
    one(sb, mb, ob); assert sb.get_scale() == 65536.0               # growth_interval = 2 good steps
    st = sb.state_dict()
    assert st == {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2, "_growth_tracker": 0}
    sc = misc.NativeScalerWithGradNormCount(fp32=False, dynamic_loss_scale=True)
    sc.load_state_dict({"scale": 1024.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 7})
    assert sc.get_scale() == 1024.0 and sc.state_dict()["_growth_tracker"] == 7
    sd = misc.NativeScalerWithGradNormCount(fp32=True)
    sd.load_state_dict(st)                                          # a bf16 / fp32 run accepts the entry and keeps scale 1
    assert sd.get_scale() == 1.0


def test_non_finite_input_gives_non_finite_loss_and_leaves_no_state_behind(golden_dir):
    """A NaN / Inf volume must come out as a non-finite loss (what the engines' guard looks for: engine_pretrain.py:153-161), without
    a hang, and must not poison the next, clean step: every kernel-side flag and workspace (the optimistic attention forward's
    give-up flag, the dQ workspace, LayerNorm partials) is re-initialised per launch, and the gradients of the bad step are zeroed
    by zero_grad like any others."""
    z, cfg, P = small(golden_dir)
    m = build(cfg, P)
    m.train()
    imgs = torch.from_numpy(z["imgs"]).to(DEV)
    noise = torch.from_numpy(z["noise"]).to(DEV)
    ratio = float(z["mask_ratio"])
    loss0, pred0, _ = m(imgs, mask_ratio=ratio, noise=noise)
    loss0.backward()
    g0 = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    for bad_value in (float("nan"), float("inf")):
        bad = imgs.clone()
        bad[0, 0, 1, 3, 5] = bad_value
        m.zero_grad()
        loss_b, pred_b, _ = m(bad, mask_ratio=ratio, noise=noise)
        loss_b.backward()
        torch.cuda.synchronize()
        assert not bool(torch.isfinite(loss_b))
        m.zero_grad()
        loss1, pred1, _ = m(imgs, mask_ratio=ratio, noise=noise)
        loss1.backward()
        assert torch.equal(pred1, pred0) and float(loss1.detach()) == float(loss0.detach())
        for k, p in m.named_parameters():
            if p.grad is not None:
                assert torch.isfinite(p.grad).all(), k
                assert rel(p.grad, g0[k]) < 1e-5 or float(g0[k].norm()) < 1e-12, k


def test_device_prefetcher_moves_one_batch_ahead_and_keeps_order():
    """misc.DevicePrefetcher (the engines' input pipeline): same batches, same order, tensors on the device -- all of them, or only
    the listed positions -- also for nested containers, a one-batch loader, an early break, and a second pass over the same loader."""
    from octcubem_amd import misc
    g = torch.Generator().manual_seed(0)
    batches = [(torch.rand(3, 1, 6, 32, 32, generator=g).pin_memory(), [torch.tensor([i]), {"name": f"p{i}", "w": torch.rand(2, generator=g)}]) for i in range(7)]
    for only in (None, (0,)):
        pf = misc.prefetched(batches, torch.device(DEV), args=None, only=only)
        assert isinstance(pf, misc.DevicePrefetcher) and len(pf) == 7
        for rep in range(2):
            seen = 0
            for i, (x, info) in enumerate(pf):
                assert x.is_cuda and torch.equal(x.cpu(), batches[i][0])
                y = (x * 2).sum()                                            # use it on the compute stream
                assert info[1]["name"] == f"p{i}"
                assert info[0].is_cuda == (only is None) and info[1]["w"].is_cuda == (only is None)
                assert torch.equal(info[1]["w"].cpu(), batches[i][1][1]["w"])
                seen += 1
                if rep == 1 and i == 3:
                    break
            assert seen == (7 if rep == 0 else 4)
            assert torch.isfinite(y)
    assert [t.item() for t in misc.DevicePrefetcher([torch.tensor(5.0)], DEV)] == [5.0]
    class A: prefetch_to_device = False
    assert misc.prefetched(batches, torch.device(DEV), args=A()) is batches


@pytest.mark.small_batch_rule
def test_small_batches_take_the_two_kernel_attention_backward(golden_dir, monkeypatch):
    """ops.attn_bwd_use_fused: one workgroup per (batch, head) cannot fill 256 CUs at a few samples, so such a backward goes to the
    dQ + dK/dV pair (the whole step at 1 volume: 42 vs 24 volumes/s).  The rule itself, and the model's gradients under it against the
    reference golden -- the same bounds as the fused path's test."""
    assert ops.ATTN_BWD_FUSED_MIN_FILL > 0.5
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    for B, H, want in ((1, 16, False), (8, 16, False), (11, 16, False), (12, 16, ncu <= 256), (16, 16, ncu <= 256), (20, 16, False),
                       (24, 16, ncu == 256), (32, 16, True), (128, 16, True)):
        if ncu == 256:
            assert ops.attn_bwd_use_fused(B, H, 32) == want, (B, H)
    calls = []
    real = ops.attn_bwd
    monkeypatch.setattr(ops, "attn_bwd", lambda *a, **k: (calls.append(k.get("fused")), real(*a, **k))[1])
    z, cfg, P = small(golden_dir)
    m = build(cfg, P)
    m.train()
    loss, pred, mask = m(torch.from_numpy(z["imgs"]).to(DEV), mask_ratio=float(z["mask_ratio"]), noise=torch.from_numpy(z["noise"]).to(DEV))
    loss.backward()
    assert calls and all(c is False for c in calls)                       # every Block took the pair
    assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-3 * abs(float(z["loss"]))
    total = float(np.sqrt(sum(float(z[f"gnorm/{k}"]) ** 2 for k, _ in m.named_parameters())))
    for k, p in m.named_parameters():
        gn = float(z[f"gnorm/{k}"])
        if gn >= 1e-3 * total:
            assert abs(float(p.grad.double().norm()) - gn) <= 1.5e-2 * gn, k


def test_operand_copy_follows_the_master_weights_through_every_kind_of_update(golden_dir):
    """The 16-bit operand copy of the parameter arena is no longer cast before every forward: FusedAdamW's kernel writes the copy of
    what it updates (octmae_mt_adamw_fused) and the arena re-casts only when a parameter's PyTorch version counter moved.  After each
    kind of update the copy must equal a fresh cast of the master weights BIT FOR BIT at the next forward."""
    from octcubem_amd import ops as _ops
    z, cfg, P = small(golden_dir)
    m = build(cfg, P).train()
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)
    opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
    scaler = misc.NativeScalerWithGradNormCount()

    def fresh():
        return _ops.cast_bf16(m.arena.flat)

    def step(clip=None):
        opt.zero_grad()
        loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        assert torch.equal(m.arena.lp, fresh()), "operand copy stale at forward time"
        return scaler(loss, opt, parameters=m.parameters(), clip_grad=clip)

    casts = []
    real = _ops.cast_bf16_into
    _ops.cast_bf16_into = lambda s_, d_: (casts.append(s_.numel()), real(s_, d_))[1]
    try:
        n1 = step()                                          # fused norm + update + copy in one pass
        assert torch.equal(m.arena.lp, fresh())              # written by the AdamW kernel
        c0 = len(casts)
        n2 = step(clip=1.0)                                  # separate norm pass (the clip coefficient must be known first)
        assert len(casts) == c0, "a step of FusedAdamW alone must not trigger a re-cast of the arena"
        assert torch.equal(m.arena.lp, fresh())
        assert float(n1) > 0 and float(n2) > 0 and torch.isfinite(n1) and torch.isfinite(n2)
        with torch.no_grad():                                # a foreign in-place update (torch.optim, EMA, clamp ...)
            m.decoder_pred.weight.mul_(1.5)
        assert not torch.equal(m.arena.lp, fresh())
        c0 = len(casts)
        step()
        assert len(casts) == c0 + 1                          # seen through the version counter: exactly one re-cast
        sd = {k: v.clone() * 0.5 for k, v in m.state_dict().items()}
        m.load_state_dict(sd, strict=True)
        step()
        assert torch.equal(m.arena.lp, fresh())
        opt.write_mirror = False                             # the A/B switch: AdamW without the copy -> the arena re-casts
        c0 = len(casts)
        step(); step()                                       # (each forward inside step() asserts a fresh copy)
        assert len(casts) == c0 + 1
        loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        assert len(casts) == c0 + 2 and torch.equal(m.arena.lp, fresh())
    finally:
        _ops.cast_bf16_into = real


def test_fused_norm_equals_the_separate_norm_pass(golden_dir):
    """NativeScalerWithGradNormCount without clipping lets the AdamW kernels accumulate the gradient norm; it must be the number the
    separate multi-tensor pass (get_grad_norm_) returns for the same gradients, and the update must be the same update."""
    z, cfg, P = small(golden_dir)
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)
    res = {}
    for mode in ("fused", "separate"):
        m = build(cfg, P).train()
        opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
        opt.zero_grad()
        loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        if mode == "fused":
            norm = misc.NativeScalerWithGradNormCount()(loss, opt, parameters=m.parameters())
        else:
            loss.backward()
            norm = misc.get_grad_norm_(m.parameters())
            opt.step()
        res[mode] = (float(norm), {k: p.detach().clone() for k, p in m.named_parameters()})
    assert abs(res["fused"][0] - res["separate"][0]) <= 1e-6 * res["separate"][0], (res["fused"][0], res["separate"][0])
    for k, p in res["fused"][1].items():
        assert torch.equal(p, res["separate"][1][k]), k


def test_a_block_used_on_its_own_follows_a_foreign_optimizer():
    """Seam 2 of SURVEY 8(b): video_vit.Block inside somebody else's model, trained by torch.optim.  The Block is the root of its own
    parameter arena; its operand copy must follow every update (until round 4 it was cast once, when the arena was bound)."""
    from functools import partial
    from octcubem_amd import video_vit
    torch.manual_seed(0)
    blk = video_vit.Block(128, 2, mlp_ratio=4.0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6)).to(DEV).train()
    x = torch.randn(2, 65, 128, device=DEV)
    opt = torch.optim.SGD(blk.parameters(), lr=0.5)
    y0 = blk(x).detach().clone()
    blk(x).square().mean().backward()
    opt.step()
    y1 = blk(x).detach().clone()
    assert not torch.equal(y0, y1), "the Block did not see the optimizer's update"
    twin = video_vit.Block(128, 2, mlp_ratio=4.0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    twin.load_state_dict({k: v.detach().cpu().clone() for k, v in blk.state_dict().items()})
    assert torch.equal(twin.to(DEV).train()(x), y1)
