"""Measures the model-level parity ledger of ONE build of the library -- the one OCTMAE_LIB selects -- against the golden vectors
produced by the real reference and writes it to --out as JSON.  tests/test_gpu_f16_parity.py runs it twice (liboctmae.so: bfloat16
operands; liboctmae_f16.so: IEEE-half operands, the reference's own default arithmetic) and compares the two ledgers.

Why a separate process per build: the shared library is loaded once per process (octcubem_amd/_lib.py).

Half needs loss scaling for its gradients (the decoder's dpred is ~1e-6 per element, below half's normal range of 6.1e-5): the loss is
multiplied by the reference's default GradScaler scale, 65536 (torch.cuda.amp.GradScaler(); custom_util/misc.py:311-312), halved
until the gradients are finite -- what NativeScalerWithGradNormCount(dynamic_loss_scale=True) does step by step -- and the gradients
are divided by it before they are compared.  The bfloat16 build runs with scale 1.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from octcubem_amd import ops  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 1))      # two of these run beside the test session: do not oversubscribe the host
DEV = "cuda"
SCALE0 = 65536.0        # torch.cuda.amp.GradScaler's init_scale; --scale0 2**24 finds the largest finite scale (its steady state)
NAMES = {}              # label -> the tensor a "worst" entry refers to
GOLDEN = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a = torch.as_tensor(a).detach().double().flatten().cpu(); b = torch.as_tensor(b).detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def backward_scaled(loss_fn, model):
    """(loss * S).backward() with S = 65536 for half (1 for bfloat16), halved while any gradient is non-finite.  Returns S."""
    S = SCALE0 if ops.LP_IS_F16 else 1.0
    while True:
        for p in model.parameters():
            if p.grad is not None:
                p.grad.zero_()
        loss = loss_fn()
        (loss * S).backward()
        torch.cuda.synchronize()
        ok = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
        if ok or S <= 1.0:
            return loss, S
        S *= 0.5


def golden_grads(model, z, S, label=""):
    """worst / median rel-L2 over the gradient tensors that carry >= 1e-3 of the global norm, and the global norm's error
    (the golden files hold `grad/<name>` -- whole tensors up to 8192 elements, every 7th element above -- and `gnorm/<name>`)."""
    total = float(np.sqrt(sum(float(z[k]) ** 2 for k in z.files if k.startswith("gnorm/"))))
    errs, sq = {}, 0.0
    for k, p in model.named_parameters():
        if f"gnorm/{k}" not in z.files:
            continue
        gn = float(z[f"gnorm/{k}"])
        if p.grad is None:
            continue
        g = p.grad.double() / S
        sq += float(g.pow(2).sum())
        if gn < 1e-3 * total:
            continue
        ref = torch.from_numpy(z[f"grad/{k}"])
        mine = g.cpu() if g.numel() <= 8192 else g.cpu().flatten()[::7]
        errs[k] = rel(mine.reshape(ref.shape), ref)
    v = sorted(errs.values())
    NAMES[f"{label}/worst_grad"] = sorted(errs, key=errs.get)[-3:]
    return {"worst_grad": v[-1], "median_grad": v[len(v) // 2], "grad_norm": abs(sq ** 0.5 - total) / total}


def case_small(out):
    import tests.test_gpu_model as TM
    z, cfg, P = TM.small(GOLDEN)
    m = TM.build(cfg, P).train()
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)
    keep = {}

    def fwd():
        (loss, fl), pred, mask = m(imgs, mask_ratio=float(z["mask_ratio"]), frame_loss=True, noise=noise)
        keep["fl"], keep["pred"], keep["mask"] = fl, pred, mask
        return loss

    loss, S = backward_scaled(fwd, m)
    assert torch.equal(keep["mask"].cpu(), torch.from_numpy(z["mask"]))
    out["small/loss"] = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    out["small/pred"] = rel(keep["pred"], z["pred"])
    out["small/frame_losses"] = rel(keep["fl"], z["frame_losses"])
    for k, v in golden_grads(m, z, S, "small").items():
        out[f"small/{k}"] = v
    out["small/loss_scale"] = S


def case_mid(out):
    import tests.test_gpu_model as TM
    from oracle import mae3d_ref as O
    cfg = O.MAEConfig(input_size=96, in_chans=1, embed_dim=256, depth=3, num_heads=4, decoder_embed_dim=128, decoder_depth=2,
                      decoder_num_heads=4, num_frames=15, t_patch_size=3, pred_t_dim=15, high_res_input_size=192)
    P = O.init_params(cfg, seed=3, bias_std=0.02)
    imgs = torch.rand(3, 1, 15, 96, 96, generator=torch.Generator().manual_seed(1))
    noise = torch.rand(3, cfg.num_patches, generator=torch.Generator().manual_seed(2))
    loss_r, pred_r, mask_r, ids_r, grads_r = O.forward_backward(P, imgs, cfg, 0.75, noise)
    m = TM.build(cfg, P)
    keep = {}

    def fwd():
        loss, pred, mask = m(imgs.to(DEV), mask_ratio=0.75, noise=noise.to(DEV))
        keep["pred"] = pred
        return loss

    loss, S = backward_scaled(fwd, m)
    out["mid/loss"] = abs(float(loss) - float(loss_r)) / float(loss_r)
    out["mid/pred"] = rel(keep["pred"], pred_r)
    total = float(O.grad_norm(grads_r.values()))
    errs = {k: rel(p.grad.double() / S, grads_r[k]) for k, p in m.named_parameters() if float(grads_r[k].norm()) >= 1e-3 * total}
    v = sorted(errs.values())
    NAMES["mid/worst_grad"] = sorted(errs, key=errs.get)[-3:]
    out["mid/worst_grad"], out["mid/median_grad"] = v[-1], v[len(v) // 2]
    out["mid/loss_scale"] = S


def case_mae2d(out):
    from functools import partial
    from octcubem_amd import models_mae_2d
    from oracle import vit_ref as V
    z = np.load(os.path.join(GOLDEN, "mae2d_small.npz"))
    cfg = V.MAE2DConfig(**json.loads(str(z["cfg"])))
    P = V.mae2d_init(cfg, seed=int(z["param_seed"]))
    m = models_mae_2d.MaskedAutoencoderViT(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
                                            embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                                            decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
                                            decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=4,
                                            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    m.load_state_dict(P, strict=True)
    m = m.to(DEV)
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)
    keep = {}

    def fwd():
        loss, pred, mask = m(imgs, mask_ratio=0.75, noise=noise)
        keep["pred"] = pred
        return loss

    loss, S = backward_scaled(fwd, m)
    out["mae2d_small/loss"] = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    out["mae2d_small/pred"] = rel(keep["pred"], z["pred"])
    for k, v in golden_grads(m, z, S, "mae2d_small").items():
        out[f"mae2d_small/{k}"] = v
    out["mae2d_small/loss_scale"] = S


def case_vit_st(out):
    from functools import partial
    from octcubem_amd import models_vit_st
    from oracle import vit_ref as V
    z = np.load(os.path.join(GOLDEN, "vit_st_small.npz"))
    cfg = V.ViTSTConfig(**json.loads(str(z["cfg"])))
    P = V.init_from_shapes(V.vit_st_param_shapes(cfg), seed=int(z["param_seed"]))
    m = models_vit_st.VisionTransformer(
        global_pool=True, num_frames=cfg.num_frames, t_patch_size=cfg.t_patch_size, img_size=cfg.img_size, patch_size=cfg.patch_size,
        in_chans=cfg.in_chans, num_classes=cfg.num_classes, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
        mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), sep_pos_embed=True, cls_embed=True)
    m.load_state_dict(P, strict=True)
    m = m.to(DEV).eval()
    x = torch.from_numpy(z["x"]).to(DEV)
    keep = {}

    def fwd():
        logits, emb = m(x, return_embeddings=True)
        keep["logits"], keep["emb"] = logits, emb
        return torch.nn.functional.cross_entropy(logits, torch.from_numpy(z["target"]).to(DEV))

    loss, S = backward_scaled(fwd, m)
    out["vit_st_small/logits"] = rel(keep["logits"], z["logits"])
    out["vit_st_small/embedding"] = rel(keep["emb"], z["embedding"])
    out["vit_st_small/loss"] = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    for k, v in golden_grads(m, z, S, "vit_st_small").items():
        out[f"vit_st_small/{k}"] = v
    out["vit_st_small/loss_scale"] = S


def pin_grads(model, z, S):
    """The full-size pins (oracle/gen_golden_fullsize.py): per-tensor gradient norms and strided samples of 20 tensors."""
    names = json.loads(str(z["grad_names"]))
    ref_norms = dict(zip(names, z["grad_norms"]))
    total = float(z["global_grad_norm"])
    params = dict(model.named_parameters())
    sq, worst_norm = 0.0, 0.0
    for k in names:
        g = params[k].grad
        gn = 0.0 if g is None else float(g.double().norm()) / S
        sq += gn * gn
        if ref_norms[k] >= 1e-3 * total:
            worst_norm = max(worst_norm, abs(gn - ref_norms[k]) / ref_norms[k])
    samples = []
    for key in z.files:
        if key.startswith("gsample/"):
            k = key[len("gsample/"):]
            if ref_norms[k] >= 1e-3 * total:
                ref = torch.from_numpy(z[key])
                samples.append(rel(params[k].grad.flatten()[::int(z[f"gstep/{k}"])][: ref.numel()].double() / S, ref))
    return {"grad_norm": abs(sq ** 0.5 - total) / total, "worst_tensor_norm": worst_norm, "grad_samples_max": max(samples),
            "grad_samples_median": float(np.median(samples))}


def case_vitl(out):
    """BASELINE config 2 at full size: ViT-L 3-D MAE on (1,1,60,256,256), forward and backward against the reference's pins."""
    from octcubem_amd import models_mae
    from oracle import mae3d_ref as O
    z = np.load(os.path.join(GOLDEN, "vitl_bwd_pins.npz"))
    pins = np.load(os.path.join(GOLDEN, "vitl_pins.npz"))
    m = models_mae.octcube_vit_large_3dmae()
    # forward pins (oracle/gen_golden.py: parameters of seed 0, volume seed 0): loss, 99 sampled elements of pred, its 2-norm
    m.load_state_dict(O.init_params(O.VIT_L, seed=0), strict=True)
    m = m.to(DEV)
    imgs0 = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(0)).to(DEV)
    torch.manual_seed(int(pins["noise_seed"]))
    noise0 = torch.rand(1, 5120).to(DEV)
    with torch.no_grad():
        loss0, pred0, _ = m(imgs0, mask_ratio=0.75, noise=noise0)
    assert torch.equal(m._ids_restore.cpu().int(), torch.from_numpy(pins["ids_restore"]))
    out["vitl/loss"] = abs(float(loss0) - float(pins["loss"])) / float(pins["loss"])
    out["vitl/pred_samples"] = rel(pred0.flatten()[torch.from_numpy(pins["pred_idx"]).to(DEV)], pins["pred_samples"])
    out["vitl/pred_l2"] = abs(float(pred0.double().norm()) - float(pins["pred_l2"])) / float(pins["pred_l2"])
    # backward pins (oracle/gen_golden_fullsize.py)
    P = O.init_params(O.VIT_L, seed=int(z["param_seed"]), bias_std=float(z["param_bias_std"]))
    m.load_state_dict({k: v.to(DEV) for k, v in P.items()}, strict=True)
    m.train()
    imgs = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(int(z["img_seed"]))).to(DEV)
    torch.manual_seed(int(z["noise_seed"]))
    noise = torch.rand(1, 5120).to(DEV)
    keep = {}

    def fwd():
        loss, pred, mask = m(imgs, mask_ratio=float(z["mask_ratio"]), noise=noise)
        keep["pred"] = pred
        return loss

    loss, S = backward_scaled(fwd, m)
    out["vitl/bwd_loss"] = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    for k, v in pin_grads(m, z, S).items():
        out[f"vitl/{k}"] = v
    out["vitl/loss_scale"] = S


def case_vit_st_l(out):
    """BASELINE config 4 at full size: the ViT-L ST fine-tune model on (1,1,60,256,256), N = 5121, head_dim 64."""
    from octcubem_amd import models_vit_st
    from oracle import vit_ref as V
    z = np.load(os.path.join(GOLDEN, "vit_st_l_pins.npz"))
    cfg = V.ViTSTConfig(**json.loads(str(z["cfg"])))
    P = V.init_from_shapes(V.vit_st_param_shapes(cfg), seed=int(z["param_seed"]), bias_std=float(z["param_bias_std"]))
    m = models_vit_st.vit_large_patch16(num_frames=60, t_patch_size=3, img_size=256, in_chans=1, num_classes=8, global_pool=True,
                                        sep_pos_embed=True, cls_embed=True, drop_path_rate=0.0)
    m.load_state_dict(P, strict=True)
    m = m.to(DEV).eval()
    x = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(int(z["img_seed"]))).to(DEV)
    keep = {}

    def fwd():
        logits, emb = m(x, return_embeddings=True)
        keep["logits"], keep["emb"] = logits, emb
        return torch.nn.functional.cross_entropy(logits, torch.from_numpy(z["target"]).to(DEV))

    loss, S = backward_scaled(fwd, m)
    out["vit_st_l/logits"] = rel(keep["logits"], z["logits"])
    out["vit_st_l/embedding"] = rel(keep["emb"], z["embedding"])
    out["vit_st_l/loss"] = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    for k, v in pin_grads(m, z, S).items():
        out[f"vit_st_l/{k}"] = v
    out["vit_st_l/loss_scale"] = S


def case_train(out):
    """Two optimizer steps through the reference-shaped loop -- NativeScalerWithGradNormCount + FusedAdamW + lr schedule -- against the
    oracle's forward / backward + AdamW, each step from the HIP model's own parameters.  On the half build the scaler runs the
    reference's GradScaler state machine (dynamic_loss_scale=True, initial scale 65536: custom_util/misc.py:311-344); on bfloat16 it is
    the identity.  Entries: loss, gradient norm as the scaler returns it (un-scaled), 1 - cos(update direction) of the two steps."""
    import tests.test_gpu_model as TM
    from octcubem_amd import misc, lr_sched, optim as foptim
    from oracle import mae3d_ref as O
    cfg = O.MAEConfig(input_size=64, in_chans=1, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64, decoder_depth=1,
                      decoder_num_heads=2, num_frames=6, t_patch_size=3, pred_t_dim=6, high_res_input_size=128)
    P = O.init_params(cfg, seed=5, bias_std=0.02)
    m = TM.build(cfg, P)
    opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
    scaler = misc.NativeScalerWithGradNormCount()          # the reference's call (main_pretrain...:456): dynamic loss scale iff the library computes on half
    assert scaler.enabled == ops.LP_IS_F16

    class A: pass
    a = A(); a.lr = 1e-3; a.min_lr = 0.0; a.warmup_epochs = 1; a.epochs = 10
    nd, _ = O.weight_decay_groups([(k, tuple(v.shape)) for k, v in P.items()], 0.05)
    nd = set(nd)
    Mr = {k: torch.zeros_like(v) for k, v in P.items()}; Vr = {k: torch.zeros_like(v) for k, v in P.items()}
    dot = n1 = n2 = 0.0
    for step in (1, 2):
        imgs = torch.rand(2, 1, 6, 64, 64, generator=torch.Generator().manual_seed(10 + step))
        noise = torch.rand(2, cfg.num_patches, generator=torch.Generator().manual_seed(20 + step))
        lr = lr_sched.adjust_learning_rate(opt, 0.5 * step, a)
        before = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        opt.zero_grad()
        loss, _, _ = m(imgs.to(DEV), mask_ratio=0.75, noise=noise.to(DEV))
        norm = scaler(loss, opt, parameters=m.parameters(), clip_grad=None)
        assert not scaler.last_step_skipped, "the dynamic loss scale skipped a step at its initial scale"
        loss_r, _, _, _, G = O.forward_backward(before, imgs, cfg, 0.75, noise)
        out[f"train/loss{step}"] = abs(float(loss) - float(loss_r)) / float(loss_r)
        gn = float(O.grad_norm(G.values()))
        out[f"train/grad_norm{step}"] = abs(float(norm) - gn) / gn
        after = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        for k in before:
            pr, Mr[k], Vr[k] = O.adamw_step(before[k], G[k], Mr[k], Vr[k], step, lr, 0.9, 0.95, 1e-8, 0.0 if k in nd else 0.05)
            if float(G[k].norm()) < 1e-3 * gn:
                continue
            du, dr = (after[k] - before[k]).double().flatten(), (pr - before[k]).double().flatten()
            dot += float(du @ dr); n1 += float(du @ du); n2 += float(dr @ dr)
        # the moments of the comparison follow the HIP model's trajectory from here on (same start for step 2)
    out["train/1-cos(update)"] = 1.0 - dot / (n1 * n2) ** 0.5
    out["train/loss_scale"] = float(scaler.get_scale())


def case_coem_l(out):
    """BASELINE config 5 at full size (round 6): the shipped tower pair -- ViT-L ST on (2,1,60,256,256), N = 5121, + ViT-L 2-D on
    (2,3,224,224), embed 512 -- through coem.create_model_from_config, one contrastive step (ClipLoss, backward through both towers)
    against the pins of the reference's own tower classes and ClipLoss (oracle/gen_golden_coem_full.py;
    retinal-COEM/src/open_clip/model.py:635-682, loss.py:21-65)."""
    from octcubem_amd import coem
    from oracle import vit_ref as V
    import tests.test_gpu_coem as TC
    z = np.load(os.path.join(GOLDEN, "coem_l_pins.npz"))
    model = coem.create_model_from_config(json.loads(json.dumps(TC.SHIPPED_CFG)), flash_semantics=False)
    PA = V.init_from_shapes(V.vit_st_param_shapes(V.ViTSTConfig(**json.loads(str(z["cfg_a"])))), seed=int(z["seed_a"]))
    PB = V.init_from_shapes(V.vit2d_param_shapes(V.ViT2DConfig(**json.loads(str(z["cfg_b"])))), seed=int(z["seed_b"]))
    model.visual.load_state_dict(PA, strict=True); model.text.load_state_dict(PB, strict=True)
    model = model.to(DEV).eval()
    vol = torch.rand(2, 1, 60, 256, 256, generator=torch.Generator().manual_seed(int(z["vol_seed"]))).to(DEV)
    ir = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(int(z["ir_seed"]))).to(DEV)
    keep = {}

    def fwd():
        fa, fb, ls = model(vol, ir)
        keep["fa"], keep["fb"] = fa, fb
        return coem.ClipLoss()(fa, fb, ls)

    loss, S = backward_scaled(fwd, model)
    out["coem_l/feat_a"] = rel(keep["fa"], z["feat_a"])
    out["coem_l/feat_b"] = rel(keep["fb"], z["feat_b"])
    out["coem_l/loss"] = abs(float(loss) - float(z["loss"])) / abs(float(z["loss"]))
    out["coem_l/logit_scale_grad"] = abs(float(model.logit_scale.grad) / S - float(z["logit_scale_grad"])) / (abs(float(z["logit_scale_grad"])) + 1e-3)
    samples = []
    for tag, tower in (("a", model.visual), ("b", model.text)):
        names = json.loads(str(z[f"grad_names_{tag}"]))
        norms = dict(zip(names, z[f"grad_norms_{tag}"]))
        grads = {k: (p.grad.double() / S if p.grad is not None else torch.zeros_like(p, dtype=torch.float64)) for k, p in tower.named_parameters()}
        tot = float(torch.sqrt(sum(g.pow(2).sum() for g in grads.values())))
        ref_tot = float(z[f"tower_grad_norm_{tag}"])
        out[f"coem_l/tower_grad_norm_{tag}"] = abs(tot - ref_tot) / ref_tot
        out[f"coem_l/worst_tensor_norm_{tag}"] = max(abs(float(grads[k].norm()) - norms[k]) / norms[k] for k in names if norms[k] >= 1e-2 * ref_tot)
        for key in z.files:
            if key.startswith(f"gsample_{tag}/"):
                k = key.split("/", 1)[1]
                if norms[k] >= 1e-2 * ref_tot:
                    samples.append(rel(grads[k].flatten()[::int(z[f"gstep_{tag}/{k}"])][:len(z[key])], z[key]))
    out["coem_l/grad_samples_max"] = max(samples)
    out["coem_l/grad_samples_median"] = float(np.median(samples))
    out["coem_l/loss_scale"] = S


def case_finetune(out):
    """The fine-tune ENGINE (round 6): engine_finetune.train_one_epoch -- lr_decay groups, accumulation, the clip branch of the scaler,
    FusedAdamW -- over the two epochs of tests/golden/finetune_small.npz, the trajectory the reference's own loop produced
    (OCTCube/engine_finetune.py:386-482 through oracle/gen_golden_finetune.py).  On the half build the scaler is the reference's
    GradScaler state machine (no step may be skipped).  Entries: worst per-iteration loss / gradient-norm error, 1 - the smallest
    cosine between this run's and the reference's total parameter update over the tensors that moved."""
    import tests.test_gpu_finetune as TF
    from octcubem_amd import engine_finetune, lr_decay, misc, optim as foptim
    z = np.load(os.path.join(GOLDEN, "finetune_small.npz"))
    cfg, P0, model = TF.build(z)
    xs = torch.rand(6, 2, 1, 12, 64, 64, generator=torch.Generator().manual_seed(int(z["data_seed"])))
    ts = torch.from_numpy(z["target"])
    opt = foptim.FusedAdamW(lr_decay.param_groups_lrd(model, 0.05, no_weight_decay_list=model.no_weight_decay(), layer_decay=0.75), lr=TF.Args.lr)
    scaler = misc.NativeScalerWithGradNormCount()
    assert scaler.enabled == ops.LP_IS_F16
    crit = torch.nn.CrossEntropyLoss()
    rec = {"loss": [], "norm": [], "skipped": 0}

    def rec_crit(o, t):
        l = crit(o, t); rec["loss"].append(float(l)); return l

    def rec_scaler(loss, optimizer, **kw):
        n = scaler(loss, optimizer, **kw)
        rec["norm"].append(-1.0 if n is None else float(n))
        rec["skipped"] += int(n is not None and scaler.last_step_skipped)
        return n
    loader = [(xs[i], ts[i]) for i in range(6)]
    for epoch in range(2):
        engine_finetune.train_one_epoch(model, rec_crit, loader, opt, torch.device(DEV), epoch, rec_scaler, 1.0, None, None, TF.Args)
    assert rec["skipped"] == 0, "the dynamic loss scale skipped a step"
    losses_, norms = np.array(rec["loss"]), np.array(rec["norm"])
    assert ((norms < 0) == (z["norms"] < 0)).all()
    out["finetune/loss_max"] = float(np.max(np.abs(losses_ - z["losses"]) / np.abs(z["losses"])))
    out["finetune/grad_norm_max"] = float(np.max(np.abs(norms[norms > 0] - z["norms"][z["norms"] > 0]) / z["norms"][z["norms"] > 0]))
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    worst = 0.0
    for k in z.files:
        if not k.startswith("final/"):
            continue
        n = k[len("final/"):]
        sub = (lambda t: t if t.numel() <= 8192 else t.flatten()[::7])
        ref, mine, init = torch.from_numpy(z[k]).flatten(), sub(sd[n]).flatten(), sub(P0[n]).flatten()
        du_ref, du = (ref - init).double(), (mine - init).double()
        if n.endswith("attn.k.bias") or float(du_ref.norm()) <= 1e-9:
            continue
        worst = max(worst, 1.0 - float((du * du_ref).sum() / (du.norm() * du_ref.norm() + 1e-30)))
    out["finetune/1-cos(update)_worst"] = worst
    out["finetune/loss_scale"] = float(scaler.get_scale())


def case_joint(out):
    """The joint pre-training LOOP (round 6): engine_pretrain.train_one_epoch_joint -- 3-D volumes + 2-D/512 B-scan triplets, summed
    loss, per-frame loss feedback, accumulation, clip, FusedAdamW -- over the epoch of tests/golden/joint_small.npz, produced by the
    reference's own loop on the same data and masking noise (Pre-training/engine_pretrain.py:29-204 through
    oracle/gen_golden_joint.py).  Entries: epoch-mean losses, worst gradient-norm and per-frame-loss error."""
    from functools import partial
    from octcubem_amd import engine_pretrain, misc, models_mae, optim as foptim
    from tests.test_oracle_joint_golden import load_joint
    z, cfg, P0, vols, imgs2d, n3, n2, frames = load_joint(GOLDEN)
    m = models_mae.MaskedAutoencoderViT(
        input_size=cfg.input_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depth=cfg.depth,
        num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
        decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio, norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps),
        num_frames=cfg.num_frames, t_patch_size=cfg.t_patch_size, sep_pos_embed=True, cls_embed=True, pred_t_dim=cfg.pred_t_dim,
        high_res_input_size=cfg.high_res_input_size)
    m.load_state_dict(P0, strict=True)
    m = m.to(DEV)
    opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
    scaler = misc.NativeScalerWithGradNormCount()
    assert scaler.enabled == ops.LP_IS_F16
    norms, skipped = [], [0]

    def rec_scaler(loss, optimizer, **kw):
        n = scaler(loss, optimizer, **kw)
        norms.append(-1.0 if n is None else float(n))
        skipped[0] += int(n is not None and scaler.last_step_skipped)
        return n
    queue = [t for pair in zip(n3, n2) for t in pair]
    noise_fn = lambda s: queue.pop(0).to(DEV)
    loader3d = [(vols[it], ([f"vol{it}_{j}" for j in range(2)], {"frames": frames[it]})) for it in range(4)]
    loader2d = [(imgs2d[it], None) for it in range(4)]
    table = {f: {} for it in range(4) for nf in range(6) for f in frames[it][nf]}

    class Args:
        accum_iter = 2; lr = 1e-3; min_lr = 1e-6; warmup_epochs = 1; epochs = 4; mask_ratio = 0.75; clip_grad = 1.0; repeat_aug = 1
    stats = engine_pretrain.train_one_epoch_joint(m, loader3d, opt, torch.device(DEV), 1, rec_scaler, loader2d, table, 0.8, args=Args,
                                                  noise_fn=noise_fn)
    assert skipped[0] == 0, "the dynamic loss scale skipped a step"
    ref = json.loads(str(z["stats"]))
    for k in ("loss", "loss_2d", "loss_all"):
        out[f"joint/{k}"] = abs(stats[k] - ref[k]) / ref[k]
    nr = np.array(norms); zr = np.asarray(z["norms"], dtype=np.float64)
    assert ((nr < 0) == (zr < 0)).all()
    out["joint/grad_norm_max"] = float(np.max(np.abs(nr[nr > 0] - zr[zr > 0]) / zr[zr > 0]))
    ref_tab = json.loads(str(z["frame_dict"]))
    out["joint/frame_loss_max"] = max(abs(table[k]["mse_loss"] - e["mse_loss"]) / abs(e["mse_loss"]) for k, e in ref_tab.items())
    out["joint/loss_scale"] = float(scaler.get_scale())


CASES = {"train": case_train, "small": case_small, "mid": case_mid, "mae2d_small": case_mae2d, "vit_st_small": case_vit_st, "vitl": case_vitl,
         "vit_st_l": case_vit_st_l, "coem_l": case_coem_l, "finetune": case_finetune, "joint": case_joint}


def main():
    global SCALE0
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--scale0", type=float, default=SCALE0, help="initial loss scale of the half build (halved while a gradient is non-finite)")
    ap.add_argument("--min-fill", type=float, default=0.0, help="ops.ATTN_BWD_FUSED_MIN_FILL (0: the fused backward, as the GPU tests run)")
    a = ap.parse_args()
    from octcubem_amd import _lib
    ops.ATTN_BWD_FUSED_MIN_FILL = a.min_fill
    SCALE0 = a.scale0
    out = {}
    for c in a.cases.split(","):
        CASES[c](out)
        torch.cuda.empty_cache()
    res = {"lib": _lib.LIB_PATH, "lp_dtype": str(ops.BF16), "min_fill": a.min_fill, "scale0": SCALE0, "entries": out, "names": NAMES}
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
