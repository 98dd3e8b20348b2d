"""GPU: the fine-tune path -- octcubem_amd.engine_finetune.train_one_epoch + lr_decay groups + FusedAdamW + the
clip branch of the loss scaler around models_vit_st -- against the trajectory the REAL reference loop produced on the same
seeded data (tests/golden/finetune_small.npz, oracle/gen_golden_finetune.py), and against the CPU oracle.

Tolerances (bf16 GEMM / attention operands, fp32 everything else): per-iteration loss <= 3e-2 relative, gradient norm <= 3e-2
relative, group learning rates exact to 1e-12, final parameters: update direction cosine >= 0.98 against the reference's
update and max deviation <= 2.5 x the largest cumulative step an element can take (AdamW's per-step move is ~lr whatever the
gradient magnitude, so an element whose tiny gradient flips sign under bf16 rounding moves by lr in the other direction)."""
import json
import os
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from octcubem_amd import models_vit_st, engine_finetune, lr_decay, misc, losses
    from octcubem_amd import optim as foptim
from oracle import vit_ref as V
from oracle import finetune_ref as FT

DEV = "cuda"


def build(z):
    cfg = V.ViTSTConfig(**json.loads(str(z["cfg"])))
    P0 = V.init_from_shapes(V.vit_st_param_shapes(cfg), seed=int(z["param_seed"]))
    m = models_vit_st.VisionTransformer(num_frames=cfg.num_frames, t_patch_size=cfg.t_patch_size, img_size=cfg.img_size,
                                        patch_size=cfg.patch_size, in_chans=cfg.in_chans, num_classes=cfg.num_classes,
                                        embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=4,
                                        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), sep_pos_embed=True, cls_embed=True,
                                        global_pool=True, drop_path_rate=0.0, dropout=0.0)
    m.load_state_dict(P0, strict=True)
    return cfg, P0, m.to(DEV)


class Args:
    accum_iter = 2; lr = 2e-4; min_lr = 1e-6; warmup_epochs = 1; epochs = 4; task_mode = "binary_cls"


def test_finetune_loop_matches_reference_trajectory(golden_dir):
    z = np.load(os.path.join(golden_dir, "finetune_small.npz"))
    cfg, P0, model = build(z)
    xs = torch.rand(6, 2, 1, 12, 64, 64, generator=torch.Generator().manual_seed(int(z["data_seed"])))
    assert abs(float(xs.double().sum()) - float(z["x_checksum"])) < 1e-6
    ts = torch.from_numpy(z["target"])
    groups = lr_decay.param_groups_lrd(model, 0.05, no_weight_decay_list=model.no_weight_decay(), layer_decay=0.75)
    id2name = {id(p): n for n, p in model.named_parameters()}
    ref_groups = json.loads(str(z["groups"]))
    key = lambda g: (g["lr_scale"], g["weight_decay"])
    for a, b in zip(sorted(groups, key=key), sorted(ref_groups, key=key)):
        assert sorted(id2name[id(p)] for p in a["params"]) == sorted(b["params"]) and key(a) == key(b)
    opt = foptim.FusedAdamW(groups, lr=Args.lr)          # torch.optim.AdamW defaults: betas (0.9, 0.999), eps 1e-8
    scaler = misc.NativeScalerWithGradNormCount()
    crit = torch.nn.CrossEntropyLoss()
    rec = {"loss": [], "norm": [], "lrs": []}

    def rec_crit(o, t):
        l = crit(o, t)
        rec["loss"].append(l.detach())
        return l

    def rec_scaler(loss, optimizer, **kw):
        n = scaler(loss, optimizer, **kw)
        rec["norm"].append(None if n is None else n.detach().clone())
        rec["lrs"].append([g["lr"] for g in optimizer.param_groups])
        return n
    loader = [(xs[i], ts[i]) for i in range(6)]
    for epoch in range(2):
        stats = engine_finetune.train_one_epoch(model, rec_crit, loader, opt, torch.device(DEV), epoch, rec_scaler, 1.0, None, None, Args)
        assert stats is not None and abs(stats["loss"] - float(z["epoch_loss"][epoch])) <= 3e-2 * float(z["epoch_loss"][epoch])
    losses_ = np.array([float(l) for l in rec["loss"]])
    norms = np.array([-1.0 if n is None else float(n) for n in rec["norm"]])
    np.testing.assert_allclose(np.sort(np.array(rec["lrs"]), axis=1), np.sort(z["lrs"], axis=1), rtol=1e-12)
    np.testing.assert_allclose(losses_, z["losses"], rtol=3e-2)
    assert ((norms < 0) == (z["norms"] < 0)).all()
    np.testing.assert_allclose(norms[norms > 0], z["norms"][z["norms"] > 0], rtol=3e-2)
    # parameters after 6 optimizer steps
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    max_step = float(np.max(z["lrs"])) * 6
    for k in z.files:
        if not k.startswith("final/"):
            continue
        n = k[len("final/"):]
        sub = (lambda t: t if t.numel() <= 8192 else t.flatten()[::7])
        ref, mine, init = torch.from_numpy(z[k]).flatten(), sub(sd[n]).flatten(), sub(P0[n]).flatten()
        du_ref, du = (ref - init).double(), (mine - init).double()
        assert float((mine - ref).abs().max()) <= 2.5 * max_step, (n, float((mine - ref).abs().max()))
        if n.endswith("attn.k.bias"):
            continue      # d loss / d k.bias == 0 exactly (softmax is shift-invariant): both runs step along rounding noise
        if float(du_ref.norm()) > 1e-9:
            cos = float((du * du_ref).sum() / (du.norm() * du_ref.norm() + 1e-30))
            assert cos >= 0.98, (n, cos)
        else:
            assert float(du.norm()) <= 1e-9, n       # norm.* never receives a gradient (models_vit_st…:247-249)


def test_finetune_step_matches_cpu_oracle_with_label_smoothing():
    """Same loop against the CPU restatement on a different seed / criterion (label smoothing 0.1, no clipping of note)."""
    cfg = V.ViTSTConfig(num_frames=6, t_patch_size=3, img_size=32, patch_size=16, in_chans=1, num_classes=5, embed_dim=64, depth=2,
                        num_heads=2, global_pool=True)
    P0 = V.init_from_shapes(V.vit_st_param_shapes(cfg), seed=77)
    g = torch.Generator().manual_seed(3)
    xs = torch.rand(4, 3, 1, 6, 32, 32, generator=g)
    ts = torch.randint(0, 5, (4, 3), generator=g)
    ref = FT.finetune_trajectory(P0, cfg, xs, ts, lr=1e-4, min_lr=1e-6, warmup_epochs=1, epochs=3, n_epochs=1, accum_iter=2, max_norm=5.0,
                                 criterion=lambda o, t: FT.label_smoothing_ce(o, t, 0.1))
    m = models_vit_st.VisionTransformer(num_frames=6, t_patch_size=3, img_size=32, patch_size=16, in_chans=1, num_classes=5, embed_dim=64,
                                        depth=2, num_heads=2, mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                                        sep_pos_embed=True, cls_embed=True, global_pool=True, dropout=0.0)
    m.load_state_dict(P0, strict=True)
    m = m.to(DEV)
    opt = foptim.FusedAdamW(lr_decay.param_groups_lrd(m, 0.05, m.no_weight_decay(), 0.75), lr=1e-4)
    scaler = misc.NativeScalerWithGradNormCount()
    crit = losses.LabelSmoothingCrossEntropy(0.1)
    got = {"loss": [], "norm": []}

    class A:
        accum_iter = 2; lr = 1e-4; min_lr = 1e-6; warmup_epochs = 1; epochs = 3; task_mode = "multi_cls"

    def rc(o, t):
        l = crit(o, t); got["loss"].append(float(l)); return l

    def rs(loss, optimizer, **kw):
        n = scaler(loss, optimizer, **kw); got["norm"].append(-1.0 if n is None else float(n)); return n
    engine_finetune.train_one_epoch(m, rc, [(xs[i], ts[i]) for i in range(4)], opt, torch.device(DEV), 0, rs, 5.0, None, None, A)
    np.testing.assert_allclose(got["loss"], ref["losses"], rtol=3e-2)
    np.testing.assert_allclose(got["norm"], ref["norms"], rtol=3e-2)


def test_evaluate_reports_loss_accuracy_and_logits():
    m = models_vit_st.VisionTransformer(num_frames=6, t_patch_size=3, img_size=32, patch_size=16, in_chans=1, num_classes=5, embed_dim=64,
                                        depth=1, num_heads=2, mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                                        sep_pos_embed=True, cls_embed=True, global_pool=True).to(DEV)
    g = torch.Generator().manual_seed(0)
    data = [(torch.rand(2, 1, 6, 32, 32, generator=g), torch.randint(0, 5, (2,), generator=g)) for _ in range(3)]
    out = engine_finetune.evaluate(data, m, torch.device(DEV))
    assert out["logits"].shape == (6, 5) and out["targets"].shape == (6,)
    lg = torch.cat([m.eval()(x.to(DEV)).float().cpu() for x, _ in data])
    assert torch.allclose(lg, out["logits"], atol=1e-6)
    assert abs(out["loss"] - float(torch.nn.functional.cross_entropy(lg, out["targets"]))) < 1e-5
    assert abs(out["acc1"] - float((lg.argmax(-1) == out["targets"]).float().mean())) < 1e-7
