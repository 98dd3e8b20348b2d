"""Autocast invariance of the drop-in modules.

The reference's engines call their models inside ``with torch.cuda.amp.autocast():`` (Pre-training/engine_pretrain.py:110,255,
OCTCube/engine_pretrain.py:57, OCTCube/engine_finetune.py:432,576, retinal-COEM/src/training/train_retclip.py:125).  The modules here
pick their own precisions, so the context must change nothing: every model is run once plainly, once inside
``torch.cuda.amp.autocast()`` (fp16) and once inside ``torch.autocast("cuda", dtype=torch.bfloat16)``, and outputs AND gradients
must be BIT-identical (the test sizes keep every weight gradient on one k slice, so the kernels themselves are bit-reproducible:
checked first by running the plain mode twice).  Plus one iteration of the reference-shaped loop:
autocast -> model(samples, mask_ratio) -> loss_scaler(loss, optimizer, parameters=...).
"""
import contextlib
import json
import os
import warnings
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from octcubem_amd import coem, misc, models_mae, models_mae_2d, models_vit, models_vit_st, video_vit
    from octcubem_amd import optim as foptim
from oracle import mae3d_ref as O
from oracle import vit_ref as V

DEV = "cuda"


def _amp_fp16():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.cuda.amp.autocast()


MODES = {
    "plain": contextlib.nullcontext,
    "plain_again": contextlib.nullcontext,
    "amp_fp16": _amp_fp16,                                             # the reference's spelling
    "autocast_bf16": lambda: torch.autocast("cuda", dtype=torch.bfloat16),
}


def _snapshot(outs, model, extra_params=()):
    outs = [o.detach().clone() for o in outs if o is not None]
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}
    return outs, grads


def _assert_bit_identical(results, what):
    ref_outs, ref_grads = results["plain"]
    for mode, (outs, grads) in results.items():
        assert len(outs) == len(ref_outs)
        for i, (a, b) in enumerate(zip(outs, ref_outs)):
            assert a.dtype == b.dtype, (what, mode, i, a.dtype, b.dtype)
            assert torch.equal(a, b), (what, mode, f"output {i}", float((a.double() - b.double()).abs().max()))
        for k in ref_grads:
            a, b = grads[k], ref_grads[k]
            assert (a is None) == (b is None), (what, mode, k)
            if a is not None:
                assert torch.equal(a, b), (what, mode, k, float((a.double() - b.double()).abs().max()))


def _run_modes(make_model, step, what):
    """make_model() -> a fresh model with fixed weights; step(model) -> (loss, outputs...) run INSIDE the mode's context; the
    backward runs outside it (as the reference's loss_scaler call does)."""
    results = {}
    for mode, ctx in MODES.items():
        m = make_model()
        with ctx():
            loss, *outs = step(m)
        loss.backward()
        torch.cuda.synchronize()
        results[mode] = _snapshot([loss] + list(outs), m)
    _assert_bit_identical(results, what)


def _mae3d(golden_dir):
    z = np.load(os.path.join(golden_dir, "mae3d_small.npz"))
    cfg = O.MAEConfig(**json.loads(str(z["cfg"])))
    P = O.init_params(cfg, seed=int(z["param_seed"]), bias_std=float(z["param_bias_std"]))

    def make():
        m = models_mae.MaskedAutoencoderViT(
            input_size=cfg.input_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depth=cfg.depth,
            num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
            decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio, norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps),
            norm_pix_loss=cfg.norm_pix_loss, num_frames=cfg.num_frames, t_patch_size=cfg.t_patch_size, sep_pos_embed=True,
            cls_embed=True, pred_t_dim=cfg.pred_t_dim, high_res_input_size=cfg.high_res_input_size)
        m.load_state_dict(P, strict=True)
        return m.to(DEV).train()
    return z, cfg, make


def test_mae3d_volume_batch_is_autocast_invariant(golden_dir):
    z, cfg, make = _mae3d(golden_dir)
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)

    def step(m):
        (loss, fl), pred, mask = m(imgs, mask_ratio=float(z["mask_ratio"]), frame_loss=True, noise=noise)
        assert pred.dtype == torch.float32 and loss.dtype == torch.float32
        return loss, pred, mask, fl
    _run_modes(make, step, "3-D MAE, volume batch")


def test_mae3d_high_res_2d_branch_is_autocast_invariant(golden_dir):
    """The 2-D / 512-style branch (t_actual == 1: high_res_patch_embed, un-interpolated spatial table) -- the branch in which a
    down-cast positional table would reach a kernel that checks for fp32."""
    z, cfg, make = _mae3d(golden_dir)
    v = np.load(os.path.join(golden_dir, "mae3d_small_variants.npz"))
    imgs, noise = torch.from_numpy(v["imgs_hr"]).to(DEV), torch.from_numpy(v["noise_hr"]).to(DEV)

    def step(m):
        loss, pred, mask = m(imgs, mask_ratio=0.75, noise=noise)
        return loss, pred, mask
    _run_modes(make, step, "3-D MAE, 2-D/512 branch")


@pytest.mark.parametrize("num_classes", [2, 8])
def test_st_vit_is_autocast_invariant(num_classes):
    """num_classes = 2 sends the head through the fp32 ATen matmul (not a multiple of 8), 8 through the bf16 GEMM."""
    cfg = V.ViTSTConfig(num_frames=6, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=num_classes, embed_dim=128,
                        depth=2, num_heads=2, global_pool=True)
    P = V.init_from_shapes(V.vit_st_param_shapes(cfg), seed=11)
    x = torch.rand(3, 1, 6, 64, 64, generator=torch.Generator().manual_seed(2)).to(DEV)
    tgt = torch.tensor([0, 1, 1], device=DEV)

    def make():
        m = models_vit_st.VisionTransformer(num_frames=6, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=num_classes,
                                            embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, sep_pos_embed=True, cls_embed=True,
                                            global_pool=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
        m.load_state_dict(P, strict=True)
        return m.to(DEV).eval()

    def step(m):
        logits, emb = m(x, return_embeddings=True)
        assert logits.dtype == torch.float32
        return torch.nn.functional.cross_entropy(logits, tgt), logits, emb
    _run_modes(make, step, f"ST ViT, {num_classes} classes")


def test_mae2d_is_autocast_invariant(golden_dir):
    z = np.load(os.path.join(golden_dir, "mae2d_small.npz"))
    cfg = V.MAE2DConfig(**json.loads(str(z["cfg"])))
    P = V.mae2d_init(cfg, seed=int(z["param_seed"]))
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)

    def make():
        m = models_mae_2d.MaskedAutoencoderViT(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
                                                embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                                                decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
                                                decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=4,
                                                norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
        m.load_state_dict(P, strict=True)
        return m.to(DEV)

    def step(m):
        loss, pred, mask = m(imgs, mask_ratio=0.75, noise=noise)
        return loss, pred, mask
    _run_modes(make, step, "2-D MAE")


def test_coem_towers_and_clip_loss_are_autocast_invariant():
    """train_retclip.py:125-131: model(images, texts) AND the loss both run inside the autocast context."""
    c3 = V.ViTSTConfig(num_frames=6, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=64, embed_dim=128, depth=2,
                       num_heads=2, global_pool=True)
    c2 = V.ViT2DConfig(img_size=64, patch_size=16, in_chans=3, num_classes=64, embed_dim=128, depth=2, num_heads=2, global_pool=True)
    P3 = V.init_from_shapes(V.vit_st_param_shapes(c3), seed=51)
    P2 = V.init_from_shapes(V.vit2d_param_shapes(c2), seed=52)
    g = torch.Generator().manual_seed(5)
    vol = torch.rand(4, 1, 6, 64, 64, generator=g).to(DEV)
    ir = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    loss_fn = coem.ClipLoss()

    def make():
        kw = dict(mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
        m3 = models_vit_st.VisionTransformer(num_frames=6, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=64,
                                             embed_dim=128, depth=2, num_heads=2, sep_pos_embed=True, cls_embed=True, global_pool=True,
                                             dropout=0.0, **kw)
        m2 = models_vit.VisionTransformer(img_size=64, patch_size=16, in_chans=3, num_classes=64, embed_dim=128, depth=2, num_heads=2,
                                          qkv_bias=True, global_pool=True, **kw)
        m3.load_state_dict(P3, strict=True); m2.load_state_dict(P2, strict=True)
        return coem.CustomTextCLIP(m3.to(DEV), m2.to(DEV)).to(DEV).train()

    def step(m):
        fa, fb, ls = m(vol, ir)
        assert fa.dtype == torch.float32 and fb.dtype == torch.float32
        return loss_fn(fa, fb, ls), fa, fb
    _run_modes(make, step, "COEM towers + ClipLoss")


def test_block_and_flash_block_seams_are_autocast_invariant():
    """Seams 1 and 2 of SURVEY 8(b) used on their own: video_vit.Block(x) and create_block(...)(hidden, residual)."""
    torch.manual_seed(3)
    x = torch.randn(2, 65, 128, device=DEV)
    for kind in ("block", "flash"):
        def make():
            torch.manual_seed(7)
            if kind == "block":
                b = video_vit.Block(128, 2, mlp_ratio=4.0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
            else:
                b = video_vit.create_block(128, 2, 4.0, True, 0.0, 0.0, drop_path1=0.0, drop_path2=0.0,
                                           norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), act_layer=torch.nn.GELU, use_flash_attn=True,
                                           fused_bias_fc=False, fused_mlp=False, fused_dropout_add_ln=False, layer_idx=0, n_layer=2,
                                           last_layer_subset=False)
            return b.to(DEV).train()

        def step(b):
            xin = x.clone().requires_grad_(True)
            if kind == "block":
                y = b(xin)
                return y.float().square().mean(), y
            h, r = b(xin, None)
            return (h.float().square().mean() + r.float().square().mean()), h, r
        _run_modes(make, step, f"seam: {kind}")


def test_reference_shaped_loop_iteration_under_autocast(golden_dir):
    """engine_pretrain.py:110-170 / OCTCube/engine_pretrain.py:57-76: forward under autocast, loss_scaler(...) outside; the updated
    parameters and the returned gradient norm equal those of the same iteration without the context, bit for bit."""
    z, cfg, make = _mae3d(golden_dir)
    imgs, noise = torch.from_numpy(z["imgs"]).to(DEV), torch.from_numpy(z["noise"]).to(DEV)
    res = {}
    for mode in ("plain", "amp_fp16"):
        m = make()
        opt = foptim.FusedAdamW(misc.add_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.95))
        scaler = misc.NativeScalerWithGradNormCount()
        opt.zero_grad()
        with MODES[mode]():
            loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        norm = scaler(loss, opt, parameters=m.parameters(), update_grad=True)
        torch.cuda.synchronize()
        res[mode] = (loss.detach().clone(), norm.detach().clone(), {k: p.detach().clone() for k, p in m.named_parameters()})
    assert torch.equal(res["plain"][0], res["amp_fp16"][0])
    # the returned norm is a sum of per-chunk partial sums added with fp32 atomics (tensors above 65 536 elements have several chunks):
    # its last bit depends on their arrival order, with or without autocast -- the update below does not use it (no clipping)
    assert abs(float(res["plain"][1]) - float(res["amp_fp16"][1])) <= 1e-6 * float(res["plain"][1])
    for k, p in res["plain"][2].items():
        assert torch.equal(p, res["amp_fp16"][2][k]), k
