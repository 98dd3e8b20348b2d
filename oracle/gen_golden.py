#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (build container only).

    python oracle/gen_golden.py            # writes tests/golden/*.npz

The reference Python under /root/reference is imported here and ONLY here; it never travels to
the GPU box.  Its un-vendored third-party imports (timm, flash_attn, iopath, simplejson) are
absent from this image, so throw-away import shims are installed into ``sys.modules`` first:

  * ``timm.models.vision_transformer.Mlp``   -- restated from timm (pinned ``timm==0.3.2`` by
    OCTCube/main_pretrain.py:27; unpinned in requirement.txt): fc1 -> act_layer() -> drop -> fc2 -> drop
  * ``timm.models.vision_transformer.DropPath`` -- stochastic depth (identity at p=0 / eval)
  * ``timm.models.layers.to_2tuple``
  * ``flash_attn.models.vit.create_block``    -- raises: the flash path (flash-attn==2.5.2, CUDA
    only) cannot run here; BASELINE pins parity to the non-flash semantics
  * ``iopath.common.file_io.g_pathmgr``, ``simplejson`` -- logging plumbing only

Everything numerical other than ``Mlp`` is executed from the reference's own source files:
Pre-training/models_mae_joint_res_flash_attn.py and Pre-training/custom_util/video_vit.py.

Golden files (all small):
  mae3d_small.npz    reduced-width model (enc 128/2 heads of 64, dec 64/2 heads of 32), weights,
                     inputs, injected noise, reference loss / pred / mask / ids_restore / grads
  mae3d_small_hr.npz same weights, high-res (2-D/512-style) branch: T=3 frames through
                     high_res_patch_embed, un-interpolated spatial table
  masking.npz        random_masking on tie-free and tie-containing noise rows at L=5120
  train_utils.npz    get_grad_norm_, add_weight_decay grouping, adjust_learning_rate samples,
                     one torch AdamW step (the optimizer the reference driver constructs)
  vitl_pins.npz      full ViT-L scalar pins (loss, mask sum, pred samples, ids checksum)
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference/Pre-training"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def install_shims():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    timm = mod("timm"); models = mod("timm.models"); layers = mod("timm.models.layers")
    vt = mod("timm.models.vision_transformer"); tl = mod("timm.layers")
    timm.models = models; models.layers = layers; models.vision_transformer = vt; timm.layers = tl

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__(); self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            return x.div(keep) * (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop = nn.Dropout(drop)

        def forward(self, x):
            return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

    class TimmAttention(nn.Module):     # timm 0.3.2 vision_transformer.Attention
        def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
            super().__init__()
            self.num_heads = num_heads
            head_dim = dim // num_heads
            self.scale = qk_scale or head_dim ** -0.5
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.attn_drop = nn.Dropout(attn_drop)
            self.proj = nn.Linear(dim, dim)
            self.proj_drop = nn.Dropout(proj_drop)

        def forward(self, x):
            B, N, C = x.shape
            qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
            attn = (q @ k.transpose(-2, -1)) * self.scale
            attn = self.attn_drop(attn.softmax(dim=-1))
            x = (attn @ v).transpose(1, 2).reshape(B, N, C)
            return self.proj_drop(self.proj(x))

    class TimmBlock(nn.Module):         # timm 0.3.2 vision_transformer.Block
        def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0,
                     act_layer=nn.GELU, norm_layer=nn.LayerNorm):
            super().__init__()
            self.norm1 = norm_layer(dim)
            self.attn = TimmAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
            self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
            self.norm2 = norm_layer(dim)
            self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

        def forward(self, x):
            x = x + self.drop_path(self.attn(self.norm1(x)))
            return x + self.drop_path(self.mlp(self.norm2(x)))

    class TimmPatchEmbed(nn.Module):    # timm 0.3.2 vision_transformer.PatchEmbed
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            img_size = to_2tuple(img_size); patch_size = to_2tuple(patch_size)
            self.img_size = img_size; self.patch_size = patch_size
            self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class TimmVisionTransformer(nn.Module):   # timm 0.3.2 vision_transformer.VisionTransformer (restated: timm is un-vendored)
        def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                     mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
                     hybrid_backbone=None, norm_layer=nn.LayerNorm):
            super().__init__()
            self.num_classes = num_classes
            self.num_features = self.embed_dim = embed_dim
            self.patch_embed = TimmPatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
            num_patches = self.patch_embed.num_patches
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
            self.pos_drop = nn.Dropout(p=drop_rate)
            dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
            self.blocks = nn.ModuleList([
                TimmBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                          attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])
            self.norm = norm_layer(embed_dim)
            self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()

        def forward_features(self, x):
            B = x.shape[0]
            x = self.patch_embed(x)
            x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
            x = self.pos_drop(x + self.pos_embed)
            for blk in self.blocks:
                x = blk(x)
            return self.norm(x)[:, 0]

        def forward(self, x):
            return self.head(self.forward_features(x))

    layers.to_2tuple = to_2tuple; tl.to_2tuple = to_2tuple
    vt.DropPath = DropPath; vt.Mlp = Mlp; vt.PatchEmbed = TimmPatchEmbed; vt.Block = TimmBlock
    vt.VisionTransformer = TimmVisionTransformer
    tv = mod("torchvision"); tvt = mod("torchvision.transforms"); tv.transforms = tvt

    class _Permissive:      # OCTCube/util/misc.py builds image transforms at import time (e.g. tf.Lambda); never called here
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): raise RuntimeError("torchvision is not installed")
    def _tv_getattr(name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Permissive
    tvt.__getattr__ = _tv_getattr
    tvt.__file__ = "<shim>"; tv.__file__ = "<shim>"

    fa = mod("flash_attn"); fam = mod("flash_attn.models"); fav = mod("flash_attn.models.vit")
    fa.models = fam; fam.vit = fav

    def create_block(*a, **k):
        raise RuntimeError("flash-attn is CUDA-only; the oracle uses the non-flash path")
    fav.create_block = create_block

    io = mod("iopath"); ioc = mod("iopath.common"); iof = mod("iopath.common.file_io")
    io.common = ioc; ioc.file_io = iof

    class _PM:
        def open(self, *a, **k):
            return open(*a, **k)
    iof.g_pathmgr = _PM()
    sj = mod("simplejson")
    sj.dumps = json.dumps; sj.loads = json.loads


def build_reference(cfg, use_grad=True):
    import models_mae_joint_res_flash_attn as ref
    from functools import partial
    model = ref.MaskedAutoencoderViT(
        input_size=cfg.input_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
        embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
        decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
        decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio,
        norm_layer=partial(nn.LayerNorm, eps=cfg.ln_eps), norm_pix_loss=cfg.norm_pix_loss,
        num_frames=cfg.num_frames, t_patch_size=cfg.t_patch_size, sep_pos_embed=True,
        cls_embed=True, pred_t_dim=cfg.pred_t_dim, high_res_input_size=cfg.high_res_input_size,
        use_flash_attn=False)
    return model


def run_reference(model, imgs, noise_seed, mask_ratio, frame_loss=False):
    """Runs the reference forward with its own torch.rand call reproducing `noise`:
    nothing before random_masking consumes RNG (models_mae…:374-406)."""
    torch.manual_seed(noise_seed)
    out = model(imgs, mask_ratio=mask_ratio, frame_loss=frame_loss)
    return out


def tie_free(noise):
    s, _ = torch.sort(noise, dim=1)
    return bool((s[:, 1:] != s[:, :-1]).all())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-vitl", action="store_true")
    args = ap.parse_args()
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    import builtins
    from oracle import mae3d_ref as O

    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(8)

    # ------------------------------------------------------------------ small model
    cfg = O.MAEConfig(input_size=64, patch_size=16, in_chans=1, embed_dim=128, depth=2, num_heads=2,
                      decoder_embed_dim=64, decoder_depth=2, decoder_num_heads=2, num_frames=12,
                      t_patch_size=3, pred_t_dim=12, high_res_input_size=128)
    model = build_reference(cfg)
    P = O.init_params(cfg, seed=7, bias_std=0.02)
    missing = model.load_state_dict(P, strict=True)
    assert set(model.state_dict().keys()) == set(P.keys()), "key mismatch vs oracle.param_shapes"
    g = torch.Generator().manual_seed(11)
    imgs = torch.rand(2, 1, 12, 64, 64, generator=g)
    noise_seed = None
    for s in range(1000, 2000):                      # seed-search a tie-free noise draw
        torch.manual_seed(s)
        nz = torch.rand(2, cfg.num_patches)
        if tie_free(nz):
            noise_seed = s; noise = nz; break
    model.train()
    (loss, fl), pred, mask = run_reference(model, imgs, noise_seed, 0.75, frame_loss=True)
    model.zero_grad()
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
             for k, p in model.named_parameters()}
    # ids_restore from the reference's own random_masking on the same noise
    torch.manual_seed(noise_seed)
    xm, m2, ids_restore, ids_keep = model.random_masking(torch.zeros(2, cfg.num_patches, 4), 0.75)
    assert torch.equal(m2, mask)
    # weights are regenerated by oracle.init_params(cfg, seed=7, bias_std=0.02) (torch CPU generator, same
    # image on both sides); a checksum pins that.  Large gradients are stored strided (every 7th) + norm.
    save = {"param_seed": 7, "param_bias_std": 0.02,
            "param_checksum": np.array([float(v.double().sum()) for v in P.values()]).sum(),
            "param_abs_checksum": np.array([float(v.double().abs().sum()) for v in P.values()]).sum()}
    for k, v in grads.items():
        save[f"gnorm/{k}"] = float(v.double().norm())
        save[f"grad/{k}"] = v.numpy() if v.numel() <= 8192 else v.flatten()[::7].numpy()
    save.update(imgs=imgs.numpy(), noise=noise.numpy(), loss=loss.detach().numpy(), pred=pred.detach().numpy(),
                mask=mask.numpy(), ids_restore=ids_restore.numpy(), ids_keep=ids_keep.numpy(),
                frame_losses=fl.detach().numpy(), cfg=json.dumps(cfg.__dict__), mask_ratio=0.75)
    np.savez_compressed(os.path.join(out_dir, "mae3d_small.npz"), **save)
    print("small: loss", float(loss), "pred", tuple(pred.shape))

    # norm_pix_loss variant + mask_ratio 0.9 (shipped launch config) on the same weights
    model.norm_pix_loss = True
    loss_np, pred_np, mask_np = run_reference(model, imgs, noise_seed, 0.9)
    model.norm_pix_loss = False
    # high-res branch (512-style 2-D batch: T = t_patch frames) -- pos-embeds NOT interpolated
    imgs_hr = torch.rand(2, 1, 3, 128, 128, generator=g)
    for s in range(3000, 4000):
        torch.manual_seed(s)
        nz = torch.rand(2, 64)
        if tie_free(nz):
            hr_seed = s; noise_hr = nz; break
    loss_hr, pred_hr, mask_hr = run_reference(model, imgs_hr, hr_seed, 0.75)
    np.savez_compressed(os.path.join(out_dir, "mae3d_small_variants.npz"),
                        loss_normpix_r90=loss_np.detach().numpy(), mask_r90=mask_np.numpy(),
                        pred_r90=pred_np.detach().numpy(),
                        imgs_hr=imgs_hr.numpy(), noise_hr=noise_hr.numpy(), loss_hr=loss_hr.detach().numpy(),
                        pred_hr=pred_hr.detach().numpy(), mask_hr=mask_hr.numpy())
    print("variants: normpix/r90", float(loss_np), "hr", float(loss_hr))

    # ------------------------------------------------------------------ masking at L=5120
    L = 5120
    rows_free, rows_tie = [], []
    s = 0
    while len(rows_free) < 4 or len(rows_tie) < 4:
        torch.manual_seed(50_000 + s); s += 1
        nz = torch.rand(1, L)
        (rows_free if tie_free(nz) else rows_tie).append(nz)
    nz_free = torch.cat(rows_free[:4]); nz_tie = torch.cat(rows_tie[:4])
    model_dummy = model

    def ref_mask(nz, ratio):
        # call the reference's random_masking with torch.rand patched to return our rows
        real = torch.rand
        torch.rand = lambda *a, **k: nz.clone()
        try:
            x = torch.arange(nz.shape[0] * L, dtype=torch.float32).view(nz.shape[0], L, 1)
            xm, mk, ir, ik = model_dummy.random_masking(x, ratio)
        finally:
            torch.rand = real
        return xm, mk, ir, ik
    xm, mk, ir, ik = ref_mask(nz_free, 0.75)
    xm9, mk9, ir9, ik9 = ref_mask(nz_free, 0.9)
    xmt, mkt, irt, ikt = ref_mask(nz_tie, 0.75)
    np.savez_compressed(os.path.join(out_dir, "masking.npz"),
                        noise_free=nz_free.numpy(), mask_free=mk.numpy(), ids_restore_free=ir.numpy(),
                        ids_keep_free=ik.numpy(), mask_free_r90=mk9.numpy(), ids_keep_free_r90=ik9.numpy(),
                        noise_tie=nz_tie.numpy(), mask_tie=mkt.numpy(), ids_restore_tie=irt.numpy(),
                        ids_keep_tie=ikt.numpy())
    print("masking: tie-free rows", nz_free.shape, "tie rows", nz_tie.shape)

    # ------------------------------------------------------------------ train utilities
    import custom_util.lr_sched as lr_sched
    import custom_util.misc as ref_misc     # imports cleanly with the shims (psutil/matplotlib present)
    ns = {"get_grad_norm_": ref_misc.get_grad_norm_, "add_weight_decay": ref_misc.add_weight_decay}
    gl = [g_ for g_ in grads.values()]

    class _P:   # parameter stand-in with .grad
        def __init__(self, g_): self.grad = g_
    gn = ns["get_grad_norm_"]([_P(g_) for g_ in gl])
    groups = ns["add_weight_decay"](model, 0.05)
    id2name = {id(p): n for n, p in model.named_parameters()}
    no_decay = [id2name[id(p)] for p in groups[0]["params"]]
    decay = [id2name[id(p)] for p in groups[1]["params"]]

    class A: pass
    a = A(); a.lr = 1.6e-3; a.min_lr = 1e-6; a.warmup_epochs = 5; a.epochs = 50
    eps_ = [0.0, 0.37, 2.5, 5.0, 5.01, 17.3, 49.99]

    class Opt:
        def __init__(self): self.param_groups = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}]
    lrs = []
    for e in eps_:
        o = Opt(); lrs.append([lr_sched.adjust_learning_rate(o, e, a), o.param_groups[0]["lr"], o.param_groups[1]["lr"]])
    # one AdamW step exactly as the driver builds the optimizer (main_pretrain…:441-455)
    m2 = build_reference(cfg); m2.load_state_dict(P)
    pg = ns["add_weight_decay"](m2, 0.05)
    opt = torch.optim.AdamW(pg, lr=1.6e-3, betas=(0.9, 0.95))
    for n_, p_ in m2.named_parameters():
        p_.grad = grads[n_].clone()
    opt.step()
    for n_, p_ in m2.named_parameters():
        p_.grad = 0.5 * grads[n_]
    opt.step()
    after = {f"adamw2/{k}": v.detach().numpy() for k, v in m2.state_dict().items()
             if k in ("blocks.0.attn.q.weight", "blocks.1.mlp.fc1.bias", "norm.weight", "mask_token",
                      "pos_embed_temporal", "decoder_blocks.1.attn.proj.weight")}
    np.savez_compressed(os.path.join(out_dir, "train_utils.npz"), grad_norm=gn.numpy(),
                        no_decay=json.dumps(no_decay), decay=json.dumps(decay), lr_epochs=np.array(eps_),
                        lr_values=np.array(lrs), **after)
    print("train utils: grad_norm", float(gn), "groups", len(no_decay), len(decay))

    # ------------------------------------------------------------------ ST fine-tune ViT + 2-D MAE (OCTCube/)
    OC = "/root/reference/OCTCube"
    for m_ in [k for k in list(sys.modules) if k == "util" or k.startswith("util.")]:
        del sys.modules[m_]
    sys.path.insert(0, OC)
    os.chdir(OC)
    from functools import partial
    from oracle import vit_ref as V
    import models_vit_st_flash_attn as ref_st
    cfgS = V.ViTSTConfig(num_frames=12, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=8, embed_dim=128,
                         depth=2, num_heads=2, global_pool=True)
    mS = ref_st.VisionTransformer(num_frames=12, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=8,
                                  embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                  sep_pos_embed=True, cls_embed=True, global_pool=True, drop_path_rate=0.0)
    PS = V.init_from_shapes(V.vit_st_param_shapes(cfgS), seed=21)
    mS.load_state_dict(PS, strict=True)
    mS.eval()
    xs = torch.rand(3, 1, 12, 64, 64, generator=torch.Generator().manual_seed(5))
    logits, emb = mS(xs, return_embeddings=True)
    tgt = torch.tensor([1, 5, 2])
    lossS = torch.nn.functional.cross_entropy(logits, tgt)
    mS.zero_grad(); lossS.backward()
    gS = {k: p.grad for k, p in mS.named_parameters() if p.grad is not None}
    mS2 = ref_st.VisionTransformer(num_frames=12, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=8,
                                   embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                   sep_pos_embed=True, cls_embed=True, global_pool=False)
    mS2.load_state_dict(PS, strict=True); mS2.eval()
    logits_cls = mS2(xs)
    saveS = {"param_seed": 21, "x": xs.numpy(), "logits": logits.detach().numpy(), "embedding": emb.detach().numpy(),
             "logits_cls": logits_cls.detach().numpy(), "target": tgt.numpy(), "loss": lossS.detach().numpy(),
             "cfg": json.dumps(cfgS.__dict__),
             "param_checksum": np.array([float(v.double().sum()) for v in PS.values()]).sum()}
    for k, v in gS.items():
        saveS[f"gnorm/{k}"] = float(v.double().norm())
        saveS[f"grad/{k}"] = v.numpy() if v.numel() <= 8192 else v.flatten()[::7].numpy()
    np.savez_compressed(os.path.join(out_dir, "vit_st_small.npz"), **saveS)
    print("ST ViT: logits", logits.detach().numpy()[0, :3], "loss", float(lossS))

    import models_mae as ref_2d
    cfg2 = V.MAE2DConfig(img_size=64, patch_size=16, in_chans=3, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=64,
                         decoder_depth=2, decoder_num_heads=2)
    m2d = ref_2d.MaskedAutoencoderViT(img_size=64, patch_size=16, in_chans=3, embed_dim=128, depth=2, num_heads=2,
                                      decoder_embed_dim=64, decoder_depth=2, decoder_num_heads=2, mlp_ratio=4,
                                      norm_layer=partial(nn.LayerNorm, eps=1e-6))
    P2 = V.mae2d_init(cfg2, seed=31)
    m2d.load_state_dict(P2, strict=True)
    assert torch.allclose(m2d.pos_embed, P2["pos_embed"]) and set(m2d.state_dict()) == set(P2)
    x2 = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(6))        # BASELINE config 1: randn B-scans
    for sd in range(7000, 8000):
        torch.manual_seed(sd)
        nz2 = torch.rand(2, cfg2.num_patches)
        if tie_free(nz2):
            break
    torch.manual_seed(sd)
    loss2, pred2, mask2 = m2d(x2, mask_ratio=0.75)
    m2d.zero_grad(); loss2.backward()
    torch.manual_seed(sd)
    _, _, ir2 = m2d.random_masking(torch.zeros(2, cfg2.num_patches, 4), 0.75)
    save2 = {"param_seed": 31, "imgs": x2.numpy(), "noise": nz2.numpy(), "loss": loss2.detach().numpy(), "pred": pred2.detach().numpy(),
             "mask": mask2.numpy(), "ids_restore": ir2.numpy(), "cfg": json.dumps(cfg2.__dict__),
             "param_checksum": np.array([float(v.double().sum()) for v in P2.values()]).sum()}
    for k, p_ in m2d.named_parameters():
        g_ = p_.grad if p_.grad is not None else torch.zeros_like(p_)
        save2[f"gnorm/{k}"] = float(g_.double().norm())
        save2[f"grad/{k}"] = g_.numpy() if g_.numel() <= 8192 else g_.flatten()[::7].numpy()
    np.savez_compressed(os.path.join(out_dir, "mae2d_small.npz"), **save2)
    print("2-D MAE: loss", float(loss2), "pred", tuple(pred2.shape))
    os.chdir(REF)

    # ------------------------------------------------------------------ ViT-L scalar pins
    if not args.skip_vitl:
        cfgL = O.VIT_L
        modelL = build_reference(cfgL)
        PL = O.init_params(cfgL, seed=0, bias_std=0.0)
        modelL.load_state_dict(PL, strict=True)
        gi = torch.Generator().manual_seed(0)
        imgsL = torch.rand(1, 1, 60, 256, 256, generator=gi)
        for s in range(1, 200):
            torch.manual_seed(s)
            nz = torch.rand(1, 5120)
            if tie_free(nz):
                seedL = s; break
        with torch.no_grad():
            lossL, predL, maskL = run_reference(modelL, imgsL, seedL, 0.75)
        torch.manual_seed(seedL)
        _, _, irL, ikL = modelL.random_masking(torch.zeros(1, 5120, 1), 0.75)
        idx = torch.arange(0, 5120 * 768, 39_989)
        np.savez_compressed(os.path.join(out_dir, "vitl_pins.npz"), loss=lossL.numpy(), noise_seed=seedL,
                            mask_sum=maskL.sum().numpy(), pred_idx=idx.numpy(),
                            pred_samples=predL.flatten()[idx].numpy(),
                            pred_l2=predL.double().norm().numpy(), ids_restore=irL.numpy().astype(np.int32),
                            n_params=sum(p.numel() for p in modelL.parameters()))
        print("ViT-L: loss", float(lossL), "mask", float(maskL.sum()), "params",
              sum(p.numel() for p in modelL.parameters()))


if __name__ == "__main__":
    main()
