"""Diagnostic: stage-by-stage distance between the HIP 3-D MAE forward and the whole-model rounding-point model
(oracle/bf16_points_mae.py) on the "mid" configuration of tests/test_gpu_rounding_model.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import torch
from octcubem_amd import models_mae
from oracle import mae3d_ref as O, bf16_points_mae as M

def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

cfg = O.MAEConfig(input_size=96, in_chans=1, embed_dim=256, depth=3, num_heads=4, decoder_embed_dim=128, decoder_depth=2,
                  decoder_num_heads=4, num_frames=15, t_patch_size=3, pred_t_dim=15, high_res_input_size=192)
P = O.init_params(cfg, seed=3, bias_std=0.02)
imgs = torch.rand(3, 1, 15, 96, 96, generator=torch.Generator().manual_seed(1))
noise = torch.rand(3, cfg.num_patches, generator=torch.Generator().manual_seed(2))
T = {}
loss_r, pred_r, mask_r, ids_r, G = M.forward_backward(P, imgs, cfg, 0.75, noise, trace=T)
m = models_mae.MaskedAutoencoderViT(
    input_size=cfg.input_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depth=cfg.depth,
    num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
    decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio, norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps),
    num_frames=cfg.num_frames, t_patch_size=cfg.t_patch_size, sep_pos_embed=True, cls_embed=True, pred_t_dim=cfg.pred_t_dim,
    high_res_input_size=cfg.high_res_input_size)
m.load_state_dict(P, strict=True)
m = m.cuda().train()
cap = {}
for i, blk in enumerate(m.blocks):
    blk.register_forward_hook(lambda mod, inp, out, i=i: cap.__setitem__(f"blocks.{i}", out))
    if i == 0:
        blk.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("enc_in", inp[0]))
for i, blk in enumerate(m.decoder_blocks):
    blk.register_forward_hook(lambda mod, inp, out, i=i: cap.__setitem__(f"decoder_blocks.{i}", out))
    if i == 0:
        blk.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("dec_in", inp[0]))
loss, pred, mask = m(imgs.cuda(), mask_ratio=0.75, noise=noise.cuda())
for k in ["enc_in"] + [f"blocks.{i}" for i in range(cfg.depth)] + ["dec_in"] + [f"decoder_blocks.{i}" for i in range(cfg.decoder_depth)]:
    print(f"{k:20s} rel {rel(cap[k], T[k]):.3e}   (HIP norm {float(cap[k].double().norm()):.4f}, model {float(T[k].norm()):.4f})")
print("pred", rel(pred, pred_r), "loss", float(loss), float(loss_r))
