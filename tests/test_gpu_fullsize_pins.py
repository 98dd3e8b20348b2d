"""Full-size parity against pins produced by the REAL reference (oracle/gen_golden_fullsize.py, run in the build container):

  * BASELINE config 2 -- ViT-L 3-D MAE, (1,1,60,256,256), mask 0.75, decoder 512x8x16 (N = 1281 / 5121, head_dim 64 / 32):
    forward AND backward of Pre-training/models_mae_joint_res_flash_attn.py:669-680 -- loss, global gradient norm, the
    2-norm of every parameter's gradient, strided samples of 20 gradient tensors (tests/golden/vitl_bwd_pins.npz);
  * BASELINE config 4 -- ViT-L spatio-temporal fine-tune model, (1,1,60,256,256) -> 8 logits, N = 5121, head_dim 64
    (OCTCube/models_vit_st_flash_attn.py:181-258): logits, pooled embedding, cross-entropy loss and the same gradient pins
    (tests/golden/vit_st_l_pins.npz).

The HIP path computes with bf16 MFMA operands (weights, activations, P, dS rounded to bf16; fp32 accumulation, residual
stream, statistics and weight gradients), the reference in fp32.  The asserted bounds are the measured errors x ~1.5 (printed
by the tests with -s; table in DESIGN.md section 2); tests/test_gpu_rounding_model.py shows where they come from: the same
computation in fp32/fp64 with bf16 rounding inserted at the HIP path's rounding points reproduces the HIP results to <= 1e-3.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from octcubem_amd import models_mae, models_vit_st
from oracle import mae3d_ref as O
from oracle import vit_ref as V

DEV = "cuda"


def rel(a, b):
    a = torch.as_tensor(a).detach().double().flatten().cpu(); b = torch.as_tensor(b).detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def grad_report(model, z, label):
    """Compares every gradient norm and the sampled gradient tensors with the pins; returns the summary numbers."""
    names = json.loads(str(z["grad_names"]))
    ref_norms = dict(zip(names, z["grad_norms"]))
    total = float(z["global_grad_norm"])
    params = dict(model.named_parameters())
    mine_sq = 0.0
    worst_norm, worst_norm_name = 0.0, None
    for k in names:
        g = params[k].grad
        gn = 0.0 if g is None else float(g.double().norm())
        mine_sq += gn * gn
        if ref_norms[k] >= 1e-3 * total:                 # tensors that carry the gradient: norm within a relative bound
            e = abs(gn - ref_norms[k]) / ref_norms[k]
            if e > worst_norm:
                worst_norm, worst_norm_name = e, k
        else:                                            # (near-)zero in the reference: absolute bound vs the global norm
            assert abs(gn - ref_norms[k]) <= 2e-3 * total, (k, gn, ref_norms[k])
    samples, tiny = {}, {}
    for key in z.files:
        if not key.startswith("gsample/"):
            continue
        k = key[len("gsample/"):]
        step = int(z[f"gstep/{k}"])
        ref = torch.from_numpy(z[key])
        mine = params[k].grad.flatten()[::step][: ref.numel()]
        if ref_norms[k] >= 1e-3 * total:
            samples[k] = rel(mine, ref)
        else:
            # A tensor whose whole gradient is < 1e-3 of the global norm (deep-layer q / k weights of a randomly initialised
            # model at N = 5121: near-uniform softmax, dS = P (dP - delta) cancels to ~2e-5 of the total) sits below the bf16
            # noise floor of the products it is a difference of: bounded ABSOLUTELY, scaled from the sample to the tensor.
            scale = (params[k].numel() / ref.numel()) ** 0.5
            tiny[k] = float((mine.cpu().double() - ref.double()).norm()) * scale / total
    gnorm = mine_sq ** 0.5
    if tiny:
        print(f"\n[{label}] tensors below 1e-3 of the global norm, absolute error / global norm: "
              + ", ".join(f"{k} {v:.2e}" for k, v in tiny.items()))
        assert max(tiny.values()) <= 1e-3
    print(f"\n[{label}] global grad norm {gnorm:.6g} (reference {total:.6g}, rel {abs(gnorm - total) / total:.2e}); "
          f"worst per-tensor norm error {worst_norm:.2e} ({worst_norm_name}); sampled gradient rel-L2: "
          + ", ".join(f"{k} {v:.2e}" for k, v in sorted(samples.items(), key=lambda kv: -kv[1])[:6])
          + f" ... median {float(np.median(list(samples.values()))):.2e}")
    return gnorm, total, worst_norm, samples


# Both attention-backward dispatches at full size: "fused" = the single-pass kernels the benchmark shapes take (the
# conftest fixture switches the fill rule off), "fill_rule" = ops.attn_bwd_use_fused as shipped -- one volume x 16 heads does not fill
# the chip, so the backward runs as the dQ + dK/dV pair with plain linear_dgrad and no delta: what the reference's shipped recipe
# (1 volume per GPU), the fine-tune recipes and config 5 at small batches execute.  Same bounds for both.
BWD_DISPATCH = pytest.mark.parametrize("dispatch", ["fused", pytest.param("fill_rule", marks=pytest.mark.small_batch_rule)])


@BWD_DISPATCH
@pytest.mark.skipif(os.environ.get("OCTMAE_SKIP_VITL", "0") == "1", reason="full-size run disabled")
def test_vitl_3dmae_backward_vs_reference_pins(golden_dir, dispatch):
    from octcubem_amd import ops
    assert ops.attn_bwd_use_fused(1, 16, 32, DEV) == (dispatch == "fused")
    z = np.load(os.path.join(golden_dir, "vitl_bwd_pins.npz"))
    P = O.init_params(O.VIT_L, seed=int(z["param_seed"]), bias_std=float(z["param_bias_std"]))
    m = models_mae.octcube_vit_large_3dmae()
    m.load_state_dict(P, strict=True)
    m = m.to(DEV).train()
    imgs = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(int(z["img_seed"])))
    torch.manual_seed(int(z["noise_seed"]))
    noise = torch.rand(1, 5120)
    loss, pred, mask = m(imgs.to(DEV), mask_ratio=float(z["mask_ratio"]), noise=noise.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    rl = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    print(f"\n[ViT-L 3-D MAE] loss {float(loss):.6f} (reference {float(z['loss']):.6f}, rel {rl:.2e})")
    assert rl <= 1e-3                                             # north-star bound
    unused = set(json.loads(str(z["unused"])))
    assert unused == {"high_res_patch_embed.proj.weight", "high_res_patch_embed.proj.bias"}
    for k in unused:
        g = dict(m.named_parameters())[k].grad
        assert g is None or float(g.abs().max()) == 0.0
    gnorm, total, worst_norm, samples = grad_report(m, z, "ViT-L 3-D MAE")
    assert abs(gnorm - total) <= 2e-3 * total                      # measured 8.6e-4 (r02, MI355X)
    assert worst_norm <= 7e-3                                      # measured 4.4e-3 (decoder_blocks.3.attn.k.weight)
    assert max(samples.values()) <= 2e-2 and float(np.median(list(samples.values()))) <= 1.5e-2   # measured 1.34e-2 / 1.03e-2


@BWD_DISPATCH
@pytest.mark.skipif(os.environ.get("OCTMAE_SKIP_VITL", "0") == "1", reason="full-size run disabled")
def test_vitl_st_finetune_model_vs_reference_pins(golden_dir, dispatch):
    from octcubem_amd import ops
    assert ops.attn_bwd_use_fused(1, 16, 64, DEV) == (dispatch == "fused")
    z = np.load(os.path.join(golden_dir, "vit_st_l_pins.npz"))
    cfg = V.ViTSTConfig(**json.loads(str(z["cfg"])))
    assert (cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.num_frames, cfg.img_size) == (1024, 24, 16, 60, 256)
    P = V.init_from_shapes(V.vit_st_param_shapes(cfg), seed=int(z["param_seed"]), bias_std=float(z["param_bias_std"]))
    m = models_vit_st.vit_large_patch16(num_frames=60, t_patch_size=3, img_size=256, in_chans=1, num_classes=8, global_pool=True,
                                        sep_pos_embed=True, cls_embed=True, drop_path_rate=0.0)
    assert set(m.state_dict()) == set(P)
    m.load_state_dict(P, strict=True)
    m = m.to(DEV).eval()                                           # dropout before the head off, as in the pin run
    x = torch.rand(1, 1, 60, 256, 256, generator=torch.Generator().manual_seed(int(z["img_seed"]))).to(DEV)
    logits, emb = m(x, return_embeddings=True)
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(z["target"]).to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    e_log, e_emb = rel(logits, z["logits"]), rel(emb, z["embedding"])
    rl = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    print(f"\n[ViT-L ST] logits rel-L2 {e_log:.2e}, embedding rel-L2 {e_emb:.2e}, loss {float(loss):.5f} "
          f"(reference {float(z['loss']):.5f}, rel {rl:.2e})")
    assert e_log <= 1.5e-2 and e_emb <= 1.5e-2 and rl <= 1.5e-2   # 5121 tokens x 24 bf16 layers
    assert set(json.loads(str(z["unused"]))) == {"norm.weight", "norm.bias"}     # computed-but-unused final norm (reference :247-249)
    gnorm, total, worst_norm, samples = grad_report(m, z, "ViT-L ST")
    assert abs(gnorm - total) <= 2e-3 * total                      # measured 5.4e-4 (r02, MI355X)
    assert worst_norm <= 3e-3                                      # measured 1.4e-3 (blocks.0.norm1.weight)
    assert max(samples.values()) <= 1.5e-2 and float(np.median(list(samples.values()))) <= 1e-2   # measured 9.1e-3 / 6.7e-3
