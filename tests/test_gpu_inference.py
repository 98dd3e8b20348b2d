"""GPU: octcubem_amd.inference_utils -- create_models + load_model + forward -- against tests/golden/inference_small.npz, which
oracle/gen_golden_inference.py produced by running the reference's own inference_utils.py (model construction from the same
``args``, checkpoint load with the temporal table interpolated 6 -> 4, eval forward).  Tolerance: bf16 operands, logits rel-L2 <= 1e-2."""
import argparse
import os
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from octcubem_amd import inference_utils as I, models_vit_st
from oracle import vit_ref as V
from tests.conftest import parity


def rel(a, b):
    a = torch.as_tensor(a).detach().double().flatten().cpu(); b = torch.as_tensor(b).detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _args(ckpt):
    return argparse.Namespace(model_type="3D_st_flash_attn", model="vit_tiny_test", num_frames=12, t_patch_size=3, input_size=64,
                              nb_classes=8, drop_path=0.0, global_pool=True, sep_pos_embed=True, cls_embed=True, ckpt=ckpt)


def test_create_models_and_load_model_match_the_reference_inference_path(golden_dir, tmp_path, monkeypatch):
    z = np.load(os.path.join(golden_dir, "inference_small.npz"))
    monkeypatch.setattr(models_vit_st, "vit_tiny_test", lambda **kw: models_vit_st.VisionTransformer(
        patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kw), raising=False)
    # the checkpoint the golden was made with: a model trained on 18 frames (6 temporal positions), same 4 x 4 spatial grid
    cfg_ck = V.ViTSTConfig(num_frames=18, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=8, embed_dim=128, depth=2,
                           num_heads=2, global_pool=True)
    sd = V.init_from_shapes(V.vit_st_param_shapes(cfg_ck), seed=int(z["ckpt_seed"]))
    ck = str(tmp_path / "ckpt.pth")
    torch.save({"model": sd}, ck)
    model = I.create_models(_args(ck))
    assert next(model.parameters()).is_cuda and not model.use_flash_attn          # the swallowed use_flash_attention keyword
    assert sorted(model.state_dict().keys()) == [str(k) for k in z["state_keys"]]
    assert torch.allclose(model.pos_embed_temporal.cpu(), torch.from_numpy(z["loaded/pos_embed_temporal"]), atol=1e-6, rtol=0)
    assert torch.equal(model.pos_embed_spatial.cpu(), torch.from_numpy(z["loaded/pos_embed_spatial"]))
    model.eval()
    x = torch.rand(2, 1, 12, 64, 64, generator=torch.Generator().manual_seed(int(z["x_seed"])))
    assert abs(float(x.double().sum()) - float(z["x_checksum"])) < 1e-6
    with torch.no_grad():
        logits = model(x.cuda())
    parity("inference_small/logits", rel(logits, z["logits"]), 1e-2)
    # a checkpoint of another SPATIAL grid: util/misc.py's interpolate_pos_embed does not resize pos_embed_spatial, the strict load fails
    cfg_bad = V.ViTSTConfig(num_frames=12, t_patch_size=3, img_size=128, patch_size=16, in_chans=1, num_classes=8, embed_dim=128, depth=2,
                            num_heads=2, global_pool=True)
    bad = str(tmp_path / "bad.pth")
    torch.save({"model": V.init_from_shapes(V.vit_st_param_shapes(cfg_bad), seed=32)}, bad)
    assert str(z["spatial_mismatch_error"]) == "RuntimeError: pos_embed_spatial"
    with pytest.raises(RuntimeError, match="pos_embed_spatial"):
        I.create_models(_args(bad))
    # no checkpoint: the model as constructed; another model type: refused
    a = _args(None)
    assert isinstance(I.create_models(a), models_vit_st.VisionTransformer)
    a.model_type = "2D"
    with pytest.raises(ValueError):
        I.create_models(a)
