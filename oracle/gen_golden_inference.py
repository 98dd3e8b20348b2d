#!/usr/bin/env python3
"""Golden vectors for the inference helpers (SURVEY §8f N2 "lets the build run real inference"), made by running the REAL
reference module /root/reference/inference_utils.py (build container only).

    python oracle/gen_golden_inference.py            # writes tests/golden/inference_small.npz

What runs from the reference's own source: inference_utils.create_models (-> OCTCube/models_vit_st_flash_attn.vit_large_patch16-style
factory call with its own keyword set, incl. the swallowed ``use_flash_attention=True``), inference_utils.load_model (checkpoint
['model'], util/misc.interpolate_pos_embed -- a no-op for the separable tables -- interpolate_temporal_pos_embed 6 -> 4, strict load;
and the size-mismatch error of that strict load for a checkpoint of another SPATIAL grid, which that interpolate_pos_embed leaves alone), a forward of the loaded model
in eval mode, and inference_utils.parse_all_output on seeded probabilities (both branches of its "Normal" rule).

Shims on top of gen_golden.install_shims(): ``monai``, ``pydicom`` (absent; data loading, never called), a stand-in for the
``OCTCube`` package object so that importing ``OCTCube.models_vit_st_flash_attn`` does not run the package's __init__ (which
imports every model family), ``OCTCube.util.PatientDataset_inhouse`` (MONAI transforms; never called), and ``nn.Module.cuda``
(the reference moves the model to the GPU; none here).  The model factory used is a reduced-width one registered under the
reference module's namespace (``vit_tiny_test``), built from the reference's own VisionTransformer class.
"""
import argparse
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
OC = REF + "/OCTCube"


def main():
    from oracle.gen_golden import install_shims
    from oracle import vit_ref as V
    install_shims()
    for name in ("monai", "pydicom"):
        sys.modules[name] = types.ModuleType(name)
    pkg = types.ModuleType("OCTCube"); pkg.__path__ = [OC]
    sys.modules["OCTCube"] = pkg
    pds = types.ModuleType("OCTCube.util.PatientDataset_inhouse"); pds.create_3d_transforms = lambda *a, **k: None
    sys.modules["OCTCube.util.PatientDataset_inhouse"] = pds
    sys.path.insert(0, OC); sys.path.insert(0, REF)
    os.chdir(OC)
    nn.Module.cuda = lambda self, *a, **k: self
    import inference_utils as R
    ref_st = sys.modules["OCTCube.models_vit_st_flash_attn"]

    def vit_tiny_test(**kwargs):
        return ref_st.VisionTransformer(patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    ref_st.vit_tiny_test = vit_tiny_test

    # checkpoint of a model trained at 64 x 64 x 18 frames (4 x 4 spatial, 6 temporal) loaded into a 64 x 64 x 12 model (4 x 4, 4)
    cfg_ck = V.ViTSTConfig(num_frames=18, t_patch_size=3, img_size=64, patch_size=16, in_chans=1, num_classes=8, embed_dim=128,
                           depth=2, num_heads=2, global_pool=True)
    sd = V.init_from_shapes(V.vit_st_param_shapes(cfg_ck), seed=31)
    ckpt_path = "/tmp/_inference_small_ckpt.pth"
    torch.save({"model": {k: v.clone() for k, v in sd.items()}}, ckpt_path)
    args = argparse.Namespace(model_type="3D_st_flash_attn", model="vit_tiny_test", num_frames=12, t_patch_size=3, input_size=64,
                              nb_classes=8, drop_path=0.0, global_pool=True, sep_pos_embed=True, cls_embed=True, ckpt=ckpt_path)
    model = R.create_models(args)
    model.eval()
    # a checkpoint of another spatial grid (8 x 8): the reference's load fails (recorded: the error type and the offending key)
    cfg_bad = V.ViTSTConfig(num_frames=12, t_patch_size=3, img_size=128, patch_size=16, in_chans=1, num_classes=8, embed_dim=128,
                            depth=2, num_heads=2, global_pool=True)
    torch.save({"model": V.init_from_shapes(V.vit_st_param_shapes(cfg_bad), seed=32)}, ckpt_path + ".bad")
    bad = argparse.Namespace(**{**vars(args), "ckpt": ckpt_path + ".bad"})
    try:
        R.create_models(bad)
        spatial_error = ""
    except RuntimeError as e:
        spatial_error = "RuntimeError" + (": pos_embed_spatial" if "pos_embed_spatial" in str(e) else "")
    os.remove(ckpt_path + ".bad")
    x = torch.rand(2, 1, 12, 64, 64, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        logits = model(x)
    loaded = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    save = {"ckpt_seed": 31, "x_seed": 5, "logits": logits.numpy(), "x_checksum": float(x.double().sum()),
            "args": str(vars(args)), "spatial_mismatch_error": spatial_error, "loaded/pos_embed_spatial": loaded["pos_embed_spatial"], "loaded/pos_embed_temporal": loaded["pos_embed_temporal"],
            "state_keys": np.array(sorted(loaded.keys()))}
    # parse_all_output: a confident case, a not-confident case, ties
    g = torch.Generator().manual_seed(6)
    cases = []
    for i in range(6):
        pos = torch.rand(8, generator=g) * (0.45 if i % 2 else 1.0)
        p = torch.stack([1 - pos, pos], dim=1).numpy().astype(np.float64 if i < 4 else np.float32)
        cases.append(p)
        save[f"parse/{i}/in"] = p
        save[f"parse/{i}/out"] = np.array(R.parse_all_output(p))
    save["disease_abbreviation"] = np.array([R.disease_abbreviation[i] for i in range(len(R.disease_abbreviation))])
    out = os.path.join(ROOT, "tests", "golden", "inference_small.npz")
    np.savez_compressed(out, **save)
    os.remove(ckpt_path)
    print("wrote", out, {k: (v.shape if hasattr(v, "shape") else v) for k, v in save.items() if not k.startswith("loaded")})
    for i in range(2):
        print(repr(str(save[f"parse/{i}/out"])))


if __name__ == "__main__":
    main()
