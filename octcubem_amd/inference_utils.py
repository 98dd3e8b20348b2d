"""Model-side inference helpers: the drop-in counterpart of the reference's ``inference_utils.py`` (repository root), for the part
that touches the hot path -- build the spatio-temporal ViT from the same ``args``, load a fine-tuned checkpoint into it with both
positional tables interpolated, and format the per-disease probabilities.  (The DICOM reading and the MONAI transform pipeline of
that file, ``create_3d_transforms``, are data loading: SURVEY section 2.1 OUT.)

  create_models(args)        inference_utils.py:43-60   -- ``args.model_type == '3D_st_flash_attn'``, ``args.model`` names a factory of
                             models_vit_st; the reference passes ``use_flash_attention=True``, a keyword its VisionTransformer does
                             not have and swallows in ``**kwargs`` (models_vit_st_flash_attn.py:51-76: the flag is ``use_flash_attn``),
                             so it builds and runs the NON-flash model; the same call here does the same
  load_model(args, model)    inference_utils.py:31-40   -- ``checkpoint['model']``, util/misc.py's interpolate_pos_embed (``pos_embed`` /
                             ``decoder_pos_embed`` only: a ``pos_embed_spatial`` of another grid fails the strict load, as there) and
                             temporal linear interpolation, then a strict load
  parse_all_output(p)        inference_utils.py:63-79   -- p: [8, 2] per-disease (negative, positive) probabilities -> one line of text
"""
from __future__ import annotations

import numpy as np
import torch

from . import models_vit_st
from .misc import interpolate_pos_embed, interpolate_temporal_pos_embed      # the util/misc.py variants, as the reference imports

# class index -> label of the released 9-way checkpoint (index 0 = no disease; the other eight are one binary head each)
disease_abbreviation = {0: "Normal", 1: "DME", 2: "AMD", 3: "POAG", 4: "EPM", 5: "DR", 6: "VD", 7: "RAO\\RVO", 8: "RNV"}


def process_dicom_array(dicom_array, val_transform):
    """A [T, H, W] array -> the transformed [1, T', H', W'] tensor and its shape (``val_transform`` is the caller's pipeline)."""
    t = val_transform({"pixel_values": torch.tensor(dicom_array).unsqueeze(0)})["pixel_values"]
    return t, t.shape


def load_model(args, model_without_ddp):
    if not getattr(args, "ckpt", None):
        print("No checkpoint for loading")
        return
    checkpoint = torch.load(args.ckpt, map_location="cpu")
    interpolate_pos_embed(model_without_ddp, checkpoint["model"])
    interpolate_temporal_pos_embed(model_without_ddp, checkpoint["model"])
    model_without_ddp.load_state_dict(checkpoint["model"])
    print("Load checkpoint %s" % args.ckpt)


def create_models(args):
    if args.model_type != "3D_st_flash_attn":
        raise ValueError(f"model_type {args.model_type!r}: only '3D_st_flash_attn' is built (the one inference_utils.py creates)")
    print("Use 3D spatio-temporal model w/ flash attention")
    model = getattr(models_vit_st, args.model)(
        num_frames=args.num_frames, t_patch_size=args.t_patch_size, img_size=args.input_size, num_classes=args.nb_classes,
        drop_path_rate=args.drop_path, global_pool=args.global_pool, sep_pos_embed=args.sep_pos_embed, cls_embed=args.cls_embed,
        use_flash_attention=True)
    model = model.cuda()
    load_model(args, model)
    return model


def parse_all_output(pred_output_cache):
    """pred_output_cache[i] = (P(not disease i+1), P(disease i+1)) for the eight binary heads.  "Normal" is one minus the largest
    disease probability when that exceeds 0.5, else the mean of the negatives."""
    p = np.asarray(pred_output_cache)
    top = int(np.argmax(p[:, 1]))
    top_prob = p[top, 1]
    normal = 1 - top_prob if top_prob > 0.5 else np.mean(p[:, 0])
    out = "Disease probability: (Disease Name: Probability) \n"
    out += f"{disease_abbreviation[0]}: {normal:.3f}" + (" " * 8 if top_prob > 0.5 else " " * 9)
    for i in range(1, len(disease_abbreviation)):
        out += f"{disease_abbreviation[i]}: {p[i - 1, 1]:.3f}" + " " * 7
    return out
