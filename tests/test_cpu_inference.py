"""CPU: the inference helpers (octcubem_amd/inference_utils.py, misc.interpolate_pos_embed) against tests/golden/inference_small.npz,
which oracle/gen_golden_inference.py wrote by running the reference's own /root/reference/inference_utils.py."""
import os
import types

import numpy as np
import torch


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "inference_small.npz"))


def test_parse_all_output_matches_the_reference_strings(golden_dir):
    from octcubem_amd import inference_utils as I
    z = _golden(golden_dir)
    assert [I.disease_abbreviation[i] for i in range(len(I.disease_abbreviation))] == [str(s) for s in z["disease_abbreviation"]]
    branches = set()
    for i in range(6):
        p = z[f"parse/{i}/in"]
        assert I.parse_all_output(p) == str(z[f"parse/{i}/out"]), i
        branches.add(bool(p[:, 1].max() > 0.5))
    assert branches == {True, False}                 # both forms of the "Normal" entry are among the cases


def test_misc_interpolate_pos_embed_is_the_util_misc_variant():
    """OCTCube/util/misc.py:1159-1222 resizes ``pos_embed`` / ``decoder_pos_embed`` and leaves ``pos_embed_spatial`` alone (the
    util/pos_embed.py function of the same name, pos_embed.interpolate_pos_embed here, resizes that one)."""
    from octcubem_amd import misc, pos_embed
    g = torch.Generator().manual_seed(0)
    model = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=16, frames=12, t_patch_size=3),
                                  pos_embed=torch.zeros(1, 17, 8), decoder_pos_embed=torch.zeros(1, 17, 8),
                                  pos_embed_spatial=torch.zeros(1, 16, 8))
    ck = {"pos_embed": torch.randn(1, 65, 8, generator=g), "decoder_pos_embed": torch.randn(1, 65, 8, generator=g),
          "pos_embed_spatial": torch.randn(1, 64, 8, generator=g), "pos_embed_temporal": torch.randn(1, 6, 8, generator=g)}
    ref = {k: v.clone() for k, v in ck.items()}
    misc.interpolate_pos_embed(model, ck)
    assert ck["pos_embed"].shape == (1, 17, 8) and ck["decoder_pos_embed"].shape == (1, 17, 8)
    assert torch.equal(ck["pos_embed"][:, :1], ref["pos_embed"][:, :1])                      # the class token's entry is kept
    want = torch.nn.functional.interpolate(ref["pos_embed"][:, 1:].reshape(1, 8, 8, 8).permute(0, 3, 1, 2), size=(4, 4), mode="bicubic",
                                           align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)
    assert torch.equal(ck["pos_embed"][:, 1:], want)
    assert torch.equal(ck["pos_embed_spatial"], ref["pos_embed_spatial"])                    # untouched by THIS variant
    ck2 = {k: v.clone() for k, v in ref.items() if k.startswith("pos_embed_")}
    model2 = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=4 * 16, frames=12, t_patch_size=3),
                                   pos_embed_spatial=torch.zeros(1, 16, 8))
    pos_embed.interpolate_pos_embed(model2, ck2)
    assert ck2["pos_embed_spatial"].shape == (1, 16, 8)
    misc.interpolate_temporal_pos_embed(model, ck)
    assert ck["pos_embed_temporal"].shape == (1, 4, 8)
