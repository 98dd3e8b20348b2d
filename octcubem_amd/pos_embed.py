"""Positional-embedding utilities with the reference's names and in-place-on-the-checkpoint-dict behaviour
(OCTCube/util/pos_embed.py): fixed 2-D sine-cosine tables (:16-63), bicubic resize of a learned spatial table to the model's
grid (``interpolate_pos_embed`` :71-107) and linear resize / centre crop of the temporal table
(``interpolate_temporal_pos_embed`` :111-141).  One-off host-side work at checkpoint-load time (ATen ``interpolate``)."""
import numpy as np
import torch


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    assert embed_dim % 2 == 0
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 2, dtype=np.float32) / (embed_dim / 2.0))
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed_from_grid(embed_dim, grid):
    assert embed_dim % 2 == 0
    return np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                           get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """[grid*grid (+1), embed_dim]; the meshgrid puts w first, each half is sin | cos."""
    g = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(g, g), axis=0).reshape([2, 1, grid_size, grid_size])
    emb = get_2d_sincos_pos_embed_from_grid(embed_dim, grid)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def _resize_square(pos_tokens, orig_size, new_size):
    c = pos_tokens.shape[-1]
    t = pos_tokens.reshape(-1, orig_size, orig_size, c).permute(0, 3, 1, 2)
    t = torch.nn.functional.interpolate(t, size=(new_size, new_size), mode="bicubic", align_corners=False)
    return t.permute(0, 2, 3, 1).flatten(1, 2)


def interpolate_pos_embed(model, checkpoint_model):
    """Resize ``pos_embed`` (extra tokens kept) or ``pos_embed_spatial`` in ``checkpoint_model`` to ``model``'s grid."""
    if "pos_embed" in checkpoint_model:
        name = "pos_embed"
        num_patches = model.patch_embed.num_patches
        num_extra = model.pos_embed.shape[-2] - num_patches
    elif "pos_embed_spatial" in checkpoint_model:
        name = "pos_embed_spatial"
        num_patches = model.patch_embed.num_patches // (model.patch_embed.frames // model.patch_embed.t_patch_size)
        num_extra = model.pos_embed_spatial.shape[-2] - num_patches
    else:
        return
    ck = checkpoint_model[name]
    orig_size = int((ck.shape[-2] - num_extra) ** 0.5)
    new_size = int(num_patches ** 0.5)
    if orig_size != new_size:
        print(f"Position interpolate {name}" + " from %dx%d to %dx%d" % (orig_size, orig_size, new_size, new_size))
        checkpoint_model[name] = torch.cat((ck[:, :num_extra], _resize_square(ck[:, num_extra:], orig_size, new_size)), dim=1)


def interpolate_temporal_pos_embed(model, checkpoint_model, smaller_interpolate_type="interp"):
    """Resize ``pos_embed_temporal`` to the model's temporal grid: linear interpolation, or a centre crop when shrinking with
    ``smaller_interpolate_type='crop'``."""
    if "pos_embed_temporal" not in checkpoint_model:
        return
    ck = checkpoint_model["pos_embed_temporal"]
    old_t = ck.shape[-2]
    new_t = model.patch_embed.frames // model.patch_embed.t_patch_size
    if old_t == new_t:
        return
    print("Position interpolate from %d to %d" % (old_t, new_t))
    t = ck.permute(0, 2, 1)
    if old_t > new_t and smaller_interpolate_type == "crop":
        start = (old_t - new_t) // 2
        t = t[:, :, start:start + new_t]
    else:
        t = torch.nn.functional.interpolate(t, size=new_t, mode="linear", align_corners=False)
    checkpoint_model["pos_embed_temporal"] = t.permute(0, 2, 1)
