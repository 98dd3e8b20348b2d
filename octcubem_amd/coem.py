"""COEM (contrastive OCT / en-face pre-training) step pieces -- SURVEY section 8f N4, counterparts of
``retinal-COEM/src/open_clip``: ``loss.gather_features`` / ``ClipLoss`` (loss.py:21-65, :148-230), ``model.CustomTextCLIP``
(model.py:635-682: two towers, ``logit_scale = log(1 / 0.07)``, L2-normalised features) and the per-step clamp of
``logit_scale`` to [0, log 100] (training/train_retclip.py).

The towers are the 3-D ST ViT (``models_vit_st``, OCT volume) and the 2-D ViT (``models_vit``, IR image) of this package; their
``head`` is the projection to the shared embedding.  The loss acts on ``[B, embed]`` features: a ``[B_global, B_global]``
logits matmul and two cross-entropies -- a few MFLOP next to the towers' TFLOPs, left to ATen on the device.  With
``world_size > 1`` the features of all ranks are exchanged by ONE all-gather per tower (RCCL over xGMI; gloo in the CPU tests),
differentiable when ``gather_with_grad`` (its backward is a reduce-scatter of the feature gradients)."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F
from ._autocast import autocast_invariant


def _native_comm(x: torch.Tensor):
    """The process's RCCL communicator behind the C ABI (comm.NativeComm), for GPU tensors when one has been created
    (misc.init_distributed_mode / bench.py do); None -> torch.distributed (gloo in the CPU tests)."""
    if not x.is_cuda:
        return None
    from . import comm as _comm
    c = _comm.get_default()
    return c if c is not None and c.world > 1 else None


def _all_gather_cat(x: torch.Tensor, group=None) -> torch.Tensor:
    """cat over ranks of x (no gradient), one collective."""
    x = x.contiguous()
    nc = _native_comm(x)
    if nc is not None:
        out = torch.empty((nc.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        nc.all_gather_async(x, out)
        nc.wait()
        return out
    world = dist.get_world_size(group)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out


class _AllGatherWithGrad(torch.autograd.Function):
    """cat(all_gather(x)) whose backward returns this rank's slice of the SUM over ranks of the incoming gradient
    (what ``torch.distributed.nn.all_gather`` computes, as one reduce-scatter instead of world_size all-reduces)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _all_gather_cat(x, group)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        nc = _native_comm(g)
        world = nc.world if nc is not None else dist.get_world_size(ctx.group)
        out = torch.empty((g.shape[0] // world,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        if nc is not None:
            from . import comm as _comm
            nc.reduce_scatter_async(g, out, _comm.SUM)
            nc.wait()
        elif g.is_cuda:
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=ctx.group)
        else:                                   # gloo has no reduce_scatter: all-reduce and slice
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
            r = dist.get_rank(ctx.group)
            out = g[r * out.shape[0]:(r + 1) * out.shape[0]].clone()
        return out, None


def gather_features(image_features, enface_features, local_loss=False, gather_with_grad=False, rank=0, world_size=1,
                    use_horovod=False, group=None):
    if use_horovod:
        raise NotImplementedError("horovod is not supported (one process per GPU under torch.distributed)")
    if gather_with_grad:
        return _AllGatherWithGrad.apply(image_features, group), _AllGatherWithGrad.apply(enface_features, group)
    gi = list(_all_gather_cat(image_features.detach(), group).chunk(world_size, dim=0))
    ge = list(_all_gather_cat(enface_features.detach(), group).chunk(world_size, dim=0))
    if not local_loss:          # keep the graph for the local rank's rows
        gi[rank] = image_features
        ge[rank] = enface_features
    return torch.cat(gi, dim=0), torch.cat(ge, dim=0)


@autocast_invariant
class ClipLoss(nn.Module):
    def __init__(self, local_loss=False, gather_with_grad=False, cache_labels=False, rank=0, world_size=1, use_horovod=False,
                 correct_label=0):
        super().__init__()
        self.local_loss = local_loss
        self.gather_with_grad = gather_with_grad
        self.cache_labels = cache_labels
        self.rank = rank
        self.world_size = world_size
        self.use_horovod = use_horovod
        self.correct_label = correct_label
        self.prev_num_logits = 0
        self.labels = {}

    def get_corrected_label(self, enface_features_i, enface_features_j, t=10 ** (-100)):
        """Soft labels that share the target mass among samples with IDENTICAL en-face features (same report)."""
        d = torch.cdist(enface_features_i.detach().float(), enface_features_j.detach().float())
        L = (d <= t).to(enface_features_i.dtype)
        return L / torch.sum(L, dim=1, keepdim=True)

    def forward(self, image_features, enface_features, logit_scale):
        device = image_features.device
        if self.world_size > 1:
            all_image, all_enface = gather_features(image_features, enface_features, self.local_loss, self.gather_with_grad,
                                                    self.rank, self.world_size, self.use_horovod)
            if self.local_loss:
                logits_per_image = logit_scale * image_features @ all_enface.T
                logits_per_enface = logit_scale * enface_features @ all_image.T
            else:
                logits_per_image = logit_scale * all_image @ all_enface.T
                logits_per_enface = logits_per_image.T
        else:
            all_enface = enface_features
            logits_per_image = logit_scale * image_features @ enface_features.T
            logits_per_enface = logit_scale * enface_features @ image_features.T
        if self.correct_label:
            src = enface_features if (self.world_size > 1 and self.local_loss) or self.world_size == 1 else all_enface
            labels = self.get_corrected_label(src, all_enface)
        else:
            num_logits = logits_per_image.shape[0]
            if self.prev_num_logits != num_logits or device not in self.labels:
                labels = torch.arange(num_logits, device=device, dtype=torch.long)
                if self.world_size > 1 and self.local_loss:
                    labels = labels + num_logits * self.rank
                if self.cache_labels:
                    self.labels[device] = labels
                    self.prev_num_logits = num_logits
            else:
                labels = self.labels[device]
        return (F.cross_entropy(logits_per_image, labels) + F.cross_entropy(logits_per_enface, labels)) / 2


def gather_features_3mod(image_features, enface1_features, enface2_features, t_weight1, t_weight2, local_loss=False,
                         gather_with_grad=False, rank=0, world_size=1, use_horovod=False, group=None):
    """open_clip/loss.py:68-146: the five per-sample tensors of the 3-modality loss (OCT, two en-face modalities, and the
    two presence weights) gathered over ranks -- with gradient, or detached with the local rank's rows spliced back in."""
    if use_horovod:
        raise NotImplementedError("horovod is not supported (one process per GPU under torch.distributed)")
    xs = (image_features, enface1_features, enface2_features, t_weight1, t_weight2)
    if gather_with_grad:
        return tuple(_AllGatherWithGrad.apply(x, group) for x in xs)
    out = []
    for x in xs:
        parts = list(_all_gather_cat(x.detach(), group).chunk(world_size, dim=0))
        if not local_loss:
            parts[rank] = x
        out.append(torch.cat(parts, dim=0))
    return tuple(out)


@autocast_invariant
class ThreeModalityClipLoss(nn.Module):
    """open_clip/loss.py:230-385: symmetric InfoNCE over the three pairs (OCT, en-face 1), (OCT, en-face 2), (en-face 1,
    en-face 2) with one temperature per pair; a sample whose modality is missing carries weight 0 in that modality's terms
    (t_weight1 / t_weight2, per sample), each term is a weighted mean over the samples present, 0 when none is; the total is
    the mean of the six directed terms."""

    def __init__(self, local_loss=False, gather_with_grad=False, cache_labels=False, rank=0, world_size=1, use_horovod=False,
                 correct_label=0):
        super().__init__()
        self.local_loss = local_loss
        self.gather_with_grad = gather_with_grad
        self.cache_labels = cache_labels
        self.rank = rank
        self.world_size = world_size
        self.use_horovod = use_horovod
        self.correct_label = correct_label
        self.prev_num_logits = 0
        self.labels = {}

    @staticmethod
    def get_corrected_label(features_i, features_j, t=1e-10):
        d = torch.sqrt(((features_i.detach().unsqueeze(1) - features_j.detach().unsqueeze(0)) ** 2).sum(dim=-1))
        L = (d <= t).to(torch.float32)
        return L / torch.sum(L, dim=1, keepdim=True)

    @staticmethod
    def _weighted(loss_vec, w):
        tot = w.sum()
        return loss_vec.new_zeros(()) if float(tot) == 0.0 else (loss_vec * w).sum() / tot

    def forward(self, image_features, enface1_features, enface2_features, logit_scale, logit_scale1, logit_scale2, t_weight1,
                t_weight2):
        device = image_features.device
        if self.world_size > 1:
            all_i, all_e1, all_e2, all_w1, all_w2 = gather_features_3mod(
                image_features, enface1_features, enface2_features, t_weight1, t_weight2, self.local_loss, self.gather_with_grad,
                self.rank, self.world_size, self.use_horovod)
        else:
            all_i, all_e1, all_e2, all_w1, all_w2 = image_features, enface1_features, enface2_features, t_weight1, t_weight2
        if self.local_loss:
            l_i_e1 = logit_scale * image_features @ all_e1.T; l_e1_i = logit_scale * enface1_features @ all_i.T
            l_i_e2 = logit_scale1 * image_features @ all_e2.T; l_e2_i = logit_scale1 * enface2_features @ all_i.T
            l_e1_e2 = logit_scale2 * enface1_features @ all_e2.T; l_e2_e1 = logit_scale2 * enface2_features @ all_e1.T
            w1, w2 = t_weight1, t_weight2
        else:
            l_i_e1 = logit_scale * all_i @ all_e1.T; l_e1_i = l_i_e1.T
            l_i_e2 = logit_scale1 * all_i @ all_e2.T; l_e2_i = l_i_e2.T
            l_e1_e2 = logit_scale2 * all_e1 @ all_e2.T; l_e2_e1 = l_e1_e2.T
            w1, w2 = all_w1, all_w2
        if self.correct_label:
            labels = self.get_corrected_label(enface1_features, all_e1).to(device)
        else:
            num_logits = all_i.shape[0]           # as the reference: the GLOBAL count, also under local_loss
            if self.prev_num_logits != num_logits or device not in self.labels:
                labels = torch.arange(num_logits, device=device, dtype=torch.long)
                if self.world_size > 1 and self.local_loss:
                    labels = labels + num_logits * self.rank
                if self.cache_labels:
                    self.labels[device] = labels
                    self.prev_num_logits = num_logits
            else:
                labels = self.labels[device]
        ce = lambda lg: F.cross_entropy(lg, labels, reduction="none")
        w12 = w1 * w2
        terms = (self._weighted(ce(l_i_e1), w1), self._weighted(ce(l_e1_i), w1), self._weighted(ce(l_i_e2), w2),
                 self._weighted(ce(l_e2_i), w2), self._weighted(ce(l_e1_e2), w12), self._weighted(ce(l_e2_e1), w12))
        return sum(terms) / 6


@autocast_invariant
class CustomTextCLIP(nn.Module):
    """Two towers + a learned temperature.  ``visual`` / ``text`` are modules mapping their input to ``[B, embed_dim]`` (here:
    models_vit_st / models_vit with ``num_classes = embed_dim``); the reference builds them from config objects
    (model.py:125-578), this class takes them ready-made."""

    def __init__(self, visual: nn.Module, text: nn.Module):
        super().__init__()
        self.visual = visual
        self.text = text
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    def encode_image(self, image, normalize: bool = False):
        features = self.visual(image).float()
        return F.normalize(features, dim=-1) if normalize else features

    def encode_text(self, text, normalize: bool = False):
        features = self.text(text).float()
        return F.normalize(features, dim=-1) if normalize else features

    def forward(self, image, text, single_modality=None):
        if single_modality is not None:
            assert single_modality in ["image", "text"], f"single_modality should be either 'image' or 'text', got {single_modality}"
            if single_modality == "image":
                return self.encode_image(image, normalize=True), None, self.logit_scale.exp()
            return None, self.encode_text(text, normalize=True), self.logit_scale.exp()
        return self.encode_image(image, normalize=True), self.encode_text(text, normalize=True), self.logit_scale.exp()


def build_towers_from_config(cfg: dict, flash_semantics: bool = True):
    """Config-driven construction of the two towers, for the model configs the reference ships for this path
    (retinal-COEM/src/open_clip/model_configs/vit_large_patch16_retFound-vit_large_patch16_OCTCube.json through
    open_clip/model.py ``_build_vision_tower`` :190-290 and ``_build_text_tower`` :462-509):

      vision_cfg.model_name  "ViT_ST" | "ViT_ST_nodrop"   -> the 3-D spatio-temporal ViT (models_vit_st), ``out_dim = embed_dim`` head,
                             sep_pos_embed / cls_embed on, dropout before the head only for "ViT_ST"
      text_cfg.vit_model_name "ViT_flash_attn"             -> the 2-D ViT (models_vit) on the en-face image

    ``use_flash_attn: true`` in the config selects, in the reference, flash-attn blocks whose final residual is dropped
    (SURVEY section 0 fact 3); ``flash_semantics`` reproduces that through ``flash_compat`` (native key layout).  The rest of
    open_clip's tower zoo (timm / HIPT / LongNet / Perceiver / HF text) is out of scope.  Checkpoints named by ``model_ckpt``
    are loaded with checkpoint.load_pretrained when the file exists (the reference raises when it does not; here the tower
    stays randomly initialised and says so)."""
    import os
    from functools import partial
    from . import checkpoint as _ck, models_vit, models_vit_st
    embed_dim = int(cfg["embed_dim"])
    v, t = dict(cfg["vision_cfg"]), dict(cfg["text_cfg"])
    name = v.get("model_name") or ""
    if name not in ("ViT_ST", "ViT_ST_nodrop"):
        raise NotImplementedError(f"vision tower {name!r}: only the ViT_ST / ViT_ST_nodrop towers of the OCTCube configs are built")
    nf = int(v.get("num_frames", -1))
    visual = models_vit_st.VisionTransformer(
        num_frames=nf if nf > 0 else 60, t_patch_size=int(v.get("t_patch_size", 3)), img_size=int(v["image_size"]),
        patch_size=int(v["patch_size"]), in_chans=int(v.get("in_chans", 1)), num_classes=embed_dim, embed_dim=int(v["width"]),
        depth=int(v["layers"]), num_heads=int(v["num_heads"]), mlp_ratio=float(v.get("mlp_ratio", 4)),
        norm_layer=partial(nn.LayerNorm, eps=float(v.get("norm_layer_eps", 1e-6))), drop_path_rate=float(v.get("drop_path_rate", 0.0)),
        dropout=float(v.get("dropout", 0.0)) if name == "ViT_ST" else 0.0, sep_pos_embed=True, cls_embed=True,
        global_pool=bool(v.get("global_pool", True)) if name == "ViT_ST_nodrop" else True,
        flash_compat=bool(v.get("use_flash_attn", False)) and flash_semantics)
    tname = t.get("vit_model_name") or ""
    if "ViT_flash_attn" not in tname or "mod" in tname:
        raise NotImplementedError(f"en-face tower {tname!r}: only ViT_flash_attn is built")
    text = models_vit.VisionTransformer(
        img_size=int(t["image_size"]), patch_size=int(t["patch_size"]), in_chans=int(t.get("in_chans", 3)), num_classes=embed_dim,
        embed_dim=int(t["width"]), depth=int(t["layers"]), num_heads=int(t["num_heads"]), mlp_ratio=float(t.get("mlp_ratio", 4)),
        qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=float(t.get("norm_layer_eps", 1e-6))),
        drop_path_rate=float(t.get("drop_path_rate", 0.0)), global_pool=bool(t.get("global_pool", True)))
    for tower, c in ((visual, v), (text, t)):
        path = c.get("model_ckpt")
        if path and os.path.exists(path):
            ck = torch.load(path, map_location="cpu")
            _ck.load_pretrained(tower, ck, filter_keys=())
        elif path:
            print(f"Warnings! No such checkpoint {path}: randomizing the model!")
    return visual, text


def create_model_from_config(cfg: dict, **kw) -> "CustomTextCLIP":
    """CustomTextCLIP(embed_dim, vision_cfg, text_cfg) of open_clip/model.py:635-646 for the configs above."""
    visual, text = build_towers_from_config(cfg, **kw)
    return CustomTextCLIP(visual, text)


def make_reducers(model, comm=None, **kw):
    """One FlatGradReducer per tower (each tower is a model with its own flat gradient arena)."""
    from .parallel import FlatGradReducer
    m = getattr(model, "module", model)
    return [FlatGradReducer(m.visual, comm=comm, **kw), FlatGradReducer(m.text, comm=comm, **kw)]


def clamp_logit_scale(model):
    """After every optimizer step (train_retclip.py): logit_scale stays within [0, ln 100]."""
    with torch.no_grad():
        getattr(model, "module", model).logit_scale.clamp_(0, math.log(100))


def train_step(model, loss_fn, images, texts, optimizers, loss_scalers=None, clip_grad=None, reducers=None):
    """One accum_freq == 1 iteration of train_retclip.train_one_epoch: forward both towers, ClipLoss, backward, optional clip,
    optimizer step(s), clamp.  ``optimizers``: one per parameter set (each tower owns its own flat arena / FusedAdamW; the
    temperature uses a plain torch optimizer).  Returns the loss (detached).

    Data parallel (the reference wraps the whole model in DistributedDataParallel, training/main_retclip.py:206): pass
    ``reducers`` = one parallel.FlatGradReducer per tower (``make_reducers``); they exchange the towers' weight gradients
    during backward, and the temperature's scalar gradient is averaged here -- without them every rank would step on its local
    gradient only and the replicas would drift apart."""
    for o in optimizers:
        o.zero_grad()
    image_features, text_features, logit_scale = model(images, texts)
    loss = loss_fn(image_features, text_features, logit_scale)
    if reducers:
        for r in reducers:
            r.begin_backward(sync=True)
    loss.backward()
    if reducers:
        for r in reducers:
            r.finish()
        g = getattr(model, "module", model).logit_scale.grad
        if g is not None and reducers[0].world > 1:
            nc = _native_comm(g)
            if nc is not None:
                from . import comm as _comm
                nc.all_reduce_async(g, _comm.AVG)
                nc.wait()
            else:
                dist.all_reduce(g)
                g.div_(reducers[0].world)
    if clip_grad is not None:
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], clip_grad)
    for o in optimizers:
        o.step()
    clamp_logit_scale(model)
    return loss.detach()
