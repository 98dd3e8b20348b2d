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


class _AllGatherWithGrad(torch.autograd.Function):
    """cat(all_gather(x)) whose backward returns this rank's slice of the SUM over ranks of the incoming gradient
    (what ``torch.distributed.nn.all_gather`` computes, as one reduce-scatter instead of world_size all-reduces)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        world = dist.get_world_size(ctx.group)
        g = g.contiguous()
        out = torch.empty((g.shape[0] // world,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        if g.is_cuda:
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
    gi = [torch.zeros_like(image_features) for _ in range(world_size)]
    ge = [torch.zeros_like(enface_features) for _ in range(world_size)]
    dist.all_gather(gi, image_features.detach().contiguous(), group=group)
    dist.all_gather(ge, enface_features.detach().contiguous(), group=group)
    if not local_loss:          # keep the graph for the local rank's rows
        gi[rank] = image_features
        ge[rank] = enface_features
    return torch.cat(gi, dim=0), torch.cat(ge, dim=0)


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


def clamp_logit_scale(model):
    """After every optimizer step (train_retclip.py): logit_scale stays within [0, ln 100]."""
    with torch.no_grad():
        getattr(model, "module", model).logit_scale.clamp_(0, math.log(100))


def train_step(model, loss_fn, images, texts, optimizers, loss_scalers=None, clip_grad=None):
    """One accum_freq == 1 iteration of train_retclip.train_one_epoch: forward both towers, ClipLoss, backward, optional clip,
    optimizer step(s), clamp.  ``optimizers``: one per parameter set (each tower owns its own flat arena / FusedAdamW; the
    temperature uses a plain torch optimizer).  Returns the loss (detached)."""
    for o in optimizers:
        o.zero_grad()
    image_features, text_features, logit_scale = model(images, texts)
    loss = loss_fn(image_features, text_features, logit_scale)
    loss.backward()
    if clip_grad is not None:
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], clip_grad)
    for o in optimizers:
        o.step()
    clamp_logit_scale(model)
    return loss.detach()
