"""Classification criteria the reference's fine-tune drivers pick from (OCTCube/main_finetune.py:305-312): timm's
``LabelSmoothingCrossEntropy`` / ``SoftTargetCrossEntropy`` (timm is not a dependency here) next to torch's own
``CrossEntropyLoss`` / ``BCEWithLogitsLoss``.  They act on ``[B, num_classes]`` logits -- host-side torch ops, not a hot kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from ._autocast import autocast_invariant


@autocast_invariant
class LabelSmoothingCrossEntropy(nn.Module):
    """NLL loss with label smoothing: mean_i[(1 - s) * nll_i + s * mean_c(-log p_ic)]."""

    def __init__(self, smoothing=0.1):
        super().__init__()
        assert smoothing < 1.0
        self.smoothing = smoothing
        self.confidence = 1.0 - smoothing

    def forward(self, x, target):
        logprobs = F.log_softmax(x.float(), dim=-1)
        nll_loss = -logprobs.gather(dim=-1, index=target.unsqueeze(1)).squeeze(1)
        smooth_loss = -logprobs.mean(dim=-1)
        return (self.confidence * nll_loss + self.smoothing * smooth_loss).mean()


@autocast_invariant
class SoftTargetCrossEntropy(nn.Module):
    """Cross entropy against a probability vector (what mixup / cutmix produce)."""

    def forward(self, x, target):
        return torch.sum(-target * F.log_softmax(x.float(), dim=-1), dim=-1).mean()
