"""Cross-entropy with label smoothing on librvit (nn.CrossEntropyLoss(label_smoothing=...) of the
reference, train_vit.py:894,937): value and gradient in one HIP kernel."""
from __future__ import annotations

import torch

from . import ops


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, label_smoothing: float):
        if not logits.is_cuda:
            raise RuntimeError("vit_amd.loss.cross_entropy runs on the GPU only (no CPU fallback)")
        lg = logits.detach().float().contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=lg.device)
        dl = torch.empty_like(lg)
        with torch.cuda.device(lg.device):
            ops.ce_loss(lg, target.contiguous(), loss, dl, None, label_smoothing, grad_scale=1.0 / lg.shape[0])
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, label_smoothing: float = 0.0) -> torch.Tensor:
    """Mean cross-entropy of fp32 logits [B,C] against int64 targets [B]."""
    return _CrossEntropy.apply(logits, target, float(label_smoothing))


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for nn.CrossEntropyLoss(label_smoothing=...) as used at train_vit.py:894."""

    def __init__(self, label_smoothing: float = 0.0):
        super().__init__()
        self.label_smoothing = float(label_smoothing)

    def forward(self, logits, target):
        return cross_entropy(logits, target, self.label_smoothing)
