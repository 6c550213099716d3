"""ORACLE — test infrastructure only.  Never imported by vit_amd/.

Sensitivity probe for the training-step parity test: the fp32 oracle (oracle/ref_model.py) run
with every stem activation (conv, BatchNorm and block outputs) and its gradient rounded to bf16 —
the storage precision of the HIP path — through a straight-through rounding op.  The distance of
ITS parameter gradients from the plain fp32 oracle's is what bf16 activation storage costs on a
given network/batch (ReLU masks of near-zero pre-activations flip); the HIP gradients are held to
a small multiple of that distance, tensor by tensor.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .ref_model import CNNViTHybridRef, _BasicBlock


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def attach_bf16_rounding(ref: CNNViTHybridRef) -> list:
    """Round the output (and incoming gradient) of every stem conv / BN / block to bf16."""
    hooks = []
    if ref.cnn_backbone is None:
        return hooks
    for mod in ref.cnn_backbone.modules():
        if isinstance(mod, (nn.Conv2d, nn.BatchNorm2d, _BasicBlock)):
            hooks.append(mod.register_forward_hook(lambda m, i, o: _RoundSTE.apply(o)))
    return hooks
