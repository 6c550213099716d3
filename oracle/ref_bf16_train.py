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

from .ref_model import CNNViTHybridRef, MobileNetV3SmallFeatures, _BasicBlock, _MbBlockRef, _SqueezeExciteRef


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


def emulate_hip_training_storage(ref: CNNViTHybridRef) -> None:
    """The fp32 oracle turned into an emulation of WHERE the HIP training path (vit_amd/trainer.py) stores bf16:
      * every conv / Linear weight is the bf16-rounded master weight (the operand copies of WeightMirror /
        _TorchPacker); gradients are taken at those rounded weights, as the kernels do;
      * the input image is bf16 (rvit_input_pack_nhwc4);
      * stem maps written to HBM are bf16, and so are their gradients: conv outputs `y`, BatchNorm(+ReLU) outputs
        `z1` / `zd`, block outputs `zo`.  bn2's output is NOT rounded on its own: rvit_bn_apply adds the residual
        and applies the ReLU in fp32 and rounds once (the block-output hook).
    The encoder keeps fp32 activations here (its bf16 storage moves gradients by < 3 %, tests/test_train_gpu.py)."""
    with torch.no_grad():
        for mod in ref.modules():
            if isinstance(mod, (nn.Conv2d, nn.Linear)):
                mod.weight.copy_(mod.weight.to(torch.bfloat16).float())
            if isinstance(mod, nn.MultiheadAttention):
                mod.in_proj_weight.copy_(mod.in_proj_weight.to(torch.bfloat16).float())
    if ref.cnn_backbone is None:
        ref.patch_embed.register_forward_pre_hook(lambda m, i: (_RoundSTE.apply(i[0]),))
        return
    cb = ref.cnn_backbone
    cb.register_forward_pre_hook(lambda m, i: (_RoundSTE.apply(i[0]),))
    rnd = lambda m, i, o: _RoundSTE.apply(o)  # noqa: E731
    if isinstance(cb, MobileNetV3SmallFeatures):
        # vit_amd/mobile_trainer.py keeps in bf16: every convolution output (depthwise, pointwise, conv_stem, the two
        # squeeze-excite projections), every activated BatchNorm output, the gated map after a squeeze-excite and
        # every block output (last BatchNorm + residual, rounded once); the BatchNorm pre-activations are recomputed
        # in fp32 and never stored, so BatchNorm modules are not hooked by themselves.
        for mod in cb.modules():
            if isinstance(mod, (nn.Conv2d, nn.ReLU, nn.Hardswish, _SqueezeExciteRef, _MbBlockRef)):
                mod.register_forward_hook(rnd)
        cb.blocks[0].register_forward_pre_hook(lambda m, i: (_RoundSTE.apply(i[0]),))   # hardswish(bn1(conv_stem)) is functional
        return
    cb.conv1.register_forward_hook(rnd)
    cb.bn1.register_forward_hook(rnd)
    for mod in cb.modules():
        if isinstance(mod, _BasicBlock):
            mod.conv1.register_forward_hook(rnd)
            mod.bn1.register_forward_hook(rnd)
            mod.conv2.register_forward_hook(rnd)
            if mod.downsample is not None:
                mod.downsample[0].register_forward_hook(rnd)
                mod.downsample[1].register_forward_hook(rnd)
            mod.register_forward_hook(rnd)


class _ForceValueSTE(torch.autograd.Function):
    """Forward: the given value; backward: the gradient passes to the module's own output unchanged."""
    @staticmethod
    def forward(ctx, own, forced):
        return forced.clone()

    @staticmethod
    def backward(ctx, g):
        return g, None


def stem_convs_in_order(ref: CNNViTHybridRef) -> list:
    """The resnet stem's convolutions in the order vit_amd/trainer.py saves their outputs: conv1, then per block
    conv1, [downsample conv], conv2."""
    cb = ref.cnn_backbone
    out = [("conv1", cb.conv1)]
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(cb, f"layer{li}")):
            out.append((f"layer{li}.{bi}.conv1", blk.conv1))
            if blk.downsample is not None:
                out.append((f"layer{li}.{bi}.downsample.0", blk.downsample[0]))
            out.append((f"layer{li}.{bi}.conv2", blk.conv2))
    return out


def force_stem_conv_outputs(ref: CNNViTHybridRef, forced: dict) -> None:
    """Teacher forcing for the training-step parity test: every stem convolution of the oracle RETURNS the map the HIP
    path stored for it (`forced[name]`, NCHW fp32 holding bf16 values) while its gradient still flows through the
    oracle's own convolution.  With the conv outputs pinned, every BatchNorm statistic, ReLU decision and max-pool
    argmax of the oracle coincides with the HIP forward's, so the two backward passes can be compared tensor by tensor
    at the precision of the gradient arithmetic itself (a free-running comparison on a random-weight network is
    dominated by ReLU decisions that flip between two bf16-rounded forwards: tests/test_train_gpu.py).  Call BEFORE
    emulate_hip_training_storage so that its bf16 rounding of the incoming gradient applies behind the forced value."""
    for name, conv in stem_convs_in_order(ref):
        val = forced[name]
        conv.register_forward_hook(lambda m, i, o, v=val: _ForceValueSTE.apply(o, v))
