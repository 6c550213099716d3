"""`CNNViTHybrid` — drop-in for the reference classifier (train_vit.py:54-185, infer.py:69-180,
deploy/infer_edge.py:77-159) whose forward/backward run on librvit HIP kernels.

The module only HOLDS parameters (same names, shapes and dtypes as the reference state_dict, so a
`cnn_vit_radar.pt` checkpoint loads unchanged); it contains no torch.nn arithmetic.
`forward` hands the parameters to `vit_amd.engine.Engine`, which enqueues HIP kernels.  Calling the
model on a CPU tensor raises: there is no fallback path.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import ops
from .engine import Engine


class _Conv(nn.Module):
    """Parameter holder named like nn.Conv2d (weight [Cout,Cin,k,k], optional bias)."""

    def __init__(self, cin: int, cout: int, k: int, stride: int, pad: int, bias: bool = False):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        if bias:
            bound = 1.0 / math.sqrt(cin * k * k)
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)


class _BatchNorm(nn.Module):
    """Parameter/buffer holder named like nn.BatchNorm2d."""

    def __init__(self, c: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Linear(nn.Module):
    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(fin)
        nn.init.uniform_(self.bias, -bound, bound)


class _LayerNorm(nn.Module):
    def __init__(self, e: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(e))
        self.bias = nn.Parameter(torch.zeros(e))


class _BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = _Conv(cin, cout, 3, stride, 1)
        self.bn1 = _BatchNorm(cout)
        self.conv2 = _Conv(cout, cout, 3, 1, 1)
        self.bn2 = _BatchNorm(cout)
        if stride != 1 or cin != cout:
            self.downsample = nn.ModuleDict({"0": _Conv(cin, cout, 1, stride, 0), "1": _BatchNorm(cout)})
        else:
            self.downsample = None


class _ResNetStem(nn.Module):
    """ResNet-18/34 trunk parameters under timm `features_only` key names (train_vit.py:87)."""
    DEPTHS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}

    def __init__(self, name: str):
        super().__init__()
        self.conv1 = _Conv(3, 64, 7, 2, 3)
        self.bn1 = _BatchNorm(64)
        cin = 64
        for li, (c, n) in enumerate(zip((64, 128, 256, 512), self.DEPTHS[name]), start=1):
            blocks = {}
            for bi in range(n):
                blocks[str(bi)] = _BasicBlock(cin, c, 2 if (bi == 0 and li > 1) else 1)
                cin = c
            setattr(self, f"layer{li}", nn.ModuleDict(blocks))

    def blocks(self):
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}").values():
                yield blk


class _SqueezeExcite(nn.Module):
    """timm SqueezeExcite parameter names: 1x1 convs WITH bias."""

    def __init__(self, c: int, rd: int):
        super().__init__()
        self.conv_reduce = _Conv(c, rd, 1, 1, 0, bias=True)
        self.conv_expand = _Conv(rd, c, 1, 1, 0, bias=True)


class _DwConv(nn.Module):
    """Depthwise conv holder named like nn.Conv2d(groups=C): weight [C,1,k,k]."""

    def __init__(self, c: int, k: int, stride: int):
        super().__init__()
        self.c, self.k, self.stride, self.pad = c, k, stride, k // 2
        self.weight = nn.Parameter(torch.empty(c, 1, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")


class _MbBlock(nn.Module):
    """One block of timm mobilenetv3_small_100 under timm's key names.
    kind 'ds' (DepthwiseSeparableConv): conv_dw, bn1, [se], conv_pw, bn2
    kind 'ir' (InvertedResidual):       conv_pw, bn1, conv_dw, bn2, [se], conv_pwl, bn3
    kind 'cn' (ConvBnAct):              conv, bn1"""

    def __init__(self, kind, cin, k, stride, mid, cout, se_rd, act):
        super().__init__()
        self.kind, self.cin, self.k, self.stride, self.mid, self.cout, self.act = kind, cin, k, stride, mid, cout, act
        self.has_residual = kind != "cn" and stride == 1 and cin == cout
        if kind == "ds":
            self.conv_dw, self.bn1 = _DwConv(cin, k, stride), _BatchNorm(cin)
            if se_rd:
                self.se = _SqueezeExcite(cin, se_rd)
            self.conv_pw, self.bn2 = _Conv(cin, cout, 1, 1, 0), _BatchNorm(cout)
        elif kind == "ir":
            self.conv_pw, self.bn1 = _Conv(cin, mid, 1, 1, 0), _BatchNorm(mid)
            self.conv_dw, self.bn2 = _DwConv(mid, k, stride), _BatchNorm(mid)
            if se_rd:
                self.se = _SqueezeExcite(mid, se_rd)
            self.conv_pwl, self.bn3 = _Conv(mid, cout, 1, 1, 0), _BatchNorm(cout)
        else:
            self.conv, self.bn1 = _Conv(cin, cout, 1, 1, 0), _BatchNorm(cout)


class _MobileNetV3SmallStem(nn.Module):
    """timm `mobilenetv3_small_100` with `features_only=True` (train_vit.py:69,87): conv_stem / bn1 and six stages
    `blocks.<stage>.<index>`; the last feature map, [B,576,7,7], is what the classifier consumes (train_vit.py:145-146).
    Architecture (Howard et al. 2019, table 2; timm's arch_def ds_r1_k3_s2_e1_c16_se0.25_nre / ir_r1_k3_s2_e4.5_c24_nre,
    ir_r1_k3_s1_e3.67_c24_nre / ir_r1_k5_s2_e4_c40_se0.25, ir_r2_k5_s1_e6_c40_se0.25 / ir_r2_k5_s1_e3_c48_se0.25 /
    ir_r3_k5_s2_e6_c96_se0.25 / cn_r1_k1_s1_c576; squeeze-excite widths from the expanded width x 0.25 rounded to 8,
    ReLU inside, hard-sigmoid gate; hard-swish from stage 2 on).  timm is not installed here: parity of this stem is
    unpinned in the same sense as the resnet stems' (DESIGN.md section 3)."""
    # (kind, k, stride, mid, cout, se_rd, act)
    STAGES = (
        (("ds", 3, 2, 16, 16, 8, "relu"),),
        (("ir", 3, 2, 72, 24, 0, "relu"), ("ir", 3, 1, 88, 24, 0, "relu")),
        (("ir", 5, 2, 96, 40, 24, "hswish"), ("ir", 5, 1, 240, 40, 64, "hswish"), ("ir", 5, 1, 240, 40, 64, "hswish")),
        (("ir", 5, 1, 120, 48, 32, "hswish"), ("ir", 5, 1, 144, 48, 40, "hswish")),
        (("ir", 5, 2, 288, 96, 72, "hswish"), ("ir", 5, 1, 576, 96, 144, "hswish"), ("ir", 5, 1, 576, 96, 144, "hswish")),
        (("cn", 1, 1, 0, 576, 0, "hswish"),),
    )

    def __init__(self):
        super().__init__()
        self.conv_stem = _Conv(3, 16, 3, 2, 1)
        self.bn1 = _BatchNorm(16)
        cin = 16
        stages = []
        for st in self.STAGES:
            blks = []
            for kind, k, stride, mid, cout, se_rd, act in st:
                blks.append(_MbBlock(kind, cin, k, stride, mid, cout, se_rd, act))
                cin = cout
            stages.append(nn.ModuleList(blks))
        self.blocks = nn.ModuleList(stages)

    def all_blocks(self):
        for st in self.blocks:
            for blk in st:
                yield blk


class _SelfAttn(nn.Module):
    """nn.MultiheadAttention parameter names and default init (torch activation.py _reset_parameters)."""

    def __init__(self, e: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * e, e))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * e))
        self.out_proj = _Linear(e, e)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _EncoderLayer(nn.Module):
    def __init__(self, e: int):
        super().__init__()
        self.self_attn = _SelfAttn(e)
        self.linear1 = _Linear(e, 4 * e)
        self.linear2 = _Linear(4 * e, e)
        self.norm1 = _LayerNorm(e)
        self.norm2 = _LayerNorm(e)


class _Encoder(nn.Module):
    def __init__(self, e: int, n: int):
        super().__init__()
        first = _EncoderLayer(e)
        layers = [first]
        for _ in range(n - 1):  # nn.TransformerEncoder deep-copies ONE layer (train_vit.py:103-116)
            clone = _EncoderLayer(e)
            clone.load_state_dict(first.state_dict())
            layers.append(clone)
        self.layers = nn.ModuleList(layers)


class CNNViTHybrid(nn.Module):
    """Hybrid CNN-ViT radar classifier; constructor signature of train_vit.py:76-77 (the
    infer.py / infer_edge.py variants omit `pretrained_backbone`, which is accepted and ignored
    here: there is no network to fetch ImageNet weights from — load a checkpoint instead).

    `backbone='patch16'` is this repository's definition of the pure-ViT `vit_radar.pt` layout
    (SURVEY.md §8c): 16x16/16 patch conv, 196+1 tokens, no cnn_backbone.* keys.
    """

    # (model_name, output_channels, feature_map_size) — train_vit.py:68-74
    BACKBONES = {
        "mobilenetv3": ("mobilenetv3_small_100", 576, 7),
        "resnet18": ("resnet18", 512, 7),
        "resnet34": ("resnet34", 512, 7),
        "efficientnet_b0": ("efficientnet_b0", 1280, 7),
        "efficientnet_b2": ("efficientnet_b2", 1408, 7),
        "patch16": ("patch16", 3, 14),
    }
    _HIP_STEMS = ("resnet18", "resnet34", "patch16", "mobilenetv3")
    _TRAIN_STEMS = ("resnet18", "resnet34", "patch16", "mobilenetv3")

    def __init__(self, num_classes, embed_dim=384, num_heads=6, num_layers=6, dropout=0.1,
                 pretrained_backbone=False, backbone="resnet34"):
        super().__init__()
        self.backbone_name = backbone
        if backbone not in self.BACKBONES:
            raise ValueError(f"Unknown backbone: {backbone}. Choose from {list(self.BACKBONES.keys())}")
        if backbone not in self._HIP_STEMS:
            # efficientnet_b0/b2: besides being unbuilt, the reference's own table is inconsistent for them: timm's
            # features_only efficientnet returns 320 / 352 channels at stride 32, not the 1280 / 1408 of its classifier
            # head that train_vit.py:72-73 feeds to patch_embed
            raise NotImplementedError(f"backbone {backbone!r}: no HIP stem yet (out of scope, SURVEY.md §8f #4)")
        if embed_dim % num_heads != 0:
            # nn.MultiheadAttention's own assertion (torch activation.py: "embed_dim must be divisible by num_heads")
            raise AssertionError("embed_dim must be divisible by num_heads")
        if embed_dim // num_heads not in ops.ATTN_HEAD_DIMS:
            raise NotImplementedError(f"HIP attention kernels cover head_dim {ops.ATTN_HEAD_DIMS} "
                                      f"(embed_dim={embed_dim}, num_heads={num_heads} gives {embed_dim // num_heads})")
        _, cnn_out, feat = self.BACKBONES[backbone]
        self.num_classes, self.embed_dim, self.num_heads, self.num_layers = num_classes, embed_dim, num_heads, num_layers
        self.dropout = float(dropout)
        self.num_patches = feat * feat
        if backbone == "patch16":
            self.cnn_backbone = None
            self.patch_embed = _Conv(3, embed_dim, 16, 16, 0, bias=True)
        else:
            self.cnn_backbone = _MobileNetV3SmallStem() if backbone == "mobilenetv3" else _ResNetStem(backbone)
            self.patch_embed = _Conv(cnn_out, embed_dim, 1, 1, 0, bias=True)
        # nn.Conv2d default init for patch_embed (reference keeps torch defaults, train_vit.py:93)
        nn.init.kaiming_uniform_(self.patch_embed.weight, a=math.sqrt(5))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim))
        self.transformer = _Encoder(embed_dim, num_layers)
        self.norm = _LayerNorm(embed_dim)
        self.head = nn.ModuleDict({"0": _Linear(embed_dim, embed_dim // 2), "3": _Linear(embed_dim // 2, num_classes)})
        self._init_weights()
        self._engine = None
        self._train_engine = None

    def _init_weights(self):  # train_vit.py:132-139
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.head.values():
            nn.init.trunc_normal_(m.weight, std=0.02)
            nn.init.zeros_(m.bias)

    # ---- engine plumbing ---------------------------------------------------------------------
    @property
    def engine(self) -> Engine:
        if self._engine is None:
            object.__setattr__(self, "_engine", Engine(self))
        return self._engine

    def _check_input(self, x: torch.Tensor) -> torch.Tensor:
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != 224 or x.shape[3] != 224:
            raise ValueError(f"expected a [B,3,224,224] tensor, got {getattr(x, 'shape', type(x))}")
        if not x.is_cuda:
            raise RuntimeError("vit_amd.CNNViTHybrid runs on the GPU through librvit only; move the model and the "
                               "input to a cuda device. There is no CPU fallback.")
        return x

    @property
    def train_engine(self):
        if self._train_engine is None:
            from .trainer import TrainEngine
            object.__setattr__(self, "_train_engine", TrainEngine(self))
        return self._train_engine

    # ---- cached parameter / buffer lists ----------------------------------------------------------------
    # Module.parameters() walks the module tree (~100 us for this model); the engines need the lists on every
    # forward (version keys, autograd inputs).  The walk is done once; every later call only re-checks that each
    # slot of each module still holds the same tensor object (a replaced Parameter rebuilds the lists).  Modules
    # added after the first forward are not noticed: call `_forget_tensor_lists()` after such surgery.
    def _tensor_slots(self):
        cache = self.__dict__.get("_slots_cache")
        if cache is not None:
            for d, n, t in cache[0]:
                if d.get(n) is not t:
                    cache = None
                    break
        if cache is None:
            slots, seen = [], set()
            mods = list(self.modules())
            for mod in mods:
                for n, t in mod._parameters.items():
                    if t is not None and id(t) not in seen:
                        seen.add(id(t))
                        slots.append((mod._parameters, n, t))
            nparam = len(slots)
            for mod in mods:
                for n, t in mod._buffers.items():
                    if t is not None and id(t) not in seen:
                        seen.add(id(t))
                        slots.append((mod._buffers, n, t))
            cache = (slots, [t for _, _, t in slots[:nparam]], [t for _, _, t in slots])
            object.__setattr__(self, "_slots_cache", cache)
        return cache

    def _live_params(self):
        """== list(self.parameters()), cached."""
        return self._tensor_slots()[1]

    def _live_tensors(self):
        """== list(self.parameters()) + list(self.buffers()), cached."""
        return self._tensor_slots()[2]

    def _forget_tensor_lists(self) -> None:
        self.__dict__.pop("_slots_cache", None)

    def invalidate_packs(self) -> None:
        """Drop the inference engine's packed weight copies.  Called automatically on every train-mode forward, on
        train()/eval() switches and (through parameter version counters) after load_state_dict; call it by hand
        after editing parameters in eval mode through an API that bypasses autograd's version counters."""
        if self._engine is not None:
            self._engine._pack = None

    def train(self, mode: bool = True):
        self.invalidate_packs()
        return super().train(mode)

    def set_inference_precision(self, precision: str, fp8_attention=None) -> None:
        """'bf16' (default) or 'mxfp8': encoder Linears on the block-scaled fp8 MFMA (eval mode only).
        fp8_attention (mxfp8 only; default False, env RVIT_FP8_ATTENTION=1): both attention products on the fp8 MFMA
        as well, instead of the bf16-MFMA attention with an MX-fp8 output epilogue."""
        self.engine.set_precision(precision, fp8_attention)

    def set_graph_capture(self, max_batch: int = 16) -> None:
        """Replay eval-mode forwards of up to `max_batch` frames from a captured HIP graph (0 disables)."""
        self.engine.set_graph_capture(max_batch)

    def set_data_parallel(self, process_group=None, bucket_mb: float = 8.0) -> None:
        """Average gradients over `process_group` (RCCL) in buckets overlapped with backward."""
        self.train_engine.set_data_parallel(process_group, bucket_mb)

    def _run(self, x, features_only: bool):
        x = self._check_input(x)
        if x.shape[0] == 0:
            # an empty batch: torch modules return an empty result in eval() and BatchNorm refuses it in train()
            if self.training and self.cnn_backbone is not None:
                raise ValueError("Expected more than 1 value per channel when training, got an empty batch")
            return x.new_zeros((0, self.embed_dim if features_only else self.num_classes), dtype=torch.float32)
        if self.pos_embed.device != x.device:
            raise RuntimeError(f"input on {x.device} but the model's parameters on {self.pos_embed.device}")
        # librvit launches on the CURRENT device's stream: make the tensors' device current for the whole forward
        # (a model on cuda:1 called while cuda:0 is current would otherwise launch on device 0 against device-1
        # pointers); a no-op when it already is
        with torch.cuda.device(x.device):
            return self._run_on_device(x, features_only)

    def _run_on_device(self, x, features_only: bool):
        if self.training and self.backbone_name not in self._TRAIN_STEMS:
            raise NotImplementedError(f"backbone {self.backbone_name!r} is built for inference: call .eval() "
                                      "(training kernels exist for the resnet stems and patch16)")
        if self.training:
            # train(): BatchNorm batch statistics; backward on librvit when gradients are recorded.
            # A training forward means the weights are about to change: the inference engine re-packs next time
            # (parameter version counters alone miss e.g. torch.optim.AdamW(fused=True) steps).
            self.invalidate_packs()
            from .trainer import HybridFunction
            if torch.is_grad_enabled():
                return HybridFunction.apply(x, self.train_engine, features_only, *self._live_params())
            return self.train_engine.forward(x, features_only)
        return self.engine.run(x, features_only)

    def forward(self, x):
        """[B,3,224,224] float (ImageNet-normalised, NCHW) -> fp32 logits [B,num_classes]."""
        return self._run(x, False)

    def forward_features(self, x):
        """-> fp32 cls-token features [B,embed_dim] (train_vit.py:173-185)."""
        return self._run(x, True)

    def config(self) -> Dict:
        """`model_config` entry of the checkpoint dict (train_vit.py:999-1007)."""
        return {"num_classes": self.num_classes, "backbone": self.backbone_name, "embed_dim": self.embed_dim,
                "num_heads": self.num_heads, "num_layers": self.num_layers, "dropout": self.dropout, "mode": "single"}
