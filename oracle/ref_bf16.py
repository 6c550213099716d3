"""ORACLE — test infrastructure only.  Never imported by vit_amd/.

bf16-rounding emulation of the HIP forward: the same arithmetic as oracle/ref_model.py
(train_vit.py:141-171) computed in fp32 on the CPU, but with weights and activations rounded to
bf16 at exactly the points where the HIP path stores bf16 (DESIGN.md "HBM layout").  It separates
"the kernels implement the reference dataflow" (HIP vs this emulation: tight tolerance, summation
order only) from "bf16 storage costs this much accuracy" (this emulation vs the fp32 oracle).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


def _bn(sd, p):
    s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + 1e-5)
    return s, sd[p + ".bias"] - sd[p + ".running_mean"] * s


def _conv_bn(x, sd, conv, bn, stride, pad, relu, res=None):
    s, b = _bn(sd, bn)
    y = F.conv2d(x, _r(sd[conv + ".weight"]), None, stride, pad) * s.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    return _r(torch.relu(y) if relu else y)


@torch.no_grad()
def forward_bf16_emulated(sd: dict, x: torch.Tensor, cfg: dict, features_only: bool = False, taps: dict = None):
    E, H, L = cfg["embed_dim"], cfg["num_heads"], cfg["num_layers"]
    B = x.shape[0]
    x = _r(x.float())
    if cfg["backbone"] == "patch16":
        t = F.conv2d(x, _r(sd["patch_embed.weight"]), sd["patch_embed.bias"], 16)
    else:
        p = "cnn_backbone."
        y = _conv_bn(x, sd, p + "conv1", p + "bn1", 2, 3, True)
        if taps is not None:
            taps["c1"] = y
        y = F.max_pool2d(y, 3, 2, 1)
        if taps is not None:
            taps["pool"] = y
        depths = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}[cfg["backbone"]]
        for li, n in enumerate(depths, start=1):
            for bi in range(n):
                q = f"{p}layer{li}.{bi}."
                stride = 2 if (bi == 0 and li > 1) else 1
                idn = y
                if q + "downsample.0.weight" in sd:
                    idn = _conv_bn(y, sd, q + "downsample.0", q + "downsample.1", stride, 0, False)
                z = _conv_bn(y, sd, q + "conv1", q + "bn1", stride, 1, True)
                y = _conv_bn(z, sd, q + "conv2", q + "bn2", 1, 1, True, res=idn)
                if taps is not None:
                    taps[f"l{li}.{bi}"] = y
        t = F.conv2d(y, _r(sd["patch_embed.weight"]), sd["patch_embed.bias"])
    t = t.flatten(2).transpose(1, 2)
    xs = torch.cat([sd["cls_token"].expand(B, -1, -1), t], 1) + sd["pos_embed"]
    S = xs.shape[1]
    if taps is not None:
        taps["tokens"] = xs
    for i in range(L):
        q = f"transformer.layers.{i}."
        ln = _r(F.layer_norm(xs, (E,), sd[q + "norm1.weight"], sd[q + "norm1.bias"]))
        qkv = _r(ln @ _r(sd[q + "self_attn.in_proj_weight"]).T + sd[q + "self_attn.in_proj_bias"])
        qh, kh, vh = [u.reshape(B, S, H, E // H).transpose(1, 2) for u in qkv.split(E, -1)]
        sc = qh @ kh.transpose(-1, -2)
        pe = torch.exp((sc - sc.amax(-1, keepdim=True)) * (E // H) ** -0.5)   # 0.125 at head_dim 64
        att = _r((_r(pe) @ vh) / pe.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, S, E)
        xs = xs + att @ _r(sd[q + "self_attn.out_proj.weight"]).T + sd[q + "self_attn.out_proj.bias"]
        ln = _r(F.layer_norm(xs, (E,), sd[q + "norm2.weight"], sd[q + "norm2.bias"]))
        hid = _r(F.gelu(ln @ _r(sd[q + "linear1.weight"]).T + sd[q + "linear1.bias"]))
        xs = xs + hid @ _r(sd[q + "linear2.weight"]).T + sd[q + "linear2.bias"]
        if taps is not None:
            taps[f"enc{i}"] = xs
    feat = F.layer_norm(xs[:, 0], (E,), sd["norm.weight"], sd["norm.bias"])
    if features_only:
        return feat
    h0 = _r(F.gelu(_r(feat) @ _r(sd["head.0.weight"]).T + sd["head.0.bias"]))
    return h0 @ _r(sd["head.3.weight"]).T + sd["head.3.bias"]
