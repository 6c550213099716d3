"""ORACLE — test infrastructure only.  Never imported by vit_amd/.

CPU emulation of the MX-fp8 inference mode (BASELINE.json configs[4]): the reference forward
(train_vit.py:141-171; torch transformer.py:945-982, functional.py:5819-5833,6623-6637) with the
inputs and weights of the four encoder Linears of every layer passed through MXFP8 fake
quantisation, and the attention products computed from e4m3 operands (attention_fp8 below), (OCP Microscaling: e4m3 elements, one E8M0 scale per 32 consecutive elements of the
contraction axis, scale = 2^(floor(log2 amax) - 8), round to nearest even, saturate at +-448) and
everything else in fp32.  The reference itself has no fp8 path; this restates the format the HIP
kernels implement (include/rvit.h) so that "the kernels implement MXFP8" and "MXFP8 costs
accuracy" can be told apart.  With quantisation switched off it must reproduce
CNNViTHybridRef.forward (tests/test_oracle.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .ref_model import CNNViTHybridRef


def mx_quantize(x: torch.Tensor):
    """x f32 [rows, K] -> (e4m3 bytes uint8 [rows, K], E8M0 uint8 [rows, K/32])."""
    rows, K = x.shape
    xb = x.float().reshape(rows, K // 32, 32)
    amax = xb.abs().amax(dim=2)
    e = (amax.contiguous().view(torch.int32) >> 23) & 0xFF
    s = (e - 8).clamp(min=0)
    inv = torch.pow(2.0, (127 - s).double()).float()
    y = (xb * inv.unsqueeze(2)).clamp(-448.0, 448.0)
    q = y.to(torch.float8_e4m3fn).view(torch.uint8).reshape(rows, K)
    return q, s.to(torch.uint8)


def mx_dequantize(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    rows, K = q.shape
    v = q.view(torch.float8_e4m3fn).float().reshape(rows, K // 32, 32)
    return (v * torch.pow(2.0, s.double() - 127.0).float().unsqueeze(2)).reshape(rows, K)


def mx_fake_quant(x: torch.Tensor) -> torch.Tensor:
    """Quantise the last axis of x to MXFP8 and back."""
    shp = x.shape
    q, s = mx_quantize(x.reshape(-1, shp[-1]))
    return mx_dequantize(q, s).reshape(shp)


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


def _e8m0(amax: torch.Tensor) -> torch.Tensor:
    """E8M0 scale exponent (biased by 127) of a block with absolute maximum amax: floor(log2 amax) - 8, >= 0."""
    e = (amax.float().contiguous().view(torch.int32) >> 23) & 0xFF
    return (e - 8).clamp(min=0)


def _to_e4m3(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """x * 2^-(s - 127) rounded to e4m3 (saturating), returned as fp32 in the SCALED domain."""
    inv = torch.pow(2.0, (127 - s).double()).float()
    return (x * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def attention_fp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """The fp8 attention of the MX-fp8 mode (vit_amd/csrc/attn.hip attn_fwd_fp8_kernel; the reference has no fp8
    path, torch functional.py:6623-6631 is the fp32 function it approximates).  q, k, v fp32 [B,H,S,Dh]:
    q and k quantised to e4m3 with one power-of-two scale per (token, head) row, v with one per (image, head),
    probabilities as e4m3(256 p) with the row sum taken before rounding.  Returns fp32 [B,H,S,Dh]."""
    Dh = q.shape[-1]
    sq, sk = _e8m0(q.abs().amax(-1)), _e8m0(k.abs().amax(-1))          # [B,H,S]
    sv = _e8m0(v.abs().amax(dim=(-1, -2)))                              # [B,H]
    q8, k8 = _to_e4m3(q, sq.unsqueeze(-1)), _to_e4m3(k, sk.unsqueeze(-1))
    v8 = _to_e4m3(v, sv[..., None, None])
    fs = lambda e: torch.pow(2.0, e.double() - 127.0).float()
    scores = (q8 @ k8.transpose(-1, -2)) * (fs(sq).unsqueeze(-1) * fs(sk).unsqueeze(-2))
    m = scores.amax(-1, keepdim=True)
    p = torch.exp2((scores - m) * ((1.0 / math.sqrt(Dh)) * 1.44269504088896340736))
    ssum = p.sum(-1, keepdim=True)
    p8 = (p * 256.0).to(torch.float8_e4m3fn).float()
    return (p8 @ v8) * (fs(sv)[..., None, None] * (1.0 / 256.0) / ssum)


@torch.no_grad()
def encoder_layer_mxfp8(L, t: torch.Tensor, quantize: bool = True, bf16_points: bool = False,
                        fp8_attention: bool = False) -> torch.Tensor:
    """One pre-LN encoder layer on the fp32 token stream t [B,S,E] with MXFP8 Linears and (quantize and
    fp8_attention) the fp8 attention above.  bf16_points rounds the tensors the HIP path stores in bf16 before it
    quantises them (LayerNorm output, qkv, attention output)."""
    fq = mx_fake_quant if quantize else (lambda u: u)
    r = _bf16 if bf16_points else (lambda u: u)
    mha = L.self_attn
    B, S, E = t.shape
    H = mha.num_heads
    Dh = E // H
    qkv = r(F.linear(fq(r(L.norm1(t))), fq(mha.in_proj_weight), mha.in_proj_bias))
    q, k, v = (u.view(B, S, H, Dh).transpose(1, 2) for u in qkv.chunk(3, dim=-1))
    if quantize and fp8_attention:
        o = attention_fp8(q, k, v)
    else:
        o = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(Dh), dim=-1) @ v
    o = r(o.transpose(1, 2).reshape(B, S, E))
    t = t + F.linear(fq(o), fq(mha.out_proj.weight), mha.out_proj.bias)
    h = F.gelu(F.linear(fq(r(L.norm2(t))), fq(L.linear1.weight), L.linear1.bias))
    return t + F.linear(fq(h), fq(L.linear2.weight), L.linear2.bias)


@torch.no_grad()
def forward_mxfp8_emulated(ref: CNNViTHybridRef, x: torch.Tensor, quantize: bool = True,
                           features_only: bool = False, taps: list = None, fp8_attention: bool = False) -> torch.Tensor:
    """taps (optional list) receives the fp32 token stream before the first layer and after every layer.
    fp8_attention: the attention products from e4m3 operands too (the HIP path's fp8_attention option); default, as
    the HIP path's default since round 3: fp32 softmax attention whose output is MX-quantised for the out_proj."""
    fq = mx_fake_quant if quantize else (lambda t: t)
    B = x.shape[0]
    t = x
    if ref.cnn_backbone is not None:
        t = ref.cnn_backbone(t)[-1]
    t = ref.patch_embed(t).flatten(2).transpose(1, 2)
    t = torch.cat([ref.cls_token.expand(B, -1, -1), t], dim=1) + ref.pos_embed
    if taps is not None:
        taps.append(t.clone())
    for L in ref.transformer.layers:
        t = encoder_layer_mxfp8(L, t, quantize, fp8_attention=fp8_attention)
        if taps is not None:
            taps.append(t.clone())
    c = ref.norm(t[:, 0])
    return c if features_only else ref.head(c)
