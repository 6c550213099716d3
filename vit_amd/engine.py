"""Kernel orchestration for `CNNViTHybrid`: packs parameters into the layouts the HIP kernels read
and enqueues the forward pass (train_vit.py:141-185) on torch's current stream.

HBM layout (all activations of one forward live in buffers cached per batch size):
  image      bf16 NHWC4, zero halo          [B, 230, 232, 4]   (patch16: [B,224,224,4], no halo)
  stem maps  bf16 NHWC                      [B, H, W, C]
  tokens x   fp32 (the residual stream)     [B, S, E]
  ln / attn  bf16                           [B*S, E]
  qkv        bf16  q | k | v                [B*S, 3E]
  mlp hidden bf16                           [B*S, 4E]
Weights: bf16 [out, in] (conv: [Cout, kh, kw, Cin]); eval BatchNorm folded to fp32 scale/bias
applied in the conv epilogue; biases / LN affine / pos_embed stay fp32.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import ops

BF16 = torch.bfloat16


def _bn_fold(bn):
    scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
    bias = bn.bias.float() - bn.running_mean.float() * scale
    return scale.contiguous(), bias.contiguous()


def _pack_conv(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,kh,kw] fp32 -> bf16 [Cout, kh*kw*Cin] (tap-major, channel-contiguous)."""
    return w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(BF16).contiguous()


def _pack_conv_px4(w: torch.Tensor, kw_pad: int) -> torch.Tensor:
    """[Cout,3,kh,kw] -> bf16 [Cout, kh * kw_pad * 4]: per kernel row a run of kw_pad pixels x 4
    channels (zero taps / zero 4th channel), matching the NHWC4 image rows."""
    cout, cin, kh, kw = w.shape
    p = torch.zeros(cout, kh, kw_pad, 4, dtype=torch.float32, device=w.device)
    p[:, :, :kw, :cin] = w.detach().permute(0, 2, 3, 1)
    return p.reshape(cout, -1).to(BF16).contiguous()


class Engine:
    """Owns packed weights and per-batch-size activation buffers of one model instance."""

    def __init__(self, model):
        self.m = model
        self._pack = None
        self._pack_key = None
        self._bufs: Dict[int, Dict[str, torch.Tensor]] = {}
        # "bf16" or "mxfp8": the four encoder Linears of every layer on the block-scaled fp8 MFMA with MX
        # (e4m3 + E8M0 per 32) weights and activations; LayerNorm, softmax, the residual stream, the stem, the
        # patch embedding and the head stay bf16 / fp32 (BASELINE.json configs[4])
        self.precision = "bf16"
        # mxfp8 mode: attention products on the fp8 MFMA too (rvit_attn_fwd_fp8) instead of the bf16-MFMA kernel with an
        # MX epilogue (rvit_attn_fwd_mx, the default since round 3: faster on gfx950, and closer to the fp32 oracle)
        self.fp8_attention = os.environ.get("RVIT_FP8_ATTENTION", "0") == "1"
        # fused encoder-layer kernels where the shape allows (RVIT_ENCODER_FUSED=0: the per-op path, for A/B runs)
        self.fused_encoder = os.environ.get("RVIT_ENCODER_FUSED", "1") != "0"
        self.FUSED_MIN_TOKENS = int(os.environ.get("RVIT_FUSED_MIN_TOKENS", "12800"))
        self.fused_stack = os.environ.get("RVIT_ENCODER_STACK", "1") != "0"   # all layers in one launch (A/B: 0 = per layer)
        self.fused_head = os.environ.get("RVIT_FUSED_HEAD", "1") != "0"        # norm + head in one launch (A/B: 0 = three)
        self.graph_max_batch = 0

    def set_precision(self, precision: str, fp8_attention=None) -> None:
        if fp8_attention is not None:
            self.fp8_attention = bool(fp8_attention)
        if precision not in ("bf16", "mxfp8"):
            raise ValueError(f"unknown inference precision {precision!r}; choose 'bf16' or 'mxfp8'")
        if precision == "mxfp8" and (self.m.embed_dim % 128 != 0):
            raise ValueError("mxfp8 needs embed_dim to be a multiple of 128")
        if precision == "mxfp8" and self.m.embed_dim // self.m.num_heads != 64:
            raise NotImplementedError("mxfp8: the MX-output attention kernels are built for head_dim 64 only")
        if precision != self.precision:
            self.precision = precision
            self._pack = None
            self._bufs.clear()

    # ---- weight packing ----------------------------------------------------------------------
    def _key(self):
        ps = self.m._live_tensors()
        return (tuple(p._version for p in ps), ps[0].data_ptr(), ps[0].device, self.m.training, self.precision)

    @torch.no_grad()
    def pack(self) -> dict:
        key = self._key()
        if self._pack is not None and key == self._pack_key:
            return self._pack
        m = self.m
        dev = m.pos_embed.device
        if dev.type != "cuda":
            raise RuntimeError("model parameters must be on a cuda device (no CPU path exists)")
        P: dict = {"device": dev}
        if m.backbone_name == "mobilenetv3":
            P["mb"] = self._pack_mobilenet(m.cnn_backbone)
            P["pe_w"] = m.patch_embed.weight.detach().reshape(m.embed_dim, -1).to(BF16).contiguous()
        elif m.cnn_backbone is not None:
            cb = m.cnn_backbone
            s, b = _bn_fold(cb.bn1)
            w1 = _pack_conv_px4(cb.conv1.weight, 8)                       # [64, 7*32]
            w1p = torch.zeros(64, 232, dtype=BF16, device=dev)             # row pitch of the fused stem kernel
            w1p[:, :224] = w1
            P["conv1"] = dict(w=w1, w232=w1p, scale=s, bias=b)
            blocks = []
            for blk in cb.blocks():
                d = {}
                for nm, conv, bn in (("c1", blk.conv1, blk.bn1), ("c2", blk.conv2, blk.bn2)):
                    s, b = _bn_fold(bn)
                    d[nm] = dict(w=_pack_conv(conv.weight), scale=s, bias=b, stride=conv.stride, cout=conv.cout)
                if blk.downsample is not None:
                    conv, bn = blk.downsample["0"], blk.downsample["1"]
                    s, b = _bn_fold(bn)
                    d["ds"] = dict(w=_pack_conv(conv.weight), scale=s, bias=b, stride=conv.stride, cout=conv.cout)
                blocks.append(d)
            P["blocks"] = blocks
            P["pe_w"] = m.patch_embed.weight.detach().reshape(m.embed_dim, -1).to(BF16).contiguous()
        else:
            P["pe_w"] = _pack_conv_px4(m.patch_embed.weight, 16)
        P["pe_b"] = m.patch_embed.bias.detach().float().contiguous()
        P["cls"] = m.cls_token.detach().float().reshape(-1).contiguous()
        P["pos"] = m.pos_embed.detach().float().reshape(-1, m.embed_dim).contiguous()
        layers = []
        for L in m.transformer.layers:
            layers.append(dict(
                ln1_w=L.norm1.weight.detach().float().contiguous(), ln1_b=L.norm1.bias.detach().float().contiguous(),
                ln2_w=L.norm2.weight.detach().float().contiguous(), ln2_b=L.norm2.bias.detach().float().contiguous(),
                qkv_w=L.self_attn.in_proj_weight.detach().to(BF16).contiguous(),
                qkv_b=L.self_attn.in_proj_bias.detach().float().contiguous(),
                out_w=L.self_attn.out_proj.weight.detach().to(BF16).contiguous(),
                out_b=L.self_attn.out_proj.bias.detach().float().contiguous(),
                fc1_w=L.linear1.weight.detach().to(BF16).contiguous(), fc1_b=L.linear1.bias.detach().float().contiguous(),
                fc2_w=L.linear2.weight.detach().to(BF16).contiguous(), fc2_b=L.linear2.bias.detach().float().contiguous(),
                eps1=L.norm1.eps, eps2=L.norm2.eps))
            if self.precision == "mxfp8":  # weights quantised once, here, from the fp32 parameters (not from
                d = layers[-1]             # their bf16 copies: double rounding flips ~3 % of the e4m3 values)
                for nm, w in (("qkv_w", L.self_attn.in_proj_weight), ("out_w", L.self_attn.out_proj.weight),
                              ("fc1_w", L.linear1.weight), ("fc2_w", L.linear2.weight)):
                    del d[nm]
                    d[nm + "_q"], d[nm + "_s"] = ops.quantize_mx(w.detach().float().contiguous())
        P["layers"] = layers
        P["norm_w"] = m.norm.weight.detach().float().contiguous()
        P["norm_b"] = m.norm.bias.detach().float().contiguous()
        P["h0_w"] = m.head["0"].weight.detach().to(BF16).contiguous()
        P["h0_b"] = m.head["0"].bias.detach().float().contiguous()
        P["h3_w"] = m.head["3"].weight.detach().to(BF16).contiguous()
        P["h3_b"] = m.head["3"].bias.detach().float().contiguous()
        self._pack, self._pack_key = P, key
        return P

    @staticmethod
    def _pack_mobilenet(cb) -> dict:
        """Folded-BatchNorm weights of the mobilenetv3_small_100 stem: pointwise convs [Cout,Cin] bf16 (GEMM B operands),
        depthwise convs [k*k, C] bf16 (tap-major), squeeze-excite projections as Linear weights with fp32 biases."""
        def pw(conv, bn):
            s, b = _bn_fold(bn)
            return dict(w=conv.weight.detach().reshape(conv.weight.shape[0], -1).to(BF16).contiguous(), scale=s, bias=b)

        def dw(conv, bn):
            s, b = _bn_fold(bn)
            c, k = conv.c, conv.k
            return dict(w=conv.weight.detach().reshape(c, k * k).t().to(BF16).contiguous(), scale=s, bias=b, k=k,
                        stride=conv.stride, pad=conv.pad)

        def se(mod):
            return dict(w1=mod.conv_reduce.weight.detach().reshape(mod.conv_reduce.weight.shape[0], -1).to(BF16).contiguous(),
                        b1=mod.conv_reduce.bias.detach().float().contiguous(),
                        w2=mod.conv_expand.weight.detach().reshape(mod.conv_expand.weight.shape[0], -1).to(BF16).contiguous(),
                        b2=mod.conv_expand.bias.detach().float().contiguous())
        s, b = _bn_fold(cb.bn1)
        # 3x3 / stride 2 stem conv straight from the fp32 input: weights f32 [27, 16], k = (kh*3 + kw)*3 + ci
        w27 = cb.conv_stem.weight.detach().float().permute(2, 3, 1, 0).reshape(27, 16).contiguous()
        out = dict(stem=dict(w=w27, scale=s, bias=b), blocks=[])
        for blk in cb.all_blocks():
            d = dict(kind=blk.kind, act=blk.act, cin=blk.cin, cout=blk.cout, mid=blk.mid, stride=blk.stride,
                     residual=blk.has_residual, se=se(blk.se) if hasattr(blk, "se") else None)
            if blk.kind == "ds":
                d["dw"], d["pwl"] = dw(blk.conv_dw, blk.bn1), pw(blk.conv_pw, blk.bn2)
            elif blk.kind == "ir":
                d["pw"], d["dw"], d["pwl"] = pw(blk.conv_pw, blk.bn1), dw(blk.conv_dw, blk.bn2), pw(blk.conv_pwl, blk.bn3)
            else:
                d["pw"] = pw(blk.conv, blk.bn1)
            out["blocks"].append(d)
        return out

    def _stem_mobilenet(self, P, b, B, xin) -> torch.Tensor:
        """mobilenetv3_small_100 features_only, last map [B,7,7,576] (NHWC bf16).  conv_stem is an implicit GEMM over the
        zero-haloed NHWC4 image; every pointwise conv and squeeze-excite projection is a GEMM over the NHWC map viewed
        as [B*H*W, C] with the folded BatchNorm, activation and residual in its epilogue; depthwise convs, the
        squeeze-excite pool and the gate multiply are the HBM-bound kernels of csrc/mobile.hip."""
        mb = P["mb"]
        dev = xin.device
        z = dict(dtype=BF16, device=dev)
        cache = b.setdefault("mb_maps", {})

        def buf(key, *shape):
            t = cache.get(key)
            if t is None or tuple(t.shape) != shape:
                t = torch.empty(*shape, **z)
                cache[key] = t
            return t
        st = mb["stem"]
        cur = buf("stem", B, 112, 112, 16)
        ops.conv3x3s2_rgb(xin, st["w"], st["scale"], st["bias"], cur, act="hswish")

        def pw(a, p, o, act=None, residual=None):   # 1x1 conv + folded BN (+ residual) (+ activation)
            fn = ops.pointwise_conv if ops.pointwise_supported(a.shape[1], o.shape[1]) else ops.gemm
            fn(a, p["w"], o, scale=p["scale"], bias=p["bias"], act=act, residual=residual)
        for i, d in enumerate(mb["blocks"]):
            Bc, H, W, C = cur.shape
            x_in = cur
            if d["kind"] == "cn":
                p = d["pw"]
                out = buf(f"o{i}", Bc, H, W, d["cout"])
                pw(cur.view(-1, C), p, out.view(-1, d["cout"]), act=d["act"])
                cur = out
                continue
            if d["kind"] == "ir":
                p = d["pw"]
                mid = buf(f"m{i}", Bc, H, W, d["mid"])
                pw(cur.view(-1, C), p, mid.view(-1, d["mid"]), act=d["act"])
                cur = mid
            q = d["dw"]
            Cm = cur.shape[3]
            Ho, Wo = (H + 2 * q["pad"] - q["k"]) // q["stride"] + 1, (W + 2 * q["pad"] - q["k"]) // q["stride"] + 1
            dwo = buf(f"d{i}", Bc, Ho, Wo, Cm)
            ops.dwconv(cur, q["w"], q["scale"], q["bias"], dwo, k=q["k"], stride=q["stride"], pad=q["pad"], act=d["act"])
            if d["se"] is not None:
                s_ = d["se"]
                rd = s_["w1"].shape[0]
                if ops.squeeze_excite_supported(Ho * Wo, Cm, rd):   # every block of mobilenetv3_small_100 at 224x224
                    ops.squeeze_excite(dwo, s_["w1"], s_["b1"], s_["w2"], s_["b2"])
                else:
                    pooled, hid, gate = buf(f"sp{i}", Bc, Cm), buf(f"sh{i}", Bc, rd), buf(f"sg{i}", Bc, Cm)
                    ops.avgpool(dwo, pooled)
                    ops.gemm(pooled, s_["w1"], hid, bias=s_["b1"], act="relu")
                    ops.gemm(hid, s_["w2"], gate, bias=s_["b2"], act="hsigmoid")
                    ops.channel_scale(dwo, gate)
            p = d["pwl"]
            out = buf(f"o{i}", Bc, Ho, Wo, d["cout"])
            pw(dwo.view(-1, Cm), p, out.view(-1, d["cout"]), residual=x_in.view(-1, d["cout"]) if d["residual"] else None)
            cur = out
        return cur

    # ---- activation buffers ------------------------------------------------------------------
    def buffers(self, B: int, dev) -> Dict[str, torch.Tensor]:
        bufs = self._bufs.get(B)
        if bufs is not None and bufs["x"].device == dev:
            return bufs
        if len(self._bufs) >= 4:  # keep the cache small: buffers are large
            self._bufs.pop(next(iter(self._bufs)))
        m = self.m
        E, S = m.embed_dim, m.num_patches + 1
        T = B * S
        z = dict(dtype=BF16, device=dev)
        b: Dict[str, torch.Tensor] = {}
        if m.backbone_name == "mobilenetv3":
            pass                                          # its stem reads the fp32 input directly; maps in b["mb_maps"]
        elif m.cnn_backbone is not None:
            b["img"] = torch.zeros(B, 230, 232, 4, **z)  # halo stays zero forever
            hw = 56
            for li, c in enumerate((64, 128, 256, 512), start=1):
                for nm in ("a", "b", "c", "d"):
                    b[f"s{li}{nm}"] = torch.empty(B, hw, hw, c, **z)
                hw //= 2
        else:
            b["img"] = torch.zeros(B, 224, 224, 4, **z)
        b["x"] = torch.empty(B, S, E, dtype=torch.float32, device=dev)
        b["ln"] = torch.empty(T, E, **z)
        b["qkv"] = torch.empty(T, 3 * E, **z)
        b["att"] = torch.empty(T, E, **z)
        if self.precision != "mxfp8":
            b["hid"] = torch.empty(T, 4 * E, **z)
        if self.precision == "mxfp8":
            u8 = dict(dtype=torch.uint8, device=dev)
            b["a_q"] = torch.empty(T, E, **u8)                        # MX copy of the LN / attention output
            b["a_s"] = torch.zeros(ops.mx_scale_shape(T, E), **u8)
            b["hid_q"] = torch.empty(T, 4 * E, **u8)                  # MX GELU output, written by linear1's epilogue
            b["hid_s"] = torch.zeros(ops.mx_scale_shape(T, 4 * E), **u8)
        b["cls_ln"] = torch.empty(B, E, **z)
        b["h0"] = torch.empty(B, E // 2, **z)
        self._bufs[B] = b
        return b

    # ---- forward -------------------------------------------------------------------------------
    def _stem(self, P, b, B, xin) -> torch.Tensor:
        c1 = P["conv1"]
        # conv7x7/2 + folded BN + ReLU + max-pool in one kernel, read straight from the fp32 NCHW input: neither the
        # 112x112x64 map nor a packed copy of the image ever reaches HBM
        cur = ops.stem_conv1_pool_f32(xin, c1["w232"], c1["scale"], c1["bias"], b["s1a"])
        stage, free = 1, ["b", "c", "d"]
        for blk in P["blocks"]:
            stride = blk["c1"]["stride"]
            if stride == 2:
                stage += 1
                free = ["a", "b", "c", "d"]
            # pick three scratch maps of this stage that are not `cur`
            names = [n for n in free if b[f"s{stage}{n}"].data_ptr() != cur.data_ptr()]
            t, o, dsb = (b[f"s{stage}{n}"] for n in names[:3])
            idn = cur
            if "ds" in blk:
                d = blk["ds"]
                idn = ops.conv2d(cur, d["w"], dsb, kh=1, kw=1, stride=d["stride"], pad=0, scale=d["scale"],
                                 bias=d["bias"])
            a = blk["c1"]
            ops.conv2d(cur, a["w"], t, kh=3, kw=3, stride=stride, pad=1, scale=a["scale"], bias=a["bias"], act="relu")
            c = blk["c2"]
            ops.conv2d(t, c["w"], o, kh=3, kw=3, stride=1, pad=1, scale=c["scale"], bias=c["bias"], residual=idn,
                       act="relu")
            cur = o
            free = ["a", "b", "c", "d"]
        return cur

    def _encoder_layer(self, L, b, B) -> None:
        """One pre-LN encoder layer on the fp32 residual stream b["x"], in place
        (torch transformer.py:945-950: x + sa(norm1(x)); x + ff(norm2(x)))."""
        m = self.m
        E, S, H = m.embed_dim, m.num_patches + 1, m.num_heads
        x2 = b["x"].view(B * S, E)
        if self.precision == "mxfp8":
            aq, as_ = b["a_q"], b["a_s"]
            # LayerNorm emits its result directly as MX-fp8 (same bytes as layernorm -> quantize_mx, one pass)
            ops.layernorm_mx(x2, L["ln1_w"], L["ln1_b"], aq, as_, L["eps1"])
            ops.gemm_mx(aq, as_, L["qkv_w_q"], L["qkv_w_s"], b["qkv"], bias=L["qkv_b"])
            # the attention result is written as the MX operand of the out_proj.  Default: the bf16-MFMA kernel with
            # an MX epilogue (un-scaled fp8 MFMAs run at the bf16 rate on gfx950 and quantising q / k / v / p costs
            # vector-ALU time: 185 vs 140 us per ViT-L layer); fp8_attention: both products on the fp8 MFMA
            if self.fp8_attention:
                ops.attention_fp8(b["qkv"], aq, as_, B, S, H)
            else:
                ops.attention_mx(b["qkv"], aq, as_, B, S, H)
            ops.gemm_mx(aq, as_, L["out_w_q"], L["out_w_s"], x2, bias=L["out_b"], residual=x2)
            ops.layernorm_mx(x2, L["ln2_w"], L["ln2_b"], aq, as_, L["eps2"])
            # linear1's epilogue emits the GELU output directly as MX-fp8 (no bf16 copy, no quantisation pass)
            ops.gemm_mx(aq, as_, L["fc1_w_q"], L["fc1_w_s"], None, bias=L["fc1_b"], act="gelu", out_q=b["hid_q"],
                        out_s=b["hid_s"])
            ops.gemm_mx(b["hid_q"], b["hid_s"], L["fc2_w_q"], L["fc2_w_s"], x2, bias=L["fc2_b"], residual=x2)
            return
        # E = 256: each half of the layer is ONE kernel that keeps its tokens on the CU from the LayerNorm to the
        # residual add (csrc/encoder.hip).  One workgroup per CU owns ~100 tokens / two images, so the fused kernels
        # need a few hundred images to fill the chip: below FUSED_MIN_TOKENS the per-op sequence is faster (B = 64:
        # 0.97 vs 1.09 ms per forward; B = 1: 0.62 vs 0.77 ms).  Both paths give the SAME BITS (the per-op attention
        # then is attention_s64, the fused kernel's core as a kernel of its own), so the choice by batch size does not
        # make a frame's logits depend on the batch it travels in (tests/test_encoder_gpu.py, test_model_gpu.py).
        shape_ok = ops.encoder_attn_supported(E, H, S) and ops.encoder_mlp_supported(E, L["fc1_w"].shape[0])
        fused = self.fused_encoder and shape_ok and B * S >= self.FUSED_MIN_TOKENS
        if fused:
            ops.encoder_attn(x2, L["ln1_w"], L["ln1_b"], L["qkv_w"], L["qkv_b"], L["out_w"], L["out_b"], B, S, H, L["eps1"])
        else:
            ops.layernorm(x2, L["ln1_w"], L["ln1_b"], b["ln"], L["eps1"])
            ops.gemm(b["ln"], L["qkv_w"], b["qkv"], bias=L["qkv_b"])
            if shape_ok:
                ops.attention_s64(b["qkv"], b["att"], B, S, H)
            else:
                ops.attention(b["qkv"], b["att"], B, S, H)
            ops.gemm(b["att"], L["out_w"], x2, bias=L["out_b"], residual=x2)
        if fused:
            ops.encoder_mlp(x2, L["ln2_w"], L["ln2_b"], L["fc1_w"], L["fc1_b"], L["fc2_w"], L["fc2_b"], L["eps2"])
        else:
            ops.layernorm(x2, L["ln2_w"], L["ln2_b"], b["ln"], L["eps2"])
            ops.gemm(b["ln"], L["fc1_w"], b["hid"], bias=L["fc1_b"], act="gelu")
            ops.gemm(b["hid"], L["fc2_w"], x2, bias=L["fc2_b"], residual=x2)

    def _encoder(self, P, b, B) -> None:
        """`self.transformer(x)` (train_vit.py:163) on b["x"].  Where the fused kernels apply and every LayerNorm shares
        one eps, the whole stack is ONE launch (csrc/encoder.hip enc_stack_kernel: a workgroup keeps its two images from
        the first LayerNorm to the last residual add; same bits as the per-layer kernels), else layer by layer."""
        m = self.m
        E, S, H = m.embed_dim, m.num_patches + 1, m.num_heads
        layers = P["layers"]
        if (self.precision != "mxfp8" and self.fused_encoder and self.fused_stack and layers
                and B * S >= self.FUSED_MIN_TOKENS
                and ops.encoder_stack_supported(E, H, S, layers[0]["fc1_w"].shape[0], len(layers))
                and len({L["eps1"] for L in layers} | {L["eps2"] for L in layers}) == 1):
            ops.encoder_stack(b["x"].view(B * S, E),
                              [(L["ln1_w"], L["ln1_b"], L["qkv_w"], L["qkv_b"], L["out_w"], L["out_b"], L["ln2_w"], L["ln2_b"],
                                L["fc1_w"], L["fc1_b"], L["fc2_w"], L["fc2_b"]) for L in layers], B, S, H, layers[0]["eps1"])
            return
        for L in layers:
            self._encoder_layer(L, b, B)

    @torch.no_grad()
    def infer(self, x: torch.Tensor, features_only: bool) -> torch.Tensor:
        m = self.m
        P = self.pack()
        if x.device != P["device"]:
            raise RuntimeError(f"input on {x.device} but model on {P['device']}")
        B = x.shape[0]
        E, S, H = m.embed_dim, m.num_patches + 1, m.num_heads
        T = B * S
        b = self.buffers(B, x.device)
        xin = x.detach()
        if xin.dtype != torch.float32:
            xin = xin.float()
        xin = xin.contiguous()
        xs = b["x"]
        x2 = xs.view(T, E)
        if m.cnn_backbone is not None:
            feat = self._stem_mobilenet(P, b, B, xin) if m.backbone_name == "mobilenetv3" else self._stem(P, b, B, xin)  # [B,7,7,C]
            ops.gemm(feat.view(B * m.num_patches, -1), P["pe_w"], x2, bias=P["pe_b"], residual=P["pos"][1:],
                     residual_period=m.num_patches, row_group=m.num_patches, row_extra=1)
        else:
            ops.input_pack(xin, b["img"], 0, 0)
            ops.conv2d_tokens(b["img"], P["pe_w"], xs, P["pos"], kh=16, kw=1, stride=16, pad=0, seg_len=64, ho=14,
                              wo=14, bias=P["pe_b"])
        ops.embed_cls(P["cls"], P["pos"], xs)
        self._encoder(P, b, B)
        cls_rows = xs.view(B, S * E)[:, :E]
        if features_only:
            out = torch.empty(B, E, dtype=torch.float32, device=x.device)
            ops.layernorm(cls_rows, P["norm_w"], P["norm_b"], out, m.norm.eps)
            return out
        logits = torch.empty(B, m.num_classes, dtype=torch.float32, device=x.device)
        if self.fused_head and ops.cls_head_supported(E, P["h0_w"].shape[0], m.num_classes):
            # final LayerNorm + both Linears of the head in one launch (csrc/head.hip; same bits as the three below)
            return ops.cls_head(cls_rows, P["norm_w"], P["norm_b"], m.norm.eps, P["h0_w"], P["h0_b"], P["h3_w"], P["h3_b"],
                                logits)
        ops.layernorm(cls_rows, P["norm_w"], P["norm_b"], b["cls_ln"], m.norm.eps)
        ops.gemm(b["cls_ln"], P["h0_w"], b["h0"], bias=P["h0_b"], act="gelu")
        ops.gemm(b["h0"], P["h3_w"], logits, bias=P["h3_b"])
        return logits

    # ---- hipGraph replay for launch-bound batch sizes ---------------------------------------------
    def set_graph_capture(self, max_batch: int) -> None:
        """Forwards of at most `max_batch` frames are captured once per (batch, output kind) into a HIP graph
        and replayed: ~65 kernel launches become one graph launch (B=1 is launch-bound).  0 disables."""
        self.graph_max_batch = int(max_batch)

    @torch.no_grad()
    def _infer_graph(self, x: torch.Tensor, features_only: bool) -> torch.Tensor:
        P = self.pack()
        if x.device != P["device"]:
            raise RuntimeError(f"input on {x.device} but model on {P['device']}")
        B = x.shape[0]
        graphs = P.setdefault("_graphs", {})     # tied to this weight pack: a re-pack drops them
        rec = graphs.get((B, features_only))
        if rec is None:
            static_in = torch.empty(B, 3, 224, 224, dtype=torch.float32, device=x.device)
            static_in.copy_(x)
            cur = torch.cuda.current_stream(x.device)
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):        # warm-up off the capture: buffers, lazy kernel attributes
                for _ in range(2):
                    self.infer(static_in, features_only)
            cur.wait_stream(side)
            bufs = self.buffers(B, x.device)     # keep them alive even if the small buffer cache evicts them
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.infer(static_in, features_only)
            rec = (graph, static_in, static_out, bufs)
            if len(graphs) >= 8:
                graphs.pop(next(iter(graphs)))
            graphs[(B, features_only)] = rec
        graph, static_in, static_out, _ = rec
        static_in.copy_(x)
        graph.replay()
        return static_out.clone()

    def run(self, x: torch.Tensor, features_only: bool) -> torch.Tensor:
        m = self.m
        if m.training:
            raise RuntimeError("call model.eval() for inference: BatchNorm batch statistics and dropout of "
                               "train mode are only available on the training path")
        if 0 < x.shape[0] <= self.graph_max_batch and ops._PROF is None and not torch.cuda.is_current_stream_capturing():
            return self._infer_graph(x, features_only)
        return self.infer(x, features_only)
