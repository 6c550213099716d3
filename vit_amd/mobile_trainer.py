"""Training pass of the MobileNetV3 stem (`backbone='mobilenetv3'`: timm mobilenetv3_small_100 features_only,
train_vit.py:68-74,87,145-146; its autograd is what `loss.backward()` runs at train_vit.py:939): train-mode forward with
saved activations and the hand-scheduled backward, on librvit kernels only.  Used by vit_amd.trainer.TrainEngine.

Per block (timm order): DepthwiseSeparableConv `ds` = conv_dw, bn1 + act, [se], conv_pw, bn2 (+ shortcut);
InvertedResidual `ir` = conv_pw, bn1 + act, conv_dw, bn2 + act, [se], conv_pwl, bn3 (+ shortcut); ConvBnAct `cn` = conv, bn1
+ act.  Every convolution output `y` (pre-BatchNorm) and every BatchNorm output `z` is kept (NHWC bf16): BatchNorm
backward recomputes the activation's argument from `y`, convolution weight gradients read the `z` in front of them.
Pointwise convolutions and the squeeze-excite projections are GEMMs over the map viewed as [B*H*W, C] (ops.gemm /
ops.wgrad / ops.linear_wgrad_bias); depthwise convolutions, BatchNorm of these widths and the squeeze-excite pieces are
csrc/mobile.hip + csrc/mobile_train.hip.  All reductions are ordered: two runs of a step give the same bits.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import ops

BF16 = torch.bfloat16
F32 = torch.float32


class MobileStemTrainer:
    def __init__(self, eng):
        self.eng = eng
        self.cb = eng.m.cnn_backbone
        self.blocks = list(self.cb.all_blocks())
        self.names = [f"cnn_backbone.blocks.{si}.{bi}." for si, st in enumerate(self.cb.blocks) for bi in range(len(st))]

    # ---- operand copies (packer interface of trainer.py: b = [N,K] bf16, t = [K,N] bf16) -------------------------
    def build_pack(self, pk) -> List[dict]:
        out = []
        for pref, blk in zip(self.names, self.blocks):
            d: Dict[str, object] = {}
            if blk.kind == "cn":
                d["pw_w"], d["pw_wt"] = pk.b(pref + "conv.weight"), pk.t(pref + "conv.weight")
            else:
                d["dw_w"] = pk.t_small(pref + "conv_dw.weight")                       # [k*k, C]: the kernels' tap-major pack
                last = "conv_pw" if blk.kind == "ds" else "conv_pwl"
                d["pwl_w"], d["pwl_wt"] = pk.b(pref + last + ".weight"), pk.t(pref + last + ".weight")
                if blk.kind == "ir":
                    d["pw_w"], d["pw_wt"] = pk.b(pref + "conv_pw.weight"), pk.t(pref + "conv_pw.weight")
                if hasattr(blk, "se"):
                    d["se_w1"], d["se_w1t"] = pk.b(pref + "se.conv_reduce.weight"), pk.t(pref + "se.conv_reduce.weight")
                    d["se_w2"], d["se_w2t"] = pk.b(pref + "se.conv_expand.weight"), pk.t(pref + "se.conv_expand.weight")
            out.append(d)
        return out

    # ---- buffers ---------------------------------------------------------------------------------------------------
    def buffers(self, B: int, dev) -> dict:
        z = dict(dtype=BF16, device=dev)
        f = dict(dtype=F32, device=dev)
        mb: Dict[str, object] = {"img": torch.zeros(B, 226, 228, 4, **z)}   # zero halo of 1 (+ row padding): conv_stem's wgrad
        mb["y0"], mb["z0"] = torch.empty(B, 112, 112, 16, **z), torch.empty(B, 112, 112, 16, **z)
        nsum = 32
        hw, cin = 112, 16
        blocks = []
        for blk in self.blocks:
            d: Dict[str, torch.Tensor] = {}
            if blk.kind == "cn":
                d["y1"], d["z1"] = torch.empty(B, hw, hw, blk.cout, **z), torch.empty(B, hw, hw, blk.cout, **z)
                nsum += 2 * blk.cout
            else:
                mid = blk.mid if blk.kind == "ir" else cin
                ho = (hw + 2 * (blk.k // 2) - blk.k) // blk.stride + 1
                if blk.kind == "ir":
                    d["y_pw"], d["z_pw"] = torch.empty(B, hw, hw, mid, **z), torch.empty(B, hw, hw, mid, **z)
                    nsum += 2 * mid
                d["y_dw"], d["z_dw"] = torch.empty(B, ho, ho, mid, **z), torch.empty(B, ho, ho, mid, **z)
                if hasattr(blk, "se"):
                    rd = blk.se.conv_reduce.weight.shape[0]
                    d["pool"], d["hid"], d["gate"] = torch.empty(B, mid, **z), torch.empty(B, rd, **z), torch.empty(B, mid, **z)
                    d["z_se"] = torch.empty(B, ho, ho, mid, **z)
                    d["d_c"], d["d_rd"], d["d_c2"] = torch.empty(B, mid, **z), torch.empty(B, rd, **z), torch.empty(B, mid, **z)
                    d["inv_hw"] = torch.full((mid,), 1.0 / (ho * ho), **f)
                d["y_out"], d["z_out"] = torch.empty(B, ho, ho, blk.cout, **z), torch.empty(B, ho, ho, blk.cout, **z)
                nsum += 2 * mid + 2 * blk.cout
                hw = ho
            cin = blk.cout
            blocks.append(d)
        mb["blocks"] = blocks
        mb["fsums"], mb["bsums"] = torch.zeros(nsum, **f), torch.zeros(nsum, **f)
        mb["ones16"], mb["zeros16"] = torch.ones(16, **f), torch.zeros(16, **f)
        mb["unit"] = {}          # per-width (ones, zeros) for the raw depthwise convolutions
        mb["scratch"] = {}
        return mb

    @staticmethod
    def _unit(mb, C, dev):
        u = mb["unit"].get(C)
        if u is None:
            u = (torch.ones(C, dtype=F32, device=dev), torch.zeros(C, dtype=F32, device=dev))
            mb["unit"][C] = u
        return u

    @staticmethod
    def _tmp(mb, key, like: torch.Tensor) -> torch.Tensor:
        t = mb["scratch"].get(key)
        if t is None or t.shape != like.shape:
            t = torch.empty_like(like)
            mb["scratch"][key] = t
        return t

    # ---- forward ---------------------------------------------------------------------------------------------------
    def _bn_fwd(self, mb, bn, y, z, act, residual=None):
        C = y.shape[-1]
        off = mb["off"]
        fs = mb["fsums"][off:off + 2 * C]
        mb["off"] = off + 2 * C
        ops.bn_generic_stats(y, fs)
        ops.bn_generic_apply(y, fs, bn.weight.detach(), bn.bias.detach(), z, eps=bn.eps, act=act, residual=residual,
                             momentum=bn.momentum, running_mean=bn.running_mean, running_var=bn.running_var)
        return off

    def forward(self, P: List[dict], mb: dict, B: int, xin: torch.Tensor) -> torch.Tensor:
        cb, dev = self.cb, xin.device
        mb["fsums"].zero_()
        mb["off"] = 0
        w27 = cb.conv_stem.weight.detach().float().permute(2, 3, 1, 0).reshape(27, 16).contiguous()
        ops.conv3x3s2_rgb(xin, w27, mb["ones16"], mb["zeros16"], mb["y0"])
        mb["xin"] = xin
        mb["stem_off"] = self._bn_fwd(mb, cb.bn1, mb["y0"], mb["z0"], "hswish")
        cur = mb["z0"]
        for blk, pk, d in zip(self.blocks, P, mb["blocks"]):
            Bc, H, W, C = cur.shape
            d["x"] = cur
            act = blk.act
            if blk.kind == "cn":
                ops.gemm(cur.view(-1, C), pk["pw_w"], d["y1"].view(-1, blk.cout))
                d["o1"] = self._bn_fwd(mb, blk.bn1, d["y1"], d["z1"], act)
                cur = d["z1"]
                continue
            t = cur
            if blk.kind == "ir":
                ops.gemm(cur.view(-1, C), pk["pw_w"], d["y_pw"].view(-1, blk.mid))
                d["o_pw"] = self._bn_fwd(mb, blk.bn1, d["y_pw"], d["z_pw"], act)
                t = d["z_pw"]
            Cm = t.shape[3]
            one, zero = self._unit(mb, Cm, dev)
            ops.dwconv(t, pk["dw_w"], one, zero, d["y_dw"], k=blk.k, stride=blk.stride, pad=blk.k // 2)
            d["o_dw"] = self._bn_fwd(mb, blk.bn1 if blk.kind == "ds" else blk.bn2, d["y_dw"], d["z_dw"], act)
            t = d["z_dw"]
            if hasattr(blk, "se"):
                se = blk.se
                ops.avgpool(t, d["pool"])
                ops.gemm(d["pool"], pk["se_w1"], d["hid"], bias=se.conv_reduce.bias.detach(), act="relu")
                ops.gemm(d["hid"], pk["se_w2"], d["gate"], bias=se.conv_expand.bias.detach(), act="hsigmoid")
                ops.channel_affine(t, d["gate"], d["z_se"])
                t = d["z_se"]
            d["pwl_in"] = t
            ops.gemm(t.view(-1, Cm), pk["pwl_w"], d["y_out"].view(-1, blk.cout))
            d["o_out"] = self._bn_fwd(mb, blk.bn2 if blk.kind == "ds" else blk.bn3, d["y_out"], d["z_out"], None,
                                      residual=cur if blk.has_residual else None)
            cur = d["z_out"]
        return cur

    # ---- backward --------------------------------------------------------------------------------------------------
    def _bn_bwd(self, mb, flat, pref, bn, dz, y, off, act, dy):
        C = y.shape[-1]
        g = self.eng._g
        ops.bn_generic_bwd(dz, y, mb["fsums"][off:off + 2 * C], mb["bsums"][off:off + 2 * C], bn.weight.detach(),
                           bn.bias.detach(), dy, g(flat, pref + "weight"), g(flat, pref + "bias"), eps=bn.eps, act=act)
        return dy

    def backward(self, P: List[dict], mb: dict, B: int, flat: torch.Tensor, dfeat: torch.Tensor) -> None:
        eng, cb = self.eng, self.cb
        g = eng._g
        mb["bsums"].zero_()
        dz = dfeat
        for blk, pk, d, pref in reversed(list(zip(self.blocks, P, mb["blocks"], self.names))):
            x = d["x"]
            Cin = x.shape[3]
            act = blk.act
            if blk.kind == "cn":
                dy = self._bn_bwd(mb, flat, pref + "bn1.", blk.bn1, dz, d["y1"], d["o1"], act, self._tmp(mb, ("dy", d["y1"].shape), d["y1"]))
                ops.wgrad(dy.view(-1, blk.cout), x.view(-1, Cin), g(flat, pref + "conv.weight").view(blk.cout, Cin))
                dx = self._tmp(mb, ("dx", x.shape), x)
                ops.gemm(dy.view(-1, blk.cout), pk["pw_wt"], dx.view(-1, Cin))
                dz = dx
                eng._maybe_reduce(flat, pref + "conv.weight")
                continue
            if blk.has_residual and blk.kind != "ir":
                raise NotImplementedError("shortcut around a depthwise-separable block (not in mobilenetv3_small_100)")
            dout = dz
            last, bn_last = ("conv_pw", "bn2") if blk.kind == "ds" else ("conv_pwl", "bn3")
            t_in = d["pwl_in"]
            Cm = t_in.shape[3]
            dy = self._bn_bwd(mb, flat, pref + bn_last + ".", getattr(blk, bn_last), dout, d["y_out"], d["o_out"], None,
                              self._tmp(mb, ("dy", d["y_out"].shape), d["y_out"]))
            ops.wgrad(dy.view(-1, blk.cout), t_in.view(-1, Cm), g(flat, pref + last + ".weight").view(blk.cout, Cm))
            dt = self._tmp(mb, ("dt", t_in.shape), t_in)
            ops.gemm(dy.view(-1, blk.cout), pk["pwl_wt"], dt.view(-1, Cm))
            if hasattr(blk, "se"):
                se, z = blk.se, d["z_dw"]
                rd = d["hid"].shape[1]
                # gate = hardsigmoid(W2 relu(W1 pool(z) + b1) + b2), z_se = z * gate
                ops.channel_dot(dt, z, d["d_c"], gate=d["gate"])                   # d(pre-hardsigmoid) [B, C]
                self._wgrad_bias(d["d_c"], d["hid"], g(flat, pref + "se.conv_expand.weight").view(Cm, rd),
                                 g(flat, pref + "se.conv_expand.bias"))
                ops.gemm(d["d_c"], pk["se_w2t"], d["d_rd"], gate=d["hid"], gate_mode="relu_mask")   # d(pre-ReLU) [B, rd]
                self._wgrad_bias(d["d_rd"], d["pool"], g(flat, pref + "se.conv_reduce.weight").view(rd, Cm),
                                 g(flat, pref + "se.conv_reduce.bias"))
                ops.gemm(d["d_rd"], pk["se_w1t"], d["d_c2"], scale=d["inv_hw"])       # d(pool) / HW [B, C]
                dzz = self._tmp(mb, ("dzz", z.shape), z)
                ops.channel_affine(dt, d["gate"], dzz, add=d["d_c2"])
                dt = dzz
            bn_dw = "bn1" if blk.kind == "ds" else "bn2"
            dy_dw = self._bn_bwd(mb, flat, pref + bn_dw + ".", getattr(blk, bn_dw), dt, d["y_dw"], d["o_dw"], act,
                                 self._tmp(mb, ("dy_dw", d["y_dw"].shape), d["y_dw"]))
            dw_in = d["z_pw"] if blk.kind == "ir" else x
            ops.dwconv_wgrad(dy_dw, dw_in, g(flat, pref + "conv_dw.weight"), k=blk.k, stride=blk.stride, pad=blk.k // 2)
            d_in = self._tmp(mb, ("d_in", dw_in.shape), dw_in)
            ops.dwconv_dgrad(dy_dw, pk["dw_w"], d_in, k=blk.k, stride=blk.stride, pad=blk.k // 2)
            if blk.kind == "ir":
                dy_pw = self._bn_bwd(mb, flat, pref + "bn1.", blk.bn1, d_in, d["y_pw"], d["o_pw"], act,
                                     self._tmp(mb, ("dy_pw", d["y_pw"].shape), d["y_pw"]))
                ops.wgrad(dy_pw.view(-1, blk.mid), x.view(-1, Cin), g(flat, pref + "conv_pw.weight").view(blk.mid, Cin))
                dx = self._tmp(mb, ("dx", x.shape), x)
                ops.gemm(dy_pw.view(-1, blk.mid), pk["pw_wt"], dx.view(-1, Cin),
                         residual=dout.view(-1, Cin) if blk.has_residual else None)   # + the shortcut's gradient
                dz = dx
            else:
                dz = d_in
            first = "conv_pw" if blk.kind == "ir" else "conv_dw"
            eng._maybe_reduce(flat, pref + first + ".weight")
        # conv_stem: BatchNorm + hard-swish backward, then the 3x3 / stride 2 weight gradient over the NHWC4 image with a
        # halo of one pixel (a kernel row = 4 pixels x 4 channels = one 16-element run; the 4th pixel / channel meet zeros)
        dy0 = self._bn_bwd(mb, flat, "cnn_backbone.bn1.", cb.bn1, dz, mb["y0"], mb["stem_off"], "hswish",
                           self._tmp(mb, ("dy0",), mb["y0"]))
        ops.input_pack(mb["xin"], mb["img"], 1, 1)
        dw1 = torch.zeros(16, 48, dtype=F32, device=dy0.device)
        ops.conv2d_wgrad(dy0, mb["img"], dw1, kh=3, kw=1, stride=2, pad=0, seg_len=16)
        g(flat, "cnn_backbone.conv_stem.weight").view(16, 3, 3, 3).copy_(dw1.view(16, 3, 4, 4)[:, :, :3, :3])

    @staticmethod
    def _wgrad_bias(dy, x, dw, db):
        ops.wgrad(dy, x, dw)
        ops.colsum(dy, db)
