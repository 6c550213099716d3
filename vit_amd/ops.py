"""Tensor-level wrappers over the librvit C ABI.

Each function enqueues one HIP kernel on torch's current stream.  Tensors must already live on
the GPU; shapes/dtypes are validated here and again inside the library.  No function in this
module has a non-HIP implementation.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, RVIT_BF16, RVIT_F32, check  # noqa: F401

_ACT = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "gelu": ACT_GELU}


# ---- optional per-launch profiling (bench.py) ----------------------------------------------------
# When a list is installed with set_profiler(), every wrapper brackets its launch with HIP events on
# the launch stream and appends (kernel_name, start_event, end_event, algorithmic_flops,
# algorithmic_bytes).  Disabled (None) in normal operation.
_PROF = None


def set_profiler(records):
    global _PROF
    _PROF = records


class _Span:
    __slots__ = ("name", "flops", "bytes", "e0")

    def __init__(self, name, flops, nbytes):
        self.name, self.flops, self.bytes = name, flops, nbytes

    def __enter__(self):
        if _PROF is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if _PROF is not None and exc[0] is None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream())
            _PROF.append((self.name, self.e0, e1, self.flops, self.bytes))
        return False


def _nbytes(*ts) -> int:
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return RVIT_BF16
    if t.dtype == torch.float32:
        return RVIT_F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _need_cuda(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vit_amd ops run only on the GPU (HIP); got a CPU tensor. There is no CPU fallback.")


def _rows2d(t: torch.Tensor, what: str):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{what}: expected a 2-D tensor with unit inner stride, got {tuple(t.shape)} "
                         f"strides {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
         scale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, residual_period: int = 0,
         act: Optional[str] = None, preact: Optional[torch.Tensor] = None, row_group: int = 0, row_extra: int = 0,
         tile: int = 0) -> torch.Tensor:
    """out = act(scale * (a @ w.T) + bias + residual); a [M,K] bf16, w [N,K] bf16."""
    _need_cuda(a, w, out, bias, scale, residual, preact)
    M, K, lda = _rows2d(a, "gemm a")
    N, K2, ldb = _rows2d(w, "gemm w")
    if K != K2:
        raise ValueError(f"gemm: K mismatch {K} vs {K2}")
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        raise TypeError("gemm: a and w must be bfloat16")
    mo, no, ldc = _rows2d(out, "gemm out")
    rows_out = M if row_group == 0 else (M // row_group) * (row_group + row_extra)
    if no != N or mo < rows_out:
        raise ValueError(f"gemm: out shape {tuple(out.shape)} incompatible with M={M} N={N}")
    ldr = 0
    r_dt = RVIT_F32
    if residual is not None:
        rr, rn, ldr = _rows2d(residual, "gemm residual")
        if rn != N or rr < (residual_period if residual_period else rows_out):
            raise ValueError("gemm: residual shape mismatch")
        r_dt = _dt(residual)
    ldp = 0
    if preact is not None:
        pr, pn, ldp = _rows2d(preact, "gemm preact")
        if pn != N or pr < rows_out or preact.dtype != torch.bfloat16:
            raise ValueError("gemm: preact must be bf16 [rows, N]")
    for v, nm in ((bias, "bias"), (scale, "scale")):
        if v is not None and (v.dtype != torch.float32 or v.numel() != N or not v.is_contiguous()):
            raise ValueError(f"gemm: {nm} must be contiguous float32 [N]")
    lib = _lib.load()
    nb = (M * K + N * K) * 2 + M * N * out.element_size() + (M * N * residual.element_size() if residual is not None
                                                             and not residual_period else 0)
    with _Span("gemm_bf16", 2 * M * N * K, nb):
        check(lib.rvit_gemm_bf16(a.data_ptr(), lda, w.data_ptr(), ldb, out.data_ptr(), ldc, _dt(out), M, N, K,
                                 _ptr(scale), _ptr(bias), _ptr(residual), ldr, r_dt, residual_period, _ACT[act],
                                 _ptr(preact), ldp, row_group, row_extra, tile, _stream()), "rvit_gemm_bf16")
    return out


def conv2d(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, kh: int, kw: int, stride: int, pad: int,
           seg_len: Optional[int] = None, scale: Optional[torch.Tensor] = None,
           bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           act: Optional[str] = None, tile: int = 0) -> torch.Tensor:
    """NHWC implicit-GEMM conv: x [B,H,W,Cpix] bf16, w [Cout, kh*kw*seg_len] bf16, out [B,Ho,Wo,Cout] bf16."""
    _need_cuda(x, w, out, scale, bias, residual)
    if x.dim() != 4 or out.dim() != 4 or not x.is_contiguous() or not out.is_contiguous() or not w.is_contiguous():
        raise ValueError("conv2d: x/out must be contiguous NHWC 4-D tensors, w contiguous")
    if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
        raise TypeError("conv2d: bf16 only")
    B, H, W, Cpix = x.shape
    Bo, Ho, Wo, Cout = out.shape
    seg = Cpix if seg_len is None else seg_len
    if Bo != B or w.shape != (Cout, kh * kw * seg):
        raise ValueError(f"conv2d: weight shape {tuple(w.shape)} != {(Cout, kh * kw * seg)} or batch mismatch")
    if residual is not None and (residual.shape != out.shape or residual.dtype != torch.bfloat16
                                 or not residual.is_contiguous()):
        raise ValueError("conv2d: residual must match out (bf16, contiguous)")
    for v, nm in ((bias, "bias"), (scale, "scale")):
        if v is not None and (v.dtype != torch.float32 or v.numel() != Cout or not v.is_contiguous()):
            raise ValueError(f"conv2d: {nm} must be contiguous float32 [Cout]")
    lib = _lib.load()
    # algorithmic work: true taps only (the 7x7 stem conv's zero taps / 4th channel are padding)
    true_k = kh * kw * seg if seg_len is None else kh * (kw * seg - seg // 8 - (seg - seg // 8) // 4)
    with _Span("conv2d_bf16", 2 * B * Ho * Wo * Cout * true_k, _nbytes(x, w, out, residual)):
        check(lib.rvit_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), out.data_ptr(), B, H, W, Cpix, seg, kh, kw, stride,
                                       pad, Ho, Wo, Cout, _ptr(scale), _ptr(bias), _ptr(residual), _ACT[act], tile,
                                       _stream()), "rvit_conv2d_fwd_bf16")
    return out


def conv2d_tokens(x: torch.Tensor, w: torch.Tensor, tokens: torch.Tensor, pos_embed: torch.Tensor, *, kh: int,
                  kw: int, stride: int, pad: int, seg_len: int, ho: int, wo: int,
                  bias: Optional[torch.Tensor] = None, tile: int = 0) -> torch.Tensor:
    """Patch conv into the fp32 token stream: tokens[b,1+p] = conv(x,w)[b,p] + bias + pos_embed[1+p]."""
    _need_cuda(x, w, tokens, pos_embed, bias)
    B, H, W, Cpix = x.shape
    Bt, S, E = tokens.shape
    if Bt != B or S != ho * wo + 1 or tokens.dtype != torch.float32 or not tokens.is_contiguous():
        raise ValueError("conv2d_tokens: tokens must be contiguous float32 [B, ho*wo+1, E]")
    if w.shape != (E, kh * kw * seg_len) or w.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 \
            or not w.is_contiguous() or not x.is_contiguous():
        raise ValueError("conv2d_tokens: bad x / w")
    if pos_embed.dtype != torch.float32 or pos_embed.numel() != S * E or not pos_embed.is_contiguous():
        raise ValueError("conv2d_tokens: pos_embed must be contiguous float32 [S,E]")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != E or not bias.is_contiguous()):
        raise ValueError("conv2d_tokens: bias must be contiguous float32 [E]")
    true_k = kh * (seg_len // 4) * 3  # px4 runs carry 3 real channels
    with _Span("conv2d_tokens", 2 * B * ho * wo * E * true_k, _nbytes(x, w, tokens)):
        check(_lib.load().rvit_conv2d_tokens_fwd(x.data_ptr(), w.data_ptr(), tokens.data_ptr(), B, H, W, Cpix,
                                                 seg_len, kh, kw, stride, pad, ho, wo, E, _ptr(bias),
                                                 pos_embed.data_ptr(), tile, _stream()), "rvit_conv2d_tokens_fwd")
    return tokens


def input_pack(x: torch.Tensor, out: torch.Tensor, pad_top: int, pad_left: int) -> torch.Tensor:
    """x fp32 [B,C<=3,H,W] -> interior of the zero-haloed bf16 NHWC4 buffer out [B,Hp,Wp,4]."""
    _need_cuda(x, out)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4:
        raise ValueError("input_pack: x must be contiguous float32 NCHW")
    if out.dtype != torch.bfloat16 or not out.is_contiguous() or out.dim() != 4 or out.shape[3] != 4:
        raise ValueError("input_pack: out must be contiguous bf16 [B,Hp,Wp,4]")
    B, C, H, W = x.shape
    if out.shape[0] != B:
        raise ValueError("input_pack: batch mismatch")
    with _Span("input_pack", 0, x.numel() * 4 + B * H * W * 8):
        check(_lib.load().rvit_input_pack_nhwc4(x.data_ptr(), out.data_ptr(), B, C, H, W, out.shape[1], out.shape[2],
                                                pad_top, pad_left, _stream()), "rvit_input_pack_nhwc4")
    return out


def maxpool3x3s2(x: torch.Tensor, out: torch.Tensor, argmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(x, out, argmax)
    B, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if tuple(out.shape) != (B, Ho, Wo, C) or x.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
        raise ValueError(f"maxpool: out must be bf16 {(B, Ho, Wo, C)}")
    if not x.is_contiguous() or not out.is_contiguous():
        raise ValueError("maxpool: contiguous NHWC required")
    if argmax is not None and (argmax.dtype != torch.uint8 or argmax.shape != out.shape or not argmax.is_contiguous()):
        raise ValueError("maxpool: argmax must be uint8 with out's shape")
    with _Span("maxpool", 0, _nbytes(x, out, argmax)):
        check(_lib.load().rvit_maxpool3x3s2_nhwc(x.data_ptr(), out.data_ptr(), _ptr(argmax), B, H, W, C, _stream()),
              "rvit_maxpool3x3s2_nhwc")
    return out


def embed_cls(cls_token: torch.Tensor, pos_embed: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """x[b,0,:] = cls_token + pos_embed[0]; x fp32 [B,S,E] contiguous."""
    _need_cuda(cls_token, pos_embed, x)
    B, S, E = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("embed_cls: x must be contiguous float32 [B,S,E]")
    if cls_token.dtype != torch.float32 or pos_embed.dtype != torch.float32 or cls_token.numel() != E \
            or pos_embed.numel() < E or not pos_embed.is_contiguous() or not cls_token.is_contiguous():
        raise ValueError("embed_cls: cls/pos must be contiguous float32")
    with _Span("embed_cls", 0, B * E * 4):
        check(_lib.load().rvit_embed_cls(cls_token.data_ptr(), pos_embed.data_ptr(), x.data_ptr(), B, S, E,
                                         _stream()), "rvit_embed_cls")
    return x


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, eps: float = 1e-5,
              mean: Optional[torch.Tensor] = None, rstd: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row LayerNorm; x [rows,E] (bf16/f32, row stride free), out [rows,E] (bf16/f32)."""
    _need_cuda(x, gamma, beta, out, mean, rstd)
    rows, E, ldx = _rows2d(x, "layernorm x")
    ro, Eo, ldy = _rows2d(out, "layernorm out")
    if ro != rows or Eo != E or gamma.numel() != E or beta.numel() != E:
        raise ValueError("layernorm: shape mismatch")
    if gamma.dtype != torch.float32 or beta.dtype != torch.float32:
        raise TypeError("layernorm: gamma/beta must be float32")
    for s in (mean, rstd):
        if s is not None and (s.dtype != torch.float32 or s.numel() != rows or not s.is_contiguous()):
            raise ValueError("layernorm: mean/rstd must be contiguous float32 [rows]")
    with _Span("layernorm_fwd", 0, rows * E * (x.element_size() + out.element_size()) + 8 * E):
        check(_lib.load().rvit_layernorm_fwd(x.data_ptr(), _dt(x), ldx, gamma.data_ptr(), beta.data_ptr(),
                                             out.data_ptr(), _dt(out), ldy, _ptr(mean), _ptr(rstd), rows, E, eps,
                                             _stream()), "rvit_layernorm_fwd")
    return out


def attention(qkv: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int, lse: Optional[torch.Tensor] = None,
              scale: Optional[float] = None) -> torch.Tensor:
    """qkv bf16 [B*S, 3*H*64] (q|k|v), out bf16 [B*S, H*64]; softmax(q k^T scale) v per head."""
    _need_cuda(qkv, out, lse)
    T, W3, ldq = _rows2d(qkv, "attention qkv")
    To, Wo, ldo = _rows2d(out, "attention out")
    if T != B * S or To != T or W3 != 3 * H * 64 or Wo != H * 64:
        raise ValueError(f"attention: shapes {tuple(qkv.shape)} / {tuple(out.shape)} do not match B={B} S={S} H={H}")
    if qkv.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
        raise TypeError("attention: bf16 only")
    if lse is not None and (lse.dtype != torch.float32 or lse.numel() != B * H * S or not lse.is_contiguous()):
        raise ValueError("attention: lse must be contiguous float32 [B,H,S]")
    sc = (1.0 / 8.0) if scale is None else float(scale)
    with _Span("attn_fwd", 4 * B * H * S * S * 64, _nbytes(qkv, out, lse)):
        check(_lib.load().rvit_attn_fwd(qkv.data_ptr(), out.data_ptr(), _ptr(lse), B, S, H, ldq, ldo, sc, _stream()),
              "rvit_attn_fwd")
    return out
