"""Tensor-level wrappers over the librvit C ABI.

Each function enqueues one HIP kernel on torch's current stream.  Tensors must already live on
the GPU; shapes/dtypes are validated here and again inside the library.  No function in this
module has a non-HIP implementation.
"""
from __future__ import annotations

import ctypes
import functools

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_RELU, GATE_GELU_GRAD, GATE_RELU_MASK, RVIT_BF16, RVIT_F32,  # noqa: F401
                   check)

_ACT = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "gelu": ACT_GELU, "hswish": _lib.ACT_HSWISH,
        "hsigmoid": _lib.ACT_HSIGMOID}
_GATE = {None: 0, "gelu_grad": GATE_GELU_GRAD, "relu_mask": GATE_RELU_MASK}
BF16 = torch.bfloat16

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
    if _PROF is None:   # only the profiler reads it
        return 0
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream
    object through several Python layers (~8 us, ~400 times per training step); the raw getters are two C calls
    (every caller has CUDA tensors in hand, so the runtime is initialised: torch.cuda.current_device()'s lazy-init
    check, three Python frames per call, is not needed)."""
    if _raw_stream is not None:
        return _raw_stream(_raw_device() if _raw_device is not None else torch.cuda.current_device())
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


_DROP_POS = {"pre_residual": _lib.DROP_PRE_RESIDUAL, "post_act": _lib.DROP_POST_ACT}


def _drop(drop):
    """(position, p, seed) -> (mode, p, seed) for the C ABI; None or p == 0 disables."""
    if drop is None or drop[1] == 0.0:
        return 0, 0.0, 0
    pos, p, seed = drop
    if pos not in _DROP_POS or not (0.0 <= p < 1.0):
        raise ValueError(f"dropout: bad spec {drop!r}")
    return _DROP_POS[pos], float(p), int(seed) & 0xFFFFFFFFFFFFFFFF


# Ordered (bit-reproducible) reductions: a scratch workspace handed to every entry point that would otherwise end a
# split reduction in fp32 atomics (include/rvit.h "Ordered reductions").  One workspace per (device, STREAM): the
# producers of a launch write their slots and the next launch on the same stream reads them, so calls that share a
# workspace must be ordered on one stream - keying it by the current stream makes that true by construction (two
# engines, or direct ops callers, on different streams of one GPU no longer share slots: ADVICE r2), and switching it
# off only affects the caller's own stream.
ORDERED_WORKSPACE_BYTES = 192 << 20
_ORDERED_WS: dict = {}


def _ows_key(device) -> tuple:
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        return (idx, _stream())


def enable_ordered_reductions(device, enable: bool = True, nbytes: int = ORDERED_WORKSPACE_BYTES) -> None:
    key = _ows_key(device)
    if not enable:
        _ORDERED_WS.pop(key, None)
    elif key not in _ORDERED_WS or _ORDERED_WS[key].numel() < nbytes:
        _ORDERED_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", key[0]))


def ordered_reductions_enabled(device) -> bool:
    return _ows_key(device) in _ORDERED_WS


def _ows(t: torch.Tensor):
    """(pointer, bytes) of the ordered-reduction workspace of t's device and the current stream, or (None, 0)."""
    w = _ORDERED_WS.get((t.device.index, _stream()))
    return (w.data_ptr(), w.numel()) if w is not None else (None, 0)


def _vec_f32(v, n, what):
    if v is not None and (v.dtype != torch.float32 or v.numel() != n or not v.is_contiguous()):
        raise ValueError(f"{what} must be contiguous float32 [{n}]")


# =================================================================================================
# forward kernels
# =================================================================================================
def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
         scale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, residual_period: int = 0,
         act: Optional[str] = None, preact: Optional[torch.Tensor] = None, row_group: int = 0, row_extra: int = 0,
         gate: Optional[torch.Tensor] = None, gate_mode: Optional[str] = None, a_rows: Optional[int] = None,
         a_row_group: int = 0, a_row_extra: int = 0, drop: Optional[Tuple[str, float, int]] = None,
         tile: int = 0) -> torch.Tensor:
    """out = act(gate(scale * (a @ w.T) + bias) + residual); a [M,K] bf16, w [N,K] bf16.

    a_rows/a_row_group/a_row_extra: use `a_rows` logical rows of `a` through the row remap.
    drop = (position, p, seed): dropout in the epilogue, position "pre_residual" | "post_act"."""
    _need_cuda(a, w, out, bias, scale, residual, preact, gate)
    Ma, K, lda = _rows2d(a, "gemm a")
    M = Ma if a_rows is None else a_rows
    if a_row_group and (M // a_row_group) * (a_row_group + a_row_extra) > Ma:
        raise ValueError("gemm: remapped A rows exceed the tensor")
    N, K2, ldb = _rows2d(w, "gemm w")
    if K != K2:
        raise ValueError(f"gemm: K mismatch {K} vs {K2}")
    if a.dtype != BF16 or w.dtype != BF16:
        raise TypeError("gemm: a and w must be bfloat16")
    mo, no, ldc = _rows2d(out, "gemm out")
    rows_out = M if row_group == 0 else (M // row_group) * (row_group + row_extra)
    if no != N or mo < rows_out:
        raise ValueError(f"gemm: out shape {tuple(out.shape)} incompatible with M={M} N={N}")
    ldr, r_dt = 0, RVIT_F32
    if residual is not None:
        rr, rn, ldr = _rows2d(residual, "gemm residual")
        if rn != N or rr < (residual_period if residual_period else rows_out):
            raise ValueError("gemm: residual shape mismatch")
        r_dt = _dt(residual)
    ldp = 0
    if preact is not None:
        pr, pn, ldp = _rows2d(preact, "gemm preact")
        if pn != N or pr < rows_out or preact.dtype != BF16:
            raise ValueError("gemm: preact must be bf16 [rows, N]")
    ldg = 0
    if gate is not None:
        gr, gn, ldg = _rows2d(gate, "gemm gate")
        if gn != N or gr < rows_out or gate.dtype != BF16 or gate_mode not in ("gelu_grad", "relu_mask"):
            raise ValueError("gemm: gate must be bf16 [rows, N] with gate_mode gelu_grad|relu_mask")
    _vec_f32(bias, N, "gemm: bias")
    _vec_f32(scale, N, "gemm: scale")
    d_mode, d_p, d_seed = _drop(drop)
    lib = _lib.load()
    nb = (M * K + N * K) * 2 + M * N * out.element_size() + (M * N * residual.element_size() if residual is not None
                                                             and not residual_period else 0)
    with _Span("gemm_bf16", 2 * M * N * K, nb):
        check(lib.rvit_gemm_bf16(a.data_ptr(), lda, a_row_group, a_row_extra, w.data_ptr(), ldb, out.data_ptr(), ldc,
                                 _dt(out), M, N, K, _ptr(scale), _ptr(bias), _ptr(gate), ldg, _GATE[gate_mode],
                                 _ptr(residual), ldr, r_dt, residual_period, _ACT[act], _ptr(preact), ldp, row_group,
                                 row_extra, d_mode, d_p, d_seed, tile, _stream()), "rvit_gemm_bf16")
    return out


def mx_scale_shape(rows: int, K: int):
    """Shape of the tiled E8M0 scale tensor of a [rows, K] MX-fp8 matrix (include/rvit.h)."""
    return ((rows + 15) // 16, (K + 511) // 512, 16, 4, 4)


def quantize_mx(x: torch.Tensor, q: Optional[torch.Tensor] = None, scales: Optional[torch.Tensor] = None):
    """MXFP8: x bf16/f32 [rows, K] -> (q uint8 [rows, K] of e4m3 bytes, scales uint8 [ceil(rows/16), ceil(K/512), 16, 4, 4])."""
    _need_cuda(x, q, scales)
    rows, K, ldx = _rows2d(x, "quantize_mx x")
    if K % 128 or x.dtype not in (BF16, torch.float32):
        raise ValueError("quantize_mx: K must be a multiple of 128, x bf16 or f32")
    if q is None:
        q = torch.empty(rows, K, dtype=torch.uint8, device=x.device)
    if scales is None:
        scales = torch.zeros(mx_scale_shape(rows, K), dtype=torch.uint8, device=x.device)
    if q.shape != (rows, K) or tuple(scales.shape) != mx_scale_shape(rows, K) or q.dtype != torch.uint8 \
            or scales.dtype != torch.uint8 or not scales.is_contiguous():
        raise ValueError("quantize_mx: q [rows,K] / tiled scales (ops.mx_scale_shape) must be uint8")
    with _Span("quantize_mx", 0, rows * K * (x.element_size() + 1)):
        check(_lib.load().rvit_quantize_mxfp8(x.data_ptr(), _dt(x), ldx, q.data_ptr(), q.stride(0), scales.data_ptr(),
                                              rows, K, _stream()), "rvit_quantize_mxfp8")
    return q, scales


def gemm_mx(a_q: torch.Tensor, a_s: torch.Tensor, w_q: torch.Tensor, w_s: torch.Tensor, out: Optional[torch.Tensor], *,
            bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act: Optional[str] = None,
            out_q: Optional[torch.Tensor] = None, out_s: Optional[torch.Tensor] = None):
    """out = act(dequant(a) @ dequant(w).T + bias) + residual on the MX-fp8 MFMA; optional MX copy (out_q, out_s)."""
    _need_cuda(a_q, a_s, w_q, w_s, out, bias, residual, out_q, out_s)
    M, K, lda = _rows2d(a_q, "gemm_mx a")
    N, K2, ldb = _rows2d(w_q, "gemm_mx w")
    if K != K2 or K % 128 or any(t.dtype != torch.uint8 for t in (a_q, a_s, w_q, w_s)) \
            or tuple(a_s.shape) != mx_scale_shape(M, K) or tuple(w_s.shape) != mx_scale_shape(N, K) \
            or not (a_s.is_contiguous() and w_s.is_contiguous()):
        raise ValueError("gemm_mx: operands must be uint8 [rows,K] (K % 128 == 0) with tiled uint8 scales")
    if out is None and out_q is None:
        raise ValueError("gemm_mx: no output")
    ldc, c_dt = 0, RVIT_BF16
    if out is not None:
        mo, no, ldc = _rows2d(out, "gemm_mx out")
        if (mo, no) != (M, N):
            raise ValueError("gemm_mx: out shape mismatch")
        c_dt = _dt(out)
    ldr, r_dt = 0, RVIT_F32
    if residual is not None:
        rr, rn, ldr = _rows2d(residual, "gemm_mx residual")
        if (rr, rn) != (M, N):
            raise ValueError("gemm_mx: residual shape mismatch")
        r_dt = _dt(residual)
    if (out_q is None) != (out_s is None) or (out_q is not None and (
            N % 128 or out_q.shape != (M, N) or tuple(out_s.shape) != mx_scale_shape(M, N)
            or out_q.dtype != torch.uint8 or out_s.dtype != torch.uint8 or not out_s.is_contiguous())):
        raise ValueError("gemm_mx: out_q [M,N] / tiled out_s must be uint8, N % 128 == 0, and come together")
    _vec_f32(bias, N, "gemm_mx: bias")
    with _Span("gemm_mxfp8", 2 * M * N * K, M * K + N * K + M * N * (out.element_size() if out is not None else 1)):
        check(_lib.load().rvit_gemm_mxfp8(a_q.data_ptr(), lda, a_s.data_ptr(), w_q.data_ptr(), ldb, w_s.data_ptr(),
                                          _ptr(out), ldc, c_dt, M, N, K, _ptr(bias), _ptr(residual), ldr, r_dt,
                                          _ACT[act], _ptr(out_q), out_q.stride(0) if out_q is not None else 0,
                                          _ptr(out_s), _stream()), "rvit_gemm_mxfp8")
    return out if out is not None else out_q


@functools.lru_cache(maxsize=256)
def _conv3x3_name(B: int, H: int, W: int, Cin: int, Cout: int) -> str:
    code = _lib.load().rvit_conv3x3_kernel(B, H, W, Cin, Cout)
    if code == 4:    # conv3x3_p8_kernel<256> / <128> are two kernels (rocprofv3 lists them apart): so do the spans
        return "conv3x3_p8_256" if Cout % 256 == 0 else "conv3x3_p8_128"
    return ("conv2d_gather", "conv3x3_direct", "conv3x3_c64", "conv3x3_c256")[code]


def conv2d(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, kh: int, kw: int, stride: int, pad: int,
           seg_len: Optional[int] = None, scale: Optional[torch.Tensor] = None,
           bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           act: Optional[str] = None, tile: int = 0) -> torch.Tensor:
    """NHWC implicit-GEMM conv: x [B,H,W,Cpix] bf16, w [Cout, kh*kw*seg_len] bf16, out [B,Ho,Wo,Cout] bf16."""
    _need_cuda(x, w, out, scale, bias, residual)
    if x.dim() != 4 or out.dim() != 4 or not x.is_contiguous() or not out.is_contiguous() or not w.is_contiguous():
        raise ValueError("conv2d: x/out must be contiguous NHWC 4-D tensors, w contiguous")
    if x.dtype != BF16 or w.dtype != BF16 or out.dtype != BF16:
        raise TypeError("conv2d: bf16 only")
    B, H, W, Cpix = x.shape
    Bo, Ho, Wo, Cout = out.shape
    seg = Cpix if seg_len is None else seg_len
    if Bo != B or w.shape != (Cout, kh * kw * seg):
        raise ValueError(f"conv2d: weight shape {tuple(w.shape)} != {(Cout, kh * kw * seg)} or batch mismatch")
    if residual is not None and (residual.shape != out.shape or residual.dtype != BF16 or not residual.is_contiguous()):
        raise ValueError("conv2d: residual must match out (bf16, contiguous)")
    _vec_f32(bias, Cout, "conv2d: bias")
    _vec_f32(scale, Cout, "conv2d: scale")
    lib = _lib.load()
    # algorithmic work: true taps only (the 7x7 stem conv's zero taps / 4th channel are padding)
    true_k = kh * kw * seg if seg_len is None else kh * (kw * seg - seg // 8 - (seg - seg // 8) // 4)
    # profile the conv kernels under separate names: the library says which one tile 0 runs for a 3x3 / stride 1 /
    # pad 1 geometry (rvit_conv3x3_kernel); a forced tile names its kernel itself
    name = "conv2d_gather"
    if kh == 3 and kw == 3 and stride == 1 and pad == 1 and seg_len is None:
        if tile == 0:
            name = _conv3x3_name(B, H, W, Cpix, Cout)
        elif 10 <= tile <= 19:
            name = {14: "conv3x3_c64", 15: "conv3x3_c256", 18: "conv3x3_c256",
                    19: "conv3x3_p8_256" if Cout % 256 == 0 else "conv3x3_p8_128"}.get(tile, "conv3x3_direct")
    with _Span(name, 2 * B * Ho * Wo * Cout * true_k,
               _nbytes(x, w, out, residual)):
        check(lib.rvit_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), out.data_ptr(), B, H, W, Cpix, seg, kh, kw, stride,
                                       pad, Ho, Wo, Cout, _ptr(scale), _ptr(bias), _ptr(residual), _ACT[act], tile,
                                       _stream()), "rvit_conv2d_fwd_bf16")
    return out


STAT_STAGE_FLOATS = 256 * 2 * 2048   # first-stage scratch of rvit_bn_stats_finalize (C <= 2048)


def conv_stats_workspace(device, mbytes: int = 32) -> torch.Tensor:
    """Workspace of conv2d_bn_stats: per-tile statistics slots followed by the finalisation's staging area."""
    return torch.empty(mbytes * (1 << 18) + STAT_STAGE_FLOATS, dtype=torch.float32, device=device)


def stem_conv1_stats_f32(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, sums: torch.Tensor,
                         stat_ws: torch.Tensor) -> torch.Tensor:
    """The stem's 7x7 / stride-2 conv of a training step straight from the fp32 NCHW input, with the batch statistics of
    the stored map: x f32 [B,3,224,224], w bf16 [64,224], out bf16 [B,112,112,64], sums f32 [128] += (sum | sumsq)."""
    _need_cuda(x, w, out, sums, stat_ws)
    B = x.shape[0]
    if x.shape != (B, 3, 224, 224) or x.dtype != torch.float32 or not x.is_contiguous() or w.shape != (64, 224) \
            or w.dtype != BF16 or not w.is_contiguous() or out.shape != (B, 112, 112, 64) or out.dtype != BF16 \
            or not out.is_contiguous():
        raise ValueError("stem_conv1_stats_f32: bad operands")
    _vec_f32(sums, 128, "stem_conv1_stats_f32: sums")
    if stat_ws.dtype != torch.float32 or not stat_ws.is_contiguous() or stat_ws.numel() <= STAT_STAGE_FLOATS:
        raise ValueError("stem_conv1_stats_f32: stat_ws must come from ops.conv_stats_workspace()")
    lib = _lib.load()
    slots = ctypes.c_int(0)
    slot_floats = stat_ws.numel() - STAT_STAGE_FLOATS   # slots first, the finalisation's staging area behind them
    with _Span("conv2d_gather", 2 * B * 112 * 112 * 64 * 147, _nbytes(x, w, out)):
        check(lib.rvit_stem_conv1_stats_f32_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), stat_ws.data_ptr(), slot_floats * 4,
                                                ctypes.byref(slots), B, _stream()), "rvit_stem_conv1_stats_f32_fwd")
    with _Span("bn_stats_finalize", 0, slots.value * 128 * 4):
        check(lib.rvit_bn_stats_finalize(stat_ws.data_ptr(), slots.value, 64, sums.data_ptr(),
                                         stat_ws.data_ptr() + slot_floats * 4, _stream()), "rvit_bn_stats_finalize")
    return out


def stem_conv1_wgrad_f32(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor) -> torch.Tensor:
    """dw f32 [64,224] += weight gradient of the stem's 7x7 / stride-2 conv from the fp32 NCHW input x [B,3,224,224] and
    dy bf16 [B,112,112,64] (column kh*32 + px*4 + ch, as conv2d_wgrad with seg_len 32 on the packed image).  Needs the
    ordered-reduction workspace of the current stream (enable_ordered_reductions)."""
    _need_cuda(dy, x, dw)
    B = x.shape[0]
    if x.shape != (B, 3, 224, 224) or x.dtype != torch.float32 or not x.is_contiguous() or dy.shape != (B, 112, 112, 64) \
            or dy.dtype != BF16 or not dy.is_contiguous() or dw.shape != (64, 224) or dw.dtype != torch.float32 \
            or not dw.is_contiguous():
        raise ValueError("stem_conv1_wgrad_f32: bad operands")
    ws, ws_bytes = _ows(dw)
    if ws is None:
        raise ValueError("stem_conv1_wgrad_f32: needs ops.enable_ordered_reductions() on this stream")
    with _Span("conv2d_wgrad", 2 * B * 112 * 112 * 64 * 224, _nbytes(dy, x, dw)):
        check(_lib.load().rvit_stem_conv1_wgrad_f32(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws, ws_bytes, B, _stream()),
              "rvit_stem_conv1_wgrad_f32")
    return dw


def conv2d_bn_stats(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, sums: torch.Tensor, stat_ws: torch.Tensor, *,
                    kh: int, kw: int, stride: int, pad: int, seg_len: Optional[int] = None,
                    bn_workspace: Optional[torch.Tensor] = None, tile: int = 0) -> bool:
    """conv2d (no scale / bias / residual / activation) + bn_stats of its output: sums[0:C] += sum out, sums[C:2C] +=
    sum out^2.  Where the conv kernel has the statistics epilogue (rvit_conv2d_fwd_stats_bf16) the sums come from
    per-tile slots written by the conv itself and the pass over the stored map disappears; otherwise bn_stats runs.
    Returns True in the first case."""
    _need_cuda(x, w, out, sums, stat_ws, bn_workspace)
    if x.dim() != 4 or out.dim() != 4 or not x.is_contiguous() or not out.is_contiguous() or not w.is_contiguous():
        raise ValueError("conv2d_bn_stats: x/out must be contiguous NHWC 4-D tensors, w contiguous")
    if x.dtype != BF16 or w.dtype != BF16 or out.dtype != BF16:
        raise TypeError("conv2d_bn_stats: bf16 only")
    B, H, W, Cpix = x.shape
    Bo, Ho, Wo, Cout = out.shape
    seg = Cpix if seg_len is None else seg_len
    if Bo != B or w.shape != (Cout, kh * kw * seg):
        raise ValueError(f"conv2d_bn_stats: weight shape {tuple(w.shape)} != {(Cout, kh * kw * seg)} or batch mismatch")
    _vec_f32(sums, 2 * Cout, "conv2d_bn_stats: sums")
    if stat_ws.dtype != torch.float32 or not stat_ws.is_contiguous() or stat_ws.numel() <= STAT_STAGE_FLOATS:
        raise ValueError("conv2d_bn_stats: stat_ws must come from ops.conv_stats_workspace()")
    lib = _lib.load()
    true_k = kh * kw * seg if seg_len is None else kh * (kw * seg - seg // 8 - (seg - seg // 8) // 4)
    name = "conv2d_gather"
    if kh == 3 and kw == 3 and stride == 1 and pad == 1 and seg_len is None and tile == 0:
        name = _conv3x3_name(B, H, W, Cpix, Cout)
    slots = ctypes.c_int(0)
    slot_floats = stat_ws.numel() - STAT_STAGE_FLOATS
    with _Span(name, 2 * B * Ho * Wo * Cout * true_k, _nbytes(x, w, out)):
        check(lib.rvit_conv2d_fwd_stats_bf16(x.data_ptr(), w.data_ptr(), out.data_ptr(), B, H, W, Cpix, seg, kh, kw, stride,
                                             pad, Ho, Wo, Cout, stat_ws.data_ptr(), slot_floats * 4, ctypes.byref(slots),
                                             tile, _stream()), "rvit_conv2d_fwd_stats_bf16")
    if slots.value > 0:
        with _Span("bn_stats_finalize", 0, slots.value * 2 * Cout * 4):
            check(lib.rvit_bn_stats_finalize(stat_ws.data_ptr(), slots.value, Cout, sums.data_ptr(),
                                             stat_ws.data_ptr() + slot_floats * 4, _stream()), "rvit_bn_stats_finalize")
        return True
    bn_stats(out, sums, bn_workspace)
    return False


def conv2d_tokens(x: torch.Tensor, w: torch.Tensor, tokens: torch.Tensor, pos_embed: torch.Tensor, *, kh: int,
                  kw: int, stride: int, pad: int, seg_len: int, ho: int, wo: int,
                  bias: Optional[torch.Tensor] = None, drop_p: float = 0.0, drop_seed: int = 0,
                  tile: int = 0) -> torch.Tensor:
    """Patch conv into the fp32 token stream: tokens[b,1+p] = drop(conv(x,w)[b,p] + bias + pos_embed[1+p])."""
    _need_cuda(x, w, tokens, pos_embed, bias)
    B, H, W, Cpix = x.shape
    Bt, S, E = tokens.shape
    if Bt != B or S != ho * wo + 1 or tokens.dtype != torch.float32 or not tokens.is_contiguous():
        raise ValueError("conv2d_tokens: tokens must be contiguous float32 [B, ho*wo+1, E]")
    if w.shape != (E, kh * kw * seg_len) or w.dtype != BF16 or x.dtype != BF16 or not w.is_contiguous() \
            or not x.is_contiguous():
        raise ValueError("conv2d_tokens: bad x / w")
    if pos_embed.dtype != torch.float32 or pos_embed.numel() != S * E or not pos_embed.is_contiguous():
        raise ValueError("conv2d_tokens: pos_embed must be contiguous float32 [S,E]")
    _vec_f32(bias, E, "conv2d_tokens: bias")
    true_k = kh * (seg_len // 4) * 3  # px4 runs carry 3 real channels
    with _Span("conv2d_tokens", 2 * B * ho * wo * E * true_k, _nbytes(x, w, tokens)):
        check(_lib.load().rvit_conv2d_tokens_fwd(x.data_ptr(), w.data_ptr(), tokens.data_ptr(), B, H, W, Cpix,
                                                 seg_len, kh, kw, stride, pad, ho, wo, E, _ptr(bias),
                                                 pos_embed.data_ptr(), float(drop_p), int(drop_seed), tile, _stream()),
              "rvit_conv2d_tokens_fwd")
    return tokens


def stem_conv1_pool(img: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, pooled: torch.Tensor,
                    relu: bool = True) -> torch.Tensor:
    """conv7x7/2 + folded BN + ReLU + maxpool3x3/2 fused: img bf16 [B,230,232,4] -> pooled bf16 [B,56,56,64]."""
    _need_cuda(img, w, scale, bias, pooled)
    B = img.shape[0]
    if img.shape != (B, 230, 232, 4) or pooled.shape != (B, 56, 56, 64) or w.shape != (64, 232) \
            or any(t.dtype != BF16 for t in (img, w, pooled)) or not (img.is_contiguous() and w.is_contiguous()
                                                                      and pooled.is_contiguous()):
        raise ValueError("stem_conv1_pool: img [B,230,232,4], w [64,232], pooled [B,56,56,64], all bf16 contiguous")
    _vec_f32(scale, 64, "stem_conv1_pool: scale")
    _vec_f32(bias, 64, "stem_conv1_pool: bias")
    with _Span("stem_conv1_pool", 2 * B * 112 * 112 * 64 * 147, _nbytes(img, w, pooled)):
        check(_lib.load().rvit_stem_conv1_pool_fwd(img.data_ptr(), w.data_ptr(), scale.data_ptr(), bias.data_ptr(),
                                                   pooled.data_ptr(), B, int(relu), _stream()),
              "rvit_stem_conv1_pool_fwd")
    return pooled


def stem_conv1_pool_f32(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, pooled: torch.Tensor,
                        relu: bool = True) -> torch.Tensor:
    """stem_conv1_pool straight from the fp32 NCHW input x [B,3,224,224]: no packed copy of the image in HBM."""
    _need_cuda(x, w, scale, bias, pooled)
    B = x.shape[0]
    if tuple(x.shape) != (B, 3, 224, 224) or x.dtype != torch.float32 or not x.is_contiguous() \
            or pooled.shape != (B, 56, 56, 64) or w.shape != (64, 232) or w.dtype != BF16 or pooled.dtype != BF16 \
            or not (w.is_contiguous() and pooled.is_contiguous()):
        raise ValueError("stem_conv1_pool_f32: x f32 [B,3,224,224], w bf16 [64,232], pooled bf16 [B,56,56,64], contiguous")
    _vec_f32(scale, 64, "stem_conv1_pool_f32: scale")
    _vec_f32(bias, 64, "stem_conv1_pool_f32: bias")
    with _Span("stem_conv1_pool", 2 * B * 112 * 112 * 64 * 147, _nbytes(x, w, pooled)):
        check(_lib.load().rvit_stem_conv1_pool_f32_fwd(x.data_ptr(), w.data_ptr(), scale.data_ptr(), bias.data_ptr(),
                                                       pooled.data_ptr(), B, int(relu), _stream()),
              "rvit_stem_conv1_pool_f32_fwd")
    return pooled


def input_pack(x: torch.Tensor, out: torch.Tensor, pad_top: int, pad_left: int) -> torch.Tensor:
    """x fp32 [B,C<=3,H,W] -> interior of the zero-haloed bf16 NHWC4 buffer out [B,Hp,Wp,4]."""
    _need_cuda(x, out)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4:
        raise ValueError("input_pack: x must be contiguous float32 NCHW")
    if out.dtype != BF16 or not out.is_contiguous() or out.dim() != 4 or out.shape[3] != 4:
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
    if tuple(out.shape) != (B, Ho, Wo, C) or x.dtype != BF16 or out.dtype != BF16:
        raise ValueError(f"maxpool: out must be bf16 {(B, Ho, Wo, C)}")
    if not x.is_contiguous() or not out.is_contiguous():
        raise ValueError("maxpool: contiguous NHWC required")
    if argmax is not None and (argmax.dtype != torch.uint8 or argmax.shape != out.shape or not argmax.is_contiguous()):
        raise ValueError("maxpool: argmax must be uint8 with out's shape")
    with _Span("maxpool", 0, _nbytes(x, out, argmax)):
        check(_lib.load().rvit_maxpool3x3s2_nhwc(x.data_ptr(), out.data_ptr(), _ptr(argmax), B, H, W, C, _stream()),
              "rvit_maxpool3x3s2_nhwc")
    return out


def embed_cls(cls_token: torch.Tensor, pos_embed: torch.Tensor, x: torch.Tensor, drop_p: float = 0.0,
              drop_seed: int = 0) -> torch.Tensor:
    """x[b,0,:] = drop(cls_token + pos_embed[0]); x fp32 [B,S,E] contiguous."""
    _need_cuda(cls_token, pos_embed, x)
    B, S, E = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("embed_cls: x must be contiguous float32 [B,S,E]")
    if cls_token.dtype != torch.float32 or pos_embed.dtype != torch.float32 or cls_token.numel() != E \
            or pos_embed.numel() < E or not pos_embed.is_contiguous() or not cls_token.is_contiguous():
        raise ValueError("embed_cls: cls/pos must be contiguous float32")
    with _Span("embed_cls", 0, B * E * 4):
        check(_lib.load().rvit_embed_cls(cls_token.data_ptr(), pos_embed.data_ptr(), x.data_ptr(), B, S, E,
                                         float(drop_p), int(drop_seed), _stream()), "rvit_embed_cls")
    return x


def dropout_apply(x: torch.Tensor, out: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    """out = x * mask(p, seed) / (1-p), element index = row * cols + col; bf16 or f32 [rows, cols]."""
    _need_cuda(x, out)
    rows, cols, ld_in = _rows2d(x, "dropout_apply x")
    ro, co, ld_out = _rows2d(out, "dropout_apply out")
    if (ro, co) != (rows, cols) or x.dtype != out.dtype or x.dtype not in (BF16, torch.float32):
        raise ValueError("dropout_apply: shape/dtype mismatch")
    with _Span("dropout_apply", 0, 2 * rows * cols * x.element_size()):
        check(_lib.load().rvit_dropout_apply(x.data_ptr(), out.data_ptr(), _dt(x), rows, cols, ld_in, ld_out, float(p),
                                             int(seed), _stream()), "rvit_dropout_apply")
    return out


def dropout_mask(n: int, p: float, seed: int, device) -> torch.Tensor:
    """uint8 [n]: 1 where element i of the dropout site (p, seed) is kept (test/debug aid)."""
    out = torch.empty(n, dtype=torch.uint8, device=device)
    _need_cuda(out)
    check(_lib.load().rvit_dropout_mask(out.data_ptr(), n, float(p), int(seed), _stream()), "rvit_dropout_mask")
    return out


def attention_dropout_mask(B: int, H: int, S: int, p: float, seed: int, device) -> torch.Tensor:
    """uint8 [B, H, S, S]: keep mask of the attention-probability dropout site (p, seed).  The kernels index that site
    as (pair * S + query) * S2 + key with the row stride S2 = S rounded up to even, so that keys (2j, 2j + 1) of a
    query share one hash (csrc/attn.hip); this view drops the padding column (test/debug aid)."""
    s2 = S + (S & 1)
    return dropout_mask(B * H * S * s2, p, seed, device).view(B, H, S, s2)[..., :S]


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, eps: float = 1e-5,
              mean: Optional[torch.Tensor] = None, rstd: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row LayerNorm; x [rows,E] (bf16/f32, row stride free), out [rows,E] (bf16/f32)."""
    _need_cuda(x, gamma, beta, out, mean, rstd)
    rows, E, ldx = _rows2d(x, "layernorm x")
    ro, Eo, ldy = _rows2d(out, "layernorm out")
    if ro != rows or Eo != E:
        raise ValueError("layernorm: shape mismatch")
    _vec_f32(gamma, E, "layernorm: gamma")
    _vec_f32(beta, E, "layernorm: beta")
    _vec_f32(mean, rows, "layernorm: mean")
    _vec_f32(rstd, rows, "layernorm: rstd")
    with _Span("layernorm_fwd", 0, rows * E * (x.element_size() + out.element_size()) + 8 * E):
        check(_lib.load().rvit_layernorm_fwd(x.data_ptr(), _dt(x), ldx, gamma.data_ptr(), beta.data_ptr(),
                                             out.data_ptr(), _dt(out), ldy, _ptr(mean), _ptr(rstd), rows, E, eps,
                                             _stream()), "rvit_layernorm_fwd")
    return out


def layernorm_mx(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, q: torch.Tensor, scales: torch.Tensor,
                 eps: float = 1e-5):
    """Row LayerNorm with MXFP8 output (== layernorm to bf16, then quantize_mx, in one pass): x [rows,E] bf16/f32,
    q uint8 [rows,E] e4m3 bytes, scales uint8 tiled (ops.mx_scale_shape(rows, E)); E % 128 == 0."""
    _need_cuda(x, gamma, beta, q, scales)
    rows, E, ldx = _rows2d(x, "layernorm_mx x")
    if E % 128 or x.dtype not in (BF16, torch.float32):
        raise ValueError("layernorm_mx: E must be a multiple of 128, x bf16 or f32")
    if q.shape != (rows, E) or q.dtype != torch.uint8 or q.stride(1) != 1 or tuple(scales.shape) != mx_scale_shape(rows, E) \
            or scales.dtype != torch.uint8 or not scales.is_contiguous():
        raise ValueError("layernorm_mx: q [rows,E] / tiled scales (ops.mx_scale_shape) must be uint8")
    _vec_f32(gamma, E, "layernorm_mx: gamma")
    _vec_f32(beta, E, "layernorm_mx: beta")
    with _Span("layernorm_mx_fwd", 0, rows * E * (x.element_size() + 1) + 8 * E):
        check(_lib.load().rvit_layernorm_mxfp8_fwd(x.data_ptr(), _dt(x), ldx, gamma.data_ptr(), beta.data_ptr(),
                                                   q.data_ptr(), q.stride(0), scales.data_ptr(), rows, E, eps, _stream()),
              "rvit_layernorm_mxfp8_fwd")
    return q, scales


def cls_head_supported(E: int, hidden: int, num_classes: int) -> bool:
    return bool(_lib.load().rvit_cls_head_supported(int(E), int(hidden), int(num_classes)))


def cls_head(x_rows: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, w0: torch.Tensor, b0: torch.Tensor,
             w3: torch.Tensor, b3: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """logits = w3 . gelu(w0 . LayerNorm(x_rows) + b0) + b3 in one launch (the bits of layernorm -> gemm(gelu) -> gemm):
    x_rows fp32 [B, E] (any row pitch), w0 bf16 [hidden, E], w3 bf16 [classes, hidden], logits fp32 [B, classes]."""
    _need_cuda(x_rows, gamma, beta, w0, b0, w3, b3, logits)
    B, E, ldx = _rows2d(x_rows, "cls_head x")
    hidden, C = w0.shape[0], w3.shape[0]
    if x_rows.dtype != torch.float32 or logits.dtype != torch.float32 or w0.dtype != BF16 or w3.dtype != BF16 \
            or w0.shape != (hidden, E) or w3.shape != (C, hidden) or logits.shape != (B, C) or logits.stride(1) != 1 \
            or not (w0.is_contiguous() and w3.is_contiguous()):
        raise ValueError("cls_head: shape / dtype mismatch")
    for v, n in ((gamma, E), (beta, E), (b0, hidden), (b3, C)):
        _vec_f32(v, n, "cls_head: vector")
    with _Span("cls_head", 2 * B * hidden * (E + C), _nbytes(x_rows, w0, w3, logits)):
        check(_lib.load().rvit_cls_head_fwd(x_rows.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                            w0.data_ptr(), b0.data_ptr(), w3.data_ptr(), b3.data_ptr(), logits.data_ptr(),
                                            logits.stride(0), B, E, hidden, C, _stream()), "rvit_cls_head_fwd")
    return logits


ATTN_HEAD_DIMS = (16, 32, 48, 64, 96, 128)


def _attn_head_dim(W3: int, Wo: int, H: int, what: str) -> int:
    """Head dim from the packed widths; 64 runs csrc/attn.hip, the others csrc/attn_dh.hip."""
    Dh = Wo // H if H > 0 else 0
    if H <= 0 or Wo != H * Dh or W3 != 3 * Wo or Dh not in ATTN_HEAD_DIMS:
        raise ValueError(f"{what}: widths {W3} / {Wo} do not give a supported head_dim for H={H} {ATTN_HEAD_DIMS}")
    return Dh


def attention(qkv: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int, lse: Optional[torch.Tensor] = None,
              scale: Optional[float] = None, drop_p: float = 0.0, drop_seed: int = 0) -> torch.Tensor:
    """qkv bf16 [B*S, 3*H*Dh] (q|k|v), out bf16 [B*S, H*Dh]; softmax(q k^T scale) v per head (scale: Dh^-0.5)."""
    _need_cuda(qkv, out, lse)
    T, W3, ldq = _rows2d(qkv, "attention qkv")
    To, Wo, ldo = _rows2d(out, "attention out")
    if T != B * S or To != T:
        raise ValueError(f"attention: shapes {tuple(qkv.shape)} / {tuple(out.shape)} do not match B={B} S={S} H={H}")
    Dh = _attn_head_dim(W3, Wo, H, "attention")
    if qkv.dtype != BF16 or out.dtype != BF16:
        raise TypeError("attention: bf16 only")
    _vec_f32(lse, B * H * S, "attention: lse")
    sc = Dh ** -0.5 if scale is None else float(scale)
    with _Span("attn_fwd", 4 * B * H * S * S * Dh, _nbytes(qkv, out, lse)):
        if Dh == 64:
            check(_lib.load().rvit_attn_fwd(qkv.data_ptr(), out.data_ptr(), _ptr(lse), B, S, H, ldq, ldo, sc,
                                            float(drop_p), int(drop_seed), _stream()), "rvit_attn_fwd")
        else:
            check(_lib.load().rvit_attn_dh_fwd(qkv.data_ptr(), out.data_ptr(), _ptr(lse), B, S, H, Dh, ldq, ldo, sc,
                                               float(drop_p), int(drop_seed), _stream()), "rvit_attn_dh_fwd")
    return out


def encoder_mlp(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor,
                w2: torch.Tensor, b2: torch.Tensor, eps: float = 1e-5, rows_per_wg: int = 0) -> torch.Tensor:
    """x += linear2(gelu(linear1(LayerNorm(x)))) in place; x f32 [T,256] contiguous, w1 bf16 [hidden,256],
    w2 bf16 [256,hidden]: one kernel, the hidden map stays in LDS (inference)."""
    _need_cuda(x, gamma, beta, w1, b1, w2, b2)
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("encoder_mlp: x must be a contiguous float32 [T, E] tensor")
    T, E = x.shape
    hidden = w1.shape[0]
    if not encoder_mlp_supported(E, hidden):
        raise ValueError(f"encoder_mlp: E={E} / hidden={hidden} not supported (E == 256, hidden % 256 == 0)")
    if w1.dtype != BF16 or w2.dtype != BF16 or tuple(w1.shape) != (hidden, E) or tuple(w2.shape) != (E, hidden) \
            or not w1.is_contiguous() or not w2.is_contiguous():
        raise ValueError("encoder_mlp: w1 [hidden,E] / w2 [E,hidden] must be contiguous bf16")
    _vec_f32(gamma, E, "encoder_mlp: gamma")
    _vec_f32(beta, E, "encoder_mlp: beta")
    _vec_f32(b1, hidden, "encoder_mlp: b1")
    _vec_f32(b2, E, "encoder_mlp: b2")
    with _Span("encoder_mlp", 4 * T * E * hidden, 8 * T * E + 4 * E * hidden):
        check(_lib.load().rvit_encoder_mlp_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), w1.data_ptr(),
                                               b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), T, E, hidden,
                                               int(rows_per_wg), _stream()), "rvit_encoder_mlp_fwd")
    return x


def encoder_attn(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, wqkv: torch.Tensor, bqkv: torch.Tensor,
                 wout: torch.Tensor, bout: torch.Tensor, B: int, S: int, H: int, eps: float = 1e-5) -> torch.Tensor:
    """x += out_proj(attention(in_proj(LayerNorm(x)))) in place; x f32 [B*S,256] contiguous, wqkv bf16 [768,256],
    wout bf16 [256,256]: one kernel, q/k/v and the attention output stay in LDS (inference)."""
    _need_cuda(x, gamma, beta, wqkv, bqkv, wout, bout)
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_contiguous() or x.shape[0] != B * S:
        raise ValueError("encoder_attn: x must be a contiguous float32 [B*S, E] tensor")
    E = x.shape[1]
    if not encoder_attn_supported(E, H, S):
        raise ValueError(f"encoder_attn: E={E} H={H} S={S} not supported (E == 256, H == 4, S <= 64)")
    if wqkv.dtype != BF16 or wout.dtype != BF16 or tuple(wqkv.shape) != (3 * E, E) or tuple(wout.shape) != (E, E) \
            or not wqkv.is_contiguous() or not wout.is_contiguous():
        raise ValueError("encoder_attn: wqkv [3E,E] / wout [E,E] must be contiguous bf16")
    _vec_f32(gamma, E, "encoder_attn: gamma")
    _vec_f32(beta, E, "encoder_attn: beta")
    _vec_f32(bqkv, 3 * E, "encoder_attn: bqkv")
    _vec_f32(bout, E, "encoder_attn: bout")
    T = B * S
    with _Span("encoder_attn", 2 * T * E * 4 * E + 4 * B * H * S * S * 64, 8 * T * E + 8 * E * E):
        check(_lib.load().rvit_encoder_attn_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), wqkv.data_ptr(),
                                                bqkv.data_ptr(), wout.data_ptr(), bout.data_ptr(), B, S, E, H, _stream()),
              "rvit_encoder_attn_fwd")
    return x


def encoder_stack(x: torch.Tensor, layers, B: int, S: int, H: int, eps: float = 1e-5) -> torch.Tensor:
    """Every layer of the pre-LN encoder in ONE launch, in place on x (f32 [B*S,256] contiguous).  `layers`: per layer the
    twelve tensors (norm1 w, b, in_proj w, b, out_proj w, b, norm2 w, b, linear1 w, b, linear2 w, b); bf16 weights, f32
    vectors.  Same bits as encoder_attn + encoder_mlp per layer."""
    import ctypes
    flat = [t for lay in layers for t in lay]
    _need_cuda(x, *flat)
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_contiguous() or x.shape[0] != B * S:
        raise ValueError("encoder_stack: x must be a contiguous float32 [B*S, E] tensor")
    E = x.shape[1]
    nl = len(layers)
    if nl == 0:
        return x
    hidden = layers[0][8].shape[0]
    if not encoder_stack_supported(E, H, S, hidden, nl):
        raise ValueError(f"encoder_stack: E={E} H={H} S={S} hidden={hidden} layers={nl} not supported")
    for lay in layers:
        if len(lay) != 12:
            raise ValueError("encoder_stack: twelve tensors per layer")
        g1, be1, wqkv, bqkv, wout, bout, g2, be2, w1, b1, w2, b2 = lay
        for w, shp in ((wqkv, (3 * E, E)), (wout, (E, E)), (w1, (hidden, E)), (w2, (E, hidden))):
            if w.dtype != BF16 or tuple(w.shape) != shp or not w.is_contiguous():
                raise ValueError(f"encoder_stack: weight of shape {tuple(w.shape)} must be contiguous bf16 {shp}")
        for v, n in ((g1, E), (be1, E), (bqkv, 3 * E), (bout, E), (g2, E), (be2, E), (b1, hidden), (b2, E)):
            _vec_f32(v, n, "encoder_stack: vector operand")
    table = (ctypes.c_void_p * (12 * nl))(*[t.data_ptr() for t in flat])
    T = B * S
    with _Span("encoder_stack", nl * (2 * T * E * 4 * E + 4 * B * H * S * S * 64 + 4 * T * E * hidden),
               8 * T * E + nl * (8 * E * E + 4 * E * hidden)):
        check(_lib.load().rvit_encoder_stack_fwd(x.data_ptr(), table, nl, float(eps), B, S, E, H, hidden, _stream()),
              "rvit_encoder_stack_fwd")
    return x


def encoder_stack_supported(E: int, H: int, S: int, hidden: int, n_layers: int) -> bool:
    return E == 256 and H == 4 and 1 <= S <= 56 and hidden > 0 and hidden % 256 == 0 and 1 <= n_layers <= 16


def encoder_attn_supported(E: int, H: int, S: int) -> bool:
    return E == 256 and H == 4 and 1 <= S <= 64


def encoder_mlp_supported(E: int, hidden: int) -> bool:
    return E == 256 and hidden > 0 and hidden % 256 == 0


def dwconv(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, *, k: int,
           stride: int, pad: int, act: Optional[str] = None) -> torch.Tensor:
    """Depthwise k x k conv + per-channel scale/bias + activation; x bf16 NHWC [B,H,W,C], w bf16 [k*k, C]."""
    _need_cuda(x, w, scale, bias, out)
    B, H, W, C = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if x.dtype != BF16 or out.dtype != BF16 or w.dtype != BF16 or not (x.is_contiguous() and out.is_contiguous() and w.is_contiguous()) \
            or tuple(w.shape) != (k * k, C) or tuple(out.shape) != (B, Ho, Wo, C) or C % 8:
        raise ValueError(f"dwconv: x {tuple(x.shape)} / w {tuple(w.shape)} / out {tuple(out.shape)} mismatch (bf16, C % 8 == 0)")
    _vec_f32(scale, C, "dwconv: scale")
    _vec_f32(bias, C, "dwconv: bias")
    with _Span("dwconv", 2 * B * Ho * Wo * C * k * k, _nbytes(x, out)):
        check(_lib.load().rvit_dwconv_fwd_bf16(x.data_ptr(), w.data_ptr(), scale.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                               B, H, W, C, k, stride, pad, _ACT[act], _stream()), "rvit_dwconv_fwd_bf16")
    return out


def avgpool(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[b, c] = mean over the pixels of x [B,H,W,C] (bf16 in, bf16 [B,C] out)."""
    _need_cuda(x, out)
    B, H, W, C = x.shape
    if x.dtype != BF16 or out.dtype != BF16 or not x.is_contiguous() or not out.is_contiguous() or tuple(out.shape) != (B, C):
        raise ValueError("avgpool: x bf16 [B,H,W,C] contiguous, out bf16 [B,C]")
    with _Span("avgpool", 0, _nbytes(x, out)):
        check(_lib.load().rvit_avgpool_nhwc(x.data_ptr(), out.data_ptr(), B, H * W, C, _stream()), "rvit_avgpool_nhwc")
    return out


def channel_scale(x: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """x[b, h, w, c] *= gate[b, c] in place (bf16)."""
    _need_cuda(x, gate)
    B, H, W, C = x.shape
    if x.dtype != BF16 or gate.dtype != BF16 or not x.is_contiguous() or not gate.is_contiguous() or tuple(gate.shape) != (B, C):
        raise ValueError("channel_scale: x bf16 [B,H,W,C] contiguous, gate bf16 [B,C]")
    with _Span("channel_scale", 0, 2 * _nbytes(x)):
        check(_lib.load().rvit_channel_scale_nhwc(x.data_ptr(), gate.data_ptr(), B, H * W, C, _stream()),
              "rvit_channel_scale_nhwc")
    return x


# ---- training pass of the MobileNetV3 stem (csrc/mobile_train.hip): every reduction ordered, workspace required ----

def _ows_required(t: torch.Tensor, what: str):
    ws, nb = _ows(t)
    if ws is None:
        raise RuntimeError(f"{what}: ordered reductions need ops.enable_ordered_reductions(device) first")
    return ws, nb


def _map2d(t: torch.Tensor, what: str):
    if t.dtype != BF16 or not t.is_contiguous() or t.dim() < 2 or t.shape[-1] % 8:
        raise ValueError(f"{what}: contiguous bf16 [..., C], C % 8 == 0")
    return t.numel() // t.shape[-1], t.shape[-1]


def bn_generic_stats(y: torch.Tensor, sums: torch.Tensor) -> torch.Tensor:
    """sums[0:C] += sum y, sums[C:2C] += sum y^2 over the rows of y bf16 [..., C] (any C % 8 == 0); ordered."""
    _need_cuda(y, sums)
    M, C = _map2d(y, "bn_generic_stats y")
    _vec_f32(sums, 2 * C, "bn_generic_stats: sums")
    ws, nb = _ows_required(y, "bn_generic_stats")
    with _Span("bn_generic_stats", 0, _nbytes(y)):
        check(_lib.load().rvit_bn_generic_stats(y.data_ptr(), M, C, sums.data_ptr(), ws, nb, _stream()), "rvit_bn_generic_stats")
    return sums


def bn_generic_apply(y: torch.Tensor, sums: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, z: torch.Tensor, *,
                     eps: float, act: Optional[str] = None, residual: Optional[torch.Tensor] = None, momentum: float = 0.1,
                     running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None) -> torch.Tensor:
    """z = act(gamma * (y - mean) / sqrt(var + eps) + beta + residual) with the batch statistics in `sums`; updates the
    running statistics when given (train-mode nn.BatchNorm2d + timm's fused activation)."""
    _need_cuda(y, sums, gamma, beta, z, residual, running_mean, running_var)
    M, C = _map2d(y, "bn_generic_apply y")
    if _map2d(z, "bn_generic_apply z") != (M, C) or (residual is not None and _map2d(residual, "bn_generic_apply residual") != (M, C)):
        raise ValueError("bn_generic_apply: shape mismatch")
    for v, n in ((sums, 2 * C), (gamma, C), (beta, C), (running_mean, C), (running_var, C)):
        _vec_f32(v, n, "bn_generic_apply: per-channel operand")
    with _Span("bn_generic_apply", 0, _nbytes(y, z, residual)):
        check(_lib.load().rvit_bn_generic_apply(y.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(residual),
                                                z.data_ptr(), M, C, float(eps), _ACT[act], float(momentum),
                                                _ptr(running_mean), _ptr(running_var), _stream()), "rvit_bn_generic_apply")
    return z


def bn_generic_bwd(dz: torch.Tensor, y: torch.Tensor, fsums: torch.Tensor, bsums: torch.Tensor, gamma: torch.Tensor,
                   beta: torch.Tensor, dy: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor, *, eps: float,
                   act: Optional[str] = None) -> torch.Tensor:
    """Backward of bn_generic_apply (activation included, recomputed from y): dy; dgamma += , dbeta += ; bsums (zeroed by
    the caller) receives (sum g, sum g * xhat)."""
    _need_cuda(dz, y, fsums, bsums, gamma, beta, dy, dgamma, dbeta)
    M, C = _map2d(y, "bn_generic_bwd y")
    if _map2d(dz, "bn_generic_bwd dz") != (M, C) or _map2d(dy, "bn_generic_bwd dy") != (M, C):
        raise ValueError("bn_generic_bwd: shape mismatch")
    for v, n in ((fsums, 2 * C), (bsums, 2 * C), (gamma, C), (beta, C), (dgamma, C), (dbeta, C)):
        _vec_f32(v, n, "bn_generic_bwd: per-channel operand")
    ws, nb = _ows_required(y, "bn_generic_bwd")
    with _Span("bn_generic_bwd", 0, 2 * _nbytes(dz, y) + _nbytes(dy)):
        check(_lib.load().rvit_bn_generic_bwd(dz.data_ptr(), y.data_ptr(), fsums.data_ptr(), bsums.data_ptr(), gamma.data_ptr(),
                                              beta.data_ptr(), dy.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), M, C,
                                              float(eps), _ACT[act], ws, nb, _stream()), "rvit_bn_generic_bwd")
    return dy


def _dw_geometry(x_shape, out_shape, k, stride, pad, what):
    B, H, W, C = x_shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if tuple(out_shape) != (B, Ho, Wo, C) or C % 8:
        raise ValueError(f"{what}: maps {tuple(x_shape)} / {tuple(out_shape)} do not match k={k} stride={stride} pad={pad}")
    return B, H, W, C


def dwconv_dgrad(dy: torch.Tensor, w: torch.Tensor, dx: torch.Tensor, *, k: int, stride: int, pad: int) -> torch.Tensor:
    """Input gradient of ops.dwconv: dy bf16 [B,Ho,Wo,C], w bf16 [k*k, C] (the forward pack), dx bf16 [B,H,W,C]."""
    _need_cuda(dy, w, dx)
    B, H, W, C = _dw_geometry(dx.shape, dy.shape, k, stride, pad, "dwconv_dgrad")
    if any(t.dtype != BF16 or not t.is_contiguous() for t in (dy, w, dx)) or tuple(w.shape) != (k * k, C):
        raise ValueError("dwconv_dgrad: bf16 contiguous operands, w [k*k, C]")
    with _Span("dwconv_dgrad", 2 * dx.numel() * k * k // (stride * stride), _nbytes(dy, dx)):
        check(_lib.load().rvit_dwconv_dgrad_bf16(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, W, C, k, stride, pad,
                                                 _stream()), "rvit_dwconv_dgrad_bf16")
    return dx


def dwconv_wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, *, k: int, stride: int, pad: int) -> torch.Tensor:
    """dw f32 [C, k*k] (the parameter's [C,1,k,k] layout) += sum over pixels of dy * shifted x; ordered."""
    _need_cuda(dy, x, dw)
    B, H, W, C = _dw_geometry(x.shape, dy.shape, k, stride, pad, "dwconv_wgrad")
    if dy.dtype != BF16 or x.dtype != BF16 or not (dy.is_contiguous() and x.is_contiguous()):
        raise ValueError("dwconv_wgrad: bf16 contiguous maps")
    _vec_f32(dw, C * k * k, "dwconv_wgrad: dw")
    ws, nb = _ows_required(x, "dwconv_wgrad")
    with _Span("dwconv_wgrad", 2 * dy.numel() * k * k, _nbytes(dy, x)):
        check(_lib.load().rvit_dwconv_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, C, k, stride, pad, ws, nb,
                                                 _stream()), "rvit_dwconv_wgrad_bf16")
    return dw


def channel_affine(x: torch.Tensor, gate: torch.Tensor, out: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b,h,w,c] = x[b,h,w,c] * gate[b,c] (+ add[b,c]); bf16; out may be x."""
    _need_cuda(x, gate, out, add)
    B, H, W, C = x.shape
    if any(t.dtype != BF16 or not t.is_contiguous() for t in (x, gate, out)) or out.shape != x.shape or tuple(gate.shape) != (B, C) \
            or C % 8 or (add is not None and (add.dtype != BF16 or not add.is_contiguous() or tuple(add.shape) != (B, C))):
        raise ValueError("channel_affine: x / out bf16 [B,H,W,C], gate / add bf16 [B,C], C % 8 == 0")
    with _Span("channel_affine", 0, _nbytes(x, out)):
        check(_lib.load().rvit_channel_affine_nhwc(x.data_ptr(), gate.data_ptr(), _ptr(add), out.data_ptr(), B, H * W, C, _stream()),
              "rvit_channel_affine_nhwc")
    return out


def channel_dot(a: torch.Tensor, x: torch.Tensor, out: torch.Tensor, gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b,c] = (sum over pixels of a * x) * hardsigmoid'(gate[b,c]) (gate = the hard-sigmoid OUTPUT; None: no factor)."""
    _need_cuda(a, x, out, gate)
    B, H, W, C = x.shape
    if any(t.dtype != BF16 or not t.is_contiguous() for t in (a, x, out)) or a.shape != x.shape or tuple(out.shape) != (B, C) \
            or C % 8 or (gate is not None and (gate.dtype != BF16 or not gate.is_contiguous() or tuple(gate.shape) != (B, C))):
        raise ValueError("channel_dot: a / x bf16 [B,H,W,C], out / gate bf16 [B,C], C % 8 == 0")
    with _Span("channel_dot", 2 * x.numel(), _nbytes(a, x)):
        check(_lib.load().rvit_channel_dot_nhwc(a.data_ptr(), x.data_ptr(), _ptr(gate), out.data_ptr(), B, H * W, C, _stream()),
              "rvit_channel_dot_nhwc")
    return out


SE_MAX_LDS_BYTES = 160 * 1024


def squeeze_excite_supported(HW: int, C: int, RD: int) -> bool:
    """Whether one image's map fits the LDS budget of squeeze_excite (include/rvit.h, RVIT_SE_MAX_LDS_BYTES)."""
    if C % 8 or RD % 8 or C > 4096 or C <= 0:
        return False
    return 2 * HW * C + 4 * ((512 // (C // 8)) * C + 2 * C + RD) <= SE_MAX_LDS_BYTES


def squeeze_excite(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """x[b,h,w,c] *= hardsigmoid(w2 . relu(w1 . mean_hw(x[b]) + b1) + b2)[c], in place, one launch (timm SqueezeExcite)."""
    _need_cuda(x, w1, b1, w2, b2)
    B, H, W, C = x.shape
    RD = w1.shape[0]
    if x.dtype != BF16 or w1.dtype != BF16 or w2.dtype != BF16 or not (x.is_contiguous() and w1.is_contiguous() and w2.is_contiguous()) \
            or tuple(w1.shape) != (RD, C) or tuple(w2.shape) != (C, RD):
        raise ValueError(f"squeeze_excite: x {tuple(x.shape)} / w1 {tuple(w1.shape)} / w2 {tuple(w2.shape)} mismatch (bf16)")
    _vec_f32(b1, RD, "squeeze_excite: b1")
    _vec_f32(b2, C, "squeeze_excite: b2")
    with _Span("squeeze_excite", 4 * B * C * RD, 2 * _nbytes(x)):
        check(_lib.load().rvit_squeeze_excite_bf16(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                                   B, H * W, C, RD, _stream()), "rvit_squeeze_excite_bf16")
    return x


def pointwise_supported(K: int, N: int) -> bool:
    """Whether pointwise_conv takes an [N, K] weight matrix (else: gemm with the same epilogue arguments)."""
    return bool(_lib.load().rvit_pointwise_conv_supported(int(K), int(N)))


def pointwise_conv(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, scale: Optional[torch.Tensor] = None,
                   bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                   act: Optional[str] = None) -> torch.Tensor:
    """out[m, n] = act(scale[n] * (x[m, :] . w[n, :]) + bias[n] + residual[m, n]); x bf16 [M, K], w bf16 [N, K], out and
    residual bf16 [M, N], all contiguous (an NHWC map viewed as rows of pixels)."""
    _need_cuda(x, w, out)
    M, K = x.shape
    N = w.shape[0]
    if x.dtype != BF16 or w.dtype != BF16 or out.dtype != BF16 or not (x.is_contiguous() and w.is_contiguous() and out.is_contiguous()) \
            or tuple(w.shape) != (N, K) or tuple(out.shape) != (M, N):
        raise ValueError(f"pointwise_conv: x {tuple(x.shape)} / w {tuple(w.shape)} / out {tuple(out.shape)} mismatch (bf16, contiguous)")
    if residual is not None and (residual.dtype != BF16 or not residual.is_contiguous() or tuple(residual.shape) != (M, N)):
        raise ValueError("pointwise_conv: residual must be bf16 [M, N] contiguous")
    if scale is not None:
        _vec_f32(scale, N, "pointwise_conv: scale")
    if bias is not None:
        _vec_f32(bias, N, "pointwise_conv: bias")
    with _Span("pointwise_conv", 2 * M * K * N, _nbytes(x, out)):
        check(_lib.load().rvit_pointwise_conv_bf16(x.data_ptr(), w.data_ptr(), scale.data_ptr() if scale is not None else None,
                                                   bias.data_ptr() if bias is not None else None,
                                                   residual.data_ptr() if residual is not None else None, out.data_ptr(),
                                                   M, K, N, _ACT[act], _stream()), "rvit_pointwise_conv_bf16")
    return out


def conv3x3s2_rgb(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, out: torch.Tensor,
                  act: Optional[str] = None) -> torch.Tensor:
    """conv 3x3 / stride 2 / pad 1 (3 -> 16) + scale/bias + activation from the fp32 NCHW input; w f32 [27, 16] with
    k = (kh*3 + kw)*3 + ci; out bf16 NHWC [B, H/2, W/2, 16]."""
    _need_cuda(x, w, scale, bias, out)
    B, Cin, H, W = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous() or Cin != 3 or H % 2 or W % 2 or w.dtype != torch.float32 \
            or tuple(w.shape) != (27, 16) or not w.is_contiguous() or out.dtype != BF16 or not out.is_contiguous() \
            or tuple(out.shape) != (B, H // 2, W // 2, 16):
        raise ValueError(f"conv3x3s2_rgb: x {tuple(x.shape)} f32 NCHW (H, W even), w f32 [27,16], out bf16 {(B, H // 2, W // 2, 16)}")
    _vec_f32(scale, 16, "conv3x3s2_rgb: scale")
    _vec_f32(bias, 16, "conv3x3s2_rgb: bias")
    with _Span("conv3x3s2_rgb", 2 * B * (H // 2) * (W // 2) * 27 * 16, _nbytes(x, out)):
        check(_lib.load().rvit_conv3x3s2_rgb_f32_fwd(x.data_ptr(), w.data_ptr(), scale.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                     B, H, W, 16, _ACT[act], _stream()), "rvit_conv3x3s2_rgb_f32_fwd")
    return out


def attention_s64(qkv: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int, scale: Optional[float] = None) -> torch.Tensor:
    """attention() for S <= 64 with the arithmetic of encoder_attn's core (bit-identical to the fused kernel)."""
    _need_cuda(qkv, out)
    T, W3, ldq = _rows2d(qkv, "attention_s64 qkv")
    To, Wo, ldo = _rows2d(out, "attention_s64 out")
    if T != B * S or To != T or W3 != 3 * H * 64 or Wo != H * 64 or S > 64 or qkv.dtype != BF16 or out.dtype != BF16:
        raise ValueError(f"attention_s64: shapes {tuple(qkv.shape)} / {tuple(out.shape)} do not match B={B} S={S} H={H} (bf16, S <= 64)")
    sc = (1.0 / 8.0) if scale is None else float(scale)
    with _Span("attn_fwd", 4 * B * H * S * S * 64, _nbytes(qkv, out)):
        check(_lib.load().rvit_attn_fwd_s64(qkv.data_ptr(), out.data_ptr(), B, S, H, ldq, ldo, sc, _stream()),
              "rvit_attn_fwd_s64")
    return out


def attention_fp8(qkv: torch.Tensor, out_q: torch.Tensor, out_s: torch.Tensor, B: int, S: int, H: int,
                  scale: Optional[float] = None):
    """fp8-MFMA attention with MXFP8 output: qkv bf16 [B*S, 3*H*64]; out_q uint8 [B*S, H*64] (e4m3), out_s the tiled
    E8M0 scales (ops.mx_scale_shape(B*S, H*64)): the A operand of the out_proj's gemm_mx."""
    _need_cuda(qkv, out_q, out_s)
    T, W3, ldq = _rows2d(qkv, "attention_fp8 qkv")
    E = H * 64
    if T != B * S or W3 != 3 * E or qkv.dtype != BF16 or E % 128:
        raise ValueError(f"attention_fp8: qkv {tuple(qkv.shape)} does not match B={B} S={S} H={H} (H even)")
    if tuple(out_q.shape) != (T, E) or out_q.dtype != torch.uint8 or out_q.stride(1) != 1 \
            or tuple(out_s.shape) != mx_scale_shape(T, E) or out_s.dtype != torch.uint8 or not out_s.is_contiguous():
        raise ValueError("attention_fp8: out_q [T,E] / tiled out_s (ops.mx_scale_shape) must be uint8")
    sc = (1.0 / 8.0) if scale is None else float(scale)
    with _Span("attn_fwd_fp8", 4 * B * H * S * S * 64, qkv.numel() * 2 + out_q.numel()):
        check(_lib.load().rvit_attn_fwd_fp8(qkv.data_ptr(), out_q.data_ptr(), out_q.stride(0), out_s.data_ptr(), B, S, H,
                                            ldq, sc, _stream()), "rvit_attn_fwd_fp8")
    return out_q, out_s


def attention_mx(qkv: torch.Tensor, out_q: torch.Tensor, out_s: torch.Tensor, B: int, S: int, H: int,
                 scale: Optional[float] = None):
    """attention() on the bf16 MFMA with the output written as MXFP8 (the bytes attention() -> quantize_mx() give):
    the MX-fp8 inference mode's default.  Arguments as attention_fp8."""
    _need_cuda(qkv, out_q, out_s)
    T, W3, ldq = _rows2d(qkv, "attention_mx qkv")
    E = H * 64
    if T != B * S or W3 != 3 * E or qkv.dtype != BF16 or E % 128:
        raise ValueError(f"attention_mx: qkv {tuple(qkv.shape)} does not match B={B} S={S} H={H} (H even)")
    if tuple(out_q.shape) != (T, E) or out_q.dtype != torch.uint8 or out_q.stride(1) != 1 \
            or tuple(out_s.shape) != mx_scale_shape(T, E) or out_s.dtype != torch.uint8 or not out_s.is_contiguous():
        raise ValueError("attention_mx: out_q [T,E] / tiled out_s (ops.mx_scale_shape) must be uint8")
    sc = (1.0 / 8.0) if scale is None else float(scale)
    with _Span("attn_fwd", 4 * B * H * S * S * 64, qkv.numel() * 2 + out_q.numel()):
        check(_lib.load().rvit_attn_fwd_mx(qkv.data_ptr(), out_q.data_ptr(), out_q.stride(0), out_s.data_ptr(), B, S, H,
                                           ldq, sc, _stream()), "rvit_attn_fwd_mx")
    return out_q, out_s


# =================================================================================================
# backward / loss kernels
# =================================================================================================
def wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, *, rows: Optional[int] = None, n_store: Optional[int] = None,
          y_row_group: int = 0, y_row_extra: int = 0, x_row_group: int = 0, x_row_extra: int = 0,
          splits: int = 0) -> torch.Tensor:
    """dw[N_store,K] += dy[M,N].T @ x[M,K]  (bf16 operands, fp32 atomics)."""
    _need_cuda(dy, x, dw)
    My, N, ldy = _rows2d(dy, "wgrad dy")
    Mx, K, ldx = _rows2d(x, "wgrad x")
    M = rows if rows is not None else min(My, Mx)
    ns, Kw, ldw = _rows2d(dw, "wgrad dw")
    n_store = ns if n_store is None else n_store
    if dy.dtype != BF16 or x.dtype != BF16 or dw.dtype != torch.float32 or Kw != K or n_store > N or n_store > ns:
        raise ValueError(f"wgrad: bad operands dy{tuple(dy.shape)} x{tuple(x.shape)} dw{tuple(dw.shape)}")
    with _Span("wgrad_bf16", 2 * M * N * K, (M * N + M * K) * 2 + n_store * K * 4):
        check(_lib.load().rvit_wgrad_bf16(dy.data_ptr(), ldy, y_row_group, y_row_extra, x.data_ptr(), ldx, x_row_group,
                                          x_row_extra, dw.data_ptr(), ldw, M, N, n_store, K, splits, *_ows(dw), _stream()),
              "rvit_wgrad_bf16")
    return dw


def linear_wgrad_bias(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, dbias: torch.Tensor, *,
                      rows: Optional[int] = None, n_store: Optional[int] = None, splits: int = 0) -> torch.Tensor:
    """dw[N_store,K] += dy[M,N].T @ x[M,K] and dbias[N_store] += dy.sum(0), one launch (two when N <= 64)."""
    _need_cuda(dy, x, dw, dbias)
    My, N, ldy = _rows2d(dy, "linear_wgrad_bias dy")
    Mx, K, ldx = _rows2d(x, "linear_wgrad_bias x")
    M = rows if rows is not None else min(My, Mx)
    ns, Kw, ldw = _rows2d(dw, "linear_wgrad_bias dw")
    n_store = ns if n_store is None else n_store
    if dy.dtype != BF16 or x.dtype != BF16 or dw.dtype != torch.float32 or Kw != K or n_store > N or n_store > ns:
        raise ValueError(f"linear_wgrad_bias: bad operands dy{tuple(dy.shape)} x{tuple(x.shape)} dw{tuple(dw.shape)}")
    _vec_f32(dbias, n_store, "linear_wgrad_bias: dbias")
    with _Span("wgrad_bf16", 2 * M * N * K, (M * N + M * K) * 2 + n_store * K * 4):
        check(_lib.load().rvit_linear_wgrad_bias_bf16(dy.data_ptr(), ldy, x.data_ptr(), ldx, dw.data_ptr(), ldw,
                                                      dbias.data_ptr(), M, N, n_store, K, splits, *_ows(dw), _stream()),
              "rvit_linear_wgrad_bias_bf16")
    return dw


def conv2d_wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, *, kh: int, kw: int, stride: int, pad: int,
                 seg_len: Optional[int] = None, splits: int = 0) -> torch.Tensor:
    """dw[Cout, kh*kw*seg] += dy^T gather(x); dy bf16 [B,Ho,Wo,Cout], x bf16 [B,H,W,Cpix], dw fp32."""
    _need_cuda(dy, x, dw)
    B, H, W, Cpix = x.shape
    Bo, Ho, Wo, Cout = dy.shape
    seg = Cpix if seg_len is None else seg_len
    if Bo != B or dw.shape != (Cout, kh * kw * seg) or dw.dtype != torch.float32 or dy.dtype != BF16 or x.dtype != BF16 \
            or not (dy.is_contiguous() and x.is_contiguous() and dw.is_contiguous()):
        raise ValueError("conv2d_wgrad: bad operands")
    with _Span("conv2d_wgrad", 2 * B * Ho * Wo * Cout * kh * kw * seg, _nbytes(dy, x, dw)):
        check(_lib.load().rvit_conv2d_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, Cpix, seg, kh, kw,
                                                 stride, pad, Ho, Wo, Cout, splits, *_ows(dw), _stream()),
              "rvit_conv2d_wgrad_bf16")
    return dw


def conv2d_dgrad(dy: torch.Tensor, w_t: torch.Tensor, dx: torch.Tensor, *, kh: int, kw: int, stride: int, pad: int,
                 residual: Optional[torch.Tensor] = None, tile: int = 0) -> torch.Tensor:
    """dx = conv_transpose(dy, w) (+ residual); dy bf16 [B,Hy,Wy,Cout], w_t bf16 [Cin, kh*kw*Cout]."""
    _need_cuda(dy, w_t, dx, residual)
    B, Hy, Wy, Cout = dy.shape
    Bx, Hx, Wx, Cin = dx.shape
    if Bx != B or w_t.shape != (Cin, kh * kw * Cout) or dy.dtype != BF16 or dx.dtype != BF16 or w_t.dtype != BF16 \
            or not (dy.is_contiguous() and dx.is_contiguous() and w_t.is_contiguous()):
        raise ValueError("conv2d_dgrad: bad operands")
    if residual is not None and (residual.shape != dx.shape or residual.dtype != BF16 or not residual.is_contiguous()):
        raise ValueError("conv2d_dgrad: residual must match dx")
    with _Span("conv2d_dgrad", 2 * B * Hy * Wy * Cout * kh * kw * Cin, _nbytes(dy, w_t, dx, residual)):
        check(_lib.load().rvit_conv2d_dgrad_bf16(dy.data_ptr(), w_t.data_ptr(), dx.data_ptr(), B, Hy, Wy, Cout, kh, kw,
                                                 stride, pad, Hx, Wx, Cin, _ptr(residual), tile, _stream()),
              "rvit_conv2d_dgrad_bf16")
    return dx


def conv2d_dgrad_bn_sums(dy: torch.Tensor, w_t: torch.Tensor, dx: torch.Tensor, *, kh: int, kw: int, stride: int, pad: int,
                         bn_y: torch.Tensor, bn_relu_mask: Optional[torch.Tensor], bn_fwd_sums: torch.Tensor,
                         bn_bwd_sums: torch.Tensor, eps: float, stat_ws: torch.Tensor,
                         residual: Optional[torch.Tensor] = None) -> bool:
    """conv2d_dgrad whose output dx enters the backward pass of a train-mode BatchNorm (+ ReLU): where the conv kernel has
    the hook (direct 3x3 / stride 1 kernels) its epilogue also produces that norm's two backward sums
    (bn_bwd_sums[0:C] += sum d, [C:2C] += rstd * sum d (y - mean), d = dx where the ReLU bit is set) and bn_bwd runs with
    reduced=True afterwards.  Returns False where it does not: dx is complete, bn_bwd_sums untouched."""
    _need_cuda(dy, w_t, dx, residual, bn_y, bn_relu_mask, bn_fwd_sums, bn_bwd_sums, stat_ws)
    B, Hy, Wy, Cout = dy.shape
    Bx, Hx, Wx, Cin = dx.shape
    if Bx != B or w_t.shape != (Cin, kh * kw * Cout) or dy.dtype != BF16 or dx.dtype != BF16 or w_t.dtype != BF16 \
            or not (dy.is_contiguous() and dx.is_contiguous() and w_t.is_contiguous()):
        raise ValueError("conv2d_dgrad_bn_sums: bad operands")
    if residual is not None and (residual.shape != dx.shape or residual.dtype != BF16 or not residual.is_contiguous()):
        raise ValueError("conv2d_dgrad_bn_sums: residual must match dx")
    if bn_y.shape != dx.shape or bn_y.dtype != BF16 or not bn_y.is_contiguous():
        raise ValueError("conv2d_dgrad_bn_sums: bn_y must match dx")
    _check_relu_mask(bn_relu_mask, B * Hx * Wx, Cin, bn_relu_mask is not None)
    _vec_f32(bn_fwd_sums, 2 * Cin, "conv2d_dgrad_bn_sums: bn_fwd_sums")
    _vec_f32(bn_bwd_sums, 2 * Cin, "conv2d_dgrad_bn_sums: bn_bwd_sums")
    if stat_ws.dtype != torch.float32 or not stat_ws.is_contiguous() or stat_ws.numel() <= STAT_STAGE_FLOATS:
        raise ValueError("conv2d_dgrad_bn_sums: stat_ws must come from ops.conv_stats_workspace()")
    lib = _lib.load()
    slots = ctypes.c_int(0)
    slot_floats = stat_ws.numel() - STAT_STAGE_FLOATS
    name = _conv3x3_name(B, Hx, Wx, Cout, Cin) if (kh == 3 and kw == 3 and stride == 1 and pad == 1) else "conv2d_dgrad"
    with _Span(name, 2 * B * Hy * Wy * Cout * kh * kw * Cin, _nbytes(dy, w_t, dx, residual, bn_y)):
        check(lib.rvit_conv2d_dgrad_bnr_bf16(dy.data_ptr(), w_t.data_ptr(), dx.data_ptr(), B, Hy, Wy, Cout, kh, kw, stride,
                                             pad, Hx, Wx, Cin, _ptr(residual), bn_y.data_ptr(), _ptr(bn_relu_mask),
                                             bn_fwd_sums.data_ptr(), eps, stat_ws.data_ptr(), slot_floats * 4,
                                             ctypes.byref(slots), _stream()), "rvit_conv2d_dgrad_bnr_bf16")
    if slots.value <= 0:
        return False
    with _Span("bn_stats_finalize", 0, slots.value * 2 * Cin * 4):
        check(lib.rvit_bn_stats_finalize(stat_ws.data_ptr(), slots.value, Cin, bn_bwd_sums.data_ptr(),
                                         stat_ws.data_ptr() + slot_floats * 4, _stream()), "rvit_bn_stats_finalize")
    return True


S2_TAP_ORDER = (4, 5, 3, 7, 1, 8, 6, 2, 0)   # K-block order of the w_s2 pack (include/rvit.h)


def conv2d_dgrad_s2(dy: torch.Tensor, w_s2: torch.Tensor, dx: torch.Tensor, *,
                    residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx = conv_transpose(dy, w) (+ residual) for a 3x3 / stride 2 / pad 1 conv, by output parity (four dense
    stride-1 convs, no zero taps); dy bf16 [B,Hy,Wy,Cout], dx bf16 [B,2Hy,2Wy,Cin], w_s2 bf16 [Cin, 9*Cout] with
    the taps in S2_TAP_ORDER."""
    _need_cuda(dy, w_s2, dx, residual)
    B, Hy, Wy, Cout = dy.shape
    Bx, Hx, Wx, Cin = dx.shape
    if Bx != B or Hx != 2 * Hy or Wx != 2 * Wy or w_s2.shape != (Cin, 9 * Cout) or dy.dtype != BF16 or dx.dtype != BF16 \
            or w_s2.dtype != BF16 or not (dy.is_contiguous() and dx.is_contiguous() and w_s2.is_contiguous()):
        raise ValueError("conv2d_dgrad_s2: bad operands")
    if residual is not None and (residual.shape != dx.shape or residual.dtype != BF16 or not residual.is_contiguous()):
        raise ValueError("conv2d_dgrad_s2: residual must match dx")
    with _Span("conv2d_dgrad", 2 * B * Hy * Wy * Cout * 9 * Cin, _nbytes(dy, w_s2, dx, residual)):
        check(_lib.load().rvit_conv2d_dgrad_s2_bf16(dy.data_ptr(), w_s2.data_ptr(), dx.data_ptr(), B, Hy, Wy, Cout, Cin,
                                                    _ptr(residual), _stream()), "rvit_conv2d_dgrad_s2_bf16")
    return dx


def conv1x1_dgrad_s2_acc(dy: torch.Tensor, w_t: torch.Tensor, dx: torch.Tensor) -> torch.Tensor:
    """dx[:, ::2, ::2, :] += dy @ w_t.T : input gradient of a 1x1 / stride 2 conv added to dx (the other pixels of dx
    get nothing from such a conv); dy bf16 [B,Hy,Wy,Cout], w_t bf16 [Cin,Cout], dx bf16 [B,2Hy,2Wy,Cin]."""
    _need_cuda(dy, w_t, dx)
    B, Hy, Wy, Cout = dy.shape
    Bx, Hx, Wx, Cin = dx.shape
    if Bx != B or Hx != 2 * Hy or Wx != 2 * Wy or w_t.shape != (Cin, Cout) or dy.dtype != BF16 or dx.dtype != BF16 \
            or w_t.dtype != BF16 or not (dy.is_contiguous() and dx.is_contiguous() and w_t.is_contiguous()):
        raise ValueError("conv1x1_dgrad_s2_acc: bad operands")
    with _Span("conv2d_dgrad", 2 * B * Hy * Wy * Cout * Cin, _nbytes(dy, w_t) + B * Hy * Wy * Cin * 4):
        check(_lib.load().rvit_conv1x1_dgrad_s2_acc_bf16(dy.data_ptr(), w_t.data_ptr(), dx.data_ptr(), B, Hy, Wy, Cout,
                                                         Cin, _stream()), "rvit_conv1x1_dgrad_s2_acc_bf16")
    return dx


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, gamma: torch.Tensor,
                  dx: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor, *, accumulate: bool,
                  dx_bf16: Optional[torch.Tensor] = None, drop: Optional[Tuple[float, int]] = None) -> torch.Tensor:
    """dx (+)= LN'(dy); dgamma/dbeta += ...; optional bf16 copy of the updated dx, with drop=(p, seed) multiplied by
    that dropout site's mask / (1 - p) (element index row * E + col): what the next Linear's backward consumes."""
    _need_cuda(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, dx_bf16)
    rows, E, lddy = _rows2d(dy, "layernorm_bwd dy")
    rx, Ex, ldx = _rows2d(x, "layernorm_bwd x")
    rd, Ed, lddx = _rows2d(dx, "layernorm_bwd dx")
    if (rx, Ex) != (rows, E) or (rd, Ed) != (rows, E) or dy.dtype != BF16 or x.dtype != torch.float32 \
            or dx.dtype != torch.float32:
        raise ValueError("layernorm_bwd: shape/dtype mismatch")
    lddb = 0
    if dx_bf16 is not None:
        rb, Eb, lddb = _rows2d(dx_bf16, "layernorm_bwd dx_bf16")
        if (rb, Eb) != (rows, E) or dx_bf16.dtype != BF16:
            raise ValueError("layernorm_bwd: dx_bf16 mismatch")
    for v, n in ((mean, rows), (rstd, rows), (gamma, E), (dgamma, E), (dbeta, E)):
        _vec_f32(v, n, "layernorm_bwd: vector")
    p_drop, seed = (float(drop[0]), int(drop[1])) if drop is not None else (0.0, 0)
    if p_drop > 0.0 and dx_bf16 is None:
        raise ValueError("layernorm_bwd: drop applies to the bf16 copy, which is missing")
    with _Span("layernorm_bwd", 0, rows * E * (2 + 4 + 8 + (2 if dx_bf16 is not None else 0))):
        check(_lib.load().rvit_layernorm_bwd_drop(dy.data_ptr(), lddy, x.data_ptr(), ldx, mean.data_ptr(), rstd.data_ptr(),
                                                  gamma.data_ptr(), dx.data_ptr(), lddx, int(accumulate), _ptr(dx_bf16),
                                                  lddb, dgamma.data_ptr(), dbeta.data_ptr(), rows, E, *_ows(dx), p_drop,
                                                  seed, _stream()),
              "rvit_layernorm_bwd_drop")
    return dx


def attention_bwd(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, dqkv: torch.Tensor,
                  B: int, S: int, H: int, scale: Optional[float] = None, drop_p: float = 0.0,
                  drop_seed: int = 0) -> torch.Tensor:
    _need_cuda(qkv, out, dout, lse, dqkv)
    T, W3, ldq = _rows2d(qkv, "attention_bwd qkv")
    _, Wo, ldo = _rows2d(out, "attention_bwd out")
    _, Wd, lddo = _rows2d(dout, "attention_bwd dout")
    Tq, Wq, lddq = _rows2d(dqkv, "attention_bwd dqkv")
    if T != B * S or Tq != T or Wq != W3 or Wd != Wo or any(t.dtype != BF16 for t in (qkv, out, dout, dqkv)):
        raise ValueError("attention_bwd: shape/dtype mismatch")
    Dh = _attn_head_dim(W3, Wo, H, "attention_bwd")
    _vec_f32(lse, B * H * S, "attention_bwd: lse")
    sc = Dh ** -0.5 if scale is None else float(scale)
    with _Span("attn_bwd", 10 * B * H * S * S * Dh, _nbytes(qkv, out, dout, dqkv)):
        if Dh == 64:
            check(_lib.load().rvit_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                            dqkv.data_ptr(), B, S, H, ldq, ldo, lddo, lddq, sc, float(drop_p),
                                            int(drop_seed), _stream()), "rvit_attn_bwd")
        else:
            check(_lib.load().rvit_attn_dh_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                               dqkv.data_ptr(), B, S, H, Dh, ldq, ldo, lddo, lddq, sc, float(drop_p),
                                               int(drop_seed), _stream()), "rvit_attn_dh_bwd")
    return dqkv


def ce_loss(logits: torch.Tensor, labels: torch.Tensor, loss_sum: Optional[torch.Tensor],
            dlogits: Optional[torch.Tensor], dlogits_bf16: Optional[torch.Tensor], label_smoothing: float,
            grad_scale: float = 1.0) -> None:
    """*loss_sum += mean smoothed CE; dlogits (f32 [B,C]) / dlogits_bf16 (bf16 [B,C_pad], zero padded)."""
    _need_cuda(logits, labels, loss_sum, dlogits, dlogits_bf16)
    B, C, ldl = _rows2d(logits, "ce_loss logits")
    if logits.dtype != torch.float32 or labels.dtype != torch.int64 or labels.numel() != B or not labels.is_contiguous():
        raise ValueError("ce_loss: logits float32 [B,C], labels int64 [B]")
    lddl = lddb = 0
    c_pad = C
    if dlogits is not None:
        _, cd, lddl = _rows2d(dlogits, "ce_loss dlogits")
        if cd != C or dlogits.dtype != torch.float32:
            raise ValueError("ce_loss: dlogits mismatch")
    if dlogits_bf16 is not None:
        _, c_pad, lddb = _rows2d(dlogits_bf16, "ce_loss dlogits_bf16")
        if c_pad < C or dlogits_bf16.dtype != BF16:
            raise ValueError("ce_loss: dlogits_bf16 mismatch")
    if loss_sum is not None and (loss_sum.dtype != torch.float32 or loss_sum.numel() != 1):
        raise ValueError("ce_loss: loss_sum must be a float32 scalar tensor")
    with _Span("ce_loss", 0, B * C * 8):
        check(_lib.load().rvit_ce_loss(logits.data_ptr(), ldl, labels.data_ptr(), _ptr(loss_sum), _ptr(dlogits), lddl,
                                       _ptr(dlogits_bf16), lddb, B, C, c_pad, float(label_smoothing),
                                       float(grad_scale), _stream()), "rvit_ce_loss")


def colsum(a: torch.Tensor, out: torch.Tensor, *, rows: Optional[int] = None, row_group: int = 0,
           row_extra: int = 0) -> torch.Tensor:
    """out[n] += sum_m a[m,n]; a bf16."""
    _need_cuda(a, out)
    Ma, N, lda = _rows2d(a, "colsum a")
    M = Ma if rows is None else rows
    if a.dtype != BF16:
        raise TypeError("colsum: a must be bf16")
    _vec_f32(out, N, "colsum: out")
    with _Span("colsum", 0, M * N * 2):
        check(_lib.load().rvit_colsum_bf16(a.data_ptr(), lda, out.data_ptr(), M, N, row_group, row_extra, *_ows(out),
                                           _stream()),
              "rvit_colsum_bf16")
    return out


def batch_sum(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[j] += sum_b x[b, j]; x fp32 [B, n] (row stride free), out fp32 [n]."""
    _need_cuda(x, out)
    B, n, ldb = _rows2d(x, "batch_sum x")
    if x.dtype != torch.float32:
        raise TypeError("batch_sum: float32 only")
    _vec_f32(out.reshape(-1), n, "batch_sum: out")
    with _Span("batch_sum", 0, B * n * 4):
        check(_lib.load().rvit_batch_sum_f32(x.data_ptr(), ldb, out.data_ptr(), B, n, _stream()), "rvit_batch_sum_f32")
    return out


# =================================================================================================
# training-mode BatchNorm (+ residual + ReLU) and max-pool backward
# =================================================================================================
def _map2d(t: torch.Tensor, what: str):
    if t.dtype != BF16 or not t.is_contiguous() or t.dim() < 2:
        raise ValueError(f"{what}: contiguous bf16 NHWC map required")
    C = t.shape[-1]
    return t.numel() // C, C


BN_WORKSPACE_BYTES = (64 + 262144) * 4   # RVIT_BN_WORKSPACE_BYTES


def bn_workspace(device) -> torch.Tensor:
    """Zeroed workspace that switches bn_stats / bn_bwd to their ordered (run-to-run bit-identical) reductions."""
    return torch.zeros(BN_WORKSPACE_BYTES // 4, dtype=torch.float32, device=device)


def _check_bn_ws(ws):
    if ws is not None and (ws.dtype != torch.float32 or not ws.is_contiguous() or ws.numel() * 4 < BN_WORKSPACE_BYTES):
        raise ValueError("bn: workspace must come from ops.bn_workspace()")


def bn_stats(y: torch.Tensor, sums: torch.Tensor, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sums[0:C] += sum y, sums[C:2C] += sum y^2 over all pixels; y bf16 [...,C].  With a workspace
    (ops.bn_workspace) the reduction is ordered: the same bits on every run."""
    _need_cuda(y, sums, workspace)
    M, C = _map2d(y, "bn_stats y")
    _vec_f32(sums, 2 * C, "bn_stats: sums")
    _check_bn_ws(workspace)
    with _Span("bn_stats", 0, M * C * 2):
        check(_lib.load().rvit_bn_stats(y.data_ptr(), M, C, sums.data_ptr(), _ptr(workspace), _stream()), "rvit_bn_stats")
    return sums


def bn_apply(y: torch.Tensor, sums: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, z: torch.Tensor, *,
             eps: float, relu: bool, residual: Optional[torch.Tensor] = None, momentum: float = 0.1,
             running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None,
             relu_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu_mask (optional, with relu): uint8 [M * C/8] receiving the sign bits of z (bit k of byte (m, c/8) =
    z[m, 8*(c/8)+k] > 0); `bn_bwd(..., relu_mask=...)` then reads it instead of z."""
    _need_cuda(y, sums, gamma, beta, z, residual, running_mean, running_var, relu_mask)
    M, C = _map2d(y, "bn_apply y")
    _check_relu_mask(relu_mask, M, C, relu)
    if z.shape != y.shape or _map2d(z, "bn_apply z") != (M, C) or (residual is not None and residual.shape != y.shape):
        raise ValueError("bn_apply: shape mismatch")
    if residual is not None:
        _map2d(residual, "bn_apply residual")
    for v, n in ((sums, 2 * C), (gamma, C), (beta, C), (running_mean, C), (running_var, C)):
        _vec_f32(v, n, "bn_apply: vector")
    with _Span("bn_apply", 0, _nbytes(y, z, residual)):
        check(_lib.load().rvit_bn_apply(y.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                        _ptr(residual), z.data_ptr(), M, C, eps, int(relu), momentum,
                                        _ptr(running_mean), _ptr(running_var), _ptr(relu_mask), _stream()),
              "rvit_bn_apply")
    return z


def bn_apply_pool(y: torch.Tensor, sums: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, pooled: torch.Tensor,
                  argmax: torch.Tensor, relu_mask: torch.Tensor, *, eps: float, momentum: float = 0.1,
                  running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None) -> torch.Tensor:
    """BatchNorm (batch statistics in `sums`) + ReLU + MaxPool2d(3, 2, 1) of y [B,H,W,C] in one pass: pooled and
    argmax [B,Ho,Wo,C], relu_mask uint8 [B*H*W*C/8]; the normalised map is not materialised."""
    _need_cuda(y, sums, gamma, beta, pooled, argmax, relu_mask, running_mean, running_var)
    if y.dim() != 4 or y.dtype != BF16 or not y.is_contiguous():
        raise ValueError("bn_apply_pool: y must be a contiguous bf16 [B,H,W,C] map")
    B, H, W, C = y.shape
    shape = (B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C)
    if tuple(pooled.shape) != shape or pooled.dtype != BF16 or not pooled.is_contiguous() \
            or tuple(argmax.shape) != shape or argmax.dtype != torch.uint8 or not argmax.is_contiguous():
        raise ValueError(f"bn_apply_pool: pooled (bf16) and argmax (uint8) must be contiguous {shape}")
    _check_relu_mask(relu_mask, B * H * W, C, True)
    for v, n in ((sums, 2 * C), (gamma, C), (beta, C), (running_mean, C), (running_var, C)):
        _vec_f32(v, n, "bn_apply_pool: vector")
    with _Span("bn_apply_pool", 0, _nbytes(y, pooled, argmax, relu_mask)):
        check(_lib.load().rvit_bn_apply_pool(y.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             pooled.data_ptr(), argmax.data_ptr(), relu_mask.data_ptr(), B, H, W, C,
                                             eps, momentum, _ptr(running_mean), _ptr(running_var), _stream()),
              "rvit_bn_apply_pool")
    return pooled


def _check_relu_mask(mask, M, C, relu):
    if mask is None:
        return
    if not relu:
        raise ValueError("bn: a ReLU mask needs relu=True")
    if mask.dtype != torch.uint8 or not mask.is_contiguous() or mask.numel() != M * (C // 8):
        raise ValueError("bn: relu_mask must be a contiguous uint8 tensor of M * C/8 bytes")


def bn_bwd(dz: torch.Tensor, z: Optional[torch.Tensor], y: torch.Tensor, fwd_sums: torch.Tensor,
           bwd_sums: torch.Tensor, gamma: torch.Tensor, dy: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor, *,
           eps: float, relu: bool, dz_masked: Optional[torch.Tensor] = None,
           relu_mask: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
           reduced: bool = False) -> torch.Tensor:
    """Both BatchNorm backward passes (reduce, then apply); bwd_sums must be zero on entry.  With relu the sign of
    the forward output comes from `relu_mask` (the bits `bn_apply` wrote) when given, else from z.  workspace
    (ops.bn_workspace): ordered reduction, as in bn_stats.  reduced: bwd_sums already hold the two sums (the conv that
    produced dz wrote them, conv2d_dgrad_bn_sums): only the apply pass runs."""
    _need_cuda(dz, z, y, fwd_sums, bwd_sums, gamma, dy, dgamma, dbeta, dz_masked, relu_mask, workspace)
    _check_bn_ws(workspace)
    M, C = _map2d(y, "bn_bwd y")
    _check_relu_mask(relu_mask, M, C, relu)
    for t in (dz, dy, z, dz_masked):
        if t is not None and (t.shape != y.shape or _map2d(t, "bn_bwd map") != (M, C)):
            raise ValueError("bn_bwd: shape mismatch")
    if relu and z is None and relu_mask is None:
        raise ValueError("bn_bwd: relu needs the forward output z or its sign bits")
    mode, zsrc = (2, relu_mask) if (relu and relu_mask is not None) else (int(relu), z)
    for v, n in ((fwd_sums, 2 * C), (bwd_sums, 2 * C), (gamma, C), (dgamma, C), (dbeta, C)):
        _vec_f32(v, n, "bn_bwd: vector")
    lib = _lib.load()
    if not reduced:
        with _Span("bn_bwd_reduce", 0, _nbytes(dz, y, z if relu else None)):
            check(lib.rvit_bn_bwd_reduce(dz.data_ptr(), _ptr(zsrc), y.data_ptr(), fwd_sums.data_ptr(), M, C, eps, mode,
                                         bwd_sums.data_ptr(), _ptr(workspace), _stream()), "rvit_bn_bwd_reduce")
    with _Span("bn_bwd_apply", 0, _nbytes(dz, y, zsrc if relu else None, dy, dz_masked)):
        check(lib.rvit_bn_bwd_apply(dz.data_ptr(), _ptr(zsrc), y.data_ptr(), fwd_sums.data_ptr(), bwd_sums.data_ptr(),
                                    gamma.data_ptr(), dy.data_ptr(), _ptr(dz_masked), dgamma.data_ptr(),
                                    dbeta.data_ptr(), M, C, eps, mode, _stream()), "rvit_bn_bwd_apply")
    return dy


def bn_bwd_pool(dpool: torch.Tensor, argmax: torch.Tensor, relu_mask: torch.Tensor, y: torch.Tensor,
                fwd_sums: torch.Tensor, bwd_sums: torch.Tensor, gamma: torch.Tensor, dy: torch.Tensor,
                dgamma: torch.Tensor, dbeta: torch.Tensor, *, eps: float,
                workspace: Optional[torch.Tensor] = None, reduced: bool = False) -> torch.Tensor:
    """maxpool3x3s2_bwd + bn_bwd(relu=True, relu_mask=...) in two launches instead of three: the un-pooled gradient
    map is gathered inside the two BatchNorm passes instead of being written and read twice (the same gradient bits per
    pixel; on even maps the passes walk 2 x 2 pixel blocks, so the two sums are added in another order than bn_bwd's).
    dpool / argmax [B,Ho,Wo,C], y / dy [B,H,W,C], relu_mask the bits bn_apply_pool wrote; bwd_sums zero on entry
    (reduced: they already hold the two sums, only the apply pass runs)."""
    _need_cuda(dpool, argmax, relu_mask, y, fwd_sums, bwd_sums, gamma, dy, dgamma, dbeta, workspace)
    _check_bn_ws(workspace)
    if y.dim() != 4:
        raise ValueError("bn_bwd_pool: y must be [B,H,W,C]")
    B, H, W, C = y.shape
    M = B * H * W
    _check_relu_mask(relu_mask, M, C, True)
    if dpool.shape != argmax.shape or tuple(dpool.shape) != (B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C) \
            or dpool.dtype != BF16 or y.dtype != BF16 or dy.dtype != BF16 or argmax.dtype != torch.uint8 \
            or dy.shape != y.shape or not all(t.is_contiguous() for t in (dpool, argmax, y, dy)) or M >= 1 << 24:
        raise ValueError("bn_bwd_pool: bad operands")
    for v, n in ((fwd_sums, 2 * C), (bwd_sums, 2 * C), (gamma, C), (dgamma, C), (dbeta, C)):
        _vec_f32(v, n, "bn_bwd_pool: vector")
    lib = _lib.load()
    if not reduced:
        with _Span("bn_bwd_reduce", 0, _nbytes(dpool, argmax, y, relu_mask)):
            check(lib.rvit_bn_bwd_reduce_pool(dpool.data_ptr(), argmax.data_ptr(), relu_mask.data_ptr(), y.data_ptr(),
                                              fwd_sums.data_ptr(), B, H, W, C, eps, bwd_sums.data_ptr(), _ptr(workspace),
                                              _stream()), "rvit_bn_bwd_reduce_pool")
    with _Span("bn_bwd_apply", 0, _nbytes(dpool, argmax, y, relu_mask, dy)):
        check(lib.rvit_bn_bwd_apply_pool(dpool.data_ptr(), argmax.data_ptr(), relu_mask.data_ptr(), y.data_ptr(),
                                         fwd_sums.data_ptr(), bwd_sums.data_ptr(), gamma.data_ptr(), dy.data_ptr(),
                                         dgamma.data_ptr(), dbeta.data_ptr(), B, H, W, C, eps, _stream()),
              "rvit_bn_bwd_apply_pool")
    return dy


def maxpool3x3s2_bwd(dpool: torch.Tensor, argmax: torch.Tensor, dx: torch.Tensor) -> torch.Tensor:
    _need_cuda(dpool, argmax, dx)
    B, H, W, C = dx.shape
    if dpool.shape != argmax.shape or dpool.dtype != BF16 or dx.dtype != BF16 or argmax.dtype != torch.uint8 \
            or not (dpool.is_contiguous() and dx.is_contiguous() and argmax.is_contiguous()) \
            or tuple(dpool.shape) != (B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C):
        raise ValueError("maxpool3x3s2_bwd: bad operands")
    with _Span("maxpool_bwd", 0, _nbytes(dpool, argmax, dx)):
        check(_lib.load().rvit_maxpool3x3s2_bwd_nhwc(dpool.data_ptr(), argmax.data_ptr(), dx.data_ptr(), B, H, W, C,
                                                     _stream()), "rvit_maxpool3x3s2_bwd_nhwc")
    return dx


def sumsq(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[0] += sum(x^2) over a flat fp32 buffer (global grad-norm, train_vit.py:942-945); out is NOT cleared here."""
    _need_cuda(x, out)
    if x.dtype != torch.float32 or out.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("sumsq: contiguous float32 expected")
    with _Span("grad_sumsq", 0, x.numel() * 4):
        check(_lib.load().rvit_sumsq_f32(_ptr(x), x.numel(), _ptr(out), *_ows(out), _stream()), "rvit_sumsq_f32")
    return out


def adamw_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, *, lr: float,
               beta1: float, beta2: float, eps: float, weight_decay: float, step: int,
               grad_norm_sq: Optional[torch.Tensor] = None, max_grad_norm: float = 0.0,
               params_bf16: Optional[torch.Tensor] = None) -> None:
    """One fused clip + AdamW update over flat fp32 buffers (torch.optim.AdamW semantics, train_vit.py:879-883)."""
    _need_cuda(params, grads, exp_avg, exp_avg_sq, grad_norm_sq)
    n = params.numel()
    for t in (params, grads, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise ValueError("adamw_step: four contiguous float32 buffers of one length expected")
    if params_bf16 is not None and (params_bf16.dtype != torch.bfloat16 or params_bf16.numel() != n
                                    or not params_bf16.is_contiguous() or not params_bf16.is_cuda):
        raise ValueError("adamw_step: params_bf16 must be a contiguous bf16 buffer of the same length")
    with _Span("adamw_step", 0, n * (28 if params_bf16 is None else 30)):
        check(_lib.load().rvit_adamw_step(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), n, float(lr),
                                          float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                          _ptr(grad_norm_sq), float(max_grad_norm), _ptr(params_bf16), _stream()),
              "rvit_adamw_step")


def cast_bf16(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out = bf16(x) over flat contiguous buffers."""
    _need_cuda(x, out)
    if x.dtype != torch.float32 or out.dtype != torch.bfloat16 or x.numel() != out.numel() \
            or not x.is_contiguous() or not out.is_contiguous():
        raise ValueError("cast_bf16: contiguous float32 -> bfloat16 buffers of one length expected")
    with _Span("cast_bf16", 0, x.numel() * 6):
        check(_lib.load().rvit_cast_bf16(_ptr(x), _ptr(out), x.numel(), _stream()), "rvit_cast_bf16")
    return out


def transpose_table(entries, device) -> tuple:
    """Validates (src_off, dst_off, rows, cols, src_ld, dst_ld) tuples and lays out the device table of
    rvit_transpose_batched_bf16; returns (table int64 [n,8], n_entries, total_tiles, src_extent, dst_extent)."""
    rows_out, tile, ext_s, ext_d = [], 0, 0, 0
    for e in entries:
        src_off, dst_off, rows, cols, src_ld, dst_ld = (int(v) for v in e)
        if min(rows, cols) <= 0 or any(v % 8 for v in (src_off, dst_off, rows, cols, src_ld, dst_ld)) \
                or src_ld < cols or dst_ld < rows:
            raise ValueError(f"transpose_table: entry {e} is not 8-element aligned / consistent")
        tr, tc = (rows + 63) // 64, (cols + 63) // 64
        rows_out.append([src_off, dst_off, rows, cols, src_ld, dst_ld, tile, tc])
        tile += tr * tc
        ext_s = max(ext_s, src_off + (rows - 1) * src_ld + cols)
        ext_d = max(ext_d, dst_off + (cols - 1) * dst_ld + rows)
    if not rows_out:
        raise ValueError("transpose_table: empty")
    return torch.tensor(rows_out, dtype=torch.int64).to(device), len(rows_out), tile, ext_s, ext_d


def transpose_batched(src: torch.Tensor, dst: torch.Tensor, table) -> torch.Tensor:
    """All transposes of `table` (from transpose_table) in one launch; src/dst flat bf16 buffers."""
    tab, n, tiles, ext_s, ext_d = table
    _need_cuda(src, dst, tab)
    if src.numel() < ext_s or dst.numel() < ext_d:
        raise ValueError("transpose_batched: table reaches beyond the buffers")
    if src.dtype != torch.bfloat16 or dst.dtype != torch.bfloat16 or not src.is_contiguous() or not dst.is_contiguous():
        raise ValueError("transpose_batched: contiguous bf16 buffers expected")
    with _Span("transpose_batched", 0, 0):
        check(_lib.load().rvit_transpose_batched_bf16(_ptr(src), _ptr(dst), _ptr(tab), n, tiles, _stream()),
              "rvit_transpose_batched_bf16")
    return dst


def preprocess_u8(frames: torch.Tensor, tables_h, tables_v, resized: int, out: torch.Tensor, tmp: torch.Tensor,
                  crop_tl: Optional[torch.Tensor] = None, noise_std: float = 0.0, noise_seed: int = 0) -> torch.Tensor:
    """uint8 frames [B,H,W,C] (or [B,H,W]) -> normalised fp32 [B,3,S,S] (rvit_preprocess_u8).  tables_* =
    (ksize, bounds int32 [resized,2], coefficients int32 [resized,ksize]) on the device."""
    _need_cuda(frames, out, tmp, crop_tl, tables_h[1], tables_h[2], tables_v[1], tables_v[2])
    if frames.dtype != torch.uint8 or not frames.is_contiguous() or frames.dim() not in (3, 4):
        raise ValueError("preprocess_u8: contiguous uint8 [B,H,W,C] or [B,H,W] frames expected")
    B, H, W = frames.shape[:3]
    C = frames.shape[3] if frames.dim() == 4 else 1
    S = out.shape[-1]
    if out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape) != (B, 3, S, S):
        raise ValueError("preprocess_u8: out must be contiguous float32 [B,3,S,S]")
    if tmp.dtype != torch.uint8 or not tmp.is_contiguous() or tmp.numel() < B * H * S:
        raise ValueError("preprocess_u8: tmp must hold B*H*S bytes")
    for (ks, bounds, coef), n_in in ((tables_h, W), (tables_v, H)):
        if bounds.dtype != torch.int32 or coef.dtype != torch.int32 or tuple(bounds.shape) != (resized, 2) \
                or tuple(coef.shape) != (resized, ks) or not bounds.is_contiguous() or not coef.is_contiguous():
            raise ValueError("preprocess_u8: malformed resampling tables")
    if crop_tl is not None and (crop_tl.dtype != torch.int32 or tuple(crop_tl.shape) != (B, 2) or not crop_tl.is_contiguous()):
        raise ValueError("preprocess_u8: crop_tl must be int32 [B,2] (top, left)")
    with _Span("preprocess_u8", 0, frames.numel() + 2 * B * H * S + out.numel() * 4):
        check(_lib.load().rvit_preprocess_u8(_ptr(frames), B, H, W, C, _ptr(tables_h[2]), _ptr(tables_h[1]), tables_h[0],
                                             _ptr(tables_v[2]), _ptr(tables_v[1]), tables_v[0], resized, _ptr(crop_tl), S,
                                             _ptr(tmp), float(noise_std), int(noise_seed) & ((1 << 64) - 1), _ptr(out),
                                             _stream()), "rvit_preprocess_u8")
    return out


PREPROCESS_STATIC_MAX_BYTES = 32 << 20   # RVIT_PREPROCESS_STATIC_MAX_BYTES


def preprocess_u8_static(frames: torch.Tensor, tables_h, tables_v, resized: int, out: torch.Tensor,
                         crop_tl: Optional[torch.Tensor] = None, noise_std: float = 0.0, noise_seed: int = 0) -> torch.Tensor:
    """preprocess_u8 through the entry point that allocates nothing (rvit_preprocess_u8_static): the intermediate
    lives in the library's static device array; B*H*S beyond 32 MiB is refused by the library.  Calls that overlap
    on one GPU race on that array."""
    _need_cuda(frames, out, crop_tl, tables_h[1], tables_h[2], tables_v[1], tables_v[2])
    if frames.dtype != torch.uint8 or not frames.is_contiguous() or frames.dim() not in (3, 4):
        raise ValueError("preprocess_u8_static: contiguous uint8 [B,H,W,C] or [B,H,W] frames expected")
    B, H, W = frames.shape[:3]
    C = frames.shape[3] if frames.dim() == 4 else 1
    S = out.shape[-1]
    if out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape) != (B, 3, S, S):
        raise ValueError("preprocess_u8_static: out must be contiguous float32 [B,3,S,S]")
    for (ks, bounds, coef), n_in in ((tables_h, W), (tables_v, H)):
        if bounds.dtype != torch.int32 or coef.dtype != torch.int32 or tuple(bounds.shape) != (resized, 2) \
                or tuple(coef.shape) != (resized, ks) or not bounds.is_contiguous() or not coef.is_contiguous():
            raise ValueError("preprocess_u8_static: malformed resampling tables")
    if crop_tl is not None and (crop_tl.dtype != torch.int32 or tuple(crop_tl.shape) != (B, 2) or not crop_tl.is_contiguous()):
        raise ValueError("preprocess_u8_static: crop_tl must be int32 [B,2] (top, left)")
    with torch.cuda.device(frames.device):   # the library uses the static array of the CURRENT device
        with _Span("preprocess_u8", 0, frames.numel() + 2 * B * H * S + out.numel() * 4):
            check(_lib.load().rvit_preprocess_u8_static(
                _ptr(frames), B, H, W, C, _ptr(tables_h[2]), _ptr(tables_h[1]), tables_h[0], _ptr(tables_v[2]),
                _ptr(tables_v[1]), tables_v[0], resized, _ptr(crop_tl), S, float(noise_std),
                int(noise_seed) & ((1 << 64) - 1), _ptr(out), _stream()), "rvit_preprocess_u8_static")
    return out
