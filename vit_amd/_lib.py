"""ctypes binding of librvit.so (the C ABI declared in include/rvit.h).

There is exactly one compute path: the HIP library.  If it is missing or a call fails, this
module raises — there is no CPU or eager-PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch  # noqa: F401  -- must be imported BEFORE librvit.so is loaded: torch ships its own
#                              libamdhip64; loading ours first would bring up a second HIP runtime
#                              that shares no device context or streams with torch's.
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librvit.so")

RVIT_BF16, RVIT_F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_HSWISH, ACT_HSIGMOID = 0, 1, 2, 3, 4
GATE_NONE, GATE_GELU_GRAD, GATE_RELU_MASK = 0, 1, 2
ABI_VERSION = 29
DROP_PRE_RESIDUAL, DROP_POST_ACT = 1, 2


class RvitError(RuntimeError):
    """A librvit entry point returned a non-zero status."""


P, I, L, F, U64 = c_void_p, c_int, c_int64, c_float, c_uint64

# name -> (argtypes, restype), exactly mirroring include/rvit.h (tests/test_abi.py cross-checks it)
SIGNATURES = {
    "rvit_version": ([], c_int),
    "rvit_last_error": ([], c_char_p),
    "rvit_gemm_bf16": ([P, L, I, I, P, L, P, L, I, L, I, I, P, P, P, L, I, P, L, I, I, I, P, L, I, I, I, F, U64, I, P],
                       c_int),
    "rvit_gemm_mxfp8": ([P, L, P, P, L, P, P, L, I, L, I, I, P, P, L, I, I, P, L, P, P], c_int),
    "rvit_quantize_mxfp8": ([P, I, L, P, L, P, L, I, P], c_int),
    "rvit_conv2d_fwd_bf16": ([P, P, P] + [I] * 12 + [P, P, P, I, I, P], c_int),
    "rvit_conv3x3_kernel": ([I, I, I, I, I], c_int),
    "rvit_conv2d_dgrad_bf16": ([P, P, P] + [I] * 11 + [P, I, P], c_int),
    "rvit_conv2d_dgrad_bnr_bf16": ([P, P, P] + [I] * 11 + [P, P, P, P, c_float, P, L, P, P], c_int),
    "rvit_conv2d_dgrad_s2_bf16": ([P, P, P, I, I, I, I, I, P, P], c_int),
    "rvit_conv1x1_dgrad_s2_acc_bf16": ([P, P, P, I, I, I, I, I, P], c_int),
    "rvit_conv2d_tokens_fwd": ([P, P, P] + [I] * 12 + [P, P, F, U64, I, P], c_int),
    "rvit_wgrad_bf16": ([P, L, I, I, P, L, I, I, P, L, L, I, I, I, I, P, L, P], c_int),
    "rvit_linear_wgrad_bias_bf16": ([P, L, P, L, P, L, P, L, I, I, I, I, P, L, P], c_int),
    "rvit_conv2d_wgrad_bf16": ([P, P, P] + [I] * 12 + [I, P, L, P], c_int),
    "rvit_stem_conv1_pool_fwd": ([P, P, P, P, P, I, I, P], c_int),
    "rvit_stem_conv1_pool_f32_fwd": ([P, P, P, P, P, I, I, P], c_int),
    "rvit_input_pack_nhwc4": ([P, P] + [I] * 8 + [P], c_int),
    "rvit_maxpool3x3s2_nhwc": ([P, P, P] + [I] * 4 + [P], c_int),
    "rvit_embed_cls": ([P, P, P, I, I, I, F, U64, P], c_int),
    "rvit_dropout_apply": ([P, P, I, L, I, L, L, F, U64, P], c_int),
    "rvit_dropout_mask": ([P, L, F, U64, P], c_int),
    "rvit_layernorm_mxfp8_fwd": ([P, I, L, P, P, P, L, P, L, I, F, P], c_int),
    "rvit_layernorm_fwd": ([P, I, L, P, P, P, I, L, P, P, L, I, F, P], c_int),
    "rvit_encoder_mlp_fwd": ([P, P, P, F, P, P, P, P, L, I, I, I, P], c_int),
    "rvit_encoder_attn_fwd": ([P, P, P, F, P, P, P, P, I, I, I, I, P], c_int),
    "rvit_encoder_stack_fwd": ([P, P, I, F, I, I, I, I, I, P], c_int),
    "rvit_layernorm_bwd": ([P, L, P, L, P, P, P, P, L, I, P, L, P, P, L, I, P, L, P], c_int),
    "rvit_layernorm_bwd_drop": ([P, L, P, L, P, P, P, P, L, I, P, L, P, P, L, I, P, L, F, U64, P], c_int),
    "rvit_attn_fwd": ([P, P, P, I, I, I, L, L, F, F, U64, P], c_int),
    "rvit_cls_head_fwd": ([P, L, P, P, F, P, P, P, P, P, L, I, I, I, I, P], c_int),
    "rvit_cls_head_supported": ([I, I, I], c_int),
    "rvit_attn_dh_fwd": ([P, P, P, I, I, I, I, L, L, F, F, U64, P], c_int),
    "rvit_attn_dh_bwd": ([P, P, P, P, P, I, I, I, I, L, L, L, L, F, F, U64, P], c_int),
    "rvit_attn_fwd_fp8": ([P, P, L, P, I, I, I, L, F, P], c_int),
    "rvit_attn_fwd_mx": ([P, P, L, P, I, I, I, L, F, P], c_int),
    "rvit_stem_conv1_stats_f32_fwd": ([P, P, P, P, L, P, I, P], c_int),
    "rvit_stem_conv1_wgrad_f32": ([P, P, P, P, L, I, P], c_int),
    "rvit_conv2d_fwd_stats_bf16": ([P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P, L, P, I, P], c_int),
    "rvit_bn_stats_finalize": ([P, I, I, P, P, P], c_int),
    "rvit_dwconv_fwd_bf16": ([P, P, P, P, P, I, I, I, I, I, I, I, I, P], c_int),
    "rvit_avgpool_nhwc": ([P, P, I, I, I, P], c_int),
    "rvit_channel_scale_nhwc": ([P, P, I, I, I, P], c_int),
    "rvit_squeeze_excite_bf16": ([P, P, P, P, P, I, I, I, I, P], c_int),
    "rvit_bn_generic_stats": ([P, L, I, P, P, L, P], c_int),
    "rvit_bn_generic_apply": ([P, P, P, P, P, P, L, I, F, I, F, P, P, P], c_int),
    "rvit_bn_generic_bwd": ([P, P, P, P, P, P, P, P, P, L, I, F, I, P, L, P], c_int),
    "rvit_dwconv_dgrad_bf16": ([P, P, P, I, I, I, I, I, I, I, P], c_int),
    "rvit_dwconv_wgrad_bf16": ([P, P, P, I, I, I, I, I, I, I, P, L, P], c_int),
    "rvit_channel_affine_nhwc": ([P, P, P, P, I, I, I, P], c_int),
    "rvit_channel_dot_nhwc": ([P, P, P, P, I, I, I, P], c_int),
    "rvit_pointwise_conv_bf16": ([P, P, P, P, P, P, L, I, I, I, P], c_int),
    "rvit_pointwise_conv_supported": ([I, I], c_int),
    "rvit_conv3x3s2_rgb_f32_fwd": ([P, P, P, P, P, I, I, I, I, I, P], c_int),
    "rvit_attn_fwd_s64": ([P, P, I, I, I, L, L, F, P], c_int),
    "rvit_attn_bwd": ([P, P, P, P, P, I, I, I, L, L, L, L, F, F, U64, P], c_int),
    "rvit_ce_loss": ([P, L, P, P, P, L, P, L, I, I, I, F, F, P], c_int),
    "rvit_colsum_bf16": ([P, L, P, L, I, L, L, P, L, P], c_int),
    "rvit_batch_sum_f32": ([P, L, P, I, L, P], c_int),
    "rvit_sumsq_f32": ([P, L, P, P, L, P], c_int),
    "rvit_adamw_step": ([P, P, P, P, L, F, F, F, F, F, I, P, F, P, P], c_int),
    "rvit_cast_bf16": ([P, P, L, P], c_int),
    "rvit_lstm_step_fwd": ([P] * 13 + [I] * 5 + [P], c_int),
    "rvit_lstm_step_bwd": ([P] * 7 + [I] * 4 + [P], c_int),
    "rvit_gru_step_fwd": ([P] * 11 + [I] * 5 + [P], c_int),
    "rvit_gru_step_bwd": ([P] * 7 + [I] * 4 + [P], c_int),
    "rvit_preprocess_u8": ([P, I, I, I, I, P, P, I, P, P, I, I, P, I, P, F, U64, P, P], c_int),
    "rvit_preprocess_u8_static": ([P, I, I, I, I, P, P, I, P, P, I, I, P, I, F, U64, P, P], c_int),
    "rvit_transpose_batched_bf16": ([P, P, P, I, L, P], c_int),
    "rvit_bn_stats": ([P, L, I, P, P, P], c_int),
    "rvit_bn_apply": ([P, P, P, P, P, P, L, I, F, I, F, P, P, P, P], c_int),
    "rvit_bn_apply_pool": ([P, P, P, P, P, P, P, I, I, I, I, F, F, P, P, P], c_int),
    "rvit_bn_bwd_reduce": ([P, P, P, P, L, I, F, I, P, P, P], c_int),
    "rvit_bn_bwd_apply": ([P, P, P, P, P, P, P, P, P, P, L, I, F, I, P], c_int),
    "rvit_bn_bwd_reduce_pool": ([P, P, P, P, P, I, I, I, I, F, P, P, P], c_int),
    "rvit_bn_bwd_apply_pool": ([P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, P], c_int),
    "rvit_maxpool3x3s2_bwd_nhwc": ([P, P, P, I, I, I, I, P], c_int),
}

_lib = None
_tls = threading.local()   # .rec: the CallRecorder standing in for the library on this thread (vit_amd/plan.py)


class CallRecorder:
    """Stands in for the loaded library on ONE thread while a launch plan is recorded (vit_amd/plan.py): every
    entry point that takes a stream (the launches; the `*_supported` queries and rvit_last_error take none) is
    appended to `calls` as [function, [arguments], name] and then called as usual."""

    def __init__(self, lib):
        self._lib = lib
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        sig = SIGNATURES.get(name)
        if sig is None or not sig[0] or sig[0][-1] is not P:
            return fn
        calls = self.calls

        def recorded(*args):
            calls.append([fn, list(args), name])
            return fn(*args)
        return recorded


def record_calls(on: bool):
    """Start (-> the CallRecorder) or stop (-> None) recording this thread's launches."""
    _tls.rec = CallRecorder(load()) if on else None
    return _tls.rec


def load() -> ctypes.CDLL:
    """Load librvit.so once; raise loudly when it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        rec = getattr(_tls, "rec", None)
        return _lib if rec is None else rec
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"librvit.so not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C vit_amd/csrc`. vit_amd has no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype
    ver = lib.rvit_version()
    if ver != ABI_VERSION:
        raise ImportError(f"librvit.so ABI version {ver} != expected {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().rvit_last_error()
        raise RvitError(f"{what} failed with status {status}: {msg.decode() if msg else '?'}")
