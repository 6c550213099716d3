"""Developer aid: MX-fp8 GEMM on the ViT-L / ViT-B Linear shapes (us per launch, TFLOP/s), epilogue forms of the encoder.
Products of >= 512 tiles of 256 x 256 run the eight-wave kernel (csrc/gemm_mx8p.hip); ROWS=<n> limits M (e.g. 16384 keeps
every shape on the 256 x 64 kernel of csrc/gemm_mx.hip for an A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = int(os.environ.get("ROWS", 256 * 197))
for name, N, K, form in (("vitL qkv", 3072, 1024, "bias"), ("vitL out", 1024, 1024, "res"), ("vitL fc1", 4096, 1024, "gelu_mx"),
                         ("vitL fc2", 1024, 4096, "res"), ("vitB fc1", 3072, 768, "gelu_mx"), ("vitB fc2", 768, 3072, "res"),
                         ("square", 8192, 8192, "bias")):
    m = 8192 if name == "square" else M
    a = torch.randn(m, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    aq, as_ = ops.quantize_mx(a)
    wq, ws = ops.quantize_mx(w)
    if form == "bias":
        out = torch.empty(m, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.gemm_mx(aq, as_, wq, ws, out, bias=bias)  # noqa: E731
    elif form == "res":
        out = torch.zeros(m, N, device=dev)
        fn = lambda: ops.gemm_mx(aq, as_, wq, ws, out, bias=bias, residual=out)  # noqa: E731
    else:
        oq = torch.empty(m, N, dtype=torch.uint8, device=dev)
        osc = torch.zeros(ops.mx_scale_shape(m, N), dtype=torch.uint8, device=dev)
        fn = lambda: ops.gemm_mx(aq, as_, wq, ws, None, bias=bias, act="gelu", out_q=oq, out_s=osc)  # noqa: E731
    us = timeit(fn)
    print(f"{name:9s} {m}x{N}x{K} {form:8s} {us:9.1f} us  {2.0*m*N*K/us/1e6:8.1f} TF/s", flush=True)
