"""Premise check for a conv kernel on the eight-wave four-phase GEMM skeleton: plain GEMMs of the stem convs' implicit
shapes on gemm8p (tile 5) and the 256x64 kernel (tile 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=30):
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


for name, M, N, K in [("l3 as-is (392 tiles)", 100352, 256, 2304), ("l3 2 rounds of 256 rows", 131072, 256, 2304),
                      ("l2 4 rounds", 262144, 256, 1152), ("l4 1 round", 32768, 512, 4608), ("l4 as-is 196 tiles", 25088, 512, 4608),
                      ("l3 1 round", 65536, 256, 2304)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    for tile in (1, 4, 5):
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: ops.gemm(a, w, o, tile=tile))
        print(f"{name:26s} {M}x{N}x{K} tile={tile} {us:8.1f} us {2.0*M*N*K/us/1e6:7.0f} TF/s", flush=True)
