"""Developer aid: bf16 GEMM tile configurations side by side (us per launch, TFLOP/s) with a cross-check of the results.
usage: bench_gemm_tiles.py [tiles, default 1,4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 4]
dev = torch.device("cuda:0")
bf = torch.bfloat16
ACT = os.environ.get("ACT") or None


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


T, TB = 512 * 50, 256 * 197
gemms = [("R qkv", T, 768, 256), ("R out", T, 256, 256), ("R fc1", T, 1024, 256), ("R fc2", T, 256, 1024),
         ("vitB qkv", TB, 2304, 768), ("vitB out", TB, 768, 768), ("vitB fc1", TB, 3072, 768), ("vitB fc2", TB, 768, 3072),
         ("vitL fc1", TB, 4096, 1024), ("big 4096^3", 4096, 4096, 4096), ("big 8192^3", 8192, 8192, 8192)]
if os.environ.get("ONLY"):
    gemms = [g for g in gemms if g[0].startswith(os.environ["ONLY"])]
for name, M, N, K in gemms:
    a = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bi = torch.randn(N, device=dev)
    ref = None
    for tile in tiles:
        o = torch.empty(M, N, device=dev, dtype=bf)
        us = timeit(lambda: ops.gemm(a, w, o, bias=bi, act=ACT, tile=tile))
        if ref is None:
            ref = o
            diff = 0.0
        else:
            diff = (o.float() - ref.float()).abs().max().item()
        print(f"{name:12s} {M}x{N}x{K} tile={tile} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TF/s  maxdiff vs first {diff:.3g}", flush=True)
