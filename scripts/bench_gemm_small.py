"""Developer aid: the E = 256 encoder's Linear shapes at the training batch (T = B * 50 tokens) per GEMM tile config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
dev = torch.device("cuda:0")
T = int(os.environ.get("T", 12800))
def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N, K in [(768, 256), (256, 256), (1024, 256), (256, 1024), (256, 768)]:
    a = torch.randn(T, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    res = []
    for tile in (0, 1, 2, 3, 6, 7):
        us = timeit(lambda: ops.gemm(a, w, out, bias=bias, act="gelu" if N == 1024 else None, tile=tile))
        res.append(f"{tile}: {us:5.1f}")
    print(f"T={T} N={N:5d} K={K:5d}  " + "  ".join(res), flush=True)
