"""Developer aid: direct (LDS-resident patch) vs implicit-GEMM conv weight gradient, per stem shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import _lib, ops
if os.environ.get("RVIT_BENCH_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["RVIT_BENCH_LIB"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, hw, c in (("l1", 56, 64), ("l2", 28, 128), ("l3", 14, 256), ("l4", 7, 512)):
    x = torch.randn(B, hw, hw, c, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, hw, hw, c, device=dev).to(torch.bfloat16)
    dw = torch.zeros(c, 9 * c, device=dev)
    fl = 2.0 * B * hw * hw * c * 9 * c
    for label, sp in (("gather", -1), ("direct", -2)):
        us = timeit(lambda: ops.conv2d_wgrad(dy, x, dw, kh=3, kw=3, stride=1, pad=1, splits=sp))
        print(f"{name} {c:3d}ch {hw:2d}x{hw:<2d} {label}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
