"""Developer aid: 3x3 / stride-1 conv weight gradients of the resnet18 stem at a training batch, direct kernel (wgrad3x3.hip)
against the implicit-GEMM gather kernel (wgrad.hip, splits=-1).  PYTHONPATH=. python scripts/bench_wgrad3x3.py [B]"""
import sys
import torch
from vit_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ops.enable_ordered_reductions(dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for c, hw in ((64, 56), (128, 28), (256, 14), (512, 7)):
    x = torch.randn(B, hw, hw, c, device=dev).bfloat16()
    dy = (torch.randn(B, hw, hw, c, device=dev) / 8).bfloat16()
    dw = torch.zeros(c, 9 * c, device=dev)
    res = []
    for sp in (0, -1):
        res.append(t(lambda: ops.conv2d_wgrad(dy, x, dw, kh=3, kw=3, stride=1, pad=1, splits=sp)))
    print(f"B={B} {c:4d} ch {hw:3d}x{hw:<3d}: direct {res[0]:6.1f} us   implicit GEMM {res[1]:6.1f} us (incl. the ordered finalize)", flush=True)
