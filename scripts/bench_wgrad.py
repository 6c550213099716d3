"""Developer aid: conv weight-gradient timing per shape over split counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0"); bf = torch.bfloat16
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
convs = [("l1 3x3 64", 56, 64, 64, 3, 1, 56), ("l2 3x3 128", 28, 128, 128, 3, 1, 28), ("l3 3x3 256", 14, 256, 256, 3, 1, 14),
         ("l4 3x3 512", 7, 512, 512, 3, 1, 7)]
for name, H, cin, cout, k, s, ho in convs:
    dy = torch.randn(B, ho, ho, cout, device=dev).to(bf); x = torch.randn(B, H, H, cin, device=dev).to(bf)
    dw = torch.zeros(cout, k * k * cin, device=dev)
    fl = 2.0 * B * ho * ho * cout * k * k * cin
    for sp in [int(v) for v in sys.argv[2].split(",")]:
        us = timeit(lambda: ops.conv2d_wgrad(dy, x, dw, kh=k, kw=k, stride=s, pad=1, splits=sp))
        print(f"wgrad {name:14s} splits={sp:4d} {us:8.1f} us {fl/us/1e6:7.1f} TF/s", flush=True)
