"""One direct-conv configuration, a few launches: target for rocprofv3 --pmc.  python scripts/c3_one.py layer tile [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops  # noqa: E402

layer, tile = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
c, hw = {"layer1": (64, 56), "layer2": (128, 28), "layer3": (256, 14), "layer4": (512, 7)}[layer]
B = 512
dev = torch.device("cuda:0")
x = torch.randn(B, hw, hw, c, device=dev).bfloat16()
w = (torch.randn(c, 9 * c, device=dev) / (3 * c ** 0.5)).bfloat16()
scale, bias = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
out = torch.empty_like(x)
for _ in range(iters):
    ops.conv2d(x, w, out, kh=3, kw=3, stride=1, pad=1, tile=tile, scale=scale, bias=bias, act="relu")
torch.cuda.synchronize()
