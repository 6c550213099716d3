"""Conv kernels timed per launch with the caches flushed between launches (a 1 GB fill), next to back-to-back timing:
how much of the gap between isolated and in-model kernel times is cache state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
dev = torch.device("cuda:0")
B = 512
junk = torch.empty(256 << 20, dtype=torch.float32, device=dev)
for name, c, hw in (("layer2", 128, 28), ("layer3", 256, 14), ("layer4", 512, 7)):
    x = torch.relu(torch.randn(B, hw, hw, c, device=dev)).bfloat16()
    w = (torch.randn(c, 9 * c, device=dev) / (3 * c ** 0.5)).bfloat16()
    res = torch.randn(B, hw, hw, c, device=dev).bfloat16()
    scale, bias = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    out = torch.empty_like(x)
    for tile in (11, 18 if c == 512 else 11, 19):
        for mode in ("warm", "cold", "producer"):
            ts = []
            for it in range(12):
                if mode == "cold":
                    junk.fill_(1.0)
                elif mode == "producer":   # the input was just written by another kernel (as in the model)
                    junk.fill_(1.0)
                    x.copy_(torch.relu(x))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.conv2d(x, w, out, kh=3, kw=3, stride=1, pad=1, tile=tile, scale=scale, bias=bias, residual=res, act="relu")
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[2:])
            print(f"{name} tile={tile} {mode:8s}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}", flush=True)
