"""Times the 3x3 / stride-1 stem convs per layer and kernel variant (tile 11 = generic direct kernel, 14 = weights
resident 64 -> 64 kernel, 0 = automatic).  python scripts/bench_conv3x3.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops  # noqa: E402


def bench(fn, iters=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda:0")
    for name, c, hw, tiles in (("layer1", 64, 56, (11, 14)), ("layer2", 128, 28, (11, 19)), ("layer3", 256, 14, (11, 15, 19)),
                               ("layer4", 512, 7, (11, 18, 19))):
        x = torch.randn(B, hw, hw, c, device=dev).bfloat16()
        w = (torch.randn(c, 9 * c, device=dev) / (3 * c ** 0.5)).bfloat16()
        res = torch.randn(B, hw, hw, c, device=dev).bfloat16()
        scale, bias = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        out = torch.empty_like(x)
        fl = 2.0 * B * hw * hw * c * c * 9
        ref_out = {}
        for tile in tiles:
            for with_res in (False, True):
                kw = dict(scale=scale, bias=bias, act="relu", residual=res if with_res else None)
                try:
                    us = bench(lambda: ops.conv2d(x, w, out, kh=3, kw=3, stride=1, pad=1, tile=tile, **kw))
                except Exception as e:  # noqa: BLE001
                    print(f"{name} tile={tile}: {str(e)[-90:]}")
                    continue
                if tile == 11:
                    ref_out[with_res] = out.clone()
                elif with_res in ref_out and not torch.equal(out, ref_out[with_res]):
                    d = (out.float() - ref_out[with_res].float()).abs()
                    print(f"   !! tile {tile} differs from tile 11: max {d.max().item():.4g}, {(d > 0).float().mean().item():.4f} of elements")
                nbytes = x.numel() * 2 * (3 if with_res else 2)
                print(f"{name} B={B} tile={tile} residual={int(with_res)}: {us:7.1f} us  {fl / us / 1e6:6.0f} TF/s  "
                      f"{nbytes / us / 1e6:5.2f} TB/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
