"""Developer aid: the stride-2 convs of the resnet18 stem (3x3 and 1x1 downsample) per gather-kernel tile config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
B = int(os.environ.get("B", 512))
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,4").split(",")]


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


for name, hw, cin, cout in [("layer2", 56, 64, 128), ("layer3", 28, 128, 256), ("layer4", 14, 256, 512)]:
    x = torch.randn(B, hw, hw, cin, device=dev).to(bf)
    for k in (3, 1):
        w = (torch.randn(cout, k * k * cin, device=dev) * 0.05).to(bf)
        sc = torch.ones(cout, device=dev)
        bi = torch.zeros(cout, device=dev)
        o = torch.empty(B, hw // 2, hw // 2, cout, device=dev, dtype=bf)
        ref = None
        for tile in tiles:
            try:
                us = timeit(lambda: ops.conv2d(x, w, o, kh=k, kw=k, stride=2, pad=k // 2, scale=sc, bias=bi, act="relu", tile=tile))
            except Exception as e:
                print(f"{name} {k}x{k}/2 tile={tile}: {e}")
                continue
            if ref is None:
                ref = o.clone()
            d = (o.float() - ref.float()).abs().max().item()
            fl = 2.0 * B * (hw // 2) ** 2 * cout * k * k * cin
            print(f"{name} {k}x{k}/2 {cin}->{cout} tile={tile}: {us:7.1f} us {fl / us / 1e6:7.1f} TF/s  maxdiff {d:.3g}", flush=True)
