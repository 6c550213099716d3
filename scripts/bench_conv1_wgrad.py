"""Stem 7x7 conv weight gradient: rolling-band kernel from the fp32 input against the implicit-GEMM path on the packed
image.  python scripts/bench_conv1_wgrad.py [B ...]"""
import sys
import torch
from vit_amd import ops

dev = torch.device("cuda:0")
ops.enable_ordered_reductions(dev)
for B in [int(a) for a in sys.argv[1:]] or [32, 128, 256, 512]:
    x = torch.randn(B, 3, 224, 224, device=dev)
    dy = (torch.randn(B, 112, 112, 64, device=dev) / 64).bfloat16()
    img = torch.zeros(B, 230, 232, 4, dtype=torch.bfloat16, device=dev)
    ops.input_pack(x, img, 3, 3)
    dw = torch.zeros(64, 224, device=dev)

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

    a = t(lambda: ops.conv2d_wgrad(dy, img, dw, kh=7, kw=1, stride=2, pad=0, seg_len=32))
    b = t(lambda: ops.stem_conv1_wgrad_f32(dy, x, dw))
    print(f"B={B}: implicit-GEMM {a:.1f} us, rolling band {b:.1f} us (incl. the ordered finalize)", flush=True)
