"""Times the four BatchNorm kernels per stem layer shape.  python scripts/bench_bn.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops  # noqa: E402


def bench(fn, iters=20):
    for _ in range(3):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    ws = ops.bn_workspace(dev) if len(sys.argv) > 2 and sys.argv[2] == "ordered" else None
    print("ordered reductions" if ws is not None else "float-atomic reductions")
    for name, hw, c in (("conv1", 112, 64), ("layer1", 56, 64), ("layer2", 28, 128), ("layer3", 14, 256), ("layer4", 7, 512)):
        y = torch.randn(B, hw, hw, c, device=dev).bfloat16()
        dz = torch.randn_like(y)
        z, dy, dzm = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
        mask = torch.empty(B * hw * hw * c // 8, dtype=torch.uint8, device=dev)
        g, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        fs, bs = torch.zeros(2 * c, device=dev), torch.zeros(2 * c, device=dev)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        mb = y.numel() * 2 / 1e6
        def once():
            ops.bn_stats(y, fs, ws)
            ops.bn_apply(y, fs, g, b, z, eps=1e-5, relu=True, relu_mask=mask)
            ops.bn_bwd(dz, None, y, fs, bs, g, dy, dg, db, eps=1e-5, relu=True, relu_mask=mask, workspace=ws)
        for _ in range(3):
            once()
        rec = []
        ops.set_profiler(rec)   # HIP events around every launch: GPU time, not host launch rate
        for _ in range(10):
            once()
        torch.cuda.synchronize()
        ops.set_profiler(None)
        tt = {}
        for n, e0, e1, _, _ in rec:
            tt[n] = tt.get(n, 0.0) + e0.elapsed_time(e1) / 10 * 1e3
        t1, t2, t3, t4 = tt["bn_stats"], tt["bn_apply"], tt["bn_bwd_reduce"], tt["bn_bwd_apply"]
        print(f"{name:7s} map {mb:7.1f} MB  stats {t1:6.1f} us ({mb / t1:5.2f} TB/s)  apply {t2:6.1f} us ({2.06 * mb / t2:5.2f})  "
              f"bwd_reduce {t3:6.1f} us ({2.06 * mb / t3:5.2f})  bwd_apply {t4:6.1f} us ({3.06 * mb / t4:5.2f})", flush=True)


if __name__ == "__main__":
    main()
