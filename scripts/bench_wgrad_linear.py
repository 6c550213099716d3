"""Developer aid: Linear weight-gradient kernel variants side by side (splits -4 = 128x128 tiles, -3 = sixteen-wave
256x256 tiles, 0 = the dispatcher's choice), with a cross-check of the results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, n=20):
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


T, TB = 256 * 50, 256 * 197
shapes = [("R qkv", T, 768, 256), ("R fc1", T, 1024, 256), ("R fc2", T, 256, 1024),
          ("vitB qkv", TB, 2304, 768), ("vitB out", TB, 768, 768), ("vitB fc1", TB, 3072, 768), ("vitB fc2", TB, 768, 3072),
          ("vitL fc1", TB, 4096, 1024)]
for name, M, N, K in shapes:
    dy = torch.randn(M, N, device=dev).to(bf)
    x = torch.randn(M, K, device=dev).to(bf)
    ref = None
    for sp in (-4, -3, 0):
        dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        ops.linear_wgrad_bias(dy, x, dw, db, splits=sp)
        if ref is None:
            ref = (dw.clone(), db.clone())
            diff = 0.0
        else:
            diff = max(((dw - ref[0]).abs().max() / ref[0].abs().max()).item(), ((db - ref[1]).abs().max() / ref[1].abs().max()).item())
        us = timeit(lambda: ops.linear_wgrad_bias(dy, x, dw, db, splits=sp))
        print(f"{name:10s} {M}x{N}x{K} splits={sp:2d} {us:9.1f} us {2.0*M*N*K/us/1e6:8.1f} TF/s  rel diff vs first {diff:.2g}", flush=True)
