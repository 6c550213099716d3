"""Diagnostic: per-workgroup phase times (shader cycles, s_memtime) of the fused stem entry kernel, from a separate build
of librvit with -DRVIT_C1_STAMPS (vit_amd/librvit_stamps.so).  python scripts/c1_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vit_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "vit_amd", "librvit_stamps.so")
from vit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 512
lib = _lib.load()
x = torch.randn(B, 3, 224, 224, device=dev)
w = (torch.randn(64, 232, device=dev) / 12).bfloat16()
scale, bias = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
pooled = torch.empty(B, 56, 56, 64, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.stem_conv1_pool_f32(x, w, scale, bias, pooled)
torch.cuda.synchronize()
n = 16384
buf = np.zeros(n * 8, dtype=np.uint64)
assert lib.rvit_debug_c1_stamps(buf.ctypes.data_as(ctypes.c_void_p), n * 8) == 0
st = buf.reshape(n, 8)[: B * 28, :6].astype(np.int64)
d = np.diff(st, axis=1)
names = ["load+convert+wait", "MFMA loop", "barrier", "scale/bias -> LDS map + barrier", "pool + stores"]
print("stem entry (fp32 input), cycles per workgroup, median / p90:")
for i, nm in enumerate(names):
    print(f"  {nm:34s} {np.median(d[:, i]):8.0f} / {np.percentile(d[:, i], 90):8.0f}")
print(f"  total {np.median(st[:, 5] - st[:, 0]):.0f}")
