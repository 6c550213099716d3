"""Diagnostic: per-section times of the eight-wave conv kernel's tap (csrc/conv3x3_p8.hip, -DRVIT_C8_STAMPS build in
vit_amd/librvit_stamps.so): R1 (reads) | barrier | M1 (32 MFMAs) | barrier | R2 (reads + DMA) | barrier | M2 (24 MFMAs) |
barrier, for wave 0 (group 0) and wave 4 (group 1).  s_memtime ticks (100 MHz) are converted with the kernel's duration."""
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
for name, c, hw in (("layer2", 128, 28), ("layer3", 256, 14), ("layer4", 512, 7)):
    x = torch.randn(B, hw, hw, c, device=dev).bfloat16()
    w = (torch.randn(c, 9 * c, device=dev) / (3 * c ** 0.5)).bfloat16()
    scale, bias = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    out = torch.empty_like(x)
    for _ in range(3):
        ops.conv2d(x, w, out, kh=3, kw=3, stride=1, pad=1, tile=19, scale=scale, bias=bias, act="relu")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d(x, w, out, kh=3, kw=3, stride=1, pad=1, tile=19, scale=scale, bias=bias, act="relu")
    e1.record()
    torch.cuda.synchronize()
    n = 256
    buf = np.zeros(n * 24, dtype=np.uint64)
    rc = lib.rvit_debug_c8_stamps(buf.ctypes.data_as(ctypes.c_void_p), n * 24)
    assert rc == 0, rc
    raw = buf.reshape(n, 2, 12).astype(np.int64)
    taps = 9 * (c // 64)
    us = e0.elapsed_time(e1) * 1e3
    print(f"{name}: kernel {us:.1f} us (stamped build); per tap, median over the first {n} workgroups, in ns (s_memtime = 100 MHz):")
    for g in (0, 1):
        a = np.median(raw[:, g, :8], axis=0) / taps * 10.0
        loop = np.median(raw[:, g, 9] - raw[:, g, 8]) * 10.0
        total = np.median(raw[:, g, 10] - raw[:, g, 8]) * 10.0
        print(f"   group {g}: R1 {a[0]:.0f} bar {a[1]:.0f} M1 {a[2]:.0f} bar {a[3]:.0f} R2 {a[4]:.0f} bar {a[5]:.0f} M2 {a[6]:.0f} bar {a[7]:.0f}"
              f"  sum {a.sum():.0f} ns/tap; loop {loop / 1e3:.1f} us, loop+epilogue {total / 1e3:.1f} us")
