"""Diagnostic: per-workgroup phase times of the direct 3x3 conv from s_memtime stamps (a separate build of librvit with
-DRVIT_C3_STAMPS, vit_amd/librvit_stamps.so; see csrc/conv3x3.hip C3_STAMP).  Phases: set-up (descriptors, prologue
DMA issue), tap loop, epilogue.  python scripts/c3_stamps.py [tile]"""
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

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 11
dev = torch.device("cuda:0")
B = 512
lib = _lib.load()
for name, c, hw in (("layer2", 128, 28), ("layer3", 256, 14), ("layer4", 512, 7)):
    if tile == 15 and c < 256:
        continue
    x = torch.randn(B, hw, hw, c, device=dev).bfloat16()
    w = (torch.randn(c, 9 * c, device=dev) / (3 * c ** 0.5)).bfloat16()
    scale, bias = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    out = torch.empty_like(x)
    for _ in range(3):
        ops.conv2d(x, w, out, kh=3, kw=3, stride=1, pad=1, tile=tile, scale=scale, bias=bias, act="relu")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d(x, w, out, kh=3, kw=3, stride=1, pad=1, tile=tile, scale=scale, bias=bias, act="relu")
    e1.record()
    torch.cuda.synchronize()
    n = min(8192, (B * hw * hw // 256) * (c // (256 if tile == 15 else 64)))
    buf = np.zeros(n * 8, dtype=np.uint64)
    rc = lib.rvit_debug_c3_stamps(buf.ctypes.data_as(ctypes.c_void_p), n * 8)
    assert rc == 0, rc
    raw = buf.reshape(n, 8).astype(np.int64)
    st = raw[:, :4]
    d = np.diff(st, axis=1)            # set-up, tap loop, epilogue in shader cycles (s_memtime), per workgroup (wave 0)
    us = e0.elapsed_time(e1) * 1e3
    med = lambda v: float(np.median(v))  # noqa: E731
    taps = 9 * (c // 64)
    print(f"{name} tile={tile}: kernel {us:.1f} us (single launch, stamped build), {n} workgroups; cycles per workgroup "
          f"(median): set-up {med(d[:, 0]):.0f}  tap loop {med(d[:, 1]):.0f} ({med(d[:, 1]) / taps:.0f} per tap: "
          f"vmcnt wait {med(raw[:, 4]) / taps:.0f}, barrier {med(raw[:, 5]) / taps:.0f}, rest "
          f"{(med(d[:, 1]) - med(raw[:, 4]) - med(raw[:, 5])) / taps:.0f})  epilogue {med(d[:, 2]):.0f}  total {med(st[:, 3] - st[:, 0]):.0f}")
