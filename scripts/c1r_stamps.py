"""Diagnostic: per-phase cycle sums (s_memtime, wave 0) of the rolling-band stem entry kernel over the bands of each
workgroup, from a -DRVIT_C1_STAMPS build (scripts/build_stamps.sh conv1 RVIT_C1_STAMPS).  python scripts/c1r_stamps.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vit_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "vit_amd", "librvit_stamps.so")
from vit_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 512))
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
st = buf.reshape(n, 8).astype(np.int64)
st = st[st[:, 1] > 0]
nb = 28 * B // len(st) if len(st) else 0
names = ["prefetch issue", "late epilogue", "MFMA loop", "early epilogue", "ring write", "barrier"]
print(f"{len(st)} workgroups, {nb} bands each; cycles per band (wave 0), median / p90 over workgroups:")
for i, nm in enumerate(names):
    v = st[:, i] / max(nb, 1)
    print(f"  {nm:26s} {np.median(v):8.0f} / {np.percentile(v, 90):8.0f}")
print(f"  sum {np.median(st[:, :6].sum(axis=1)) / max(nb, 1):.0f} per band")
