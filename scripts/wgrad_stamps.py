"""Where wave 0 of a workgroup of the sixteen-wave weight-gradient kernel spends its M loop (ViT-B linear1 shape), from
a -DRVIT_WG_STAMPS build of csrc/wgrad.hip linked as vit_amd/librvit_alt.so.  s_memtime ticks = shader cycles."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib, ops  # noqa: E402

_lib.LIB_PATH = os.path.abspath("vit_amd/librvit_alt.so")


def main():
    dev = torch.device("cuda:0")
    M, N, K = 256 * 197, 3072, 768
    g = torch.Generator().manual_seed(0)
    dy = torch.randn(M, N, generator=g).to(dev).to(torch.bfloat16)
    x = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    for _ in range(3):
        ops.linear_wgrad_bias(dy, x, dw, db)
    torch.cuda.synchronize()
    n = 252 * 4
    buf = (ctypes.c_ulonglong * n)()
    assert _lib.load().rvit_debug_wgrad_stamps(buf, n) == 0
    st = np.array(buf, dtype=np.float64).reshape(252, 4)
    nmt = st[:, 3].mean()
    print(f"M tiles per workgroup {nmt:.0f}; cycles per 64-row tile: total {(st[:, 2] / st[:, 3]).mean():.0f}, "
          f"counted wait {(st[:, 0] / st[:, 3]).mean():.0f}, barrier {(st[:, 1] / st[:, 3]).mean():.0f}, "
          f"reads + MFMAs + DMA issue {((st[:, 2] - st[:, 0] - st[:, 1]) / st[:, 3]).mean():.0f} (2,048 of matrix pipe per SIMD)")


if __name__ == "__main__":
    main()
