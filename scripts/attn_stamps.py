"""Per-workgroup timeline of the attention backward (ViT-B/16 shape) from a -DRVIT_ATTN_STAMPS build of the library:
    hipcc ... -DRVIT_ATTN_STAMPS -c attn.hip, linked as vit_amd/librvit_alt.so;  RVIT_BENCH_LIB=vit_amd/librvit_alt.so
The figures are s_memtime ticks of wave 0 = shader cycles on gfx950 (MI355X_MICROARCH.md), printed in kilocycles."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib, ops  # noqa: E402

_lib.LIB_PATH = os.path.abspath(os.environ.get("RVIT_BENCH_LIB", "vit_amd/librvit_alt.so"))


def main():
    dev = torch.device("cuda:0")
    B, S, H = 256, 197, 12
    E = H * 64
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B * S, 3 * E, generator=g) * 0.7).to(dev).to(torch.bfloat16)
    dout = torch.randn(B * S, E, generator=g).to(dev).to(torch.bfloat16)
    out = torch.empty(B * S, E, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H * S, device=dev)
    dqkv = torch.empty_like(qkv)
    lib = _lib.load()
    for p in (0.0, 0.1):
        ops.attention(qkv, out, B, S, H, lse=lse, drop_p=p, drop_seed=7)
        for _ in range(3):
            ops.attention_bwd(qkv, out, dout, lse, dqkv, B, S, H, drop_p=p, drop_seed=7)
        torch.cuda.synchronize()
        n = 3072 * 8
        buf = (ctypes.c_ulonglong * n)()
        rc = lib.rvit_debug_attn_stamps(buf, n)
        assert rc == 0, rc
        st = np.array(buf, dtype=np.int64).reshape(3072, 8)[:, :7]
        d = np.diff(st, axis=1).astype(np.float64) / 1000.0   # kilocycles
        names = ["stage loads+LDS stores", "barrier", "pass A sweep (wave 0)", "pass A stores", "pass B sweep (wave 0)", "pass B stores"]
        print(f"p={p}: workgroup life mean {(st[:, 6] - st[:, 0]).mean() / 1000.0:.2f} kcycles")
        for i, nm in enumerate(names):
            print(f"   {nm:28s} mean {d[:, i].mean():7.2f} kcyc   p10 {np.percentile(d[:, i], 10):7.2f}   p90 {np.percentile(d[:, i], 90):7.2f}")


if __name__ == "__main__":
    main()
