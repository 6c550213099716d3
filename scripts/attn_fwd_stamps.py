"""Per-workgroup timeline of the attention forward (ViT-B/16 shape), wave 0, from a build of the library with
-DRVIT_ATTN_STAMPS -DRVIT_ATTN_STAMPS_FWD linked as vit_amd/librvit_alt.so (see scripts/attn_stamps.py for the backward).
s_memtime ticks = shader cycles; printed in kilocycles."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vit_amd import _lib, ops
_lib.LIB_PATH = os.path.abspath("vit_amd/librvit_alt.so")
dev = torch.device("cuda:0"); B, S, H = 256, 197, 12; E = H * 64
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * S, 3 * E, generator=g) * 0.7).to(dev).to(torch.bfloat16)
out = torch.empty(B * S, E, device=dev, dtype=torch.bfloat16); lse = torch.empty(B * H * S, device=dev)
lib = _lib.load()
for p in (0.0, 0.1):
    for _ in range(3): ops.attention(qkv, out, B, S, H, lse=lse, drop_p=p, drop_seed=7)
    torch.cuda.synchronize()
    n = 3072 * 8; buf = (ctypes.c_ulonglong * n)(); assert lib.rvit_debug_attn_stamps(buf, n) == 0
    st = np.array(buf, dtype=np.int64).reshape(3072, 8)
    d = np.diff(st, axis=1) / 1000.0
    names = ["issue DMA -> barrier passed", "Q load (block 1)", "S^T MFMAs", "max/exp/sum/dropout", "PV MFMAs", "store", "second block (all)"]
    print(f"p={p}: WG life {((st[:,7]-st[:,0]).mean()/1000):.2f} kcyc")
    for i, nm in enumerate(names): print(f"   {nm:30s} {d[:, i].mean():7.2f} kcyc  p10 {np.percentile(d[:, i],10):6.2f} p90 {np.percentile(d[:, i],90):6.2f}")
