"""Developer aid: per-workgroup timeline of the 8-phase GEMM (a build of csrc/gemm8p.hip with -DP8_STAMPS).
usage: RVIT_G8_STAMPS=/tmp/st.bin python scripts/g8_stamps.py M N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vit_amd import ops

M, N, K = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
path = os.environ["RVIT_G8_STAMPS"]
for _ in range(3):
    ops.gemm(a, w, o, tile=5)
torch.cuda.synchronize()
st = np.fromfile(path, dtype=np.uint64).reshape(-1, 16)
t = st[:, :4].astype(np.int64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0            # us (100 MHz counter)
hw = st[:, 4]
cu_key = hw                      # xcc id << 32 | HW_ID (se, sh, cu, simd, wave bits): mask the wave / simd bits
cu_key = (hw >> np.uint64(32)) * np.uint64(1 << 16) + ((hw & np.uint64(0xffffffff)) >> np.uint64(8) & np.uint64(0xff)) + (((hw & np.uint64(0xffffffff)) >> np.uint64(13)) & np.uint64(0x7)) * np.uint64(256)
print(f"tiles {len(t)}  kernel span {t[:,3].max():.1f} us  distinct CUs {len(np.unique(cu_key))}")
print(f"prologue  (start -> first barrier): mean {np.mean(t[:,1]-t[:,0]):.2f} us  p10 {np.percentile(t[:,1]-t[:,0],10):.2f} p90 {np.percentile(t[:,1]-t[:,0],90):.2f}")
print(f"K loop    : mean {np.mean(t[:,2]-t[:,1]):.2f} us  p10 {np.percentile(t[:,2]-t[:,1],10):.2f} p90 {np.percentile(t[:,2]-t[:,1],90):.2f}  per K tile {np.mean(t[:,2]-t[:,1])/(K//64):.3f}")
print(f"epilogue  : mean {np.mean(t[:,3]-t[:,2]):.2f} us  p10 {np.percentile(t[:,3]-t[:,2],10):.2f} p90 {np.percentile(t[:,3]-t[:,2],90):.2f}")
gaps = []
for k in np.unique(cu_key):
    idx = np.where(cu_key == k)[0]
    idx = idx[np.argsort(t[idx, 0])]
    gaps += list(t[idx[1:], 0] - t[idx[:-1], 3])
gaps = np.array(gaps if gaps else [0.0])
print(f"gap between workgroups on one CU: mean {gaps.mean():.2f} us  p10 {np.percentile(gaps,10):.2f} p90 {np.percentile(gaps,90):.2f}  (n={len(gaps)})")
order = np.argsort(t[:, 0])
print("first workgroups by start time (start, prologue end, loop end, end):")
for i in order[:4]:
    print("   ", np.round(t[i], 2))
# how synchronised are the rounds: histogram of epilogue-start times
h, e = np.histogram(t[:, 2], bins=40)
print("epilogue starts per time bin:", h.tolist())
ep = t[:, 3] - t[:, 2]
o = np.argsort(t[:, 2])
n = len(o)
for lo, hi in ((0, min(256, n)), (max(n - 64, 0), n)):
    sel = o[lo:hi]
    print(f"epilogue of workgroups {lo}..{hi} by epilogue start: mean {ep[sel].mean():.2f} us, prologue {np.mean(t[sel,1]-t[sel,0]):.2f}, K loop {np.mean(t[sel,2]-t[sel,1]):.2f}")
x = (st[:, 5:8].astype(np.int64) - t0) / 100.0
print(f"epilogue split: loop end -> cols loaded {np.mean(x[:,0]-t[:,2]):.2f} us, first fragment row {np.mean(x[:,1]-x[:,0]):.2f}, rows 1..7 {np.mean(x[:,2]-x[:,1]):.2f}, final drain {np.mean(t[:,3]-x[:,2]):.2f}")
cy = st[:, 8:16].astype(np.int64)
def mhz(a, b, wa, wb):
    return np.mean((cy[:, b] - cy[:, a]) / np.maximum((st[:, wb].astype(np.int64) - st[:, wa].astype(np.int64)) / 100.0, 1e-3))
print(f"s_memtime ticks per us: K loop {mhz(1,2,1,2):.0f}, epilogue rows 1..7 {mhz(6,7,6,7):.0f}; ticks per fragment row {np.mean(cy[:,7]-cy[:,6])/7:.0f}")
