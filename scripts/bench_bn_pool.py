"""Times the stem-entry BatchNorm passes of a training step (bn_apply_pool forward; the pooled backward reduce / apply).
PYTHONPATH=. python scripts/bench_bn_pool.py [batch]"""
import sys
import torch
from vit_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
H = W = 112
C = 64
y = (torch.randn(B, H, W, C, device=dev) * 1.3).bfloat16()
g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
ws = ops.bn_workspace(dev)
fs = torch.zeros(2 * C, device=dev)
ops.bn_stats(y, fs, ws)
pooled = torch.empty(B, 56, 56, C, dtype=torch.bfloat16, device=dev)
arg = torch.empty(B, 56, 56, C, dtype=torch.uint8, device=dev)
mask = torch.empty(B * H * W * C // 8, dtype=torch.uint8, device=dev)
dpool = torch.randn(B, 56, 56, C, device=dev).bfloat16()
dy = torch.empty_like(y)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)


def once():
    ops.bn_apply_pool(y, fs, g, b, pooled, arg, mask, eps=1e-5)
    bs = torch.zeros(2 * C, device=dev)
    ops.bn_bwd_pool(dpool, arg, mask, y, fs, bs, g, dy, dg, db, eps=1e-5, workspace=ws)


for _ in range(3):
    once()
rec = []
ops.set_profiler(rec)
for _ in range(10):
    once()
torch.cuda.synchronize()
ops.set_profiler(None)
tt = {}
for n, e0, e1, _, _ in rec:
    tt[n] = tt.get(n, 0.0) + e0.elapsed_time(e1) / 10 * 1e3
print(f"B={B}: " + "  ".join(f"{k} {v:.1f} us" for k, v in tt.items()), flush=True)
