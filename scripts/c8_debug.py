"""Developer aid: where does the eight-wave conv kernel (tile 19) differ from the generic direct kernel (tile 11)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
dev = torch.device("cuda:0")
B, c, hw, tl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(0)
x = torch.randn(B, hw, hw, c, device=dev).bfloat16()
w = (torch.randn(c, 9 * c, device=dev) / (3 * c ** 0.5)).bfloat16()
o1, o2 = torch.empty_like(x), torch.empty_like(x)
ops.conv2d(x, w, o1, kh=3, kw=3, stride=1, pad=1, tile=11)
ops.conv2d(x, w, o2, kh=3, kw=3, stride=1, pad=1, tile=19)
torch.cuda.synchronize()
d = (o1.float() - o2.float()).abs().reshape(-1, c) > 0
print("differing fraction", d.float().mean().item())
M = d.shape[0]
pix = d.float().mean(1)          # per pixel
per_in_tile = torch.zeros(tl)
cnt = torch.zeros(tl)
idx = (torch.arange(M) % tl)
per_in_tile.index_add_(0, idx, pix.cpu())
cnt.index_add_(0, idx, torch.ones(M))
r = per_in_tile / cnt
print("by pixel-in-tile (groups of 14):", [round(float(r[i:i + 14].mean()), 2) for i in range(0, tl, 14)])
ch = d.float().mean(0).cpu()
print("by channel (groups of 16):", [round(float(ch[i:i + 16].mean()), 2) for i in range(0, c, 16)])
tiles = d.reshape(-1, tl, c).float().mean((1, 2)).cpu() if M % tl == 0 else None
if tiles is not None:
    print("by tile: first 16", [round(float(v), 2) for v in tiles[:16]], "min", float(tiles.min()), "max", float(tiles.max()))
