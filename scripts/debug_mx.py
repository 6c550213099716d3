import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
dev = torch.device("cuda:0")
M, N, K = 256, 64, 128
def fp8(x): return x.to(torch.float8_e4m3fn).view(torch.uint8)
a = torch.zeros(M, K); a[torch.arange(128), torch.arange(128)] = 1.0
w = torch.ones(N, K)
sa = torch.zeros(M, 4, dtype=torch.uint8); sa[:, 0] = 127; sa[:, 1] = 128; sa[:, 2] = 129; sa[:, 3] = 130
sw = torch.full((N, 4), 127, dtype=torch.uint8)
out = torch.empty(M, N, device=dev)
ops.gemm_mx(fp8(a).to(dev), sa.to(dev), fp8(w).to(dev), sw.to(dev), out)
blk = torch.log2(out[:128, 0].cpu()).round().int().tolist()
print("A: scale block applied to byte k (k=0..127):")
print(blk)
# same for W: one-hot W rows, A all ones
a2 = torch.ones(M, K); w2 = torch.zeros(N, K); 
res = []
for half in range(2):
    w2.zero_(); w2[torch.arange(64), torch.arange(64) + 64 * half] = 1.0
    sw2 = torch.zeros(N, 4, dtype=torch.uint8); sw2[:, 0] = 127; sw2[:, 1] = 128; sw2[:, 2] = 129; sw2[:, 3] = 130
    sa2 = torch.full((M, 4), 127, dtype=torch.uint8)
    ops.gemm_mx(fp8(a2).to(dev), sa2.to(dev), fp8(w2).to(dev), sw2.to(dev), out)
    res += torch.log2(out[0, :64].cpu()).round().int().tolist()
print("W: scale block applied to byte k:")
print(res)
