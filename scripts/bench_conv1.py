"""Stem entry (csrc/conv1.hip): the rolling-band kernel against the band kernel on the packed input (bit equality for
every workgroups-per-image split) and its time per launch.  RVIT_CONV1_ROLL=0 times the band kernel instead."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vit_amd import _lib
if os.environ.get('RVIT_LIB'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vit_amd', os.environ['RVIT_LIB'])
from vit_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
w = torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)
wp = torch.zeros(64, 7, 8, 4)
wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)
w232 = torch.zeros(64, 232, dtype=torch.bfloat16)
w232[:, :224] = wp.to(torch.bfloat16).reshape(64, 224)
w232 = w232.to(dev)
scale = (torch.rand(64, generator=g) + 0.5).to(dev)
scale[5] = -0.7   # a negative folded-BN scale: the affine map must come before the maximum
bias = (torch.randn(64, generator=g) * 0.3).to(dev)
for B in ([1, 3, 37, 74, 130, 260, 512] if not os.environ.get('QUICK') else [3, 512]):
    x = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    xp = torch.zeros(B, 230, 232, 4, dtype=torch.bfloat16, device=dev)
    ops.input_pack(x, xp, 3, 3)
    ref = torch.empty(B, 56, 56, 64, dtype=torch.bfloat16, device=dev)
    ops.stem_conv1_pool(xp, w232, scale, bias, ref)
    out = torch.full_like(ref, float("nan"))
    ops.stem_conv1_pool_f32(x, w232, scale, bias, out)
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    nbad = (out.float() != ref.float()).sum().item() if not same else 0
    print(f"B={B}: equal={same} mismatches={nbad}", flush=True)
    if not same:
        d = (out.float() - ref.float()).abs()
        idx = torch.nonzero(d > 0)[:5]
        print(idx.tolist(), "nan:", torch.isnan(out.float()).sum().item())
B = 512
x = torch.randn(B, 3, 224, 224, generator=g).to(dev)
out = torch.empty(B, 56, 56, 64, dtype=torch.bfloat16, device=dev)
for _ in range(5): ops.stem_conv1_pool_f32(x, w232, scale, bias, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.stem_conv1_pool_f32(x, w232, scale, bias, out)
e1.record(); torch.cuda.synchronize()
print(f"B=512 stem_conv1_pool_f32 (RVIT_CONV1_ROLL={os.environ.get('RVIT_CONV1_ROLL', '1')} lib={os.environ.get('RVIT_LIB', 'librvit.so')}): {e0.elapsed_time(e1) / 50 * 1000:.1f} us per launch")
