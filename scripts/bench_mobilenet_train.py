"""Developer aid: training step time of the mobilenetv3 hybrid (forward in train mode, CE loss, backward, FusedAdamW).
usage: bench_mobilenet_train.py [batch, default 256] [backbone, default mobilenetv3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd.loss import cross_entropy
from vit_amd.model import CNNViTHybrid
from vit_amd.optim import FusedAdamW
from vit_amd.synth import micro_doppler_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
backbone = sys.argv[2] if len(sys.argv) > 2 else "mobilenetv3"
dev = torch.device("cuda:0")
m = CNNViTHybrid(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone=backbone).to(dev).train()
opt = FusedAdamW(m, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)
x = micro_doppler_batch(min(B, 64), seed=1).to(dev).repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous()
y = (torch.arange(B) % 11).to(dev)


def step():
    opt.zero_grad()
    loss = cross_entropy(m(x), y, label_smoothing=0.05)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{backbone} B={B}: {dt * 1e3:.2f} ms/step  {B / dt:.0f} samples/s  loss {loss.item():.3f}")
