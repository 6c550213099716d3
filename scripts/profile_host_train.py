"""Where the host time of a small-batch training step goes (cProfile over the enqueue of N steps).
Usage on the GPU box: PYTHONPATH=. python scripts/profile_host_train.py [batch] [steps]"""
import cProfile
import pstats
import sys
import time

import torch

from vit_amd.loss import CrossEntropyLoss
from vit_amd.model import CNNViTHybrid
from vit_amd.optim import FusedAdamW
from vit_amd.synth import micro_doppler_batch, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
cfg = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone="resnet18")
model = CNNViTHybrid(**cfg)
model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed=0))
model = model.to(dev).train()
x = micro_doppler_batch(B, seed=5).to(dev)
y = torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(0)).to(dev)
opt = FusedAdamW(model, lr=1e-4, weight_decay=0.05, max_grad_norm=1.0)
crit = CrossEntropyLoss(label_smoothing=0.05)


def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model(x), y)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, with the final sync {1e3 * (t2 - t0) / N:.3f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
