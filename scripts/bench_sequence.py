"""Developer aid: throughput of the sequence model (CNNViTSequence, reference defaults: resnet18 + E256/H4/L4,
bi-LSTM 256 x 2, sequence_length 8) - inference and one training step, frames/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd.sequence import CNNViTSequence, StreamingClassifier
dev = torch.device("cuda:0")
cfg = dict(backbone="resnet18", embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, temporal_hidden=256, temporal_layers=2,
           temporal_dropout=0.3, temporal_bidirectional=True, classifier_dropout=0.4, sequence_length=8, mode="sequence")
torch.manual_seed(0)
m = CNNViTSequence(11, cfg).to(dev)
def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for B in (1, 16, 64):
    x = torch.randn(B, 8, 3, 224, 224, device=dev)
    m.eval()
    with torch.no_grad():
        dt = timed(lambda: m(x), 20)
        f = m.spatial_encoder.forward_features(x.view(B * 8, 3, 224, 224)).view(B, 8, -1)
        dtt = timed(lambda: m.forward_temporal_only(f), 20)
    print(f"inference B={B:3d} T=8: {dt*1e3:7.3f} ms  {B*8/dt:9.0f} frames/s  (temporal encoder + classifier alone {dtt*1e3:.3f} ms)", flush=True)
B = 16
x = torch.randn(B, 8, 3, 224, 224, device=dev); y = torch.randint(0, 11, (B,), device=dev)
m.train(); opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(m(x), y).backward()
    opt.step()
dt = timed(step, 10)
print(f"training  B={B:3d} T=8: {dt*1e3:7.3f} ms/step  {B*8/dt:9.0f} frames/s", flush=True)
sc = StreamingClassifier(model=m, classes=[str(i) for i in range(11)], config=cfg)
print("streaming:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in sc.benchmark(50).items()})
