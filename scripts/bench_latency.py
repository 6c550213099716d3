"""Developer aid: forward latency / throughput of the R hybrid over batch sizes (BASELINE configs[1] sweep);
RVIT_BACKBONE picks the stem (resnet18 | resnet34 | mobilenetv3 | none)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd.model import CNNViTHybrid
from vit_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
m = CNNViTHybrid(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone=os.environ.get("RVIT_BACKBONE", "resnet18"))
m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 0))
m = m.to(dev).eval()
if os.environ.get("RVIT_GRAPHS"):
    m.set_graph_capture(int(os.environ["RVIT_GRAPHS"]))
for B in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8,64,512").split(",")]:
    x = torch.randn(B, 3, 224, 224, device=dev)
    with torch.no_grad():
        for _ in range(5): m(x)
        torch.cuda.synchronize()
        n = 50 if B <= 64 else 10
        t0 = time.perf_counter()
        for _ in range(n): m(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:5d}  {dt*1e3:8.3f} ms/forward  {B/dt:10.1f} frames/s", flush=True)
