"""Eager vs hipGraph replay of the B=512 inference forward: are the ~0.3 ms between the sum of kernel times and the step
time host launch gaps (a graph removes them) or device-side kernel boundaries (it does not)?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd.model import CNNViTHybrid  # noqa: E402
from vit_amd.synth import micro_doppler_batch, synth_state_dict  # noqa: E402

dev = torch.device("cuda:0")
m = CNNViTHybrid(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone="resnet18")
m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 0))
m = m.to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.cat([micro_doppler_batch(64, seed=i) for i in range(B // 64)]).to(dev)


def timed(n=20):
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3


e, eh = timed()
m.set_graph_capture(B)
g, gh = timed()
print(f"B={B}: eager {e:.3f} ms/step (host enqueue {eh:.3f}), graph replay incl. input copy {g:.3f} ms/step (host {gh:.3f})")
