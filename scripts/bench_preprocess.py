"""Developer aid: throughput of the device input pipeline on full-size frames, beside Pillow on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vit_amd.pipeline import DevicePreprocessor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
H, W = 804, 1475
frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev)
for train in (False, True):
    pre = DevicePreprocessor((H, W), train=train)
    for _ in range(3):
        pre(frames)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        pre(frames)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    S = 224
    alg = B * (H * W * 3 + 2 * H * S + 3 * S * S * 4)
    print(f"{'train_tf' if train else 'eval_tf '} B={B}: {ms:.3f} ms  {B / ms * 1e3:,.0f} frames/s  "
          f"{alg / ms / 1e6:.0f} GB/s algorithmic ({alg / B / 1e6:.2f} MB/frame)")
# host baseline: Pillow convert + resize + numpy normalise, one thread
from PIL import Image
from vit_amd.infer import preprocess
img = Image.fromarray(frames[0].cpu().numpy())
t0 = time.perf_counter()
for _ in range(20):
    preprocess(img)
dt = (time.perf_counter() - t0) / 20
print(f"host (Pillow, 1 thread): {dt * 1e3:.2f} ms/frame = {1 / dt:,.0f} frames/s")
