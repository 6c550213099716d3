"""Per-frame latency of StreamingClassifier.update on full-size radar frames (804 x 1475 RGB uint8, the size of the
reference's spectrogram PNGs), host (Pillow) transform against the on-device one.  python scripts/bench_streaming.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd.sequence import CNNViTSequence, StreamingClassifier  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = dict(backbone="resnet18", embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, sequence_length=8,
               temporal_hidden=256, temporal_layers=2, bidirectional=True, temporal_model="lstm")
    model = CNNViTSequence(num_classes=11, config=cfg).to(dev).eval()
    classes = [f"c{i}" for i in range(11)]
    rng = np.random.default_rng(0)
    frames = [(rng.random((804, 1475, 3)) * 255).astype(np.uint8) for _ in range(8)]
    for name, kw in (("host transform (Pillow)", {}), ("device transform", {"device_preprocess": True})):
        sc = StreamingClassifier(model=model, classes=classes, config=cfg, **kw)
        for i in range(24):
            sc.update(frames[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for i in range(n):
            sc.update(frames[i % 8])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:26s} {dt * 1e3:6.2f} ms per frame  ({1 / dt:6.0f} frames/s)", flush=True)


if __name__ == "__main__":
    main()
