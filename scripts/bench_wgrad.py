"""Times the weight-gradient kernel (rvit_linear_wgrad_bias) on the four Linear shapes of a ViT-B/16 layer
(M = 256 x 197 tokens).  RVIT_BENCH_LIB selects another build of the library for an A/B on one box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib, ops  # noqa: E402

if os.environ.get("RVIT_BENCH_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["RVIT_BENCH_LIB"])


def main():
    dev = torch.device("cuda:0")
    M = 256 * 197
    g = torch.Generator().manual_seed(0)
    out = []
    for name, N, K in (("in_proj", 2304, 768), ("out_proj", 768, 768), ("linear1", 3072, 768), ("linear2", 768, 3072)):
        dy = torch.randn(M, N, generator=g).to(dev).to(torch.bfloat16)
        x = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
        dw = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)
        for _ in range(3):
            ops.linear_wgrad_bias(dy, x, dw, db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear_wgrad_bias(dy, x, dw, db)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out.append(f"{name} {us:.0f} us ({2.0 * M * N * K / us / 1e6:.0f} TF/s)")
    print(os.path.basename(_lib.LIB_PATH) + ": " + "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
