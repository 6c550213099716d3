"""Times rvit_attn_fwd / rvit_attn_bwd on the ViT-B/16 training shape (B 256, S 197, H 12), with and without
probability dropout, and prints a checksum of dqkv (two builds of the library can be compared on one box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import _lib, ops  # noqa: E402

if os.environ.get("RVIT_BENCH_LIB"):   # an alternative build of the library (A/B on one box)
    _lib.LIB_PATH = os.path.abspath(os.environ["RVIT_BENCH_LIB"])


def main():
    dev = torch.device("cuda:0")
    B, S, H = 256, 197, 12
    E = H * 64
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B * S, 3 * E, generator=g) * 0.7).to(dev).to(torch.bfloat16)
    dout = torch.randn(B * S, E, generator=g).to(dev).to(torch.bfloat16)
    out = torch.empty(B * S, E, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H * S, device=dev)
    dqkv = torch.empty_like(qkv)
    for p in (0.1, 0.0):
        def fwd():
            ops.attention(qkv, out, B, S, H, lse=lse, drop_p=p, drop_seed=7)

        def bwd():
            ops.attention_bwd(qkv, out, dout, lse, dqkv, B, S, H, drop_p=p, drop_seed=7)

        res = {}
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{os.path.basename(_lib.LIB_PATH)} p={p}: fwd {res['fwd']:.1f} us  bwd {res['bwd']:.1f} us  "
              f"dqkv checksum {dqkv.float().abs().sum().item():.6e}", flush=True)


if __name__ == "__main__":
    main()
