import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from vit_amd import ops
from test_encoder_gpu import _mlp_weights, _mlp_per_op
dev = torch.device("cuda:0")
for T, hidden, rows in [(1999, 1024, 50), (777, 256, 33), (800, 1024, 16), (640, 256, 64), (640, 256, 48), (640, 256, 80), (640,256,96)]:
    W = _mlp_weights(256, hidden, dev, seed=T + hidden)
    x = (torch.randn(T, 256, generator=torch.Generator().manual_seed(T)) * 1.5 + 0.3).to(dev)
    want = _mlp_per_op(x, W, 1e-5)
    got = x.clone()
    ops.encoder_mlp(got, W["ln_w"], W["ln_b"], W["w1"], W["b1"], W["w2"], W["b2"], 1e-5, rows_per_wg=rows)
    d = (got - want).abs()
    bad_rows = (d.max(1).values > 0).nonzero().flatten()
    print(T, hidden, rows, "max", d.max().item(), "bad rows", bad_rows.numel(), (bad_rows % rows).unique().tolist()[:40],
          "bad cols", (d.max(0).values > 0).sum().item())
