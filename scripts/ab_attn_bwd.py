"""Attention backward at ViT-B/16 scale (B 256, S 197, H 12): time per launch, and the outputs saved to / compared with a
file so that two runs (RVIT_ATTN_BWD_PIPE=0 / 1: the one-barrier kernel / the pipelined one) can be checked bit for bit.
usage: python scripts/ab_attn_bwd.py save|check path [drop_p]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vit_amd import ops

mode, path = sys.argv[1], sys.argv[2]
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
dev = torch.device("cuda:0")
res = {}
for (B, S, H) in [(256, 197, 12), (3, 160, 2), (2, 256, 4), (5, 129, 3)]:
    E = H * 64
    g = torch.Generator().manual_seed(S)
    qkv = (torch.randn(B * S, 3 * E, generator=g) * 0.8).to(torch.bfloat16).to(dev)
    do = (torch.randn(B * S, E, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    out = torch.empty(B * S, E, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H * S, dtype=torch.float32, device=dev)
    dqkv = torch.full((B * S, 3 * E), float("nan"), dtype=torch.bfloat16, device=dev)
    for pp in (0.0, p):
        ops.attention(qkv, out, B, S, H, lse=lse, drop_p=pp, drop_seed=77)
        ops.attention_bwd(qkv, out, do, lse, dqkv, B, S, H, drop_p=pp, drop_seed=77)
        torch.cuda.synchronize()
        res[f"{B}_{S}_{H}_{pp}"] = dqkv.cpu().clone()
        if B == 256:
            for _ in range(3):
                ops.attention_bwd(qkv, out, do, lse, dqkv, B, S, H, drop_p=pp, drop_seed=77)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.attention_bwd(qkv, out, do, lse, dqkv, B, S, H, drop_p=pp, drop_seed=77)
            e1.record(); torch.cuda.synchronize()
            print(f"attn_bwd B={B} S={S} H={H} p={pp}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us  (RVIT_ATTN_BWD_PIPE={os.environ.get('RVIT_ATTN_BWD_PIPE', '1')})", flush=True)
if mode == "save":
    torch.save(res, path)
else:
    ref = torch.load(path, weights_only=True)
    for k, v in res.items():
        same = torch.equal(v.view(torch.int16), ref[k].view(torch.int16))
        print(f"{k}: bit-identical={same}" + ("" if same else f" max|diff| {(v.float() - ref[k].float()).abs().max():.4g} nan {torch.isnan(v.float()).sum().item()}"))
