"""us per launch of the attention kernels by head dim (csrc/attn.hip for 64, csrc/attn_dh.hip for the others)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(5): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for B, S, E in [(256, 197, 768), (512, 50, 256)]:
    for Dh in (16, 32, 48, 64, 96, 128):
        if E % Dh: continue
        H = E // Dh
        qkv = (torch.randn(B * S, 3 * E, device=dev) * 0.7).to(torch.bfloat16)
        out = torch.empty(B * S, E, device=dev, dtype=torch.bfloat16)
        dout = torch.randn(B * S, E, device=dev).to(torch.bfloat16)
        lse = torch.empty(B * H * S, device=dev)
        dqkv = torch.empty_like(qkv)
        f = timeit(lambda: ops.attention(qkv, out, B, S, H, lse=lse))
        bw = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, dqkv, B, S, H))
        fl = 4 * B * H * S * S * Dh
        print(f"B={B} S={S} E={E} H={H} Dh={Dh}: fwd {f:7.1f} us ({fl / f * 1e-6:6.1f} TFLOP/s)  bwd {bw:7.1f} us ({2.5 * fl / bw * 1e-6:6.1f} TFLOP/s)", flush=True)
