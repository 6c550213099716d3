"""Developer aid: MX-fp8 GEMM vs bf16 GEMM timing on ViT-L shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K in [("vitL qkv", 197*256, 3072, 1024), ("vitL fc1", 197*256, 4096, 1024), ("vitL fc2", 197*256, 1024, 4096), ("big 4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
    aq, as_ = ops.quantize_mx(a); wq, ws = ops.quantize_mx(w)
    ab, wb = a.to(torch.bfloat16), w.to(torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us8 = timeit(lambda: ops.gemm_mx(aq, as_, wq, ws, o))
    us16 = timeit(lambda: ops.gemm(ab, wb, o))
    usq = timeit(lambda: ops.quantize_mx(ab, aq, as_))
    print(f"{name:12s} mxfp8 {us8:8.1f} us {2.0*M*N*K/us8/1e6:7.1f} TF/s | bf16 {us16:8.1f} us {2.0*M*N*K/us16/1e6:7.1f} TF/s | quantize A {usq:7.1f} us", flush=True)
