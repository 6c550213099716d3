"""Developer aid: per-shape timing of the conv / GEMM kernels at the bench shapes (B per GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
dev = torch.device("cuda:0")
bf = torch.bfloat16

def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

convs = [("conv1 7x7/2", 230, 232, 4, 64, 7, 1, 2, 0, 32, 112),
         ("l1 3x3 64", 56, 56, 64, 64, 3, 3, 1, 1, None, 56),
         ("l2 3x3/2 64>128", 56, 56, 64, 128, 3, 3, 2, 1, None, 28),
         ("l2 3x3 128", 28, 28, 128, 128, 3, 3, 1, 1, None, 28),
         ("l2 ds 1x1/2", 56, 56, 64, 128, 1, 1, 2, 0, None, 28),
         ("l3 3x3 256", 14, 14, 256, 256, 3, 3, 1, 1, None, 14),
         ("l4 3x3 512", 7, 7, 512, 512, 3, 3, 1, 1, None, 7)]
for name, H, W, cin, cout, kh, kw, s, p, seg, ho in convs:
    x = torch.randn(B, H, W, cin, device=dev).to(bf)
    seg_len = cin if seg is None else seg
    w = (torch.randn(cout, kh * kw * seg_len, device=dev) * 0.05).to(bf)
    y = torch.empty(B, ho, ho, cout, device=dev, dtype=bf)
    sc = torch.ones(cout, device=dev); bi = torch.zeros(cout, device=dev)
    true_k = 147 if seg else kh * kw * cin
    fl = 2.0 * B * ho * ho * cout * true_k
    nbytes = x.numel() * 2 + y.numel() * 2
    for tile in tiles:
        try:
            us = timeit(lambda: ops.conv2d(x, w, y, kh=kh, kw=kw, stride=s, pad=p, seg_len=seg, scale=sc, bias=bi, act="relu", tile=tile))
        except Exception:
            continue
        print(f"{name:18s} tile={tile} {us:9.1f} us  {fl/us/1e6:8.1f} TF/s  {nbytes/us/1e3:8.1f} GB/s(alg)")
T = B * 50
gemms = [("qkv", T, 768, 256), ("out", T, 256, 256), ("fc1", T, 1024, 256), ("fc2", T, 256, 1024), ("patch", B * 49, 256, 512),
         ("vitB qkv", 197 * 64, 2304, 768), ("vitB fc1", 197 * 64, 3072, 768), ("vitB fc2", 197 * 64, 768, 3072), ("big 4096^3", 4096, 4096, 4096)]
for name, M, N, K in gemms:
    a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    o = torch.empty(M, N, device=dev, dtype=bf); bi = torch.zeros(N, device=dev)
    for tile in tiles:
        us = timeit(lambda: ops.gemm(a, w, o, bias=bi, tile=tile))
        print(f"{name:18s} tile={tile} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TF/s  {(M*K+N*K+M*N)*2/us/1e3:8.1f} GB/s(alg)")

if os.environ.get("RVIT_BENCH_BWD", "1") == "1":
    print("---- backward: dgrad / wgrad per shape ----")
    for name, H, W, cin, cout, kh, kw, s, p, seg, ho in convs:
        dy = torch.randn(B, ho, ho, cout, device=dev).to(bf)
        x = torch.randn(B, H, W, cin, device=dev).to(bf)
        seg_len = cin if seg is None else seg
        dw = torch.zeros(cout, kh * kw * seg_len, device=dev)
        fl = 2.0 * B * ho * ho * cout * kh * kw * seg_len
        us = timeit(lambda: ops.conv2d_wgrad(dy, x, dw, kh=kh, kw=kw, stride=s, pad=p, seg_len=seg))
        print(f"wgrad {name:18s} {us:9.1f} us  {fl/us/1e6:8.1f} TF/s  {(dy.numel()+x.numel())*2/us/1e3:8.1f} GB/s(alg)")
        if seg is None:
            w_t = (torch.randn(cin, kh * kw * cout, device=dev) * 0.05).to(bf)
            dx = torch.empty(B, H, W, cin, device=dev, dtype=bf)
            for tile in tiles:
                try:
                    us = timeit(lambda: ops.conv2d_dgrad(dy, w_t, dx, kh=kh, kw=kw, stride=s, pad=p, tile=tile))
                except Exception:
                    continue
                print(f"dgrad {name:18s} tile={tile} {us:9.1f} us  {fl/us/1e6:8.1f} TF/s  {(dy.numel()+dx.numel())*2/us/1e3:8.1f} GB/s(alg)")
    for name, M, N, K in gemms[:5]:
        dy = torch.randn(M, N, device=dev).to(bf); x = torch.randn(M, K, device=dev).to(bf)
        dw = torch.zeros(N, K, device=dev)
        us = timeit(lambda: ops.wgrad(dy, x, dw))
        print(f"wgrad {name:18s} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TF/s  {(M*K+M*N)*2/us/1e3:8.1f} GB/s(alg)")
