"""Developer aid: fused encoder kernels against the per-op path at the R-hybrid scale (us per layer half)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops, _lib
if os.environ.get("RVIT_LIB_PATH"):   # an ablation build (scripts/ablate_encoder.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["RVIT_LIB_PATH"])
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_encoder_gpu import _mlp_weights, _mlp_per_op, _attn_weights

dev = torch.device("cuda:0")
B, S, E, hidden = int(os.environ.get("B", 512)), 50, 256, 1024
T = B * S


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


W = _mlp_weights(E, hidden, dev, 1)
x = torch.randn(T, E, device=dev)
ln = torch.empty(T, E, dtype=torch.bfloat16, device=dev)
hid = torch.empty(T, hidden, dtype=torch.bfloat16, device=dev)
y = x.clone()


def per_op():
    ops.layernorm(y, W["ln_w"], W["ln_b"], ln, 1e-5)
    ops.gemm(ln, W["w1"], hid, bias=W["b1"], act="gelu")
    ops.gemm(hid, W["w2"], y, bias=W["b2"], residual=y)


print(f"mlp per-op  T={T}: {timeit(per_op):8.1f} us", flush=True)
for rows in [int(r) for r in os.environ.get("ROWS", "0,112,96,64,50").split(",")]:
    y.copy_(x)
    us = timeit(lambda: ops.encoder_mlp(y, W["ln_w"], W["ln_b"], W["w1"], W["b1"], W["w2"], W["b2"], 1e-5, rows_per_wg=rows))
    print(f"mlp fused   T={T} rows_per_wg={rows:3d}: {us:8.1f} us  {4.0 * T * E * hidden / us / 1e6:7.1f} TF/s", flush=True)

WA = _attn_weights(E, dev, 2)
qkv = torch.empty(T, 3 * E, dtype=torch.bfloat16, device=dev)
att = torch.empty(T, E, dtype=torch.bfloat16, device=dev)


def attn_per_op():
    ops.layernorm(y, WA["ln_w"], WA["ln_b"], ln, 1e-5)
    ops.gemm(ln, WA["wqkv"], qkv, bias=WA["bqkv"])
    ops.attention(qkv, att, B, S, 4)
    ops.gemm(att, WA["wout"], y, bias=WA["bout"], residual=y)


y.copy_(x)
print(f"attn per-op T={T}: {timeit(attn_per_op):8.1f} us", flush=True)
y.copy_(x)
us = timeit(lambda: ops.encoder_attn(y, WA["ln_w"], WA["ln_b"], WA["wqkv"], WA["bqkv"], WA["wout"], WA["bout"], B, S, 4, 1e-5))
print(f"attn fused  T={T}: {us:8.1f} us", flush=True)

# the whole stack (4 layers) in one launch against eight launches
from test_encoder_gpu import _attn_weights as _aw, _mlp_weights as _mw  # noqa: E402
lays, refs = [], []
for li in range(4):
    A, M = _aw(E, dev, 10 * li + 1), _mw(E, hidden, dev, 10 * li + 2)
    lays.append((A["ln_w"], A["ln_b"], A["wqkv"], A["bqkv"], A["wout"], A["bout"], M["ln_w"], M["ln_b"], M["w1"], M["b1"], M["w2"], M["b2"]))
    refs.append((A, M))


def eight():
    for A, M in refs:
        ops.encoder_attn(y, A["ln_w"], A["ln_b"], A["wqkv"], A["bqkv"], A["wout"], A["bout"], B, S, 4, 1e-5)
        ops.encoder_mlp(y, M["ln_w"], M["ln_b"], M["w1"], M["b1"], M["w2"], M["b2"], 1e-5)


y.copy_(x)
print(f"4 layers, eight launches T={T}: {timeit(eight, 20):8.1f} us", flush=True)
y.copy_(x)
print(f"4 layers, one launch    T={T}: {timeit(lambda: ops.encoder_stack(y, lays, B, S, 4, 1e-5), 20):8.1f} us", flush=True)
