"""Ordered (bit-reproducible) reductions of the training path (include/rvit.h "Ordered reductions"; the reference asks
for deterministic kernels, train_vit.py:50): kernel level against the atomic forms and torch, and a whole training
step run twice from the same state must give the same bits in every gradient and every parameter."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture()
def ordered(dev):
    from vit_amd import ops
    was = ops.ordered_reductions_enabled(dev)
    ops.enable_ordered_reductions(dev, True)
    yield
    ops.enable_ordered_reductions(dev, was)


def _twice(fn):
    a = fn()
    b = fn()
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("M,N,K,kw", [(25600, 256, 1024, {}), (25600, 1024, 256, {}), (50432, 768, 768, {}), (512, 64, 128, {}),
                                      (4096, 16, 512, dict(n_store=11)), (3000, 256, 256, dict(splits=7))])
def test_linear_wgrad_ordered_is_reproducible_and_right(dev, ordered, M, N, K, kw):
    from vit_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).to(BF16).to(dev)
    x = torch.randn(M, K, generator=g).to(BF16).to(dev)
    ns = kw.get("n_store", N)

    def run():
        dw = torch.full((ns, K), 0.5, device=dev)
        ops.wgrad(dy, x, dw, **kw)
        return dw
    a, b = _twice(run)
    assert torch.equal(a, b)
    want = dy.float()[:, :ns].T @ x.float() + 0.5
    assert ((a - want).abs().max() / want.abs().max()).item() < 2e-5
    if N > 64:
        def run_b():
            dw = torch.zeros(ns, K, device=dev)
            db = torch.full((ns,), -1.0, device=dev)
            ops.linear_wgrad_bias(dy, x, dw, db)
            return torch.cat([dw.flatten(), db])
        a, b = _twice(run_b)
        assert torch.equal(a, b)
        assert torch.allclose(a[-ns:], dy.float().sum(0) - 1.0, rtol=1e-4, atol=1e-2)
        assert ((a[:-ns].view(ns, K) - (want - 0.5)).abs().max() / want.abs().max()).item() < 2e-5


@pytest.mark.parametrize("B,hw,ci,co,k,stride", [(16, 56, 64, 64, 3, 1), (16, 28, 128, 128, 3, 1), (16, 7, 512, 512, 3, 1),
                                                 (16, 56, 64, 128, 3, 2), (16, 28, 128, 256, 1, 2)])
def test_conv_wgrad_ordered_is_reproducible_and_equals_atomic_form(dev, B, hw, ci, co, k, stride):
    from vit_amd import ops
    g = torch.Generator().manual_seed(hw + ci + co)
    x = torch.randn(B, hw, hw, ci, generator=g).to(BF16).to(dev)
    ho = hw // stride
    dy = torch.randn(B, ho, ho, co, generator=g).to(BF16).to(dev)

    def run():
        dw = torch.zeros(co, k * k * ci, device=dev)
        ops.conv2d_wgrad(dy, x, dw, kh=k, kw=k, stride=stride, pad=k // 2)
        return dw
    was = ops.ordered_reductions_enabled(dev)
    try:
        ops.enable_ordered_reductions(dev, False)
        atomic = run()
        ops.enable_ordered_reductions(dev, True)
        a, b = _twice(run)
    finally:
        ops.enable_ordered_reductions(dev, was)
    assert torch.equal(a, b)
    assert ((a - atomic).abs().max() / atomic.abs().max()).item() < 1e-5   # same sums, another association


def test_small_reductions_ordered(dev, ordered):
    from vit_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randn(20000, 768, generator=g).to(BF16).to(dev)

    def cs():
        o = torch.ones(768, device=dev)
        ops.colsum(a, o)
        return o
    u, v = _twice(cs)
    assert torch.equal(u, v) and torch.allclose(u, a.float().sum(0) + 1.0, rtol=1e-4, atol=1e-2)
    x = torch.randn(11_000_003, generator=g).to(dev)

    def ss():
        o = torch.zeros(1, device=dev)
        ops.sumsq(x, o)
        return o
    u, v = _twice(ss)
    assert torch.equal(u, v) and abs(u.item() / (x.double() ** 2).sum().item() - 1.0) < 1e-5
    rows, E = 5000, 256
    dy = torch.randn(rows, E, generator=g).to(BF16).to(dev)
    xx = torch.randn(rows, E, generator=g).to(dev)
    mean, rstd = xx.mean(1), 1.0 / torch.sqrt(xx.var(1, unbiased=False) + 1e-5)
    gamma = (torch.rand(E, generator=g) + 0.5).to(dev)

    def ln():
        dx = torch.zeros(rows, E, device=dev)
        dg, db = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
        ops.layernorm_bwd(dy, xx, mean, rstd, gamma, dx, dg, db, accumulate=False)
        return torch.cat([dg, db])
    u, v = _twice(ln)
    assert torch.equal(u, v)

    def ln_adjacent():   # weight and bias gradients back to back (a flat gradient buffer): one finalisation launch for both
        dx = torch.zeros(rows, E, device=dev)
        flat = torch.zeros(2 * E, device=dev)
        ops.layernorm_bwd(dy, xx, mean, rstd, gamma, dx, flat[:E], flat[E:], accumulate=False)
        return flat
    w = ln_adjacent()
    if ordered:
        assert torch.equal(u, w)   # same association per output as the two-launch form
    else:
        assert torch.allclose(u, w, rtol=1e-4, atol=1e-3)
    xh = (xx - mean[:, None]) * rstd[:, None]
    assert torch.allclose(u[:E], (dy.float() * xh).sum(0), rtol=1e-3, atol=5e-2)
    assert torch.allclose(u[E:], dy.float().sum(0), rtol=1e-3, atol=5e-2)


@pytest.mark.parametrize("backbone,B", [("resnet18", 8), ("patch16", 6)])
def test_training_step_is_bit_reproducible(dev, backbone, B):
    """Two runs of forward + loss + backward + fused clip/AdamW from the same state: every gradient, every parameter
    and the loss must be the same bits (ordered BatchNorm sums + ordered split reductions; dropout masks are
    counter-based and seeded)."""
    from vit_amd.loss import cross_entropy
    from vit_amd.model import CNNViTHybrid
    from vit_amd.optim import FusedAdamW
    from vit_amd.synth import micro_doppler_batch, synth_state_dict
    cfg = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=2, dropout=0.1, backbone=backbone)
    x = micro_doppler_batch(B, seed=77).to(dev)
    y = (torch.arange(B) % 11).to(dev)
    outs = []
    for _ in range(2):
        m = CNNViTHybrid(**cfg)
        m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 3))
        m = m.to(dev).train()
        opt = FusedAdamW(m, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
        assert m.train_engine.ordered and m.train_engine.ordered_bn
        torch.manual_seed(1234)
        losses = []
        for _step in range(2):
            opt.zero_grad()
            loss = cross_entropy(m(x), y, label_smoothing=0.05)
            loss.backward()
            grads = m.train_engine.last_flat.clone()
            opt.step()
            losses.append(loss.detach().clone())
        outs.append((torch.stack(losses), grads, torch.cat([p.detach().flatten() for p in m.parameters()])))
    for u, v in zip(outs[0], outs[1]):
        assert torch.isfinite(u).all()
        assert torch.equal(u, v), f"max diff {(u - v).abs().max().item():.3g}"
