"""GPU parity of the training-pass kernels of the MobileNetV3 stem (csrc/mobile_train.hip; the reference trains its
`backbone='mobilenetv3'` model through timm + torch autograd, train_vit.py:68-74,87,145-146,939) against torch fp32 /
autograd on the same bf16-rounded operands: train-mode BatchNorm of any width with timm's fused activations, depthwise
convolution gradients, the squeeze-excite multiply and the gate's gradient.  Outputs are bf16 (one rounding, 2^-8
relative); per-channel sums are fp32 ordered reductions (bit-reproducible: two runs must give the same bits)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _bf(t):
    return t.to(BF16)


def _close(got, ref, rtol, atol, what):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs()
    bad = (err > atol + rtol * ref.abs()).sum().item()
    assert bad == 0, f"{what}: {bad}/{err.numel()} out of tolerance, max err {err.max().item():.4g}"


def _act(u, act):
    return {"relu": F.relu, "hswish": F.hardswish, None: lambda v: v}[act](u)


@pytest.fixture()
def ordered(dev):
    from vit_amd import ops
    ops.enable_ordered_reductions(dev, True)
    yield ops
    ops.enable_ordered_reductions(dev, False)


@pytest.mark.parametrize("C,M,act,with_res", [(16, 3000, "hswish", False), (72, 1568, "relu", False), (24, 777, None, True),
                                              (576, 392, "hswish", False), (96, 50, None, False), (240, 6272, "hswish", False)])
def test_bn_generic_forward_backward(dev, ordered, C, M, act, with_res):
    ops = ordered
    g = torch.Generator().manual_seed(C * 31 + M)
    y = _bf(torch.randn(M, C, generator=g) * (0.5 + torch.rand(C, generator=g)) + torch.randn(C, generator=g))
    res = _bf(torch.randn(M, C, generator=g)) if with_res else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5
    dz = _bf(torch.randn(M, C, generator=g))
    eps, mom = 1e-5, 0.1
    rm0, rv0 = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    # ---- reference (fp32, autograd)
    yr = y.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    mean, var = yr.mean(0), yr.var(0, unbiased=False)
    u = (yr - mean) / torch.sqrt(var + eps) * gr + br
    if res is not None:
        u = u + res.float()
    z_ref = _act(u, act)
    z_ref.backward(dz.float())
    # ---- kernels
    yd, fs = y.to(dev), torch.zeros(2 * C, device=dev)
    ops.bn_generic_stats(yd, fs)
    _close(fs[:C] / M, mean.detach(), 1e-5, 1e-5, "mean")
    _close(fs[C:] / M - (fs[:C] / M) ** 2, var.detach(), 1e-3, 1e-4, "var")
    fs2 = torch.zeros(2 * C, device=dev)
    ops.bn_generic_stats(yd, fs2)
    assert torch.equal(fs, fs2), "ordered statistics must be bit-reproducible"
    z = torch.empty(M, C, dtype=BF16, device=dev)
    rm, rv = rm0.to(dev), rv0.to(dev)
    ops.bn_generic_apply(yd, fs, gamma.to(dev), beta.to(dev), z, eps=eps, act=act, residual=None if res is None else res.to(dev),
                         momentum=mom, running_mean=rm, running_var=rv)
    _close(z, z_ref.detach(), 1.0 / 128, 2e-3, "bn apply")
    _close(rm, (1 - mom) * rm0 + mom * mean.detach(), 1e-5, 1e-5, "running_mean")
    _close(rv, (1 - mom) * rv0 + mom * yr.detach().var(0, unbiased=True), 1e-3, 1e-4, "running_var")
    bs = torch.zeros(2 * C, device=dev)
    dy = torch.empty(M, C, dtype=BF16, device=dev)
    dgamma, dbeta = torch.ones(C, device=dev), torch.full((C,), 2.0, device=dev)   # the kernels accumulate (+=)
    ops.bn_generic_bwd(dz.to(dev), yd, fs, bs, gamma.to(dev), beta.to(dev), dy, dgamma, dbeta, eps=eps, act=act)
    scale = yr.grad.abs().max().item()
    _close(dy, yr.grad, 1.0 / 100, 5e-3 * scale, "bn dy")
    _close(dgamma - 1.0, gr.grad, 2e-3, 2e-3 * gr.grad.abs().max().item(), "dgamma")
    _close(dbeta - 2.0, br.grad, 2e-3, 2e-3 * br.grad.abs().max().item(), "dbeta")


@pytest.mark.parametrize("C,hw,k,stride,B", [(16, 112, 3, 2, 2), (72, 56, 3, 2, 2), (88, 28, 3, 1, 3), (96, 28, 5, 2, 2),
                                              (240, 14, 5, 1, 4), (576, 7, 5, 1, 5)])
def test_dwconv_gradients(dev, ordered, C, hw, k, stride, B):
    ops = ordered
    g = torch.Generator().manual_seed(C + hw + k)
    pad = k // 2
    x = _bf(torch.randn(B, C, hw, hw, generator=g))
    w = _bf(torch.randn(C, 1, k, k, generator=g) / k)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    out = F.conv2d(xr, wr, stride=stride, padding=pad, groups=C)
    dy = _bf(torch.randn(out.shape, generator=g))
    out.backward(dy.float())
    x_n, dy_n = x.permute(0, 2, 3, 1).contiguous().to(dev), dy.permute(0, 2, 3, 1).contiguous().to(dev)
    w_pack = w.reshape(C, k * k).t().contiguous().to(dev)           # [k*k, C], the forward kernel's pack
    dx = torch.empty_like(x_n)
    ops.dwconv_dgrad(dy_n, w_pack, dx, k=k, stride=stride, pad=pad)
    _close(dx.permute(0, 3, 1, 2), xr.grad, 1.0 / 100, 1e-2, "dwconv dgrad")
    dw = torch.ones(C * k * k, device=dev)
    ops.dwconv_wgrad(dy_n, x_n, dw, k=k, stride=stride, pad=pad)
    ref = wr.grad.reshape(C, k * k)
    _close(dw.view(C, k * k) - 1.0, ref, 2e-3, 2e-3 * ref.abs().max().item(), "dwconv wgrad")
    dw2 = torch.ones(C * k * k, device=dev)
    ops.dwconv_wgrad(dy_n, x_n, dw2, k=k, stride=stride, pad=pad)
    assert torch.equal(dw, dw2), "ordered weight gradient must be bit-reproducible"


@pytest.mark.parametrize("C,hw,B", [(16, 56, 3), (96, 14, 4), (576, 7, 6)])
def test_squeeze_excite_backward_pieces(dev, C, hw, B):
    """out = z * gate (separate output) and its backward: d(z) = dout * gate + dpool / HW, d(gate pre-activation) =
    (sum_p dout * z) * hardsigmoid'(pre), hardsigmoid' = 1/6 strictly inside the gate's (0, 1) range."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(C + hw)
    z = _bf(torch.randn(B, hw, hw, C, generator=g))
    pre = torch.randn(B, C, generator=g) * 3
    gate = _bf(F.hardsigmoid(pre))
    dout = _bf(torch.randn(B, hw, hw, C, generator=g))
    add = _bf(torch.randn(B, C, generator=g) / (hw * hw))
    zd, gd, dd = z.to(dev), gate.to(dev), dout.to(dev)
    out = torch.empty_like(zd)
    ops.channel_affine(zd, gd, out)
    _close(out, z.float() * gate.float()[:, None, None, :], 1.0 / 128, 1e-3, "gate multiply")
    dz = torch.empty_like(zd)
    ops.channel_affine(dd, gd, dz, add=add.to(dev))
    _close(dz, dout.float() * gate.float()[:, None, None, :] + add.float()[:, None, None, :], 1.0 / 128, 1e-3, "gate backward")
    dpre = torch.empty(B, C, dtype=BF16, device=dev)
    ops.channel_dot(dd, zd, dpre, gate=gd)
    inside = (gate.float() > 0) & (gate.float() < 1)
    ref = (dout.float() * z.float()).sum(dim=(1, 2)) * inside / 6.0
    _close(dpre, ref, 1.0 / 100, 1e-2 * ref.abs().max().item(), "gate gradient")
    dot = torch.empty(B, C, dtype=BF16, device=dev)
    ops.channel_dot(dd, zd, dot)
    ref2 = (dout.float() * z.float()).sum(dim=(1, 2))
    _close(dot, ref2, 1.0 / 100, 1e-2 * ref2.abs().max().item(), "channel dot")
