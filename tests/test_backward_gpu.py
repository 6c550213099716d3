"""GPU parity of the backward / loss kernels against torch autograd (fp32, CPU) on the same
bf16-rounded inputs.  Tolerances: fp32-accumulated results 1e-3 relative to the gradient scale;
bf16 outputs one bf16 ulp (2^-7 relative) plus the attention-specific P rounding."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


def _close(got, ref, rtol, atol, what=""):
    got = got.float().cpu()
    ref = ref.float()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = (err > bound).sum().item()
    assert bad == 0, f"{what}: {bad}/{err.numel()} out of tolerance, max err {err.max().item():.4g} " \
                     f"(ref scale {ref.abs().max().item():.3g})"


@pytest.mark.parametrize("M,N,K,splits", [(1000, 256, 256, 0), (777, 768, 264, 3), (64, 16, 128, 1), (5000, 1024, 256, 0),
                                          (300, 136, 72, 2),
                                          # >= 32 M tiles per split: the eight-wave variant (ragged M, partial N / K tiles)
                                          (4100, 256, 384, 1), (5000, 1024, 256, 2), (3000, 136, 200, 1)])
def test_wgrad_plain(dev, M, N, K, splits):
    from vit_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    dy = _bf(torch.randn(M, N, generator=g))
    x = _bf(torch.randn(M, K, generator=g))
    init = torch.randn(N, K, generator=g)
    dw = init.clone().to(dev)
    ops.wgrad(dy.to(dev), x.to(dev), dw, splits=splits)
    ref = init + dy.float().T @ x.float()
    _close(dw, ref, 1e-3, 1e-3 * math.sqrt(M), "wgrad")


@pytest.mark.parametrize("M,N,K,splits", [(1000, 256, 256, 0), (777, 768, 264, 3), (300, 136, 72, 2), (64, 16, 128, 1),
                                          (37, 128, 256, 0),
                                          # eight-wave variant; ragged M, partial N / K tiles
                                          (4100, 256, 384, 1), (5000, 1024, 256, 2), (3000, 136, 200, 1)])
def test_linear_wgrad_bias(dev, M, N, K, splits):
    """Weight and bias gradient in one launch == wgrad + colsum (train_vit.py:141-171 Linears)."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(M + N + K + 1)
    dy = _bf(torch.randn(M, N, generator=g))
    x = _bf(torch.randn(M, K, generator=g))
    init, binit = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    dw, db = init.clone().to(dev), binit.clone().to(dev)
    ops.linear_wgrad_bias(dy.to(dev), x.to(dev), dw, db, splits=splits)
    _close(dw, init + dy.float().T @ x.float(), 1e-3, 1e-3 * math.sqrt(M), "wgrad")
    _close(db, binit + dy.float().sum(0), 1e-3, 1e-3 * math.sqrt(M), "bias grad")
    # same weight gradient as the plain entry (atomics order aside), same sums as colsum
    dw2, db2 = init.clone().to(dev), binit.clone().to(dev)
    ops.wgrad(dy.to(dev), x.to(dev), dw2, splits=splits)
    ops.colsum(dy.to(dev), db2)
    _close(dw, dw2.cpu(), 1e-5, 1e-4 * math.sqrt(M), "wgrad vs plain entry")
    _close(db, db2.cpu(), 1e-5, 1e-4 * math.sqrt(M), "bias grad vs colsum")


@pytest.mark.parametrize("M,N,K", [(700, 264, 520), (64, 256, 256), (4100, 1024, 384), (50432, 768, 768), (1000, 72, 1032)])
def test_wgrad_sixteen_wave_tile(dev, M, N, K):
    """The 256x256-tile weight-gradient variant (splits=-3 forces it; what ViT-B/L Linears run) against the 128x128
    kernel (splits=-4) and against torch: ragged M, partial N / K tiles in both sub-tiles, with and without bias sums."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(M + N + K + 2)
    dy = _bf(torch.randn(M, N, generator=g)).to(dev)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    init, binit = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    res = []
    for sp in (-3, -4):
        dw, db, dw2 = init.clone().to(dev), binit.clone().to(dev), init.clone().to(dev)
        ops.linear_wgrad_bias(dy, x, dw, db, splits=sp)
        ops.wgrad(dy, x, dw2, splits=sp)
        res.append((dw, db, dw2))
    tol = 1e-4 * math.sqrt(M)
    _close(res[0][0], res[1][0].cpu(), 1e-5, tol, "W16 dw vs 128x128")
    _close(res[0][1], res[1][1].cpu(), 1e-5, tol, "W16 dbias vs 128x128")
    _close(res[0][2], res[0][0].cpu(), 1e-5, tol, "W16 dw with / without bias sums")
    if M <= 5000:
        _close(res[0][0], init + dy.float().cpu().T @ x.float().cpu(), 1e-3, 1e-3 * math.sqrt(M), "W16 dw vs torch")
        _close(res[0][1], binit + dy.float().cpu().sum(0), 1e-3, 1e-3 * math.sqrt(M), "W16 dbias vs torch")


def test_wgrad_sixteen_wave_exact(dev):
    """Small integers: exact sums, any layout / sub-tile mix-up shows."""
    from vit_amd import ops
    M, N, K = 4096 + 40, 512, 384
    dy = _bf((torch.arange(M * N).reshape(M, N) % 7 - 3).float())
    x = _bf((torch.arange(M * K).reshape(M, K) % 5 - 2).float())
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.linear_wgrad_bias(dy.to(dev), x.to(dev), dw, db, splits=-3)
    assert torch.equal(dw.cpu(), dy.float().T @ x.float())
    assert torch.equal(db.cpu(), dy.float().sum(0))


def test_linear_wgrad_bias_exact_and_padded(dev):
    """Small integers (exact sums) and the padded-logit head (N = 16, 11 stored rows)."""
    from vit_amd import ops
    M, N, K = 4096 + 40, 256, 128
    dy = _bf((torch.arange(M * N).reshape(M, N) % 7 - 3).float())
    x = _bf((torch.arange(M * K).reshape(M, K) % 5 - 2).float())
    for splits in (1, 4):
        dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        ops.linear_wgrad_bias(dy.to(dev), x.to(dev), dw, db, splits=splits)
        assert torch.equal(dw.cpu(), dy.float().T @ x.float())
        assert torch.equal(db.cpu(), dy.float().sum(0))
    g = torch.Generator().manual_seed(5)
    B, C = 37, 11
    dl = torch.zeros(B, 16)
    dl[:, :C] = torch.randn(B, C, generator=g)
    dl = _bf(dl)
    h = _bf(torch.randn(B, K, generator=g))
    dw, db = torch.zeros(C, K, device=dev), torch.zeros(C, device=dev)
    ops.linear_wgrad_bias(dl.to(dev), h.to(dev), dw, db, n_store=C)
    _close(dw, dl[:, :C].float().T @ h.float(), 1e-3, 1e-3, "head wgrad")
    _close(db, dl[:, :C].float().sum(0), 1e-3, 1e-3, "head bias grad")


def test_wgrad_exact_asymmetric(dev):
    """Small integers: every product and sum is exact, so any layout/permutation error shows."""
    from vit_amd import ops
    M, N, K = 256, 128, 128
    dy = _bf((torch.arange(M * N).reshape(M, N) % 7 - 3).float())
    x = _bf((torch.arange(M * K).reshape(M, K) % 5 - 2).float())
    dw = torch.zeros(N, K, device=dev)
    ops.wgrad(dy.to(dev), x.to(dev), dw, splits=2)
    assert torch.equal(dw.cpu(), dy.float().T @ x.float())


def test_wgrad_padded_logits_and_row_remap(dev):
    from vit_amd import ops
    g = torch.Generator().manual_seed(4)
    B, C, K = 37, 11, 128
    dl = torch.zeros(B, 16)
    dl[:, :C] = torch.randn(B, C, generator=g)
    dl = _bf(dl)
    h = _bf(torch.randn(B, K, generator=g))
    dw = torch.zeros(C, K, device=dev)
    ops.wgrad(dl.to(dev), h.to(dev), dw, n_store=C)
    _close(dw, dl[:, :C].float().T @ h.float(), 1e-3, 1e-3, "head wgrad")
    # patch_embed wgrad: dY = token gradients (rows past the cls row), X = stem features
    Bq, P, E, Cf = 3, 49, 64, 128
    dtok = _bf(torch.randn(Bq * (P + 1), E, generator=g))
    feat = _bf(torch.randn(Bq * P, Cf, generator=g))
    dwp = torch.zeros(E, Cf, device=dev)
    ops.wgrad(dtok.to(dev), feat.to(dev), dwp, rows=Bq * P, y_row_group=P, y_row_extra=1)
    ref = dtok.float().reshape(Bq, P + 1, E)[:, 1:].reshape(-1, E).T @ feat.float()
    _close(dwp, ref, 1e-3, 2e-2, "patch wgrad")


@pytest.mark.parametrize("cin,cout,hw,k,stride,pad", [(64, 64, 56, 3, 1, 1), (64, 128, 56, 3, 2, 1), (64, 128, 56, 1, 2, 0),
                                                      (256, 512, 14, 3, 2, 1), (128, 128, 28, 3, 1, 1)])
def test_conv_wgrad_and_dgrad(dev, cin, cout, hw, k, stride, pad):
    from vit_amd import ops
    B = 2
    g = torch.Generator().manual_seed(cin + cout + hw + k)
    x = _bf(torch.randn(B, hw, hw, cin, generator=g))
    w = _bf(torch.randn(cout, k, k, cin, generator=g) / math.sqrt(k * k * cin))
    ho = (hw + 2 * pad - k) // stride + 1
    dy = _bf(torch.randn(B, ho, ho, cout, generator=g))
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.float().permute(0, 3, 1, 2).requires_grad_(True)
    y = F.conv2d(xr, wr, None, stride, pad)
    y.backward(dy.float().permute(0, 3, 1, 2))
    dw = torch.zeros(cout, k * k * cin, device=dev)
    ops.conv2d_wgrad(dy.to(dev), x.to(dev), dw, kh=k, kw=k, stride=stride, pad=pad)
    _close(dw.reshape(cout, k, k, cin), wr.grad.permute(0, 2, 3, 1), 2e-3, 2e-2, "conv wgrad")
    w_t = w.permute(3, 1, 2, 0).reshape(cin, k * k * cout).contiguous()
    res = _bf(torch.randn(B, hw, hw, cin, generator=g))
    dx = torch.empty(B, hw, hw, cin, dtype=torch.bfloat16, device=dev)
    ops.conv2d_dgrad(dy.to(dev), w_t.to(dev), dx, kh=k, kw=k, stride=stride, pad=pad, residual=res.to(dev))
    _close(dx, xr.grad.permute(0, 2, 3, 1) + res.float(), 1.0 / 128, 2e-2, "conv dgrad")


@pytest.mark.parametrize("cin,cout,hw,B", [(64, 64, 56, 2), (128, 128, 28, 3), (256, 256, 14, 5), (512, 512, 7, 11),
                                            (64, 128, 14, 1), (128, 64, 7, 40), (192, 64, 28, 2), (64, 64, 12, 3)])
def test_conv3x3_wgrad_direct(dev, cin, cout, hw, B):
    """The LDS-resident-patch weight-gradient kernel (csrc/wgrad3x3.hip, splits=-2) against torch autograd and against
    the implicit-GEMM kernel (splits=-1): row-aligned tiles (56), tiles straddling image rows and images (28, 14, 7,
    12), short last tiles, accumulation into a non-zero dW."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(cin + cout + hw + B)
    x = _bf(torch.randn(B, hw, hw, cin, generator=g))
    dy = _bf(torch.randn(B, hw, hw, cout, generator=g))
    wr = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), wr, None, 1, 1).backward(dy.float().permute(0, 3, 1, 2))
    want = wr.grad.permute(0, 2, 3, 1)
    init = torch.randn(cout, 9 * cin, generator=g)
    dw = init.clone().to(dev)
    ops.conv2d_wgrad(dy.to(dev), x.to(dev), dw, kh=3, kw=3, stride=1, pad=1, splits=-2)
    _close((dw.cpu() - init).reshape(cout, 3, 3, cin), want, 2e-3 * math.sqrt(B * hw * hw / 6272), 2e-2, "direct conv wgrad")
    dw2 = torch.zeros(cout, 9 * cin, device=dev)
    ops.conv2d_wgrad(dy.to(dev), x.to(dev), dw2, kh=3, kw=3, stride=1, pad=1, splits=-1)
    dw3 = torch.zeros(cout, 9 * cin, device=dev)
    ops.conv2d_wgrad(dy.to(dev), x.to(dev), dw3, kh=3, kw=3, stride=1, pad=1)          # automatic: the direct kernel
    # same bf16 products, fp32 sums in a different order
    scale = dw2.abs().max().item()
    assert (dw3 - dw2).abs().max().item() <= 1e-4 * scale + 1e-5


def test_conv3x3_wgrad_direct_rejects_other_geometry(dev):
    from vit_amd import ops
    from vit_amd._lib import RvitError
    x = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16, device=dev)
    dy = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device=dev)
    dw = torch.zeros(64, 9 * 64, device=dev)
    with pytest.raises(RvitError):
        ops.conv2d_wgrad(dy, x, dw, kh=3, kw=3, stride=2, pad=1, splits=-2)


def test_conv1_wgrad_px4(dev):
    from vit_amd import ops
    B, H = 2, 224
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, 3, H, H, generator=g)
    xp = torch.zeros(B, H + 6, H + 8, 4, dtype=torch.bfloat16, device=dev)
    ops.input_pack(x.to(dev), xp, 3, 3)
    dy = _bf(torch.randn(B, 112, 112, 64, generator=g))
    wr = torch.zeros(64, 3, 7, 7, requires_grad=True)
    F.conv2d(_bf(x).float(), wr, None, 2, 3).backward(dy.float().permute(0, 3, 1, 2))
    dw = torch.zeros(64, 224, device=dev)
    ops.conv2d_wgrad(dy.to(dev), xp, dw, kh=7, kw=1, stride=2, pad=0, seg_len=32)
    got = dw.reshape(64, 7, 8, 4)[:, :, :7, :3].permute(0, 3, 1, 2)
    _close(got, wr.grad, 2e-3, 5e-2, "conv1 wgrad")


@pytest.mark.parametrize("rows,E", [(300, 256), (1000, 768), (7, 384)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_layernorm_bwd(dev, rows, E, accumulate):
    from vit_amd import ops
    g = torch.Generator().manual_seed(rows + E)
    x = torch.randn(rows, E, generator=g) * 2 + 0.3
    gam = torch.randn(E, generator=g)
    bet = torch.randn(E, generator=g)
    dy = _bf(torch.randn(rows, E, generator=g))
    base = torch.randn(rows, E, generator=g)
    xr = x.clone().requires_grad_(True)
    gr = gam.clone().requires_grad_(True)
    br = bet.clone().requires_grad_(True)
    F.layer_norm(xr, (E,), gr, br, 1e-5).backward(dy.float())
    mean = x.mean(1)
    rstd = 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
    dx = base.clone().to(dev)
    dxb = torch.empty(rows, E, dtype=torch.bfloat16, device=dev)
    dg = torch.zeros(E, device=dev)
    db = torch.zeros(E, device=dev)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), mean.to(dev), rstd.to(dev), gam.to(dev), dx, dg, db,
                      accumulate=accumulate, dx_bf16=dxb)
    ref = xr.grad + (base if accumulate else 0)
    _close(dx, ref, 1e-4, 1e-4, "ln dx")
    _close(dxb, ref, 1.0 / 128, 1e-3, "ln dx bf16")
    _close(dg, gr.grad, 1e-3, 1e-3 * math.sqrt(rows), "ln dgamma")
    _close(db, br.grad, 1e-3, 1e-3 * math.sqrt(rows), "ln dbeta")
    # rvit_layernorm_bwd_drop: the bf16 copy with the consuming site's dropout on it = rvit_dropout_apply on the plain
    # copy, bit for bit; the fp32 gradient and the affine gradients are untouched
    p_drop, seed = 0.25, 0x1234_5678_9ABC
    dx2 = base.clone().to(dev)
    dxb2 = torch.empty(rows, E, dtype=torch.bfloat16, device=dev)
    dg2, db2 = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), mean.to(dev), rstd.to(dev), gam.to(dev), dx2, dg2, db2,
                      accumulate=accumulate, dx_bf16=dxb2, drop=(p_drop, seed))
    want = ops.dropout_apply(dxb, torch.empty_like(dxb), p_drop, seed)
    assert torch.equal(dxb2, want), "fused dropout on the bf16 copy differs from rvit_dropout_apply"
    assert torch.equal(dx2, dx)
    kept = (dxb2 != 0).float().mean().item()
    assert abs(kept - (1 - p_drop)) < 0.05, kept
    with pytest.raises(ValueError):
        ops.layernorm_bwd(dy.to(dev), x.to(dev), mean.to(dev), rstd.to(dev), gam.to(dev), dx2, dg2, db2,
                          accumulate=accumulate, drop=(p_drop, seed))


@pytest.mark.parametrize("B,S,H", [(3, 50, 4), (2, 197, 3), (4, 17, 2), (1, 64, 1), (2, 100, 2), (1, 256, 1)])
def test_attention_bwd(dev, B, S, H):
    from vit_amd import ops
    E = H * 64
    g = torch.Generator().manual_seed(B * 1000 + S)
    qkv = _bf(torch.randn(B * S, 3 * E, generator=g))
    dout = _bf(torch.randn(B * S, E, generator=g))
    qkv_r = qkv.float().requires_grad_(True)
    q, k, v = [t.reshape(B, S, H, 64).transpose(1, 2) for t in qkv_r.split(E, dim=1)]
    o = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(B * S, E)
    o.backward(dout.float())
    out = torch.empty(B * S, E, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, S, device=dev)
    qd = qkv.to(dev)
    ops.attention(qd, out, B, S, H, lse=lse)
    dqkv = torch.zeros(B * S, 3 * E, dtype=torch.bfloat16, device=dev)
    ops.attention_bwd(qd, out, dout.to(dev), lse, dqkv, B, S, H)
    ref = qkv_r.grad
    scale = ref.abs().max().item()
    _close(dqkv, ref, 1.0 / 32, 0.02 * scale, "attention dqkv")
    # relative Frobenius error: the per-element bound above is loose, this one is not
    rel = (dqkv.float().cpu() - ref).norm() / ref.norm()
    assert rel < 0.02, f"attention backward relative error {rel:.4f}"


def test_ce_loss(dev):
    from vit_amd import ops
    g = torch.Generator().manual_seed(2)
    B, C = 300, 11
    z = torch.randn(B, C, generator=g) * 3
    y = torch.randint(0, C, (B,), generator=g)
    zr = z.clone().requires_grad_(True)
    loss = F.cross_entropy(zr, y, label_smoothing=0.05)
    loss.backward()
    ls = torch.zeros(1, device=dev)
    dl = torch.empty(B, C, device=dev)
    dlb = torch.full((B, 16), 7.0, dtype=torch.bfloat16, device=dev)
    ops.ce_loss(z.to(dev), y.to(dev), ls, dl, dlb, 0.05, grad_scale=1.0 / B)
    _close(ls, loss.detach().reshape(1), 1e-5, 1e-5, "loss")
    _close(dl, zr.grad, 1e-4, 1e-7, "dlogits")
    _close(dlb[:, :C], zr.grad, 1.0 / 128, 1e-6, "dlogits bf16")
    assert float(dlb[:, C:].abs().sum()) == 0.0


@pytest.mark.parametrize("name,c,hw,B,res,relu", [
    ("c64", 64, 56, 5, False, True), ("c64-res", 64, 56, 24, True, True), ("direct64-l4", 512, 7, 37, True, True),
    ("direct64-l3", 256, 14, 3, False, False), ("p8-128", 128, 28, 256, True, True), ("p8-256", 256, 14, 256, False, True),
    ("c256", 512, 7, 512, True, True)])
def test_dgrad_with_bn_backward_sums(dev, name, c, hw, B, res, relu):
    """rvit_conv2d_dgrad_bnr_bf16 (round 3): the 3x3 dgrad whose output enters a BatchNorm's backward pass writes that
    pass's two sums from its epilogue.  dx bit-identical to rvit_conv2d_dgrad_bf16's; the sums equal to those of the
    stored dx (exact arithmetic in float64) up to fp32 round-off, and to rvit_bn_bwd_reduce's; two runs identical."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(c + hw + B)
    dy = _bf(torch.randn(B, hw, hw, c, generator=g) / 8).to(dev)
    w_t = _bf(torch.randn(c, 9 * c, generator=g) / math.sqrt(9 * c)).to(dev)
    r = _bf(torch.randn(B, hw, hw, c, generator=g) / 4).to(dev) if res else None
    y = _bf(torch.randn(B, hw, hw, c, generator=g) * 1.3 + 0.2).to(dev)
    M = B * hw * hw
    mask = torch.randint(0, 256, (M * (c // 8),), generator=g, dtype=torch.uint8).to(dev) if relu else None
    yf = y.float().reshape(M, c)
    fs = torch.cat([yf.sum(0), (yf * yf).sum(0)]).contiguous()
    eps = 1e-5
    dx_ref = torch.empty(B, hw, hw, c, dtype=torch.bfloat16, device=dev)
    ops.conv2d_dgrad(dy, w_t, dx_ref, kh=3, kw=3, stride=1, pad=1, residual=r)
    ws = ops.conv_stats_workspace(dev)
    runs = []
    for _ in range(2):
        dx = torch.full_like(dx_ref, float("nan"))
        bs = torch.zeros(2 * c, device=dev)
        fused = ops.conv2d_dgrad_bn_sums(dy, w_t, dx, kh=3, kw=3, stride=1, pad=1, residual=r, bn_y=y, bn_relu_mask=mask,
                                         bn_fwd_sums=fs, bn_bwd_sums=bs, eps=eps, stat_ws=ws)
        assert fused, f"{name}: no hook"
        assert torch.equal(dx, dx_ref)
        runs.append(bs)
    assert torch.equal(runs[0], runs[1])
    d = dx_ref.double().reshape(M, c)
    if relu:
        bits = ((mask.reshape(M, c // 8, 1).int() >> torch.arange(8, device=dev).int()) & 1).reshape(M, c)
        d = d * bits
    mean = (fs[:c] / M).double()
    var = (fs[c:] / M).double() - mean * mean
    rstd = 1.0 / torch.sqrt(var.clamp_min(0) + eps)
    yc = yf.double() - mean
    exact = torch.cat([d.sum(0), rstd * (d * yc).sum(0)]).float()
    scale = torch.cat([d.abs().sum(0), rstd * (d * yc).abs().sum(0)]).float().clamp_min(1e-6)
    err = ((runs[0] - exact).abs() / scale).max().item()
    assert err <= 2e-5, err
    # and the pass it replaces
    dyo = torch.empty_like(dx_ref)
    bs2 = torch.zeros(2 * c, device=dev)
    ops.bn_bwd(dx_ref, None, y, fs, bs2, torch.ones(c, device=dev), dyo, torch.zeros(c, device=dev), torch.zeros(c, device=dev),
               eps=eps, relu=relu, relu_mask=mask, workspace=ops.bn_workspace(dev))
    err2 = ((runs[0] - bs2).abs() / scale).max().item()
    assert err2 <= 4e-5, err2
    dyf = torch.empty_like(dx_ref)
    ops.bn_bwd(dx_ref, None, y, fs, runs[0].clone(), torch.ones(c, device=dev), dyf, torch.zeros(c, device=dev),
               torch.zeros(c, device=dev), eps=eps, relu=relu, relu_mask=mask, reduced=True)
    assert (dyf.float() - dyo.float()).abs().max().item() <= 2.0 ** -7 * dyo.float().abs().max().item()


def test_colsum_and_batch_sum(dev):
    from vit_amd import ops
    g = torch.Generator().manual_seed(6)
    a = _bf(torch.randn(1234, 200, generator=g))
    out = torch.ones(200, device=dev)
    ops.colsum(a.to(dev), out)
    _close(out, 1 + a.float().sum(0), 1e-4, 1e-3, "colsum")
    x = torch.randn(9, 50 * 64, generator=g)
    o2 = torch.ones(50 * 64, device=dev)
    ops.batch_sum(x.to(dev), o2)
    _close(o2, 1 + x.sum(0), 1e-5, 1e-5, "batch_sum")
    for B, n in ((300, 4 * 37), (129, 64), (1, 12800), (257, 50 * 256)):   # ragged column blocks, several row rounds
        x = torch.randn(B, n, generator=g)
        runs = []
        for _ in range(2):
            o = torch.ones(n, device=dev)
            ops.batch_sum(x.to(dev), o)
            runs.append(o)
        assert torch.equal(runs[0], runs[1])
        _close(runs[0], 1 + x.double().sum(0).float(), 1e-5, 2e-5, f"batch_sum {B}x{n}")


def test_gemm_gate_and_a_remap(dev):
    from vit_amd import ops
    g = torch.Generator().manual_seed(12)
    M, N, K = 150, 512, 128
    dy = _bf(torch.randn(M, K, generator=g))
    wt = _bf(torch.randn(N, K, generator=g) / 8)
    u = _bf(torch.randn(M, N, generator=g))
    ur = u.float().requires_grad_(True)
    F.gelu(ur).backward(dy.float() @ wt.float().T)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(dy.to(dev), wt.to(dev), out, gate=u.to(dev), gate_mode="gelu_grad")
    _close(out, ur.grad, 1.0 / 128, 2e-3, "gelu-grad gate")
    ops.gemm(dy.to(dev), wt.to(dev), out, gate=u.to(dev), gate_mode="relu_mask")
    _close(out, (dy.float() @ wt.float().T) * (u.float() > 0), 1.0 / 128, 2e-3, "relu-mask gate")
    # A rows taken past the cls row of every sample
    B, P, E = 4, 49, 64
    tok = _bf(torch.randn(B * (P + 1), E, generator=g))
    w2 = _bf(torch.randn(128, E, generator=g) / 8)
    o2 = torch.empty(B * P, 128, dtype=torch.float32, device=dev)
    ops.gemm(tok.to(dev), w2.to(dev), o2, a_rows=B * P, a_row_group=P, a_row_extra=1)
    ref = tok.float().reshape(B, P + 1, E)[:, 1:].reshape(-1, E) @ w2.float().T
    _close(o2, ref, 1e-4, 1e-4, "A row remap")


@pytest.mark.parametrize("M,C", [(1000, 64), (2 * 28 * 28, 128), (333, 512), (5 * 112 * 112, 64)])
@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False)])
def test_batchnorm_train_fwd_bwd(dev, M, C, relu, with_res):
    """Training-mode BatchNorm (+ residual + ReLU) forward and backward (csrc/bn.hip) against torch autograd on the
    same bf16-rounded operands; with relu the backward is run twice, from the forward output z and from the
    sign-bit mask `bn_apply` writes, and the two must agree bit for bit."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(M + C)
    y = (torch.randn(M, C, generator=g) * 1.5 + 0.3).bfloat16()
    res = torch.randn(M, C, generator=g).bfloat16() if with_res else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    dz = torch.randn(M, C, generator=g).bfloat16()
    # reference
    yr = y.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    mean, var = yr.mean(0), yr.var(0, unbiased=False)
    zr = (yr - mean) / torch.sqrt(var + 1e-5) * gr + br
    if with_res:
        zr = zr + res.float()
    pre = zr.detach()
    if relu:
        zr = torch.relu(zr)
    zr.backward(dz.float())
    # elements whose pre-activation is within rounding of 0 may take either ReLU branch (the batch statistics are float
    # atomics, so their last bits move from run to run): their dy is not compared, the channel sums hardly notice them
    sure = pre.abs() > 1e-3 if relu else torch.ones_like(pre, dtype=torch.bool)
    # device
    yd, dzd = y.to(dev).view(1, 1, M, C), dz.to(dev).view(1, 1, M, C)
    fs = torch.zeros(2 * C, device=dev)
    ops.bn_stats(yd, fs)
    z = torch.empty_like(yd)
    mask = torch.zeros(M * (C // 8), dtype=torch.uint8, device=dev) if relu else None
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ops.bn_apply(yd, fs, gamma.to(dev), beta.to(dev), z, eps=1e-5, relu=relu,
                 residual=None if res is None else res.to(dev).view(1, 1, M, C), running_mean=rm, running_var=rv,
                 relu_mask=mask)
    assert (z.view(M, C).float().cpu() - zr.detach()).abs().max().item() <= 2.0 ** -7 * max(1.0, zr.abs().max().item())
    assert torch.allclose(rm.cpu(), 0.1 * mean.detach(), atol=1e-4)
    assert torch.allclose(rv.cpu(), 0.9 + 0.1 * yr.detach().var(0, unbiased=True), rtol=1e-3, atol=1e-4)
    if relu:
        bits = ((mask.view(M, C // 8, 1) >> torch.arange(8, device=dev, dtype=torch.uint8)) & 1).view(M, C).bool()
        assert torch.equal(bits, z.view(M, C) > 0)

    def backward(**kw):
        bs = torch.zeros(2 * C, device=dev)
        dy = torch.empty_like(yd)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dzm = torch.empty_like(yd)
        ops.bn_bwd(dzd, kw.get("z"), yd, fs, bs, gamma.to(dev), dy, dg, db, eps=1e-5, relu=relu, dz_masked=dzm,
                   relu_mask=kw.get("relu_mask"))
        return dy, dg, db, dzm

    dy, dg, db, dzm = backward(z=z)
    scale = max(1.0, yr.grad.abs().max().item())
    assert ((dy.view(M, C).float().cpu() - yr.grad).abs() * sure).max().item() <= 0.02 * scale
    assert (~sure).sum().item() <= 2e-3 * sure.numel()
    assert torch.allclose(dg.cpu(), gr.grad, rtol=2e-2, atol=0.05 * max(1.0, gr.grad.abs().max().item()) * 0.1 + 0.5)
    assert torch.allclose(db.cpu(), br.grad, rtol=2e-2, atol=0.5)
    if relu:
        dy2, dg2, db2, dzm2 = backward(relu_mask=mask)
        # the masked gradient is identical; the per-channel sums are float atomics (same addends, order may differ),
        # so dy may move by an ulp of bf16
        assert torch.equal(dzm, dzm2)
        assert (dy.float() - dy2.float()).abs().max().item() <= 2.0 ** -7 * scale
        assert torch.allclose(dg, dg2, rtol=1e-4, atol=1e-2) and torch.allclose(db, db2, rtol=1e-4, atol=1e-2)
        with pytest.raises(ValueError):
            ops.bn_bwd(dzd, None, yd, fs, torch.zeros(2 * C, device=dev), gamma.to(dev), dy, dg, db, eps=1e-5, relu=True)
        with pytest.raises(ValueError):
            ops.bn_bwd(dzd, None, yd, fs, torch.zeros(2 * C, device=dev), gamma.to(dev), dy, dg, db, eps=1e-5, relu=True,
                       relu_mask=mask[:-1])


@pytest.mark.parametrize("M,C", [(1000, 64), (40 * 28 * 28, 128), (333, 512), (64 * 49, 512), (8 * 112 * 112, 64),
                                 (5000, 1024), (9 * 196, 256)])
def test_batchnorm_ordered_reductions_are_bit_reproducible(dev, M, C):
    """bn_stats / bn_bwd with a workspace run the ORDERED reductions (no float atomics): ten runs give the same bits,
    the totals agree with a float64 sum, and the workspace can be shared by consecutive calls of different shapes."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(M * 3 + C)
    y = (torch.randn(M, C, generator=g) * 1.5 + 0.3).bfloat16().to(dev).view(1, 1, M, C)
    dz = torch.randn(M, C, generator=g).bfloat16().to(dev).view(1, 1, M, C)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    ws = ops.bn_workspace(dev)
    other = torch.randn(1, 1, 777, 128, device=dev).bfloat16()     # a different shape through the same workspace

    def run():
        fs = torch.zeros(2 * C, device=dev)
        ops.bn_stats(y, fs, ws)
        ops.bn_stats(other, torch.zeros(256, device=dev), ws)
        z = torch.empty_like(y)
        mask = torch.zeros(M * (C // 8), dtype=torch.uint8, device=dev)
        ops.bn_apply(y, fs, gamma, torch.zeros(C, device=dev), z, eps=1e-5, relu=True, relu_mask=mask)
        bs = torch.zeros(2 * C, device=dev)
        dy = torch.empty_like(y)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ops.bn_bwd(dz, None, y, fs, bs, gamma, dy, dg, db, eps=1e-5, relu=True, relu_mask=mask, workspace=ws)
        return fs, bs, dy, dg, db, z

    first = run()
    for _ in range(9):
        again = run()
        for a, b in zip(first, again):
            assert torch.equal(a, b)
    fs = first[0].double().cpu()
    yf = y.view(M, C).double().cpu()
    assert torch.allclose(fs[:C], yf.sum(0), rtol=1e-5, atol=1e-2)
    assert torch.allclose(fs[C:], (yf * yf).sum(0), rtol=1e-5, atol=1e-2)
    assert int(ws[:1].view(torch.int32).item()) == 0            # the ticket counter is re-armed
    # same results (to float round-off) as the atomic form
    fs2 = torch.zeros(2 * C, device=dev)
    ops.bn_stats(y, fs2)
    assert torch.allclose(first[0], fs2, rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("B,H,W,C", [(2, 112, 112, 64), (3, 10, 14, 64), (1, 7, 9, 128)])
def test_bn_apply_pool_fused(dev, B, H, W, C):
    """BatchNorm + ReLU + MaxPool2d(3,2,1) in one pass (the training stem entry) against the two-kernel path
    (bn_apply, then maxpool3x3s2): pooled map, argmax taps, sign bits and running statistics all bit-identical;
    odd map sizes exercise the clipped windows."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(B * H + W)
    y = (torch.randn(B, H, W, C, generator=g) * 1.3 - 0.1).bfloat16().to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    fs = torch.zeros(2 * C, device=dev)
    ops.bn_stats(y, fs)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    z = torch.empty_like(y)
    m_ref = torch.zeros(B * H * W * (C // 8), dtype=torch.uint8, device=dev)
    rm1, rv1 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ops.bn_apply(y, fs, gamma, beta, z, eps=1e-5, relu=True, running_mean=rm1, running_var=rv1, relu_mask=m_ref)
    p_ref = torch.empty(B, Ho, Wo, C, dtype=torch.bfloat16, device=dev)
    a_ref = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=dev)
    ops.maxpool3x3s2(z, p_ref, a_ref)
    p = torch.full_like(p_ref, float("nan"))
    a = torch.full_like(a_ref, 255)
    m = torch.full_like(m_ref, 0xAA)
    rm2, rv2 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ops.bn_apply_pool(y, fs, gamma, beta, p, a, m, eps=1e-5, running_mean=rm2, running_var=rv2)
    assert torch.equal(p, p_ref) and torch.equal(a, a_ref) and torch.equal(m, m_ref)
    assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
    # and against torch
    zr = torch.relu(torch.nn.functional.batch_norm(y.float().permute(0, 3, 1, 2).cpu(), None, None, gamma.cpu(), beta.cpu(),
                                                   True, 0.1, 1e-5))
    pr = torch.nn.functional.max_pool2d(zr, 3, 2, 1).permute(0, 2, 3, 1)
    assert (p.float().cpu() - pr).abs().max().item() <= 2.0 ** -7 * max(1.0, pr.abs().max().item())


@pytest.mark.parametrize("B,H,W,C", [(3, 112, 112, 64), (2, 15, 9, 64), (5, 8, 8, 128), (1, 7, 30, 16), (2, 2, 2, 64),
                                     (7, 6, 20, 32)])
@pytest.mark.parametrize("ordered", [False, True])
def test_bn_bwd_behind_maxpool_fused(dev, B, H, W, C, ordered):
    """bn_bwd_pool (the max-pool's gradient gathered inside the two BatchNorm backward passes) against
    maxpool3x3s2_bwd + bn_bwd(relu_mask=...).  Given the same two sums, dy is bit-identical (the gathered gradient of a
    pixel is the value maxpool_bwd stores).  The sums themselves: bit-identical with the ordered reductions on odd maps
    (per-pixel kernels, bn_bwd's partition); on even maps the passes walk 2 x 2 pixel blocks and add in another order:
    equal to fp32 round-off, identical from run to run.  Odd sizes: clipped windows, rows in one window only."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(B * H + W + C)
    y = (torch.randn(B, H, W, C, generator=g) * 1.3 - 0.1).bfloat16().to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    ws = ops.bn_workspace(dev) if ordered else None
    fs = torch.zeros(2 * C, device=dev)
    ops.bn_stats(y, fs, workspace=ws)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    pooled = torch.empty(B, Ho, Wo, C, dtype=torch.bfloat16, device=dev)
    arg = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=dev)
    mask = torch.empty(B * H * W * (C // 8), dtype=torch.uint8, device=dev)
    ops.bn_apply_pool(y, fs, gamma, beta, pooled, arg, mask, eps=1e-5)
    dpool = torch.randn(B, Ho, Wo, C, generator=g).bfloat16().to(dev)

    dx = torch.empty_like(y)
    ops.maxpool3x3s2_bwd(dpool, arg, dx)
    bs1, dg1, db1 = torch.zeros(2 * C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy1 = torch.empty_like(y)
    ops.bn_bwd(dx, None, y, fs, bs1, gamma, dy1, dg1, db1, eps=1e-5, relu=True, relu_mask=mask, workspace=ws)

    bs2, dg2, db2 = torch.zeros(2 * C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy2 = torch.full_like(y, float("nan"))
    ops.bn_bwd_pool(dpool, arg, mask, y, fs, bs2, gamma, dy2, dg2, db2, eps=1e-5, workspace=ws)
    # the apply pass alone on bn_bwd's sums: the same bits
    dy3 = torch.full_like(y, float("nan"))
    ops.bn_bwd_pool(dpool, arg, mask, y, fs, bs1.clone(), gamma, dy3, torch.zeros(C, device=dev), torch.zeros(C, device=dev),
                    eps=1e-5, reduced=True)
    assert torch.equal(dy1, dy3)
    if ordered and ((H | W) & 1):
        assert torch.equal(bs1, bs2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
        assert torch.equal(dy1, dy2)
    elif ordered:
        bs3 = torch.zeros(2 * C, device=dev)
        ops.bn_bwd_pool(dpool, arg, mask, y, fs, bs3, gamma, torch.empty_like(y), torch.zeros(C, device=dev),
                        torch.zeros(C, device=dev), eps=1e-5, workspace=ws)
        assert torch.equal(bs2, bs3)
        scale = max(1.0, bs1.abs().max().item())
        assert (bs1 - bs2).abs().max().item() <= 2e-5 * scale * math.sqrt(B * H * W)
        assert (dy1.float() - dy2.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, dy1.float().abs().max().item())
    else:
        assert torch.allclose(bs1, bs2, rtol=1e-4, atol=1e-3)
        assert (dy1.float() - dy2.float()).abs().max().item() <= 2e-2 * max(1.0, dy1.float().abs().max().item())


@pytest.mark.parametrize("B,Hy,Wy,cin,cout", [(2, 28, 28, 64, 128), (3, 7, 7, 256, 512), (1, 5, 9, 64, 64)])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_dgrad_stride2_by_parity(dev, B, Hy, Wy, cin, cout, with_res):
    """Input gradient of a 3x3 / stride 2 / pad 1 conv decomposed by output parity (four dense stride-1 convs over dY)
    and the accumulating 1x1 / stride 2 form, against torch autograd and against the generic transposed gather."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(B + Hy + cin)
    w = _bf(torch.randn(cout, 3, 3, cin, generator=g) / math.sqrt(9 * cin))            # [Cout, kh, kw, Cin]
    dy = _bf(torch.randn(B, Hy, Wy, cout, generator=g))
    res = _bf(torch.randn(B, 2 * Hy, 2 * Wy, cin, generator=g)) if with_res else None
    xr = torch.zeros(B, cin, 2 * Hy, 2 * Wy, requires_grad=True)
    F.conv2d(xr, w.float().permute(0, 3, 1, 2), None, 2, 1).backward(dy.float().permute(0, 3, 1, 2))
    want = xr.grad.permute(0, 2, 3, 1) + (res.float() if with_res else 0)
    taps = w.permute(3, 1, 2, 0).reshape(cin, 9, cout)                                  # [Cin, tap, Cout]
    w_s2 = taps[:, list(ops.S2_TAP_ORDER), :].reshape(cin, 9 * cout).contiguous().to(dev)
    dx = torch.full((B, 2 * Hy, 2 * Wy, cin), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.conv2d_dgrad_s2(dy.to(dev), w_s2, dx, residual=None if res is None else res.to(dev))
    _close(dx, want, 1.0 / 128, 2e-2, "stride-2 dgrad by parity")
    dx_old = torch.empty_like(dx)
    ops.conv2d_dgrad(dy.to(dev), taps.reshape(cin, 9 * cout).contiguous().to(dev), dx_old, kh=3, kw=3, stride=2, pad=1,
                     residual=None if res is None else res.to(dev))
    assert (dx.float() - dx_old.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, dx_old.float().abs().max().item())
    # 1x1 / stride 2, accumulated into dx: only the (even, even) pixels change
    w1 = _bf(torch.randn(cout, cin, generator=g) / math.sqrt(cin))                      # [Cout, Cin]
    before = dx.clone()
    ops.conv1x1_dgrad_s2_acc(dy.to(dev), w1.t().contiguous().to(dev), dx)
    add = dy.float() @ w1.float()                                                       # [B,Hy,Wy,Cin]
    want2 = before.float().cpu().clone()
    want2[:, ::2, ::2, :] += add
    _close(dx, want2, 1.0 / 128, 2e-2, "1x1 stride-2 dgrad, accumulated")
    mask = torch.ones(2 * Hy, 2 * Wy, dtype=torch.bool)
    mask[::2, ::2] = False
    assert torch.equal(dx[:, mask.to(dev)], before[:, mask.to(dev)])
