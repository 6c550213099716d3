"""The reference's `backbone='mobilenetv3'` (timm mobilenetv3_small_100 features_only, train_vit.py:68-74,87): the
HBM-bound kernels of csrc/mobile.hip against torch on the same bf16 operands, the new GEMM activations, and the whole
classifier against the fp32 CPU oracle (oracle/ref_model.py::MobileNetV3SmallFeatures; parity at the timm boundary is
unpinned, as for the resnet stems).  The training pass: tests/test_mobilenet_train_gpu.py, test_mobilenet_train_step_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _act(t, name):
    return {"relu": torch.relu, "hswish": F.hardswish, "hsigmoid": F.hardsigmoid, None: lambda u: u}[name](t)


# Wo % 7 == 0 runs the strip kernel (7 outputs per thread), anything else the one-output-per-thread kernel
@pytest.mark.parametrize("B,H,C,k,stride,act", [(3, 56, 16, 3, 2, "relu"), (2, 28, 96, 5, 2, "hswish"), (2, 14, 240, 5, 1, "hswish"),
                                               (5, 7, 576, 5, 1, "hswish"), (1, 9, 8, 3, 1, None), (2, 15, 24, 5, 2, "relu"),
                                               (2, 112, 16, 3, 2, "relu"), (2, 28, 88, 3, 1, "relu"), (3, 21, 40, 5, 1, None),
                                               (1, 14, 8, 3, 2, "hswish")])
def test_dwconv_matches_torch(dev, B, H, C, k, stride, act):
    from vit_amd import ops
    g = torch.Generator().manual_seed(H * C + k)
    x = torch.randn(B, H, H, C, generator=g).to(BF16)
    w = (torch.randn(C, 1, k, k, generator=g) * 0.3).to(BF16)
    scale, bias = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(B, Ho, Ho, C, dtype=BF16, device=dev)
    ops.dwconv(x.to(dev), w.reshape(C, k * k).t().contiguous().to(dev), scale.to(dev), bias.to(dev), out, k=k, stride=stride,
               pad=k // 2, act=act)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, stride, k // 2, groups=C)
    ref = _act(ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), act).permute(0, 2, 3, 1)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= 2 ** -7 * max(1.0, ref.abs().max().item()), err    # one bf16 rounding of the output


def test_avgpool_channel_scale_and_new_gemm_activations(dev):
    from vit_amd import ops
    g = torch.Generator().manual_seed(3)
    for B, H, C in ((4, 56, 16), (3, 14, 240), (2, 7, 576), (5, 28, 96)):
        x = torch.randn(B, H, H, C, generator=g).to(BF16)
        pooled = torch.empty(B, C, dtype=BF16, device=dev)
        ops.avgpool(x.to(dev), pooled)
        ref = x.float().mean((1, 2))
        assert (pooled.float().cpu() - ref).abs().max().item() <= 2 ** -8 * max(1.0, ref.abs().max().item()) + 1e-3
        gate = torch.rand(B, C, generator=g).to(BF16)
        xs = x.to(dev).clone()
        ops.channel_scale(xs, gate.to(dev))
        want = (x.float() * gate.float().view(B, 1, 1, C)).to(BF16)
        assert torch.equal(xs.cpu(), want)
    a = torch.randn(300, 96, generator=g).to(BF16)
    w = (torch.randn(40, 96, generator=g) * 0.3).to(BF16)
    bias, scale = torch.randn(40, generator=g), torch.rand(40, generator=g) + 0.5
    for act in ("hswish", "hsigmoid"):
        out = torch.empty(300, 40, dtype=BF16, device=dev)
        ops.gemm(a.to(dev), w.to(dev), out, scale=scale.to(dev), bias=bias.to(dev), act=act)
        ref = _act((a.float() @ w.float().T) * scale + bias, act)
        assert (out.float().cpu() - ref).abs().max().item() <= 2 ** -7 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,H,C,RD", [(3, 28, 16, 8), (5, 14, 240, 64), (2, 7, 576, 144), (4, 14, 96, 24), (1, 14, 120, 32)])
def test_squeeze_excite_one_launch_matches_the_four_launch_form(dev, B, H, C, RD):
    """Same rounding points (pooled mean, ReLU output, gate in bf16); the projections sum in a different order than the
    MFMA GEMMs, so a gate may differ by one bf16 step and the product by one more rounding of it."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(C + RD)
    x = torch.randn(B, H, H, C, generator=g).to(BF16)
    w1, b1 = (torch.randn(RD, C, generator=g) * 0.2).to(BF16), torch.randn(RD, generator=g) * 0.1
    w2, b2 = (torch.randn(C, RD, generator=g) * 0.3).to(BF16), torch.randn(C, generator=g) * 0.5
    assert ops.squeeze_excite_supported(H * H, C, RD)
    fused = x.to(dev).clone()
    ops.squeeze_excite(fused, w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev))
    ref4 = x.to(dev).clone()
    pooled, hid, gate = (torch.empty(B, n, dtype=BF16, device=dev) for n in (C, RD, C))
    ops.avgpool(ref4, pooled)
    ops.gemm(pooled, w1.to(dev), hid, bias=b1.to(dev), act="relu")
    ops.gemm(hid, w2.to(dev), gate, bias=b2.to(dev), act="hsigmoid")
    ops.channel_scale(ref4, gate)
    xf = x.float()
    p = xf.mean((1, 2)).to(BF16).float()
    h = torch.relu(p @ w1.float().T + b1).to(BF16).float()
    gt = F.hardsigmoid(h @ w2.float().T + b2).to(BF16).float()
    want = xf * gt.view(B, 1, 1, C)
    for got in (fused, ref4):
        err = (got.float().cpu() - want).abs()
        assert (err <= 2 ** -6 * xf.abs() + 1e-6).all(), err.max().item()   # gate off by <= 1 bf16 step + 1 rounding
    assert (fused.float() - ref4.float()).abs().max().item() <= 2 ** -6 * xf.abs().max().item()
    assert (fused == ref4).float().mean().item() >= 0.9
    assert not ops.squeeze_excite_supported(56 * 56, 72, 24)      # 451 KB map: the engine falls back to four launches
    big = torch.zeros(1, 56, 56, 72, dtype=BF16, device=dev)
    with pytest.raises(Exception, match="LDS"):
        ops.squeeze_excite(big, torch.zeros(24, 72, dtype=BF16, device=dev), torch.zeros(24, device=dev),
                           torch.zeros(72, 24, dtype=BF16, device=dev), torch.zeros(72, device=dev))


@pytest.mark.parametrize("B,H,W,act", [(3, 224, 224, "hswish"), (2, 32, 48, None), (1, 6, 4, "relu")])
def test_conv3x3s2_rgb_matches_torch(dev, B, H, W, act):
    from vit_amd import ops
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(16, 3, 3, 3, generator=g) * 0.3
    scale, bias = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.2
    out = torch.empty(B, H // 2, W // 2, 16, dtype=BF16, device=dev)
    ops.conv3x3s2_rgb(x.to(dev), w.permute(2, 3, 1, 0).reshape(27, 16).contiguous().to(dev), scale.to(dev), bias.to(dev), out, act=act)
    ref = _act(F.conv2d(x, w, None, 2, 1) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), act).permute(0, 2, 3, 1)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= 2 ** -8 * max(1.0, ref.abs().max().item()) + 1e-5, err   # fp32 arithmetic, one bf16 rounding of the output


@pytest.mark.parametrize("M,K,N,act,res", [(3 * 56 * 56, 16, 16, None, False), (2 * 56 * 56 + 5, 16, 72, "relu", False),
                                            (28 * 28, 72, 24, None, False), (1000, 88, 24, None, True), (777, 24, 96, "hswish", False),
                                            (14 * 14 * 3, 240, 40, None, True), (100, 40, 240, "hswish", False),
                                            (49 * 4, 144, 48, None, True), (1, 48, 144, "hswish", False), (16, 120, 48, None, False),
                                            (70000, 40, 120, "hswish", False)])
def test_pointwise_conv_matches_torch_and_the_general_gemm(dev, M, K, N, act, res):
    from vit_amd import ops
    g = torch.Generator().manual_seed(M + K * N)
    x = torch.randn(M, K, generator=g).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.3).to(BF16)
    scale, bias = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    r = torch.randn(M, N, generator=g).to(BF16) if res else None
    assert ops.pointwise_supported(K, N)
    out, out_g = (torch.full((M, N), 7.0, dtype=BF16, device=dev) for _ in range(2))
    rd = r.to(dev) if res else None
    ops.pointwise_conv(x.to(dev), w.to(dev), out, scale=scale.to(dev), bias=bias.to(dev), residual=rd, act=act)
    ops.gemm(x.to(dev), w.to(dev), out_g, scale=scale.to(dev), bias=bias.to(dev), residual=rd, act=act)
    ref = (x.float() @ w.float().T) * scale + bias
    ref = _act(ref + r.float() if res else ref, act)
    tol = 2 ** -7 * max(1.0, ref.abs().max().item())
    assert (out.float().cpu() - ref).abs().max().item() <= tol
    assert (out.float() - out_g.float()).abs().max().item() <= tol
    assert (out == out_g).float().mean().item() >= 0.98    # same products, fp32 sums in a different order


def test_pointwise_conv_refuses_what_it_cannot_hold(dev):
    from vit_amd import ops
    assert not ops.pointwise_supported(96, 576) and not ops.pointwise_supported(576, 96) and not ops.pointwise_supported(20, 16)
    with pytest.raises(Exception, match="rvit_gemm_bf16"):
        ops.pointwise_conv(torch.zeros(8, 576, dtype=BF16, device=dev), torch.zeros(96, 576, dtype=BF16, device=dev),
                           torch.zeros(8, 96, dtype=BF16, device=dev))


def _pair(dev, seed=7):
    from oracle.ref_model import CNNViTHybridRef
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import synth_state_dict
    cfg = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone="mobilenetv3")
    m = CNNViTHybrid(**cfg)
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed)
    # synth_state_dict's scales were chosen for the resnet stems; through this stem's hard-swish / squeeze-excite chain
    # they let the signal decay 4x per stage until the 576-channel map (and with it every logit) no longer depends on
    # the input (batch std 0.003) and any parity check becomes vacuous.  These gains keep every stage O(1) with a batch
    # std as large as its mean magnitude (measured on the oracle: 0.28/0.31, 0.20/0.22, 0.78/0.87, 1.1/1.3, 1.4/1.8).
    for k in sd:
        if k.startswith("cnn_backbone.blocks.") and k.endswith("conv_dw.weight"):
            sd[k] = sd[k] * 1.5
        if k.startswith("cnn_backbone.blocks.") and k.endswith(("conv_pwl.weight", "conv_pw.weight", ".conv.weight")):
            sd[k] = sd[k] * 1.25
        if k.startswith("cnn_backbone.") and ".bn" in k and k.endswith(".bias"):
            sd[k] = sd[k] * 0.3
    m.load_state_dict(sd)
    ref = CNNViTHybridRef(**cfg)
    ref.load_state_dict(sd)
    return m.to(dev).eval(), ref.eval()


# Fixed bounds at 1.5x the error measured on MI355X for this fixture (bf16 storage of 52 maps and every GEMM operand
# against the fp32 oracle): max |dlogit| 0.545 on logits that reach 10.0, feature map max diff 0.189.
# That is the size bf16 storage alone produces: the fp32 oracle with only its BN / activation / Linear outputs rounded to
# bf16 (CPU, fp32 weights) moves by 0.345 / 0.120 / rel L2 0.030 on the same inputs; the HIP path also rounds weights.
MB_LOGIT_BOUND = 0.82
MB_FEATURE_BOUND = 0.29
MB_FEATURE_REL_L2 = 0.05


def test_mobilenetv3_model_matches_oracle(dev):
    from vit_amd.synth import micro_doppler_batch
    torch.set_num_threads(16)
    m, ref = _pair(dev)
    x = micro_doppler_batch(12, seed=21)
    with torch.no_grad():
        want = ref(x)
        feat_want = ref.forward_features(x)
        got = m(x.to(dev)).cpu()
        feat = m.forward_features(x.to(dev)).cpu()
        again = m(x.to(dev)).cpu()
        one = m(x[3:4].to(dev)).cpu()
    err = (got - want).abs().max().item()
    print(f"mobilenetv3: max |dlogit| {err:.4f} (bound {MB_LOGIT_BOUND}), |logit| max {want.abs().max():.2f}, "
          f"features max diff {(feat - feat_want).abs().max():.4f}, oracle classes {sorted(set(want.argmax(1).tolist()))}")
    assert torch.isfinite(got).all() and err <= MB_LOGIT_BOUND
    rel = ((feat - feat_want).norm() / feat_want.norm()).item()
    print(f"mobilenetv3: features rel L2 {rel:.4f}, |feature| max {feat_want.abs().max():.2f}")
    assert (feat - feat_want).abs().max().item() <= MB_FEATURE_BOUND and rel <= MB_FEATURE_REL_L2
    srt = want.sort(dim=1).values
    decided = (srt[:, -1] - srt[:, -2]) > 2 * MB_LOGIT_BOUND
    assert (got.argmax(1) == want.argmax(1))[decided].all() and decided.float().mean() >= 0.3
    assert torch.equal(got, again) and torch.allclose(got[3:4], one, atol=1e-5, rtol=0)   # deterministic, batch-invariant
    assert want.std(0).mean().item() > 0.5 and len(set(want.argmax(1).tolist())) >= 3   # the logits depend on the input


def test_mobilenetv3_checkpoint_round_trip(dev, tmp_path):
    from vit_amd.infer import load_model
    from vit_amd.synth import micro_doppler_batch, synth_checkpoint
    m, _ = _pair(dev)
    path = tmp_path / "cnn_vit_radar.pt"
    ck = synth_checkpoint(m, 7)
    ck["model"] = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    torch.save(ck, path)
    m2, classes, _ = load_model(str(path), dev)
    x = micro_doppler_batch(4, seed=2).to(dev)
    with torch.no_grad():
        assert torch.equal(m(x), m2(x))
    assert len(classes) == 11
