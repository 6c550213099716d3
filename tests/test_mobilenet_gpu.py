"""The reference's `backbone='mobilenetv3'` (timm mobilenetv3_small_100 features_only, train_vit.py:68-74,87): the
HBM-bound kernels of csrc/mobile.hip against torch on the same bf16 operands, the new GEMM activations, and the whole
classifier against the fp32 CPU oracle (oracle/ref_model.py::MobileNetV3SmallFeatures; parity at the timm boundary is
unpinned, as for the resnet stems).  Inference only: train() forwards are refused."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _act(t, name):
    return {"relu": torch.relu, "hswish": F.hardswish, "hsigmoid": F.hardsigmoid, None: lambda u: u}[name](t)


@pytest.mark.parametrize("B,H,C,k,stride,act", [(3, 56, 16, 3, 2, "relu"), (2, 28, 96, 5, 2, "hswish"), (2, 14, 240, 5, 1, "hswish"),
                                               (5, 7, 576, 5, 1, "hswish"), (1, 9, 8, 3, 1, None), (2, 15, 24, 5, 2, "relu")])
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


def test_mobilenetv3_checkpoint_round_trip_and_training_is_refused(dev, tmp_path):
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
    m.train()
    try:
        with pytest.raises(NotImplementedError):
            m(x)
    finally:
        m.eval()
