"""CPU-only checks that pin what can be pinned about the oracle (oracle/ref_model.py):
  * its tail is the reference's own torch.nn modules/arguments (train_vit.py:93-127): state_dict
    keys and shapes equal the list the reference source implies;
  * torch's eval fast path and the train-mode slow path of the encoder agree;
  * vit_amd.CNNViTHybrid exposes exactly the same keys/shapes (drop-in checkpoint layout);
  * the bf16 emulation stays within the documented distance of the fp32 oracle.
The stem (timm resnet18) is PARITY UNPINNED: no reference-held vector covers it (see the oracle
header)."""
import torch

from oracle.ref_model import CNNViTHybridRef

R = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone="resnet18")


def _expected_tail(E, L, S, C, ncls):
    exp = {"cls_token": (1, 1, E), "pos_embed": (1, S, E), "patch_embed.weight": (E, C, 1, 1), "patch_embed.bias": (E,)}
    for i in range(L):
        p = f"transformer.layers.{i}."
        exp.update({p + "self_attn.in_proj_weight": (3 * E, E), p + "self_attn.in_proj_bias": (3 * E,),
                    p + "self_attn.out_proj.weight": (E, E), p + "self_attn.out_proj.bias": (E,),
                    p + "linear1.weight": (4 * E, E), p + "linear1.bias": (4 * E,),
                    p + "linear2.weight": (E, 4 * E), p + "linear2.bias": (E,),
                    p + "norm1.weight": (E,), p + "norm1.bias": (E,), p + "norm2.weight": (E,), p + "norm2.bias": (E,)})
    exp.update({"norm.weight": (E,), "norm.bias": (E,), "head.0.weight": (E // 2, E), "head.0.bias": (E // 2,),
                "head.3.weight": (ncls, E // 2), "head.3.bias": (ncls,)})
    return exp


def test_tail_state_dict_layout_matches_reference_source():
    ref = CNNViTHybridRef(**R)
    sd = {k: tuple(v.shape) for k, v in ref.state_dict().items() if not k.startswith("cnn_backbone.")}
    assert sd == _expected_tail(256, 4, 50, 512, 11)
    assert len(sd) == 22 + 12 * 3  # 22 + 12*(L-1) tensors


def test_stem_key_names_follow_timm_feature_list_net():
    ref = CNNViTHybridRef(**R)
    keys = [k for k in ref.state_dict() if k.startswith("cnn_backbone.")]
    assert "cnn_backbone.conv1.weight" in keys and "cnn_backbone.bn1.running_var" in keys
    assert "cnn_backbone.layer2.0.downsample.0.weight" in keys and "cnn_backbone.layer4.1.bn2.num_batches_tracked" in keys
    assert not any(".fc." in k for k in keys)
    n_params = sum(v.numel() for k, v in ref.state_dict().items()
                   if k.startswith("cnn_backbone.") and "running" not in k and "num_batches" not in k)
    assert n_params == 11_176_512  # resnet18 minus fc


def test_product_module_has_the_same_checkpoint_layout():
    from vit_amd.model import CNNViTHybrid
    for cfg in (R, dict(R, backbone="resnet34"), dict(R, backbone="patch16", embed_dim=768, num_heads=12, num_layers=2)):
        ref, m = CNNViTHybridRef(**cfg), CNNViTHybrid(**cfg)
        a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert a == b and list(a) == list(b)
    assert sum(p.numel() for p in CNNViTHybrid(**R).parameters()) == 14_514_763


def test_encoder_fast_and_slow_paths_agree():
    from vit_amd.synth import synth_state_dict
    cfg = dict(R, backbone="patch16")
    ref = CNNViTHybridRef(**cfg)
    ref.load_state_dict(synth_state_dict({k: v.shape for k, v in ref.state_dict().items()}, 3))
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    ref.eval()
    with torch.no_grad():
        fast = ref(x)
    slow = ref(x)  # grad enabled -> torch takes the per-op slow path
    assert (fast - slow.detach()).abs().max() < 5e-5


def test_unknown_backbone_raises_value_error():
    import pytest
    from vit_amd.model import CNNViTHybrid
    with pytest.raises(ValueError):
        CNNViTHybrid(11, backbone="vgg")
    with pytest.raises(ValueError):
        CNNViTHybridRef(11, backbone="vgg")


def test_bf16_emulation_distance_to_fp32_oracle():
    from oracle.ref_bf16 import forward_bf16_emulated
    from vit_amd.synth import micro_doppler_batch, synth_state_dict
    ref = CNNViTHybridRef(**R).eval()
    sd = synth_state_dict({k: v.shape for k, v in ref.state_dict().items()}, 1)
    ref.load_state_dict(sd)
    x = micro_doppler_batch(4, seed=77)
    with torch.no_grad():
        want = ref(x)
    emu = forward_bf16_emulated(sd, x, R)
    assert (emu - want).abs().max() < 0.35
    assert (emu.argmax(1) == want.argmax(1)).all()


def test_mask_replay_restatement_equals_module_forward():
    """oracle/ref_dropout.py with all-ones masks must reproduce CNNViTHybridRef.forward (the torch modules
    the reference instantiates), for the hybrid and the patch16 layouts, features and logits."""
    from oracle.ref_dropout import forward_with_masks, mask_shapes
    from oracle.ref_model import CNNViTHybridRef
    from vit_amd.synth import micro_doppler_batch, synth_state_dict
    for cfg in (dict(num_classes=11, embed_dim=128, num_heads=2, num_layers=2, dropout=0.0, backbone="patch16"),
                dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=2, dropout=0.0, backbone="resnet18")):
        ref = CNNViTHybridRef(**cfg)
        ref.load_state_dict(synth_state_dict({k: v.shape for k, v in ref.state_dict().items()}, 3))
        ref.eval()
        x = micro_doppler_batch(2, seed=8)
        ones = {k: torch.ones(sh) for k, sh in mask_shapes(ref, 2).items()}
        with torch.no_grad():
            assert torch.allclose(forward_with_masks(ref, x, ones, 0.0), ref(x), rtol=1e-4, atol=1e-5)
            assert torch.allclose(forward_with_masks(ref, x, ones, 0.0, features_only=True), ref.forward_features(x),
                                  rtol=1e-4, atol=1e-5)


def test_mxfp8_emulation_restatement_and_format():
    """oracle/ref_mxfp8.py: without quantisation it reproduces the module forward; the MXFP8 round trip obeys
    the format (exact on powers of two and on e4m3-representable blocks, error bounded by the block maximum)."""
    from oracle.ref_mxfp8 import forward_mxfp8_emulated, mx_dequantize, mx_fake_quant, mx_quantize
    from oracle.ref_model import CNNViTHybridRef
    from vit_amd.synth import micro_doppler_batch, synth_state_dict
    cfg = dict(num_classes=11, embed_dim=128, num_heads=2, num_layers=2, dropout=0.0, backbone="patch16")
    ref = CNNViTHybridRef(**cfg)
    ref.load_state_dict(synth_state_dict({k: v.shape for k, v in ref.state_dict().items()}, 3))
    ref.eval()
    x = micro_doppler_batch(2, seed=8)
    with torch.no_grad():
        assert torch.allclose(forward_mxfp8_emulated(ref, x, quantize=False), ref(x), rtol=1e-4, atol=1e-5)
        d = (forward_mxfp8_emulated(ref, x) - ref(x)).abs().max().item()
    assert 0.0 < d < 1.0                                   # quantisation changes the logits, moderately
    # format checks
    blk = torch.tensor([[1.0, -2.0, 0.5, 448.0] + [0.0] * 28])          # representable with scale 2^0
    q, s = mx_quantize(blk)
    assert int(s[0, 0]) == 127 and torch.equal(mx_dequantize(q, s), blk)
    big = blk * 2.0 ** 20
    q, s = mx_quantize(big)
    assert int(s[0, 0]) == 147 and torch.equal(mx_dequantize(q, s), big)
    z = torch.zeros(1, 32)
    q, s = mx_quantize(z)
    assert int(s[0, 0]) == 0 and torch.equal(mx_dequantize(q, s), z)
    r = torch.randn(64, 256) * torch.exp(torch.randn(64, 1) * 4)
    err = (mx_fake_quant(r) - r).reshape(64, 8, 32).abs().amax(2)
    assert (err <= r.reshape(64, 8, 32).abs().amax(2) * 2.0 ** -3).all()


def test_init_weights_semantics():
    """`_init_weights` (train_vit.py:132-139) and the torch defaults the reference keeps everywhere else: trunc-normal
    std 0.02 on pos_embed / cls_token / head Linear weights, zero head biases; MultiheadAttention in_proj
    xavier-uniform with zero biases, out_proj bias zero; and - because nn.TransformerEncoder deep-copies ONE layer
    (train_vit.py:103-116) - all encoder layers IDENTICAL at init.  Checked on the HIP model's parameter holder and
    against the oracle (the same torch.nn constructors as the reference) built under the same seed statistics."""
    import math
    import torch
    from oracle.ref_model import CNNViTHybridRef
    from vit_amd.model import CNNViTHybrid
    cfg = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone="resnet18")
    torch.manual_seed(0)
    m = CNNViTHybrid(**cfg)
    torch.manual_seed(0)
    ref = CNNViTHybridRef(**cfg)
    sd, sr = m.state_dict(), ref.state_dict()
    for k in ("pos_embed", "cls_token", "head.0.weight", "head.3.weight"):
        for t in (sd[k], sr[k]):
            assert abs(t.mean().item()) < 0.01 and t.abs().max().item() <= 2.0
        # std 0.02 within sampling error (cls_token has only 256 values)
        assert abs(sd[k].std().item() - 0.02) < (0.004 if sd[k].numel() < 1000 else 0.0015), k
        assert abs(sr[k].std().item() - 0.02) < (0.004 if sr[k].numel() < 1000 else 0.0015), k
    for k in ("head.0.bias", "head.3.bias"):
        assert torch.count_nonzero(sd[k]) == 0 and torch.count_nonzero(sr[k]) == 0
    E = cfg["embed_dim"]
    for s in (sd, sr):
        # identical layers at init
        for i in range(1, cfg["num_layers"]):
            for leaf in ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                         "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                         "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"):
                assert torch.equal(s[f"transformer.layers.0.{leaf}"], s[f"transformer.layers.{i}.{leaf}"]), (i, leaf)
        w = s["transformer.layers.0.self_attn.in_proj_weight"]
        bound = math.sqrt(6.0 / (E + 3 * E))                       # xavier_uniform_ on [3E, E]
        assert w.abs().max().item() <= bound + 1e-6 and abs(w.std().item() - bound / math.sqrt(3)) < 0.002
        assert torch.count_nonzero(s["transformer.layers.0.self_attn.in_proj_bias"]) == 0
        assert torch.count_nonzero(s["transformer.layers.0.self_attn.out_proj.bias"]) == 0
        # nn.Linear default (kaiming_uniform a=sqrt(5)): U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias
        for leaf, fan_in in (("linear1", E), ("linear2", 4 * E), ("self_attn.out_proj", E)):
            w = s[f"transformer.layers.0.{leaf}.weight"]
            assert w.abs().max().item() <= 1.0 / math.sqrt(fan_in) + 1e-6
            assert abs(w.std().item() - 1.0 / math.sqrt(3 * fan_in)) < 0.1 / math.sqrt(3 * fan_in)
        assert torch.equal(s["norm.weight"], torch.ones(E)) and torch.count_nonzero(s["norm.bias"]) == 0
        pe = s["patch_embed.weight"]                               # nn.Conv2d default init, train_vit.py:93
        assert pe.abs().max().item() <= 1.0 / math.sqrt(512) + 1e-6


def test_mobilenetv3_small_restatement_matches_the_published_size():
    """oracle MobileNetV3SmallFeatures = timm mobilenetv3_small_100 features_only (train_vit.py:69,87) restated without
    timm: the trunk's parameter count plus timm's conv_head (576 -> 1024, bias) and classifier (1024 -> 1000) must give
    the published 2.54 M of mobilenetv3_small_100; feature maps at strides 2..32 have 16, 16, 24, 48, 576 channels (576
    at 7x7 is what train_vit.py:69 declares); product and oracle state_dicts have the same keys and shapes."""
    from oracle.ref_model import CNNViTHybridRef, MobileNetV3SmallFeatures
    from vit_amd.model import CNNViTHybrid
    f = MobileNetV3SmallFeatures().eval()
    trunk = sum(p.numel() for p in f.parameters())
    assert trunk == 927_008
    assert trunk + (576 * 1024 + 1024) + (1024 * 1000 + 1000) == 2_542_856
    with torch.no_grad():
        shapes = [tuple(t.shape[1:]) for t in f(torch.zeros(1, 3, 224, 224))]
    assert shapes == [(16, 112, 112), (16, 56, 56), (24, 28, 28), (48, 14, 14), (576, 7, 7)]
    cfg = dict(num_classes=11, embed_dim=128, num_heads=2, num_layers=1, backbone="mobilenetv3")
    a = {k: tuple(v.shape) for k, v in CNNViTHybridRef(**cfg).state_dict().items()}
    b = {k: tuple(v.shape) for k, v in CNNViTHybrid(**cfg).state_dict().items()}
    assert a == b
    keys = set(a)
    assert {"cnn_backbone.conv_stem.weight", "cnn_backbone.blocks.0.0.conv_dw.weight", "cnn_backbone.blocks.0.0.se.conv_reduce.bias",
            "cnn_backbone.blocks.1.0.conv_pwl.weight", "cnn_backbone.blocks.4.2.bn3.running_var",
            "cnn_backbone.blocks.5.0.conv.weight"} <= keys
    assert "cnn_backbone.blocks.1.0.se.conv_reduce.weight" not in keys      # stage 1 has no squeeze-excite


def test_dropout_generator_restatement():
    """oracle/ref_dropout.keep_mask_numpy (the kernels' counter-based dropout generator restated in NumPy; the GPU
    suite holds rvit_dropout_mask to it bit for bit): keep rate, determinism, one decision pair per 32-bit hash, and the
    first decisions of one site as a regression vector."""
    import numpy as np
    from oracle.ref_dropout import keep_mask_numpy
    n, p, seed = 1 << 20, 0.1, 0x1234_5678_9ABC_DEF0
    m = keep_mask_numpy(n, p, seed)
    assert m.dtype == np.uint8 and m.shape == (n,)
    assert abs(m.mean() - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5
    assert np.array_equal(m, keep_mask_numpy(n, p, seed))
    assert [int(v) for v in m[:16]] == [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1]
    assert np.array_equal(keep_mask_numpy(n - 1, p, seed), m[:-1])          # a prefix does not depend on n
    assert keep_mask_numpy(1000, 0.0, seed).all()
    for delta in (1, 1 << 32, 1 << 40):      # related seeds (consecutive, one key word equal) still give independent sites
        both = (m & keep_mask_numpy(n, p, seed ^ delta)).mean()
        assert abs(both - (1 - p) ** 2) < 6 * (p * (1 - p) / n) ** 0.5, delta
