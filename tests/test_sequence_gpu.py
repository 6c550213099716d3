"""Sequence model on librvit (vit_amd/sequence.py, csrc/lstm.hip) against the oracle (oracle/ref_sequence.py =
torch.nn.LSTM + the reference's module graph, fp32 CPU): temporal encoder forward (fp32 kernels: tight), classifier
and full model (bf16 GEMM operands: bf16 tolerances), gradients, dropout by mask replay, streaming wrapper."""
import pytest
import torch

from oracle import ref_sequence as rs

pytestmark = pytest.mark.gpu

CFG = dict(backbone="patch16", embed_dim=128, num_heads=2, num_layers=1, dropout=0.0, temporal_hidden=64,
           temporal_layers=2, temporal_dropout=0.3, temporal_bidirectional=True, classifier_dropout=0.4,
           sequence_length=4, mode="sequence")


def _pair(dev, cfg=CFG, seed=0):
    from vit_amd.sequence import CNNViTSequence
    torch.manual_seed(seed)
    ref = rs.CNNViTSequenceRef(11, cfg)
    with torch.no_grad():      # livelier than the default init: biases and head weights away from zero
        for n, p in ref.named_parameters():
            if n.startswith(("temporal_encoder", "classifier")):
                p.copy_(torch.randn_like(p) * (0.15 if p.dim() > 1 else 0.1))
    mine = CNNViTSequence(11, cfg)
    mine.load_state_dict(ref.state_dict())
    return mine.to(dev), ref


@pytest.mark.parametrize("bidir,layers,B,T", [(True, 2, 3, 5), (False, 1, 2, 1), (True, 1, 9, 3), (False, 2, 17, 4)])
def test_temporal_encoder_forward_fp32(dev, bidir, layers, B, T):
    from vit_amd.sequence import TemporalEncoder
    torch.manual_seed(1)
    ref = rs.TemporalEncoderRef(32, 24, layers, 0.3, bidir).eval()
    mine = TemporalEncoder(32, 24, layers, 0.3, bidir)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev).eval()
    x = torch.randn(B, T, 32)
    with torch.no_grad():
        want = ref(x)
        got = mine(x.to(dev)).cpu()
    assert got.shape == want.shape
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-4), (got - want).abs().max().item()


@pytest.mark.parametrize("bidir,layers,B,T", [(True, 2, 3, 5), (False, 1, 2, 1), (False, 2, 17, 4)])
def test_gru_encoder_forward_fp32(dev, bidir, layers, B, T):
    from vit_amd.sequence import TemporalEncoder
    torch.manual_seed(2)
    ref = rs.TemporalEncoderRef(32, 24, layers, 0.3, bidir, "gru").eval()
    mine = TemporalEncoder(32, 24, layers, 0.3, bidir, "gru")
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev).eval()
    x = torch.randn(B, T, 32)
    with torch.no_grad():
        want, got = ref(x), mine(x.to(dev)).cpu()
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-4), (got - want).abs().max().item()


@pytest.mark.parametrize("model_type", ["lstm", "gru"])
def test_temporal_head_eval_and_gradients(dev, model_type):
    cfg = dict(CFG, temporal_model=model_type)
    mine, ref = _pair(dev, cfg)
    assert mine.temporal_encoder.model_type == model_type and type(ref.temporal_encoder.rnn).__name__.lower() == model_type
    B, T, E = 5, 4, CFG["embed_dim"]
    f = torch.randn(B, T, E)
    y = torch.tensor([0, 3, 5, 10, 2])
    # eval forward
    mine.eval(); ref.eval()
    with torch.no_grad():
        want = ref.forward_temporal_only(f)
        got = mine.forward_temporal_only(f.to(dev)).cpu()
    assert torch.allclose(got, want, atol=2e-2, rtol=2e-2), (got - want).abs().max().item()
    # gradients with every dropout probability at 0 (exact same function on both sides)
    cfg0 = dict(cfg, temporal_dropout=0.0, classifier_dropout=0.0)
    mine, ref = _pair(dev, cfg0)
    mine.train(); ref.train()
    fr = f.clone().requires_grad_(True)
    torch.nn.functional.cross_entropy(ref.forward_temporal_only(fr), y).backward()
    fm = f.to(dev).requires_grad_(True)
    torch.nn.functional.cross_entropy(mine.forward_temporal_only(fm), y.to(dev)).backward()
    def rel(a, b):
        return ((a.cpu() - b).norm() / (b.norm() + 1e-12)).item()
    assert rel(fm.grad, fr.grad) < 2e-2
    refp = dict(ref.named_parameters())
    for n, p in mine.named_parameters():
        if n.startswith("spatial_encoder"):
            assert p.grad is None
            continue
        assert p.grad is not None and p.grad.shape == p.shape, n
        assert rel(p.grad, refp[n].grad) < 3e-2, (n, rel(p.grad, refp[n].grad))


def test_temporal_dropout_mask_replay(dev):
    """Train-mode forward with the kernels' own dropout masks replayed in the oracle: same logits, and the masks
    have the right keep rates (LSTM inter-layer p=0.3, classifier p=0.4; train_vit.py:205,275)."""
    from vit_amd import ops
    from vit_amd.sequence import _site_seed
    mine, ref = _pair(dev)
    mine.train(); ref.train()
    B, T, E, H = 16, 4, CFG["embed_dim"], CFG["temporal_hidden"]
    f = torch.randn(B, T, E)
    torch.manual_seed(77)
    got = mine.forward_temporal_only(f.to(dev).requires_grad_(True))
    torch.manual_seed(77)
    base = int(torch.randint(0, 2 ** 62, (1,)).item())
    m_lstm = ops.dropout_mask(B * T * 2 * H, 0.3, _site_seed(base, 100), dev).view(B, T, 2 * H).cpu()
    m_cls = ops.dropout_mask(B * H, 0.4, _site_seed(base, 1), dev).view(B, H).cpu()
    assert abs(m_lstm.float().mean().item() - 0.7) < 0.03 and abs(m_cls.float().mean().item() - 0.6) < 0.06
    with torch.no_grad():
        want = rs.temporal_forward_with_masks(ref, f, {0: m_lstm}, 0.3, m_cls, 0.4)
    assert torch.allclose(got.detach().cpu(), want, atol=3e-2, rtol=3e-2), (got.detach().cpu() - want).abs().max().item()
    # and backward through the dropped path runs and differs from the eval-mode gradient structure
    got.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in mine.named_parameters()
               if not n.startswith("spatial_encoder"))


@pytest.mark.parametrize("backbone", ["patch16", "resnet18"])
def test_full_sequence_model_forward(dev, backbone):
    cfg = dict(CFG, backbone=backbone)
    mine, ref = _pair(dev, cfg, seed=3)
    mine.eval(); ref.eval()
    from vit_amd.synth import micro_doppler_batch
    B, T = 2, 3
    x = micro_doppler_batch(B * T, seed=5).view(B, T, 3, 224, 224)
    with torch.no_grad():
        want = ref(x)
        got = mine(x.to(dev)).cpu()
        fs = mine.forward_spatial_only(x[0].to(dev))
        again = mine.forward_temporal_only(fs.unsqueeze(0)).cpu()
    assert torch.allclose(got, want, atol=0.06, rtol=0.05), (got - want).abs().max().item()
    assert torch.allclose(again[0], got[0], atol=1e-3)          # streaming split == fused forward


def test_full_sequence_model_trains(dev):
    from vit_amd.synth import micro_doppler_batch
    mine, _ = _pair(dev, dict(CFG, backbone="resnet18", dropout=0.1), seed=4)
    mine.train()
    mine.freeze_backbone()
    assert all(not p.requires_grad for p in mine.spatial_encoder.cnn_backbone.parameters())
    mine.unfreeze_backbone()
    B, T = 4, 3
    x = micro_doppler_batch(B * T, seed=6).view(B, T, 3, 224, 224).to(dev)
    y = torch.tensor([1, 4, 7, 9], device=dev)
    opt = torch.optim.AdamW(mine.parameters(), lr=3e-4)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(mine(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mine.parameters())
    assert losses[-1] < losses[0], losses


def test_streaming_classifier(dev):
    from vit_amd.sequence import StreamingClassifier
    mine, _ = _pair(dev, seed=5)
    classes = [f"c{i}" for i in range(11)]
    sc = StreamingClassifier(model=mine, classes=classes, config=CFG, smoothing_window=2)
    assert sc.mode == "sequence" and sc.sequence_length == 4
    frames = [torch.randn(3, 224, 224) for _ in range(6)]
    res = [sc.update(fr) for fr in frames]
    assert res[:3] == [None, None, None] and all(r is not None for r in res[3:])
    assert res[3].frame_count == 4 and res[5].frame_count == 6
    with torch.no_grad():
        direct = torch.softmax(mine(torch.stack(frames[0:4]).unsqueeze(0).to(dev)), dim=1)[0].cpu()
    top = max(range(11), key=lambda i: direct[i].item())
    assert res[3].prediction == classes[top] and abs(res[3].confidence - direct[top].item()) < 2e-3
    assert len(res[4].all_probs) == 11 and abs(sum(p for _, p in res[4].all_probs) - 1.0) < 1e-4
    sc.reset()
    assert sc.update(frames[0]) is None and sc.frame_count == 1
    import numpy as np
    assert sc.update((np.random.default_rng(0).random((300, 400)) * 255).astype(np.uint8)) is None   # luma array input
    with pytest.raises(TypeError):
        sc.update(3.14)
    b = sc.benchmark(n_frames=6)
    assert b["mode"] == "sequence" and b["fps"] > 0


def test_streaming_classifier_device_preprocess(dev):
    """device_preprocess=True: uint8 frames are resized / normalised on the GPU; the tensor the model sees equals the
    host transform's bit for bit, so the predictions are the same."""
    import numpy as np
    from vit_amd.infer import preprocess
    from vit_amd.sequence import StreamingClassifier
    mine, _ = _pair(dev, seed=5)
    classes = [f"c{i}" for i in range(11)]
    host = StreamingClassifier(model=mine, classes=classes, config=CFG, smoothing_window=1)
    gpu = StreamingClassifier(model=mine, classes=classes, config=CFG, smoothing_window=1, device_preprocess=True)
    rng = np.random.default_rng(3)
    frames = [(rng.random((201, 368, 3)) * 255).astype(np.uint8) for _ in range(5)]
    from PIL import Image
    want = preprocess(Image.fromarray(frames[0]))
    got = gpu._preprocess(frames[0]).cpu()
    assert torch.equal(want, got)
    assert torch.equal(preprocess(frames[1][:, :, 0]), gpu._preprocess(np.ascontiguousarray(frames[1][:, :, 0])).cpu())  # luma
    for fr in frames:
        a, b = host.update(fr), gpu.update(fr)
        assert (a is None) == (b is None)
        if a is not None:
            assert a.prediction == b.prediction and abs(a.confidence - b.confidence) < 1e-6


def test_sequence_checkpoint_round_trip(dev, tmp_path):
    """A checkpoint in the reference layout (train_vit.py:996-1019) with mode 'sequence' loads through
    vit_amd.infer.load_model and through StreamingClassifier(model_path) with the restricted unpickler."""
    from vit_amd.infer import load_model
    from vit_amd.sequence import CNNViTSequence, StreamingClassifier
    mine, _ = _pair(dev, seed=6)
    classes = [f"c{i}" for i in range(11)]
    path = tmp_path / "cnn_vit_lstm_radar.pt"
    torch.save({"model": {k: v.cpu() for k, v in mine.state_dict().items()}, "classes": classes,
                "model_config": dict(CFG, num_classes=11), "metadata": {"confidence_threshold": 0.4}}, path)
    m2, cls2, meta = load_model(str(path), dev)
    assert isinstance(m2, CNNViTSequence) and cls2 == classes and meta["mode"] == "sequence"
    x = torch.randn(1, 4, 3, 224, 224, device=dev)
    with torch.no_grad():
        assert torch.equal(m2(x), mine.eval()(x))
    sc = StreamingClassifier(str(path))
    assert sc.sequence_length == 4 and sc.confidence_threshold == 0.4
