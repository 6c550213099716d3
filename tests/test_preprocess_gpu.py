"""The device input pipeline (vit_amd/pipeline.py, csrc/preproc.hip) against the oracle (oracle/ref_preprocess.py,
itself pinned bit-for-bit against Pillow in tests/test_preprocess.py): eval and train transforms of
train_vit.py:722-756 on full-size 1475x804 frames and odd sizes.  Integer part bit-exact; the float tail
(x/255 - mean)/std in fp32 must be bit-identical too (same two IEEE operations)."""
import numpy as np
import pytest
import torch

from oracle import ref_preprocess as rp

pytestmark = pytest.mark.gpu


def _frames(B, H, W, C, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, (B, H, W, C), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    f[0, ..., 0] = (xx * 255 // max(W - 1, 1)).astype(np.uint8)       # one structured frame
    if B > 1:
        f[1] = 0
        f[1, ::2] = 255                                                # saturating stripes
    return f


@pytest.mark.parametrize("H,W,C,B", [(804, 1475, 3, 3), (333, 517, 3, 2), (100, 180, 3, 2), (224, 224, 3, 1),
                                     (804, 1475, 1, 2), (240, 320, 4, 2)])
def test_eval_transform_bit_exact(dev, H, W, C, B):
    from vit_amd.pipeline import DevicePreprocessor
    f = _frames(B, H, W, C, H + W)
    pre = DevicePreprocessor((H, W), train=False)
    x = torch.from_numpy(f if C > 1 else f[..., 0].copy()).to(dev)
    out = pre(x).cpu().numpy()
    for b in range(B):
        want = rp.eval_transform(f[b] if C > 1 else f[b, ..., 0])
        assert out[b].shape == want.shape
        assert np.array_equal(out[b], want), (b, np.abs(out[b] - want).max())


def test_train_transform_crop_and_noise(dev):
    from vit_amd.pipeline import DevicePreprocessor
    H, W, B = 804, 1475, 4
    f = _frames(B, H, W, 3, 5)
    x = torch.from_numpy(f).to(dev)
    clean = DevicePreprocessor((H, W), train=True, noise_factor=0.0)
    crops = torch.tensor([[0, 0], [22, 22], [7, 19], [13, 0]], dtype=torch.int32)
    out = clean(x, crop_tl=crops).cpu().numpy()
    for b in range(B):
        want = rp.train_transform(f[b], int(crops[b, 0]), int(crops[b, 1]))
        assert np.array_equal(out[b], want), b
    # random crops come from the given generator: reproducible, within range, not all equal
    g = torch.Generator().manual_seed(11)
    a1 = clean(x, generator=g)
    g = torch.Generator().manual_seed(11)
    a2 = clean(x, generator=g)
    assert torch.equal(a1, a2)
    with pytest.raises(ValueError):
        clean(x, crop_tl=torch.tensor([[0, 23]] * B, dtype=torch.int32))
    # noise: N(0, 0.05^2) added after normalisation, independent per channel, reproducible from the generator
    noisy = DevicePreprocessor((H, W), train=True, noise_factor=0.05)
    g = torch.Generator().manual_seed(3)
    n1 = noisy(x, generator=g, crop_tl=crops)
    g = torch.Generator().manual_seed(3)
    n2 = noisy(x, generator=g, crop_tl=crops)
    n3 = noisy(x, generator=torch.Generator().manual_seed(4), crop_tl=crops)
    assert torch.equal(n1, n2) and not torch.equal(n1, n3)
    d = (n1.cpu() - torch.from_numpy(out)).double()
    assert abs(d.mean().item()) < 5e-4 and abs(d.std().item() - 0.05) < 5e-4
    z = d / 0.05
    assert abs((z ** 3).mean().item()) < 0.02 and abs((z ** 4).mean().item() - 3.0) < 0.05     # skewness, kurtosis
    assert abs(z.abs().max().item()) < 6.5
    c01 = (z[:, 0] * z[:, 1]).mean().item()
    cshift = (z[:, :, :, 1:] * z[:, :, :, :-1]).mean().item()
    assert abs(c01) < 0.01 and abs(cshift) < 0.01


def test_pipeline_feeds_the_model(dev):
    """frames -> DevicePreprocessor -> CNNViTHybrid equals host preprocessing (vit_amd.infer.preprocess) -> model."""
    from PIL import Image
    from vit_amd.infer import preprocess
    from vit_amd.model import CNNViTHybrid
    from vit_amd.pipeline import DevicePreprocessor
    from vit_amd.synth import synth_state_dict
    f = _frames(3, 804, 1475, 3, 9)
    m = CNNViTHybrid(num_classes=11, embed_dim=128, num_heads=2, num_layers=1, dropout=0.0, backbone="resnet18")
    m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 2))
    m = m.to(dev).eval()
    with torch.no_grad():
        a = m(DevicePreprocessor((804, 1475))(torch.from_numpy(f).to(dev)))
        b = m(torch.stack([preprocess(Image.fromarray(f[i])) for i in range(3)]).to(dev))
    assert torch.equal(a, b)


def test_shape_errors(dev):
    from vit_amd.pipeline import DevicePreprocessor
    pre = DevicePreprocessor((100, 180))
    with pytest.raises(ValueError):
        pre(torch.zeros(1, 101, 180, 3, dtype=torch.uint8, device=dev))
    with pytest.raises(ValueError):
        pre(torch.zeros(1, 100, 180, 3, dtype=torch.float32, device=dev))
