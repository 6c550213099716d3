"""Pins oracle/ref_preprocess.py against Pillow itself (the library behind the reference's torchvision transforms,
train_vit.py:722-756) and checks the product's host-side coefficient tables against the oracle's.  CPU only."""
import numpy as np
import pytest

from oracle import ref_preprocess as rp


def _images():
    rng = np.random.default_rng(7)
    yield "noise", rng.integers(0, 256, (804, 1475, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:804, 0:1475]
    grad = np.stack([(xx * 255 // 1474), (yy * 255 // 803), ((xx + yy) % 256)], -1).astype(np.uint8)
    yield "gradient", grad
    yield "small-upscale", rng.integers(0, 256, (100, 180, 3), dtype=np.uint8)
    yield "odd", rng.integers(0, 256, (333, 517, 3), dtype=np.uint8)
    yield "exact", rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    sat = np.zeros((600, 900, 3), dtype=np.uint8); sat[::2] = 255
    yield "stripes", sat


@pytest.mark.parametrize("name,img", list(_images()), ids=[n for n, _ in _images()])
def test_oracle_equals_pillow(name, img):
    from PIL import Image
    pil = Image.fromarray(img)
    luma = np.asarray(pil.convert("L"))
    assert np.array_equal(rp.rgb2l(img), luma)
    for side in (224, 246):
        want = np.asarray(pil.convert("L").resize((side, side), Image.BILINEAR))
        got = rp.resize_bilinear(luma, (side, side))
        assert np.array_equal(got, want), (name, side, np.abs(got.astype(int) - want.astype(int)).max())


def test_oracle_tensor_path_matches_host_preprocess():
    """The product's host path (vit_amd.infer.preprocess: Pillow + numpy) and the oracle agree bit-for-bit."""
    from vit_amd.infer import preprocess
    from PIL import Image
    img = next(iter(_images()))[1]
    want = preprocess(Image.fromarray(img)).numpy()
    assert np.array_equal(rp.eval_transform(img), want)


def test_rgba_and_luma_inputs():
    from PIL import Image
    rng = np.random.default_rng(3)
    rgba = rng.integers(0, 256, (50, 70, 4), dtype=np.uint8)
    assert np.array_equal(rp.rgb2l(rgba), np.asarray(Image.fromarray(rgba, "RGBA").convert("L")))
    luma = rng.integers(0, 256, (300, 400), dtype=np.uint8)
    assert np.array_equal(rp.eval_transform(luma), rp.to_tensor_normalize(
        np.asarray(Image.fromarray(luma).resize((224, 224), Image.BILINEAR))))


@pytest.mark.parametrize("n_in,n_out", [(1475, 224), (804, 224), (1475, 246), (804, 246), (100, 224), (224, 224), (225, 224)])
def test_product_tables_equal_oracle(n_in, n_out):
    from vit_amd.pipeline import resample_tables
    ks, bounds, kk = rp.precompute_coeffs(n_in, n_out)
    pk, pb, pc = resample_tables(n_in, n_out)
    assert pk == ks and np.array_equal(pb, bounds) and np.array_equal(pc, kk)
    assert (kk.sum(axis=1) - (1 << rp.PRECISION_BITS)).__abs__().max() <= ks   # rows sum to 1.0 up to rounding


def test_pipeline_refuses_cpu():
    import torch
    from vit_amd.pipeline import DevicePreprocessor
    with pytest.raises(RuntimeError):
        DevicePreprocessor((804, 1475), device="cpu")
