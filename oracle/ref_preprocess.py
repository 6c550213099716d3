"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's image transforms.

    eval : Grayscale(3) -> Resize((224,224)) -> ToTensor -> Normalize           train_vit.py:751-756, infer.py:346-353
    train: Grayscale(3) -> Resize((246,246)) -> RandomCrop(224) -> ToTensor
           -> Normalize -> AddGaussianNoise(0.05)                               train_vit.py:722-749, 557-567

The arithmetic lives in the reference's dependencies: torchvision (absent here; its Grayscale / Resize on a PIL
image are `img.convert("L")` and `img.resize(size[::-1], BILINEAR)`, ToTensor is `uint8 / 255`, Normalize is
`(x - mean) / std` in fp32) and Pillow (present: 12.x).  This file restates Pillow's published algorithm in
scalar numpy/Python, following Pillow's C sources by name:
  - `rgb2l`   (libImaging/Convert.c, macro L24):  L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16
  - `precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`, `ImagingResampleVertical_8bpc`
    (libImaging/Resample.c): separable triangle filter whose support grows with the down-scaling factor, 22-bit
    fixed-point coefficients, a uint8 intermediate image after the horizontal pass.
PINNED: tests/test_preprocess.py checks every function here bit-for-bit against Pillow itself (the library the
reference calls) on random and structured images of several sizes, so parity of the HIP pipeline with this oracle is
parity with the reference's transform.  The Gaussian noise is random in the reference (torch.randn_like): only its
distribution can be matched.
"""
from __future__ import annotations

import math

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
PRECISION_BITS = 32 - 8 - 2


def rgb2l(rgb: np.ndarray) -> np.ndarray:
    """uint8 [H,W,3|4] -> uint8 [H,W] (Convert.c rgb2l; alpha ignored as in rgba -> l)."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def _triangle(x: float) -> float:   # bilinear_filter, Resample.c
    if x < 0.0:
        x = -x
    return 1.0 - x if x < 1.0 else 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """-> (ksize, bounds [out,2] int (first tap, tap count), coefficients [out, ksize] int32 with 22 fractional bits)."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_triangle((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(v: int) -> int:
    v >>= PRECISION_BITS
    return 0 if v < 0 else 255 if v > 255 else v


def resample_1d(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One pass of ImagingResample over uint8 [H,W]: axis 1 = horizontal, 0 = vertical."""
    a = img if axis == 1 else img.T
    _, bounds, kk = precompute_coeffs(a.shape[1], out_size)
    out = np.empty((a.shape[0], out_size), dtype=np.uint8)
    a64 = a.astype(np.int64)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (a64[:, x0:x0 + n] * kk[xx, :n].astype(np.int64)).sum(axis=1) + (1 << (PRECISION_BITS - 1))
        out[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out if axis == 1 else out.T


def resize_bilinear(luma: np.ndarray, out_hw) -> np.ndarray:
    """Image.resize((w, h), BILINEAR) on mode L: horizontal pass, uint8 intermediate, vertical pass; a pass whose
    size does not change is skipped (Resample.c ImagingResampleInner need_horizontal / need_vertical)."""
    oh, ow = out_hw
    t = luma
    if ow != t.shape[1]:
        t = resample_1d(t, ow, 1)
    if oh != t.shape[0]:
        t = resample_1d(t, oh, 0)
    return np.ascontiguousarray(t)


def to_tensor_normalize(luma224: np.ndarray) -> np.ndarray:
    """ToTensor + Normalize on a 3x replicated L image -> float32 [3,H,W]."""
    a = luma224.astype(np.float32) / np.float32(255.0)
    return ((a[None] - MEAN[:, None, None]) / STD[:, None, None]).astype(np.float32)


def eval_transform(img: np.ndarray, image_size: int = 224) -> np.ndarray:
    luma = rgb2l(img) if img.ndim == 3 else img
    return to_tensor_normalize(resize_bilinear(luma, (image_size, image_size)))


def train_transform(img: np.ndarray, crop_top: int, crop_left: int, image_size: int = 224) -> np.ndarray:
    """Without the noise (random in the reference); crop offsets as RandomCrop.get_params returns them (i, j)."""
    luma = rgb2l(img) if img.ndim == 3 else img
    s = int(image_size * 1.1)
    r = resize_bilinear(luma, (s, s))
    return to_tensor_normalize(r[crop_top:crop_top + image_size, crop_left:crop_left + image_size])
