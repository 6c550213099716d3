"""Input pipeline on device: the reference's `train_tf` / `eval_tf` torchvision pipelines
(train_vit.py:722-756, infer.py:346-353) from decoded uint8 frames to the normalised tensor, in two HIP launches.

    eval : Grayscale(3) -> Resize((224,224)) -> ToTensor -> Normalize
    train: Grayscale(3) -> Resize((246,246)) -> RandomCrop(224) -> ToTensor -> Normalize -> AddGaussianNoise(0.05)

PNG decoding stays on the host (PIL / DataLoader workers, as in the reference); the frames are uploaded as uint8
(3.6 MB per 1475x804 RGB frame instead of nothing-smaller-than 602 KB of fp32 AFTER a host-side resize that costs
~10 ms of CPU per frame).  The integer part (luma, antialiased bilinear resize) is bit-exact with Pillow; the
random crop draws its offsets like torchvision's RandomCrop.get_params (torch.randint on the CPU generator); the
Gaussian noise is a counter-based generator seeded from torch's CPU RNG, not torch.randn's stream.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops

PRECISION_BITS = 32 - 8 - 2


def resample_tables(in_size: int, out_size: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the triangle ("bilinear") filter, vectorised:
    -> (ksize, bounds int32 [out,2] = (first tap, tap count), coefficients int32 [out,ksize], 22 fractional bits).
    Same double-precision operations in the same order as libImaging/Resample.c (the sums run left to right)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast truncates toward zero;
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)    # negative values clamp to 0 anyway
    n = xmax - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    arg = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(arg < 1.0, 1.0 - arg, 0.0)
    w[np.arange(ksize)[None, :] >= n[:, None]] = 0.0
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):            # left-to-right accumulation, as the C loop does
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)
    kk[np.arange(ksize)[None, :] >= n[:, None]] = 0
    bounds = np.stack([xmin, n], axis=1).astype(np.int32)
    return ksize, bounds, kk.astype(np.int32)


class DevicePreprocessor:
    """`eval_tf` (train=False) or `train_tf` (train=True) of train_vit.py:722-756 for frames of one size.

    __call__(frames): uint8 CUDA tensor [B,H,W,3] (RGB), [B,H,W,4] (RGBA) or [B,H,W] (luma) -> float32
    [B,3,image_size,image_size], ready for CNNViTHybrid.forward."""

    def __init__(self, in_hw: Tuple[int, int], train: bool = False, image_size: int = 224, random_crop: bool = True,
                 noise_factor: float = 0.05, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("DevicePreprocessor runs on the GPU through librvit only (host path: vit_amd.infer.preprocess)")
        self.H, self.W = int(in_hw[0]), int(in_hw[1])
        self.train, self.image_size = bool(train), int(image_size)
        self.crop = self.train and random_crop
        self.resized = int(image_size * 1.1) if self.crop else self.image_size      # train_vit.py:728
        self.noise = float(noise_factor) if self.train else 0.0                     # train_vit.py:746-747
        self.device = dev
        self._th = self._upload(resample_tables(self.W, self.resized))
        self._tv = self._upload(resample_tables(self.H, self.resized))
        self._tmp: Optional[torch.Tensor] = None

    def _upload(self, t):
        ks, bounds, coef = t
        return ks, torch.from_numpy(bounds).to(self.device), torch.from_numpy(coef).to(self.device)

    def __call__(self, frames: torch.Tensor, generator: Optional[torch.Generator] = None,
                 crop_tl: Optional[torch.Tensor] = None) -> torch.Tensor:
        if frames.dim() not in (3, 4) or tuple(frames.shape[1:3]) != (self.H, self.W):
            raise ValueError(f"expected uint8 frames [B,{self.H},{self.W}(,C)], got {tuple(frames.shape)}")
        B, S = frames.shape[0], self.image_size
        need = B * self.H * S
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, 3, S, S, dtype=torch.float32, device=self.device)
        seed = 0
        if self.crop:
            if crop_tl is None:      # RandomCrop.get_params: i ~ U{0..h-th}, j ~ U{0..w-tw}, per image
                crop_tl = torch.randint(0, self.resized - S + 1, (B, 2), generator=generator, dtype=torch.int32)
            host = crop_tl.detach().cpu()        # validated on the host: the kernel indexes its tables with them
            if tuple(host.shape) != (B, 2) or int(host.min()) < 0 or int(host.max()) > self.resized - S:
                raise ValueError("crop_tl must be [B,2] offsets (top, left) within the resized image")
            crop_tl = host.to(device=self.device, dtype=torch.int32).contiguous()
        else:
            crop_tl = None
        if self.noise > 0:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())
        with torch.cuda.device(self.device):
            return ops.preprocess_u8(frames, self._th, self._tv, self.resized, out, self._tmp, crop_tl, self.noise, seed)
