"""Deterministic synthetic checkpoints and synthetic micro-Doppler-shaped inputs.

Neither author checkpoint (`cnn_vit_radar.pt`, `vit_radar.pt`) ships with the reference
(.MISSING_LARGE_BLOBS) and there is no network, so benchmarks and parity tests use weights drawn
here from numpy's PCG64 (independent of torch's RNG, identical on every box) in the documented
checkpoint layout (train_vit.py:996-1038).  Scales are chosen so activations stay O(1) through
the stem and encoder and the logits have real top-1 margins (SURVEY.md §7.3 "argmax parity").
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

CLASSES = ["Away", "Bend", "Crawl", "Kneel", "Limp", "Pick", "SStep", "Scissor", "Sit", "Toes", "Towards"]


def synth_state_dict(shapes: Dict[str, torch.Size], seed: int = 0, logit_gain: float = 6.0) -> Dict[str, torch.Tensor]:
    """Values for every key of a CNNViTHybrid state_dict, drawn per key from PCG64(seed, key index)."""
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        rng = np.random.Generator(np.random.PCG64([seed, i]))
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.tensor(100, dtype=torch.long)
            continue
        if leaf == "running_mean":
            v = rng.normal(0.0, 0.2, shape)
        elif leaf == "running_var":
            v = rng.uniform(0.6, 1.4, shape)
        elif "bn" in name or "downsample.1" in name:                 # BatchNorm affine
            hi = 0.8 if ".bn2." in name else 1.3      # damp the residual branch: maps stay O(1)
            v = rng.uniform(hi - 0.5, hi, shape) if leaf == "weight" else rng.normal(0.0, 0.1, shape)
        elif "norm" in name:                                          # LayerNorm affine
            v = rng.uniform(0.8, 1.2, shape) if leaf == "weight" else rng.normal(0.0, 0.05, shape)
        elif name in ("cls_token", "pos_embed"):
            v = rng.normal(0.0, 0.3, shape)
        elif leaf in ("bias", "in_proj_bias"):
            v = rng.normal(0.0, 0.05, shape)
        else:                                                         # conv / linear weights
            fan_in = int(np.prod(shape[1:]))
            gain = math.sqrt(2.0) if len(shape) == 4 and "patch_embed" not in name else 1.0
            if name == "head.3.weight":
                gain = logit_gain
            v = rng.normal(0.0, gain / math.sqrt(fan_in), shape)
            if len(shape) == 4:
                # zero-DC filters: post-ReLU maps have a large positive mean; without this every
                # token is dominated by one common vector and the logits barely depend on the input
                v = v - v.reshape(shape[0], -1).mean(1).reshape(-1, 1, 1, 1)
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape).copy())
    return out


def synth_checkpoint(model, seed: int = 0) -> dict:
    """A checkpoint dict in the reference layout (train_vit.py:996-1019) with synthetic weights."""
    sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed)
    return {"model": sd, "classes": list(CLASSES[: model.num_classes]), "model_config": model.config(),
            "metadata": {"model_version": "2.0.0", "trained_at": "synthetic", "config_hash": f"synth{seed:04d}",
                         "pytorch_version": torch.__version__, "best_val_accuracy": 0.0, "epoch": 0,
                         "train_samples": 0, "val_samples": 0, "confidence_threshold": 0.5, "seed": seed}}


def micro_doppler_batch(batch: int, seed: int = 1234, size: int = 224) -> torch.Tensor:
    """Synthetic micro-Doppler-shaped spectrograms, ImageNet-normalised, fp32 [B,3,size,size].

    Each image is a sum of 3-6 sinusoidal Doppler tracks over time plus a Gaussian noise floor,
    rendered as grayscale in [0,1], replicated to 3 channels and normalised with the reference's
    mean/std (infer.py:346-353).  Generated with numpy PCG64 on the host.
    """
    rng = np.random.Generator(np.random.PCG64([seed, batch]))
    t = np.linspace(0.0, 1.0, size, dtype=np.float32)[None, None, :]          # time on the x axis
    f = np.linspace(-1.0, 1.0, size, dtype=np.float32)[None, :, None]         # Doppler on the y axis
    img = rng.normal(0.12, 0.04, (batch, size, size)).astype(np.float32)
    ntracks = rng.integers(3, 7, batch)
    for k in range(6):
        on = (ntracks > k).astype(np.float32)[:, None, None]
        amp = rng.uniform(0.2, 0.8, (batch, 1, 1)).astype(np.float32)
        freq = rng.uniform(1.0, 6.0, (batch, 1, 1)).astype(np.float32)
        phase = rng.uniform(0.0, 2 * np.pi, (batch, 1, 1)).astype(np.float32)
        off = rng.uniform(-0.3, 0.3, (batch, 1, 1)).astype(np.float32)
        width = rng.uniform(0.02, 0.06, (batch, 1, 1)).astype(np.float32)
        power = rng.uniform(0.4, 1.0, (batch, 1, 1)).astype(np.float32)
        centre = off + amp * np.sin(2 * np.pi * freq * t + phase)
        img += on * power * np.exp(-0.5 * ((f - centre) / width) ** 2)
    img = np.clip(img, 0.0, 1.0)
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[None, :, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[None, :, None, None]
    return torch.from_numpy((img[:, None] - mean) / std)
