"""Fixture helper: the synthetic checkpoints whose classifier head was fitted on the CPU oracle's features so that
predictions cover all 11 classes (scripts/make_golden.py `fitted`; tests/golden/fitted_head_*.npz hold the head, the
rest regenerates from the seed; fitted_vectors_*.npz hold the oracle's logits)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FITTED = {
    "R": (dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="resnet18"), 0),
    "P": (dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="patch16"), 5),
    # BASELINE configs[3] / configs[4] at their own size (scripts/make_golden.py fitted_big: only head.3 is fitted)
    "B": (dict(num_classes=11, embed_dim=768, num_heads=12, num_layers=12, dropout=0.0, backbone="patch16"), 8),
    "L": (dict(num_classes=11, embed_dim=1024, num_heads=16, num_layers=24, dropout=0.0, backbone="patch16"), 9),
}


def fitted_state_dict(name, shapes):
    from vit_amd.synth import synth_state_dict
    cfg, seed = FITTED[name]
    sd = synth_state_dict(shapes, seed)
    head = np.load(os.path.join(GOLD, f"fitted_head_{name}.npz"))
    for k in head.files:
        assert tuple(sd[k].shape) == head[k].shape, k
        sd[k] = torch.from_numpy(head[k].copy())
    return sd


def fitted_vectors(name):
    return np.load(os.path.join(GOLD, f"fitted_vectors_{name}.npz"))


def fitted_inputs():
    """(11 preprocessed golden spectrograms, the 33 fitted synthetic frames, 16 held-out synthetic frames)."""
    from PIL import Image
    from vit_amd.infer import preprocess
    from vit_amd.synth import CLASSES, micro_doppler_batch
    ximg = torch.stack([preprocess(Image.open(os.path.join(GOLD, "images", c + ".png"))) for c in CLASSES])
    return ximg, micro_doppler_batch(33, seed=4242), micro_doppler_batch(16, seed=777)
