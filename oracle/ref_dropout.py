"""ORACLE — test infrastructure only.  Never imported by vit_amd/.

Training-mode forward of the reference model with EXPLICIT dropout masks, so that a test can replay
the masks the HIP kernels drew (vit_amd.ops.dropout_mask) and compare losses and gradients with
autograd.  torch's own dropout cannot be steered from outside (Philox stream), so
the six dropout sites are restated here on top of the oracle module's parameters:

  pos_drop                         train_vit.py:100,160
  attention-probability dropout    torch functional.py:6569-6570 (scaled_dot_product_attention dropout_p)
  dropout1 (after out_proj)        torch transformer.py:977   (_sa_block)
  dropout  (after GELU in the FFN) torch transformer.py:982   (_ff_block)
  dropout2 (after linear2)         torch transformer.py:982
  head dropout                     train_vit.py:125

Everything else calls the same torch.nn modules / functionals as oracle/ref_model.py.  Pinned by
tests/test_oracle.py: with all-ones masks (p = 0) it must reproduce CNNViTHybridRef.forward.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from .ref_model import CNNViTHybridRef


def keep_mask_numpy(n: int, p: float, seed: int):
    """The counter-based generator of vit_amd/csrc/common.h (DropKey / rvit_drop_hash / drop_mult) restated in NumPy:
    uint8 [n], 1 where element i of the site (p, seed) is kept.  The key (k0, k1) = halves of splitmix64(seed); one 32-bit hash per element PAIR (i >> 1) - two rounds
    of {24-bit multiply of the low 24 bits + 24-bit multiply of bits 8..31, xor-shift} - whose low 16 bits decide the
    even element and high 16 bits the odd one; kept iff those bits >= round(p * 2^16).  Integer arithmetic: the kernels'
    masks (rvit_dropout_mask) must equal this bit for bit (tests/test_dropout_gpu.py).  The attention site indexes its
    elements (pair * S + query) * S2 + key with S2 = S rounded up to even."""
    import numpy as np
    m32 = np.uint64(0xFFFFFFFF)
    m64 = (1 << 64) - 1                      # key words: the halves of splitmix64(seed) (make_drop_key, on the host)
    z = (seed + 0x9E3779B97F4A7C15) & m64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m64
    z ^= z >> 31
    k0, k1 = np.uint64(z & 0xFFFFFFFF), np.uint64(z >> 32)
    thresh = int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)) if p > 0 else 0
    if thresh == 0:
        return np.ones(n, dtype=np.uint8)

    def mul24(a, c):
        return ((a & np.uint64(0xFFFFFF)) * np.uint64(c)) & m32

    def mix24(x, c_lo, c_hi):
        return (mul24(x, c_lo) + mul24(x >> np.uint64(8), c_hi)) & m32

    pair = np.arange((n + 1) // 2, dtype=np.uint64)
    h = mix24((pair + k0) & m32, 0xB5297B, 0x68E31D)
    h ^= h >> np.uint64(15)
    h = mix24(h ^ k1, 0x9E3779, 0x7FEB35)
    h ^= h >> np.uint64(16)
    bits = np.stack([h & np.uint64(0xFFFF), h >> np.uint64(16)], axis=1).reshape(-1)[:n]
    return (bits >= np.uint64(thresh)).astype(np.uint8)


def mask_shapes(ref: CNNViTHybridRef, B: int) -> Dict[object, tuple]:
    """Site -> mask shape, in the element order the kernels index (row-major)."""
    E = ref.pos_embed.shape[-1]
    S = ref.num_patches + 1
    H = ref.transformer.layers[0].self_attn.num_heads
    shapes: Dict[object, tuple] = {"pos": (B, S, E), "head": (B, E // 2)}
    for i in range(len(ref.transformer.layers)):
        shapes[(i, "attn")] = (B, H, S, S)
        shapes[(i, "proj")] = (B, S, E)
        shapes[(i, "ffn")] = (B, S, 4 * E)
        shapes[(i, "fc2")] = (B, S, E)
    return shapes


def forward_with_masks(ref: CNNViTHybridRef, x: torch.Tensor, masks: Dict[object, torch.Tensor], p: float,
                       features_only: bool = False) -> torch.Tensor:
    """Logits (or features) of `ref` with dropout sites multiplied by masks[site] / (1 - p)."""
    k = 1.0 / (1.0 - p)
    B = x.shape[0]
    t = x
    if ref.cnn_backbone is not None:
        t = ref.cnn_backbone(t)[-1]
    t = ref.patch_embed(t).flatten(2).transpose(1, 2)
    t = torch.cat([ref.cls_token.expand(B, -1, -1), t], dim=1) + ref.pos_embed
    t = t * masks["pos"] * k
    for i, L in enumerate(ref.transformer.layers):
        mha = L.self_attn
        H = mha.num_heads
        S, E = t.shape[1], t.shape[2]
        Dh = E // H
        h = L.norm1(t)
        qkv = F.linear(h, mha.in_proj_weight, mha.in_proj_bias)
        q, kk, v = (u.view(B, S, H, Dh).transpose(1, 2) for u in qkv.chunk(3, dim=-1))
        a = torch.softmax(q @ kk.transpose(-1, -2) / math.sqrt(Dh), dim=-1)
        a = a * masks[(i, "attn")] * k
        o = (a @ v).transpose(1, 2).reshape(B, S, E)
        t = t + mha.out_proj(o) * masks[(i, "proj")] * k
        h = L.norm2(t)
        f = F.gelu(L.linear1(h)) * masks[(i, "ffn")] * k
        t = t + L.linear2(f) * masks[(i, "fc2")] * k
    c = ref.norm(t[:, 0])
    if features_only:
        return c
    hd = F.gelu(ref.head[0](c)) * masks["head"] * k
    return ref.head[3](hd)
