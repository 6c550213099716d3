"""The one entry point that allocates nothing, `rvit_preprocess_u8_static` (include/rvit.h): the uint8 intermediate of
the device input pipeline lives in a static device array inside librvit instead of a caller-owned buffer.  Called
through the C ABI (ctypes, vit_amd.ops.preprocess_u8_static) and checked BIT FOR BIT against the existing entry
point `rvit_preprocess_u8` (which tests/test_preprocess_gpu.py checks against the oracle) on two calls in a row, the
second with a smaller input: the array keeps the first call's bytes, and nothing of them may leak into the second
result.  Sizes beyond the array are refused with a clear error.  This file is collected last on purpose."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _frames(B, H, W, C, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.integers(0, 256, (B, H, W, C), dtype=np.uint8))


def _both(dev, frames, train, crops=None, seed=0):
    from vit_amd import ops
    from vit_amd.pipeline import DevicePreprocessor
    B, H, W = frames.shape[:3]
    pre = DevicePreprocessor((H, W), train=train, noise_factor=0.05 if train else 0.0, device=dev)
    S = pre.image_size
    x = frames.to(dev)
    crop = None if crops is None else crops.to(device=dev, dtype=torch.int32).contiguous()
    tmp = torch.empty(B * H * S, dtype=torch.uint8, device=dev)
    want = torch.empty(B, 3, S, S, dtype=torch.float32, device=dev)
    ops.preprocess_u8(x, pre._th, pre._tv, pre.resized, want, tmp, crop, pre.noise, seed)
    got = torch.full((B, 3, S, S), float("nan"), dtype=torch.float32, device=dev)
    before = torch.cuda.memory_allocated(dev)
    ops.preprocess_u8_static(x, pre._th, pre._tv, pre.resized, got, crop, pre.noise, seed)
    assert torch.cuda.memory_allocated(dev) == before      # torch's allocator was not touched by the call
    torch.cuda.synchronize(dev)
    return got, want


def test_static_scratch_matches_existing_entry_twice(dev):
    # call 1: full-size 1475x804 RGB frames, eval transform (804 x 224 bytes of intermediate per frame)
    got, want = _both(dev, _frames(6, 804, 1475, 3, 1), train=False)
    assert torch.equal(got, want)
    # call 2, smaller input, train transform with crops and noise: must not see call 1's leftovers
    crops = torch.tensor([[0, 0], [22, 22], [7, 19]], dtype=torch.int32)
    got, want = _both(dev, _frames(3, 333, 517, 3, 2), train=True, crops=crops, seed=1234)
    assert torch.equal(got, want)
    # call 3: larger again (RGBA), the array is reused from its start
    got, want = _both(dev, _frames(2, 240, 320, 4, 3), train=False)
    assert torch.equal(got, want)


def test_static_scratch_rejects_oversize(dev):
    from vit_amd import ops
    from vit_amd._lib import RvitError
    from vit_amd.pipeline import DevicePreprocessor
    H, W, S = 804, 1475, 224
    B = ops.PREPROCESS_STATIC_MAX_BYTES // (H * S) + 1       # one frame too many for the 32 MiB array
    pre = DevicePreprocessor((H, W), train=False, device=dev)
    x = torch.zeros(B, H, W, 3, dtype=torch.uint8, device=dev)
    out = torch.empty(B, 3, S, S, dtype=torch.float32, device=dev)
    with pytest.raises(RvitError, match="exceed the static scratch"):
        ops.preprocess_u8_static(x, pre._th, pre._tv, pre.resized, out)
    # one frame fewer fits
    ops.preprocess_u8_static(x[:B - 1], pre._th, pre._tv, pre.resized, out[:B - 1])
    torch.cuda.synchronize(dev)
