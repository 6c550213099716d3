"""The oracle on the paths bench.py actually times (VERDICT r2 "next" item 1).

Every other oracle-compared model test runs B <= 33, where the dispatcher takes the per-op encoder kernels and the
small tiles.  The driver-timed runs are B=512 inference (fused `enc_attn` / `enc_mlp` above 12,800 tokens,
vit_amd/engine.py FUSED_MIN_TOKENS; the 256x256 tile on the stride-2 convs once the batch fills the chip), B=256
training, ViT-B/16-scale and ViT-L/16-scale at B=256 (the eight-wave GEMMs need >= 512 output tiles).  Here:

  (a) B=512 inference of the R hybrid (reference call site infer.py:393-396): the 11 golden spectrograms and the 33
      fitted frames scattered through a 512-frame batch: every label equals the oracle's, the logits are within the
      fixed bf16 bound of tests/test_golden_gpu.py, and equal the same frames run at B=44 to 1e-5;
  (b) one B=256 training step of the R hybrid against the fp32 oracle's autograd (train_vit.py:935-939);
  (c) ViT-B/16-scale and ViT-L/16-scale (E1024 / H16 / all 24 layers) fitted-head fixtures, in bf16 AND MX-fp8 mode, at
      B=256: labels equal the fp32 oracle's on all 44 fitted inputs, logits within FIXED bounds (1.5 x measured), and
      equal to the same frames at B=60 (the small-tile kernels) to the stated tolerance.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

R_BOUND = 0.31            # tests/test_golden_gpu.py FITTED_BOUND["R"]
# (c): fixed bounds = 1.5 x the largest |HIP - fp32 oracle| logit difference measured on the round-3 tree (the test
# prints measured / bound).  MX-fp8: the CPU emulation of the format itself sits 1.39 (B) / 1.28 (L) from the fp32
# oracle on these inputs (tests/golden/fitted_vectors_*.npz mx_emulation_logits); logits of magnitude ~6.
BIG_BOUND = {("B", "bf16"): 0.36, ("B", "mxfp8"): 2.15, ("L", "bf16"): 0.51, ("L", "mxfp8"): 2.5}   # measured 0.237 / 1.43 / 0.337 / 1.667


def _fitted_model(name, dev):
    from tests.fitted import FITTED, fitted_state_dict
    from vit_amd.model import CNNViTHybrid
    cfg, _ = FITTED[name]
    m = CNNViTHybrid(**cfg)
    m.load_state_dict(fitted_state_dict(name, {k: v.shape for k, v in m.state_dict().items()}))
    return m.to(dev).eval()


def test_b512_inference_labels_and_logits_on_the_benched_path(dev):
    from tests.fitted import fitted_inputs, fitted_vectors
    from vit_amd.synth import micro_doppler_batch
    m = _fitted_model("R", dev)
    assert m.engine.fused_encoder and 512 * 50 >= m.engine.FUSED_MIN_TOKENS > 44 * 50   # the two batches take different paths
    ximg, xfit, _ = fitted_inputs()
    known = torch.cat([ximg, xfit])                                     # 44 frames with oracle logits on file
    vec = fitted_vectors("R")
    want = np.concatenate([vec["image_logits"], vec["fit_logits"]])
    x = micro_doppler_batch(512, seed=2024)
    pos = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).permutation(512)[:44].copy())
    x[pos] = known
    with torch.no_grad():
        big = m(x.to(dev)).cpu()
        small = m(known.to(dev)).cpu()
    got = big[pos]
    err = float(np.abs(got.numpy() - want).max())
    print(f"\n[R B=512] max |logit - oracle| {err:.4f} / bound {R_BOUND}; max |B=512 - B=44| {(got - small).abs().max():.2e}")
    assert got.argmax(1).tolist() == want.argmax(1).tolist() == list(range(11)) + [j % 11 for j in range(33)]
    assert err <= R_BOUND
    assert torch.allclose(got, small, atol=1e-5, rtol=0)
    # the filler frames are frames too: nothing non-finite, and not one constant prediction
    assert torch.isfinite(big).all() and len(set(big.argmax(1).tolist())) >= 6


def test_b256_training_step_matches_autograd(dev):
    """One training step at the batch bench.py's `train` leg times (B=256, dropout 0): loss within 2 %, encoder / head
    tensors within 6 % relative L2 of the fp32 oracle, stem tensors against the bf16-storage oracle
    (tests/test_train_gpu.py for the tolerances and why)."""
    from tests.test_train_gpu import check_training_step
    cfg = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="resnet18")
    check_training_step(dev, "R-hybrid-B256", cfg, 256, 0.06, 0.90)


@pytest.mark.parametrize("name", ["B", "L"])
def test_large_width_fitted_labels_bf16_and_mxfp8(dev, name):
    from tests.fitted import fitted_inputs, fitted_vectors
    m = _fitted_model(name, dev)
    ximg, xfit, xheld = fitted_inputs()
    xs = torch.cat([ximg, xfit, xheld])                                   # 60 frames with oracle logits on file
    vec = fitted_vectors(name)
    want = np.concatenate([vec["image_logits"], vec["fit_logits"], vec["heldout_logits"]])
    idx = (torch.arange(256) * 7 + 3) % 60                               # every frame 4-5 times, scattered
    x256 = xs[idx].to(dev)
    first = [int((idx == j).nonzero()[0]) for j in range(60)]
    for mode in ("bf16", "mxfp8"):
        m.set_inference_precision(mode)
        with torch.no_grad():
            big = m(x256).cpu()
            small = m(xs.to(dev)).cpu()
        got = big[first]
        err = float(np.abs(got.numpy() - want).max())
        bound = BIG_BOUND[(name, mode)]
        dup = max(float((big[idx == j] - got[j]).abs().max()) for j in range(60))
        bs = float((got - small).abs().max())
        print(f"\n[{name} {mode} B=256] max |logit - fp32 oracle| measured {err:.4f} / bound {bound}; copies of a frame in "
              f"one batch differ by {dup:.2e}; B=256 vs B=60 {bs:.2e}")
        assert got[:44].argmax(1).tolist() == want[:44].argmax(1).tolist() == list(range(11)) + [j % 11 for j in range(33)]
        assert err <= bound
        srt = np.sort(want[44:], axis=1)
        decided = (srt[:, -1] - srt[:, -2]) > 2 * bound
        assert (got[44:].argmax(1).numpy() == want[44:].argmax(1))[decided].all()
        assert dup <= 1e-5                  # a frame's logits do not depend on its position in the batch
        assert bs <= 1e-5                   # nor on the batch size (the large- and small-tile GEMMs give the same bits)
    m.set_inference_precision("bf16")
