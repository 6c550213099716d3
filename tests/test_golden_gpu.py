"""GPU: the HIP path against the committed golden vectors and on the 11 real spectrograms
(BASELINE.json configs[0] plumbing: image file -> label), through the reference-shaped caller API
(vit_amd/infer.py).  Tolerance: |logit - golden| <= 0.15 + 0.05*|golden| (bf16 compute vs the fp32
oracle, the bound of tests/test_model_gpu.py); argmax must agree wherever the golden top-1 margin
exceeds twice that bound."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ATOL, RTOL = 0.15, 0.05


def _gold():
    with open(os.path.join(GOLD, "manifest.json")) as fh:
        man = json.load(fh)
    return man, np.load(os.path.join(GOLD, "oracle_vectors.npz"))


def _model(cfg, seed, dev):
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import synth_state_dict
    m = CNNViTHybrid(**cfg)
    m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed))
    return m.to(dev).eval()


@pytest.mark.parametrize("name", ["tiny", "R", "legacy"])
def test_hip_logits_match_golden(dev, name):
    from vit_amd.synth import micro_doppler_batch
    man, vec = _gold()
    cfg, B, seed = man[f"{name}/config"], man[f"{name}/batch"], man[f"{name}/seed"]
    m = _model(cfg, seed, dev)
    with torch.no_grad():
        logits = m(micro_doppler_batch(B, seed=100 + seed).to(dev)).cpu().numpy()
    ref = vec[f"{name}/logits_eval"]
    err = np.abs(logits - ref)
    print(f"\n[{name}] max |err| {err.max():.4f}")
    assert (err <= ATOL + RTOL * np.abs(ref)).all()


def test_real_images_to_labels(dev, tmp_path):
    from PIL import Image
    from vit_amd.infer import load_model, predict_batch, predict_folder, predict_single
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import CLASSES, synth_checkpoint
    man, vec = _gold()
    cfg = man["R/config"]
    # a synthetic checkpoint in the reference layout, written and re-read through load_model
    ck = synth_checkpoint(CNNViTHybrid(**cfg), seed=man["R/seed"])
    path = tmp_path / "cnn_vit_radar.pt"
    torch.save(ck, path)
    model, classes, meta = load_model(str(path), dev)
    assert classes == CLASSES and meta["mode"] == "single" and not model.training
    imgs = [Image.open(os.path.join(GOLD, "images", c + ".png")) for c in CLASSES]
    res = predict_batch(model, imgs, classes, dev, top_k=3)
    ref = vec["R/image_logits"]
    got = np.concatenate([r.raw_logits for r in res])
    bound = ATOL + RTOL * np.abs(ref)
    assert (np.abs(got - ref) <= bound).all(), np.abs(got - ref).max()
    srt = np.sort(ref, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    for i, r in enumerate(res):
        if margin[i] > 2 * bound[i].max():
            assert r.prediction == CLASSES[int(ref[i].argmax())]
        assert 0.0 < r.confidence <= 1.0 and len(r.top_k) == 3
    # file-based entry points (infer.py:356-430) agree with the batched call
    one = predict_single(model, os.path.join(GOLD, "images", "Away.png"), classes, dev)
    assert one is not None and np.allclose(one.raw_logits, res[0].raw_logits, atol=0.05)
    assert predict_single(model, str(tmp_path / "nope.png"), classes, dev) is None
    folder = predict_folder(model, os.path.join(GOLD, "images"), classes, dev)
    assert sorted(folder) == sorted(c + ".png" for c in CLASSES)
    for c, r in zip(CLASSES, res):
        assert folder[c + ".png"].prediction == r.prediction
    # the same calls with the resize / normalise done on the GPU: the model sees the same tensor, bit for bit
    # (the fixtures are 224x224 already; an up-sized copy exercises the resampling)
    res_dev = predict_batch(model, imgs, classes, dev, top_k=3, device_preprocess=True)
    for a, b in zip(res, res_dev):
        assert np.array_equal(a.raw_logits, b.raw_logits)
    big = [im.convert("RGB").resize((368, 201), Image.BILINEAR) for im in imgs[:4]]
    r_host = predict_batch(model, big, classes, dev, top_k=1)
    r_dev = predict_batch(model, big, classes, dev, top_k=1, device_preprocess=True)
    for a, b in zip(r_host, r_dev):
        assert np.array_equal(a.raw_logits, b.raw_logits)
    mixed = [big[0], imgs[1]]   # different sizes: falls back to the host transform
    assert len(predict_batch(model, mixed, classes, dev, device_preprocess=True)) == 2
    folder_dev = predict_folder(model, os.path.join(GOLD, "images"), classes, dev, device_preprocess=True)
    assert {k: v.prediction for k, v in folder_dev.items()} == {k: v.prediction for k, v in folder.items()}


# Fixed logit bounds for the fitted fixtures: 1.5 x the largest |HIP - fp32 oracle| measured on the round-2 tree
# (printed by the test: "measured / bound").  bf16 storage through 20 conv layers + 4 encoder layers (R) / the 16x16
# patch conv + 4 encoder layers (P); logits of magnitude ~5.
FITTED_BOUND = {"R": 0.31, "P": 0.46}   # measured 0.203 / 0.307


@pytest.mark.parametrize("name", ["R", "P"])
def test_fitted_checkpoint_labels_agree_exactly(dev, tmp_path, name):
    """BASELINE configs[0] / configs[1] as written: a checkpoint file in the reference layout (`cnn_vit_radar.pt`:
    hybrid; `vit_radar.pt`: the pure-ViT patch16 layout) -> load_model -> the 11 spectrogram files -> labels
    (infer.py:268-406).  The checkpoint's head is fitted so that the oracle predicts class i for image i and class
    j % 11 for synthetic frame j with margins > 4: the HIP path must reproduce EVERY label (an input-independent path
    would get 1 of 11), and the logits within the fixed bound."""
    from PIL import Image
    from tests.fitted import FITTED, fitted_inputs, fitted_state_dict, fitted_vectors
    from vit_amd.infer import load_model, predict_batch, predict_folder
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import CLASSES, synth_checkpoint
    cfg, seed = FITTED[name]
    proto = CNNViTHybrid(**cfg)
    ck = synth_checkpoint(proto, seed=seed)
    ck["model"] = fitted_state_dict(name, {k: v.shape for k, v in proto.state_dict().items()})
    assert ck["model_config"]["backbone"] == cfg["backbone"]
    path = tmp_path / ("cnn_vit_radar.pt" if name == "R" else "vit_radar.pt")
    torch.save(ck, path)
    model, classes, meta = load_model(str(path), dev)
    assert classes == CLASSES and not model.training
    assert (model.cnn_backbone is None) == (name == "P")
    vec = fitted_vectors(name)
    bound = FITTED_BOUND[name]
    # (1) files -> labels through the reference-shaped callers
    imgs = [Image.open(os.path.join(GOLD, "images", c + ".png")) for c in CLASSES]
    res = predict_batch(model, imgs, classes, dev, top_k=3)
    assert [r.prediction for r in res] == CLASSES
    folder = predict_folder(model, os.path.join(GOLD, "images"), classes, dev)
    assert all(folder[c + ".png"].prediction == c for c in CLASSES)
    got_img = np.concatenate([r.raw_logits for r in res])
    # (2) the synthetic frames, fitted and held out
    ximg, xfit, xheld = fitted_inputs()
    with torch.no_grad():
        got_fit = model(xfit.to(dev)).cpu().numpy()
        got_held = model(xheld.to(dev)).cpu().numpy()
    assert got_fit.argmax(1).tolist() == [j % 11 for j in range(33)]
    worst = 0.0
    for got, key in ((got_img, "image_logits"), (got_fit, "fit_logits"), (got_held, "heldout_logits")):
        ref = vec[key]
        err = np.abs(got - ref).max()
        worst = max(worst, err)
        srt = np.sort(ref, axis=1)
        decided = (srt[:, -1] - srt[:, -2]) > 2 * bound
        assert (got.argmax(1) == ref.argmax(1))[decided].all(), key
    print(f"\n[fitted {name}] max |logit - oracle| measured {worst:.4f} / bound {bound}")
    assert worst <= bound
    assert len(set(got_held.argmax(1).tolist())) >= 6 and len(set(got_img.argmax(1).tolist())) == 11
