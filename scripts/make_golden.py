"""Generates tests/golden/: run in the build container (reads /root/reference for the images only).

  images/<Class>.png     one real spectrogram per class: the first file (sorted) of
                         /root/reference/splits/test/<Class>/, converted to 8-bit luma and resized
                         to 224x224 bilinear (the reference's own preprocessing up to ToTensor,
                         infer.py:346-353) -- data, ~30 KB each
  manifest.json          source file name of every image, torch version, thread count
  oracle_vectors.npz     outputs of the CPU oracle (oracle/ref_model.py, the same torch.nn modules the
                         reference instantiates) on synthetic checkpoints from vit_amd.synth (numpy PCG64,
                         identical on every box): eval logits, train-mode loss, per-parameter gradient
                         L2 norms, and the logits of the 11 images

  fitted_head_{R,P}.npz  (python scripts/make_golden.py fitted)  head.0 / head.3 weights fitted on the CPU (AdamW on the
  fitted_vectors_*.npz   oracle's fp32 cls features, everything below the head stays the seed-0 synthetic checkpoint)
                         so that the ORACLE's predictions cover all 11 classes: image i -> class i on the 11 real
                         spectrograms, synthetic frame j (micro_doppler_batch(33, seed=4242)) -> class j % 11, with
                         top-1 margins > 4 against a bf16 logit error of ~0.2.  With the plain synthetic weights every
                         input predicts one class and an argmax assertion cannot fail.  The vectors hold the
                         oracle's logits on the 11 images, the 33 fitted frames and 16 held-out frames (seed 777).

The reference itself cannot be imported (timm / torchvision absent: ModuleNotFoundError), so these
pin the ORACLE across machines and torch builds; they are not outputs of the reference's code.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_model import CNNViTHybridRef  # noqa: E402
from vit_amd.infer import preprocess  # noqa: E402
from vit_amd.synth import CLASSES, micro_doppler_batch, synth_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SPLIT = "/root/reference/splits/test"

CONFIGS = {
    "tiny": (dict(num_classes=11, embed_dim=128, num_heads=2, num_layers=2, dropout=0.0, backbone="patch16"), 2, 11),
    "R": (dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="resnet18"), 2, 0),
    "legacy": (dict(num_classes=11, embed_dim=384, num_heads=6, num_layers=6, dropout=0.0, backbone="resnet18"), 1, 1),
}


def main():
    torch.set_num_threads(8)
    os.makedirs(os.path.join(GOLD, "images"), exist_ok=True)
    manifest = {"torch": torch.__version__, "threads": torch.get_num_threads(), "images": {}}
    imgs = []
    for cls in CLASSES:
        d = os.path.join(SPLIT, cls)
        f = sorted(os.listdir(d))[0]
        g = Image.open(os.path.join(d, f)).convert("L").resize((224, 224), Image.BILINEAR)
        g.save(os.path.join(GOLD, "images", cls + ".png"), optimize=True)
        manifest["images"][cls] = f
        imgs.append(preprocess(Image.open(os.path.join(GOLD, "images", cls + ".png"))))
    ximg = torch.stack(imgs)
    out = {}
    for name, (cfg, B, seed) in CONFIGS.items():
        ref = CNNViTHybridRef(**cfg)
        sd = synth_state_dict({k: v.shape for k, v in ref.state_dict().items()}, seed)
        ref.load_state_dict(sd)
        x = micro_doppler_batch(B, seed=100 + seed)
        y = torch.arange(B) % 11
        ref.eval()
        with torch.no_grad():
            out[f"{name}/logits_eval"] = ref(x).numpy()
            if name == "R":
                out["R/image_logits"] = ref(ximg).numpy()
        ref.train()
        loss = F.cross_entropy(ref(x), y, label_smoothing=0.05)
        loss.backward()
        out[f"{name}/loss_train"] = np.float32(loss.item())
        names = [n for n, _ in ref.named_parameters()]
        out[f"{name}/grad_norms"] = np.array([p.grad.norm().item() for _, p in ref.named_parameters()], np.float32)
        manifest[f"{name}/param_names"] = names
        manifest[f"{name}/config"] = cfg
        manifest[f"{name}/batch"] = B
        manifest[f"{name}/seed"] = seed
    np.savez_compressed(os.path.join(GOLD, "oracle_vectors.npz"), **out)
    with open(os.path.join(GOLD, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)
    print({k: getattr(v, "shape", v) for k, v in out.items()})
    print("image argmax:", [CLASSES[i] for i in out["R/image_logits"].argmax(1)])
    srt = np.sort(out["R/image_logits"], axis=1)
    print("image top-1 margins:", np.round(srt[:, -1] - srt[:, -2], 2))


# fitted fixtures: name -> (model config, synthetic seed of everything below the head)
FITTED = {
    "R": (dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="resnet18"), 0),
    # BASELINE configs[0]: the pure-ViT `vit_radar.pt` layout (build-defined, backbone='patch16')
    "P": (dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="patch16"), 5),
}
HEAD_KEYS = ("head.0.weight", "head.0.bias", "head.3.weight", "head.3.bias")
# large-width fixtures (python scripts/make_golden.py fitted_big): BASELINE configs[3] / configs[4] at their own size.
# Only head.3 (11 x E/2 weights + 11 biases, ~20 KB) is fitted and committed; head.0 stays synthetic.
FITTED_BIG = {
    "B": (dict(num_classes=11, embed_dim=768, num_heads=12, num_layers=12, dropout=0.0, backbone="patch16"), 8),
    "L": (dict(num_classes=11, embed_dim=1024, num_heads=16, num_layers=24, dropout=0.0, backbone="patch16"), 9),
}


def make_fitted(name):
    """Fits the classifier head of a synthetic checkpoint on the oracle's own features (CPU) and records the oracle's
    logits with it.  Only the head (34k floats) is committed; the other weights regenerate from the seed."""
    import torch.nn as nn
    torch.set_num_threads(8)
    cfg, seed = FITTED[name]
    ref = CNNViTHybridRef(**cfg)
    sd = synth_state_dict({k: v.shape for k, v in ref.state_dict().items()}, seed)
    ref.load_state_dict(sd)
    ref.eval()
    ximg = torch.stack([preprocess(Image.open(os.path.join(GOLD, "images", c + ".png"))) for c in CLASSES])
    xfit = micro_doppler_batch(33, seed=4242)
    xheld = micro_doppler_batch(16, seed=777)
    y = torch.cat([torch.arange(11), torch.arange(33) % 11])
    with torch.no_grad():
        f = ref.forward_features(torch.cat([ximg, xfit]))
    torch.manual_seed(0)
    E = cfg["embed_dim"]
    head = nn.Sequential(nn.Linear(E, E // 2), nn.GELU(), nn.Dropout(0.0), nn.Linear(E // 2, 11))   # train_vit.py:122-127
    for m in head:
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            nn.init.zeros_(m.bias)
    opt = torch.optim.AdamW(head.parameters(), lr=3e-3, weight_decay=0.05)
    for _ in range(400):
        opt.zero_grad()
        # feature noise of the size of the bf16 storage error: the margins must survive it
        F.cross_entropy(head(f + 0.03 * torch.randn_like(f)), y, label_smoothing=0.05).backward()
        opt.step()
    fitted = {"head.0.weight": head[0].weight, "head.0.bias": head[0].bias, "head.3.weight": head[3].weight,
              "head.3.bias": head[3].bias}
    fitted = {k: v.detach().clone() for k, v in fitted.items()}
    np.savez_compressed(os.path.join(GOLD, f"fitted_head_{name}.npz"), **{k: v.numpy() for k, v in fitted.items()})
    sd.update(fitted)
    ref.load_state_dict(sd)
    with torch.no_grad():
        out = {"image_logits": ref(ximg).numpy(), "fit_logits": ref(xfit).numpy(), "heldout_logits": ref(xheld).numpy()}
    np.savez_compressed(os.path.join(GOLD, f"fitted_vectors_{name}.npz"), **out)
    for k, v in out.items():
        srt = np.sort(v, axis=1)
        print(name, k, "argmax", v.argmax(1).tolist(), "min margin %.2f" % (srt[:, -1] - srt[:, -2]).min())
    assert out["image_logits"].argmax(1).tolist() == list(range(11))
    assert out["fit_logits"].argmax(1).tolist() == [j % 11 for j in range(33)]


def make_fitted_big(name):
    """ViT-B/16- and ViT-L/16-scale fixtures (E768/H12/L12, E1024/H16/L24): the last head Linear is fitted on the
    oracle's fp32 cls features AND on the features of the CPU MX-fp8 emulation (oracle/ref_mxfp8.py), both with added
    noise of the size of the fp8 format's feature error (rms 0.1 per element, measured: 0.09-0.10), so that the fp32
    oracle's labels (image i -> class i, frame j -> class j % 11) have margins that survive either precision.  The
    vectors hold the fp32 oracle's logits (the parity target) and the emulation's (the format's cost, for the record)."""
    from oracle.ref_mxfp8 import forward_mxfp8_emulated
    torch.set_num_threads(8)
    cfg, seed = FITTED_BIG[name]
    ref = CNNViTHybridRef(**cfg)
    sd = synth_state_dict({k: v.shape for k, v in ref.state_dict().items()}, seed)
    ref.load_state_dict(sd)
    ref.eval()
    ximg = torch.stack([preprocess(Image.open(os.path.join(GOLD, "images", c + ".png"))) for c in CLASSES])
    xfit = micro_doppler_batch(33, seed=4242)
    xheld = micro_doppler_batch(16, seed=777)
    xs = torch.cat([ximg, xfit, xheld])
    y = torch.cat([torch.arange(11), torch.arange(33) % 11])
    with torch.no_grad():
        f = ref.forward_features(xs)
        fe = forward_mxfp8_emulated(ref, xs, features_only=True)
    W0, b0 = sd["head.0.weight"], sd["head.0.bias"]
    hid = lambda u: F.gelu(F.linear(u, W0, b0))   # noqa: E731  (train_vit.py:122-124)
    torch.manual_seed(0)
    W3 = torch.zeros(11, cfg["embed_dim"] // 2, requires_grad=True)
    b3 = torch.zeros(11, requires_grad=True)
    opt = torch.optim.AdamW([W3, b3], lr=1e-2, weight_decay=0.05)
    for _ in range(800):
        opt.zero_grad()
        u = torch.cat([f[:44] + 0.1 * torch.randn(44, f.shape[1]), fe[:44] + 0.1 * torch.randn(44, f.shape[1])])
        F.cross_entropy(F.linear(hid(u), W3, b3), torch.cat([y, y]), label_smoothing=0.01).backward()
        opt.step()
    fitted = {"head.3.weight": W3.detach().clone(), "head.3.bias": b3.detach().clone()}
    np.savez_compressed(os.path.join(GOLD, f"fitted_head_{name}.npz"), **{k: v.numpy() for k, v in fitted.items()})
    sd.update(fitted)
    ref.load_state_dict(sd)
    with torch.no_grad():
        lg = ref.head(f).numpy()
        assert np.allclose(lg[:11], ref(ximg).numpy(), atol=1e-4)
        lge = ref.head(fe).numpy()
    out = {"image_logits": lg[:11], "fit_logits": lg[11:44], "heldout_logits": lg[44:],
           "mx_emulation_logits": lge}
    np.savez_compressed(os.path.join(GOLD, f"fitted_vectors_{name}.npz"), **out)
    for k in ("image_logits", "fit_logits", "heldout_logits"):
        srt = np.sort(out[k], axis=1)
        print(name, k, "argmax", out[k].argmax(1).tolist(), "min margin %.2f" % (srt[:, -1] - srt[:, -2]).min())
    print(name, "max |emulation - fp32| logits %.3f" % np.abs(lge - lg).max(), "emulation labels equal:",
          bool((lge[:44].argmax(1) == lg[:44].argmax(1)).all()))
    assert out["image_logits"].argmax(1).tolist() == list(range(11))
    assert out["fit_logits"].argmax(1).tolist() == [j % 11 for j in range(33)]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fitted":
        for nm in FITTED:
            make_fitted(nm)
    elif len(sys.argv) > 1 and sys.argv[1] == "fitted_big":
        for nm in FITTED_BIG:
            make_fitted_big(nm)
    else:
        main()
