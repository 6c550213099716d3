"""Inference callers of the hot path, mirroring the function-level API of the reference's infer.py
(load_model :268-320, validate_input :323-343, get_transform :346-353, predict_single :356-406,
predict_folder :409-430, benchmark_inference :433-469) and the post-processing of SURVEY.md §8(a)
row 17 (softmax over the classes, top-k), on top of vit_amd.model.CNNViTHybrid.

Differences, all deliberate:
  * the model runs on librvit (HIP, bf16) and therefore needs a GPU device; there is no CPU path;
  * preprocessing is restated with PIL + numpy (torchvision is not a dependency):
    grayscale (ITU-R 601 luma, PIL "L") replicated to 3 channels -> bilinear resize to 224x224
    -> /255 -> ImageNet mean/std normalisation;
  * checkpoints are read with torch.load(weights_only=True): the documented layout
    (train_vit.py:996-1019: model / classes / model_config / metadata) holds only tensors and plain
    containers, nothing needs to be unpickled;
  * predict_batch / predict_folder batch the images through one forward (the reference loops).
Checkpoints with mode == "sequence" load as vit_amd.sequence.CNNViTSequence (streaming: StreamingClassifier there).
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .model import CNNViTHybrid

IMAGE_EXTENSIONS = {".png", ".jpg", ".jpeg", ".bmp", ".tiff", ".gif"}
_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


@dataclass
class PredictionResult:  # infer.py:48-55
    prediction: str
    confidence: float
    top_k: List[Tuple[str, float]]
    is_reliable: bool
    raw_logits: Optional[np.ndarray] = None


def load_model(checkpoint_path: str, device: torch.device) -> Tuple[torch.nn.Module, list, dict]:
    """(model in eval mode on `device`, class names, metadata) from a reference-layout checkpoint."""
    if not os.path.exists(checkpoint_path):
        raise FileNotFoundError(f"Checkpoint not found: {checkpoint_path}")
    # the reference stores `torch.__version__` (a str subclass, train_vit.py:1012) in the metadata: allow-list
    # that one class so the restricted unpickler still executes nothing from the file
    from torch.torch_version import TorchVersion
    with torch.serialization.safe_globals([TorchVersion]):
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    classes = ckpt["classes"]
    config = ckpt.get("model_config", {"num_classes": len(classes), "backbone": "resnet18", "embed_dim": 384,
                                       "num_heads": 6, "num_layers": 6, "dropout": 0.1, "mode": "single"})
    metadata = dict(ckpt.get("metadata", {"confidence_threshold": 0.5, "trained_at": "unknown",
                                          "config_hash": "unknown"}))
    mode = config.get("mode", "single")
    if mode == "sequence":     # infer.py:301-302
        from .sequence import CNNViTSequence
        model = CNNViTSequence(num_classes=config["num_classes"], config=config)
    else:
        model = CNNViTHybrid(num_classes=config["num_classes"], backbone=config.get("backbone", "resnet18"),
                             embed_dim=config["embed_dim"], num_heads=config["num_heads"],
                             num_layers=config["num_layers"], dropout=config["dropout"])
    model.load_state_dict(ckpt["model"])
    model.to(device)
    model.eval()
    metadata["mode"] = mode
    return model, classes, metadata


def validate_input(image_path: str) -> Tuple[bool, str]:
    from PIL import Image
    if not os.path.exists(image_path):
        return False, f"File not found: {image_path}"
    ext = Path(image_path).suffix.lower()
    if ext not in IMAGE_EXTENSIONS:
        return False, f"Invalid file type: {ext}. Supported: {IMAGE_EXTENSIONS}"
    try:
        with Image.open(image_path) as img:
            img.verify()
    except Exception as e:  # noqa: BLE001 - any decoder error means "corrupted"
        return False, f"Corrupted image file: {e}"
    return True, ""


def preprocess(img) -> torch.Tensor:
    """PIL image (any mode) or uint8 [H,W] luma array -> float32 [3,224,224], ImageNet-normalised."""
    from PIL import Image
    if isinstance(img, np.ndarray):
        img = Image.fromarray(img)
    g = img.convert("L")
    if g.size != (224, 224):
        g = g.resize((224, 224), Image.BILINEAR)
    a = np.asarray(g, dtype=np.float32) / 255.0
    x = (a[None, :, :] - _MEAN[:, None, None]) / _STD[:, None, None]
    return torch.from_numpy(np.ascontiguousarray(x))


def postprocess(logits: torch.Tensor, classes: Sequence[str], confidence_threshold: float = 0.5,
                top_k: int = 3) -> List[PredictionResult]:
    """softmax + top-k per row of `logits` [B, C] (infer.py:394-406)."""
    probs = torch.softmax(logits.float(), dim=1)
    k = min(top_k, len(classes))
    tp, ti = probs.topk(k, dim=1)
    tp, ti, lg = tp.cpu(), ti.cpu(), logits.detach().float().cpu().numpy()
    out = []
    for r in range(logits.shape[0]):
        conf = float(tp[r, 0])
        out.append(PredictionResult(prediction=classes[int(ti[r, 0])], confidence=conf,
                                    top_k=[(classes[int(i)], float(p)) for p, i in zip(tp[r], ti[r])],
                                    is_reliable=conf >= confidence_threshold, raw_logits=lg[r:r + 1]))
    return out


_DEVICE_PRE: dict = {}


def _device_batch(images: Sequence, device: torch.device) -> Optional[torch.Tensor]:
    """uint8 frames of one size -> the model input, resized / normalised on the GPU (vit_amd.pipeline: the arithmetic
    of `preprocess`, bit for bit); None when the frames differ in size or are not 8-bit."""
    arrs = []
    for im in images:
        a = im if isinstance(im, np.ndarray) else np.asarray(im if im.mode in ("L", "RGB", "RGBA") else im.convert("RGB"))
        if a.dtype != np.uint8 or a.ndim not in (2, 3) or (arrs and a.shape != arrs[0].shape):
            return None
        arrs.append(a)
    from .pipeline import DevicePreprocessor
    hw = tuple(arrs[0].shape[:2])
    key = (hw, str(device))
    pre = _DEVICE_PRE.get(key)
    if pre is None:
        pre = _DEVICE_PRE[key] = DevicePreprocessor(hw, train=False, device=device)
    return pre(torch.from_numpy(np.stack(arrs)).to(device, non_blocking=True))


def predict_batch(model: CNNViTHybrid, images: Sequence, classes: Sequence[str], device: torch.device,
                  confidence_threshold: float = 0.5, top_k: int = 3, device_preprocess: bool = False) -> List[PredictionResult]:
    """One batched forward over PIL images / luma arrays.  device_preprocess: resize / normalise same-sized 8-bit
    frames on the GPU instead of with Pillow on the host (same tensor)."""
    if len(images) == 0:
        return []
    x = _device_batch(images, device) if device_preprocess else None
    if x is None:
        x = torch.stack([preprocess(im) for im in images]).to(device)
    with torch.no_grad():
        logits = model(x)
    return postprocess(logits, classes, confidence_threshold, top_k)


def predict_single(model: CNNViTHybrid, image_path: str, classes: list, device: torch.device,
                   confidence_threshold: float = 0.5, top_k: int = 3) -> Optional[PredictionResult]:
    from PIL import Image
    ok, msg = validate_input(image_path)
    if not ok:
        print(f"Input validation failed: {msg}")
        return None
    try:
        img = Image.open(image_path).convert("RGB")
    except Exception as e:  # noqa: BLE001
        print(f"Error loading image {image_path}: {e}")
        return None
    return predict_batch(model, [img], classes, device, confidence_threshold, top_k)[0]


def predict_folder(model: CNNViTHybrid, folder_path: str, classes: list, device: torch.device,
                   confidence_threshold: float = 0.5, batch_size: int = 64,
                   device_preprocess: bool = False) -> Dict[str, PredictionResult]:
    from PIL import Image
    paths = [p for p in sorted(Path(folder_path).iterdir()) if p.suffix.lower() in IMAGE_EXTENSIONS
             and validate_input(str(p))[0]]
    results: Dict[str, PredictionResult] = {}
    for i in range(0, len(paths), batch_size):
        chunk = paths[i:i + batch_size]
        imgs = [Image.open(p).convert("RGB") for p in chunk]
        for p, r in zip(chunk, predict_batch(model, imgs, classes, device, confidence_threshold, top_k=1,
                                             device_preprocess=device_preprocess)):
            results[p.name] = r
    return results


def benchmark_inference(model: CNNViTHybrid, device: torch.device, num_runs: int = 100, batch_size: int = 1) -> dict:
    """Latency / throughput of the forward on random input (infer.py:433-469: 10 warm-up runs, then timed)."""
    x = torch.randn(batch_size, 3, 224, 224, device=device)
    with torch.no_grad():
        for _ in range(10):
            model(x)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(num_runs):
            model(x)
        torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    return {"batch_size": batch_size, "num_runs": num_runs, "latency_ms": dt / num_runs * 1e3,
            "fps": batch_size * num_runs / dt}
