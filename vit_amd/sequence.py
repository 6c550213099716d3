"""Sequence model of the reference: CNN-ViT per frame + bidirectional LSTM across frames +
classifier, with the streaming wrapper -- `TemporalEncoder` / `CNNViTSequence` (train_vit.py:188-318,
infer.py:186-265) and `StreamingClassifier` (infer.py:479-630) -- on librvit.

The modules only HOLD parameters, under the reference's state_dict names (`spatial_encoder.*`,
`temporal_encoder.rnn.weight_ih_l0[_reverse]` ..., `classifier.0/.3.*`), so a `cnn_vit_lstm_radar.pt` checkpoint
loads unchanged.  Arithmetic:
  * frames -> features: `CNNViTHybrid.forward_features` (engine / training engine of this package);
  * LSTM / GRU: one fused HIP launch per time step for both directions (csrc/lstm.hip, fp32 straight from the fp32
    parameters), the step's backward likewise; weight / bias / input gradients from the saved gate gradients
    with the package's GEMM kernels;
  * classifier: the same fused GEMM epilogues as the single-frame head.
Both `model_type='lstm'` (the reference default, train_vit.py:267) and 'gru' are built.
There is no CPU path.
"""
from __future__ import annotations

import math
import time
from collections import deque
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from .model import CNNViTHybrid, _Linear
from .ops import _ptr, _stream, check

BF16 = torch.bfloat16
F32 = torch.float32
_M64 = (1 << 64) - 1


class _LSTMParams(nn.Module):
    """Parameter holder with nn.LSTM's / nn.GRU's names, shapes and default init (uniform +-1/sqrt(hidden));
    `gates` = 4 (LSTM: i,f,g,o) or 3 (GRU: r,z,n)."""

    def __init__(self, input_dim: int, hidden: int, num_layers: int, bidirectional: bool, gates: int = 4):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.bidirectional = input_dim, hidden, num_layers, bidirectional
        self.gates = gates
        k = 1.0 / math.sqrt(hidden)
        ndir = 2 if bidirectional else 1
        for layer in range(num_layers):
            fin = input_dim if layer == 0 else hidden * ndir
            for sfx in ("", "_reverse")[:ndir]:
                for name, shape in ((f"weight_ih_l{layer}{sfx}", (gates * hidden, fin)),
                                    (f"weight_hh_l{layer}{sfx}", (gates * hidden, hidden)),
                                    (f"bias_ih_l{layer}{sfx}", (gates * hidden,)), (f"bias_hh_l{layer}{sfx}", (gates * hidden,))):
                    self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def direction(self, layer: int, rev: bool):
        s = "_reverse" if rev else ""
        return tuple(getattr(self, f"{n}_l{layer}{s}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))


class TemporalEncoder(nn.Module):
    """train_vit.py:188-229: (bi)LSTM over frame features; returns [forward h at t = T-1 | backward h at t = 0]."""

    def __init__(self, input_dim, hidden_dim=256, num_layers=2, dropout=0.3, bidirectional=True, model_type="lstm"):
        super().__init__()
        if model_type not in ("lstm", "gru"):
            raise ValueError(f"Unknown model_type: {model_type}")
        if hidden_dim % 8 or input_dim % 4:
            raise NotImplementedError("the recurrent kernels need hidden_dim % 8 == 0 and input_dim % 4 == 0")
        self.model_type, self.bidirectional, self.hidden_dim = model_type, bidirectional, hidden_dim
        self.num_layers = num_layers
        self.dropout = float(dropout) if num_layers > 1 else 0.0
        self.rnn = _LSTMParams(input_dim, hidden_dim, num_layers, bidirectional, gates=4 if model_type == "lstm" else 3)
        self.output_dim = hidden_dim * (2 if bidirectional else 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _TemporalFunction.apply(x, self, None, self.training and torch.is_grad_enabled(), *self.parameters())


def _site_seed(base: int, site: int) -> int:
    z = (base + 0x9E3779B97F4A7C15 * (site + 1)) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _lstm_forward(enc: TemporalEncoder, x: torch.Tensor, train: bool, base_seed: int):
    """x [B,T,In] fp32 -> (top-layer output [B,T,ndir*H], saved tensors per layer)."""
    lib = _lib.load()
    B, T, _ = x.shape
    H, ndir = enc.hidden_dim, 2 if enc.bidirectional else 1
    saved = []
    cur = x.contiguous()
    for layer in range(enc.num_layers):
        In = cur.shape[2]
        out = torch.empty(B, T, ndir * H, dtype=F32, device=x.device)
        lstm = enc.model_type == "lstm"
        c = torch.empty(ndir, B, H, dtype=F32, device=x.device) if lstm else None
        gates = torch.empty(B, T, ndir, 4 * H, dtype=F32, device=x.device) if train else None   # GRU: (r, z, n, hn)
        cseq = torch.empty(B, T, ndir, H, dtype=F32, device=x.device) if train and lstm else None
        wf = [p.detach() for p in enc.rnn.direction(layer, False)]
        wr = [p.detach() for p in enc.rnn.direction(layer, True)] if ndir == 2 else [None] * 4
        with ops._Span("rnn_fwd", 2 * B * T * ndir * enc.rnn.gates * H * (In + H), 0):
            for s in range(T):
                if lstm:
                    check(lib.rvit_lstm_step_fwd(_ptr(cur), *[_ptr(w) for w in wf], *[_ptr(w) for w in wr], _ptr(out),
                                                 _ptr(c), _ptr(gates), _ptr(cseq), B, T, In, H, s, _stream()),
                          "rvit_lstm_step_fwd")
                else:
                    check(lib.rvit_gru_step_fwd(_ptr(cur), *[_ptr(w) for w in wf], *[_ptr(w) for w in wr], _ptr(out),
                                                _ptr(gates), B, T, In, H, s, _stream()), "rvit_gru_step_fwd")
        nxt = out
        seed = 0
        if train and enc.dropout > 0 and layer < enc.num_layers - 1:   # nn.LSTM: dropout on every layer's output but the last
            seed = _site_seed(base_seed, 100 + layer)
            nxt = torch.empty_like(out)
            ops.dropout_apply(out.view(B * T, ndir * H), nxt.view(B * T, ndir * H), enc.dropout, seed)
        saved.append(dict(x=cur, out=out, gates=gates, cseq=cseq, seed=seed))
        cur = nxt
    return cur, saved


def _lstm_backward(enc: TemporalEncoder, saved, dtop: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """dtop [B,T,ndir*H] -> (dx of layer 0 [B,T,In], {parameter name: gradient})."""
    lib = _lib.load()
    H, ndir = enc.hidden_dim, 2 if enc.bidirectional else 1
    grads: Dict[str, torch.Tensor] = {}
    dout = dtop.contiguous()
    for layer in reversed(range(enc.num_layers)):
        sv = saved[layer]
        x, out = sv["x"], sv["out"]
        B, T, In = x.shape
        dev = x.device
        lstm = enc.model_type == "lstm"
        G = enc.rnn.gates * H                         # gate rows: 4H (LSTM) or 3H (GRU)
        # LSTM: one set of pre-activation gradients; GRU: input-side then hidden-side (they differ in the n gate)
        dG = torch.empty(B, T, ndir, G if lstm else 2 * G, dtype=F32, device=dev)
        carry = torch.empty(ndir, B, H, dtype=F32, device=dev)
        whh_t = [enc.rnn.direction(layer, d == 1)[1].detach().t().contiguous() for d in range(ndir)]
        with ops._Span("rnn_bwd", 2 * B * T * ndir * G * H, 0):
            for s in reversed(range(T)):
                if lstm:
                    check(lib.rvit_lstm_step_bwd(_ptr(dout), _ptr(whh_t[0]), _ptr(whh_t[1]) if ndir == 2 else None,
                                                 _ptr(sv["gates"]), _ptr(sv["cseq"]), _ptr(dG), _ptr(carry), B, T, H, s,
                                                 _stream()), "rvit_lstm_step_bwd")
                else:
                    check(lib.rvit_gru_step_bwd(_ptr(dout), _ptr(whh_t[0]), _ptr(whh_t[1]) if ndir == 2 else None,
                                                _ptr(sv["gates"]), _ptr(out), _ptr(dG), _ptr(carry), B, T, H, s,
                                                _stream()), "rvit_gru_step_bwd")
        x16 = x.reshape(B * T, In).to(BF16)
        dx = torch.empty(B * T, In, dtype=F32, device=dev)
        for d in range(ndir):
            sfx = "_reverse" if d else ""
            w_ih = enc.rnn.direction(layer, d == 1)[0].detach()
            dGd = dG[:, :, d, :G].reshape(B * T, G)                       # fp32, contiguous copy of this direction
            dGh = dGd if lstm else dG[:, :, d, G:].reshape(B * T, G)      # hidden-side gradients (GRU)
            dG16 = dGd.to(BF16)
            dGh16 = dG16 if lstm else dGh.to(BF16)
            hprev = torch.zeros(B, T, H, dtype=BF16, device=dev)          # h of the previous step of this direction
            if T > 1:
                if d == 0:
                    hprev[:, 1:] = out[:, :-1, :H]
                else:
                    hprev[:, :-1] = out[:, 1:, H:]
            g_ih = torch.zeros(G, In, dtype=F32, device=dev)
            g_hh = torch.zeros(G, H, dtype=F32, device=dev)
            g_b = torch.zeros(G, dtype=F32, device=dev)
            ops.wgrad(dG16, x16, g_ih)
            ops.wgrad(dGh16, hprev.view(B * T, H), g_hh)
            ops.batch_sum(dGd, g_b)
            if lstm:
                g_bh = g_b.clone()
            else:
                g_bh = torch.zeros(G, dtype=F32, device=dev)
                ops.batch_sum(dGh, g_bh)
            grads[f"weight_ih_l{layer}{sfx}"], grads[f"weight_hh_l{layer}{sfx}"] = g_ih, g_hh
            grads[f"bias_ih_l{layer}{sfx}"], grads[f"bias_hh_l{layer}{sfx}"] = g_b, g_bh
            # dx (+)= dG_d @ W_ih: gemm computes a @ w.T with w [N=In, K=4H]
            ops.gemm(dG16, w_ih.t().contiguous().to(BF16), dx, residual=dx if d else None)
        dx = dx.view(B, T, In)
        if layer > 0:
            seed = saved[layer - 1]["seed"]
            if seed:
                dd = torch.empty_like(dx)
                ops.dropout_apply(dx.view(B * T, In), dd.view(B * T, In), enc.dropout, seed)
                dx = dd
        dout = dx
    return dout, grads


def _select_last(enc: TemporalEncoder, top: torch.Tensor) -> torch.Tensor:
    H = enc.hidden_dim
    if enc.bidirectional:     # train_vit.py:225-228
        return torch.cat([top[:, -1, :H], top[:, 0, H:]], dim=1)
    return top[:, -1, :].contiguous()


def _scatter_last(enc: TemporalEncoder, dtf: torch.Tensor, B: int, T: int) -> torch.Tensor:
    H = enc.hidden_dim
    ndir = 2 if enc.bidirectional else 1
    dtop = torch.zeros(B, T, ndir * H, dtype=F32, device=dtf.device)
    dtop[:, -1, :H] = dtf[:, :H]
    if enc.bidirectional:
        dtop[:, 0, H:] = dtf[:, H:]
    return dtop


class _TemporalFunction(torch.autograd.Function):
    """features [B,T,E] -> temporal feature [B,out_dim] (classifier=None) or logits [B,classes]."""

    @staticmethod
    def forward(ctx, feats, enc: TemporalEncoder, cls, train: bool, *params):
        if not feats.is_cuda:
            raise RuntimeError("the temporal encoder runs on the GPU through librvit only; there is no CPU fallback")
        with torch.cuda.device(feats.device):   # librvit launches on the current device's stream
            return _TemporalFunction._forward(ctx, feats, enc, cls, train)

    @staticmethod
    def _forward(ctx, feats, enc: TemporalEncoder, cls, train: bool):
        if feats.dim() != 3 or feats.shape[2] != enc.rnn.input_size:
            raise ValueError(f"expected features [B,T,{enc.rnn.input_size}], got {tuple(feats.shape)}")
        base = int(torch.randint(0, 2 ** 62, (1,)).item()) if train else 0
        top, saved = _lstm_forward(enc, feats.detach().float(), train, base)
        tf = _select_last(enc, top)
        ctx.enc, ctx.cls, ctx.saved, ctx.shape = enc, cls, saved, tuple(feats.shape)
        if cls is None:
            return tf
        D = enc.output_dim
        B = tf.shape[0]
        l0, l3 = cls["0"], cls["3"]
        C = l3.weight.shape[0]
        tf16 = tf.to(BF16)
        h = torch.empty(B, D // 2, dtype=BF16, device=tf.device)
        u0 = torch.empty(B, D // 2, dtype=BF16, device=tf.device) if train else None
        p = cls.drop_p if train else 0.0
        ctx.cls_seed = _site_seed(base, 1) if p > 0 else 0
        ops.gemm(tf16, l0.weight.detach().to(BF16), h, bias=l0.bias.detach(), act="gelu", preact=u0,
                 drop=("post_act", p, ctx.cls_seed) if p > 0 else None)
        logits = torch.empty(B, C, dtype=F32, device=tf.device)
        ops.gemm(h, l3.weight.detach().to(BF16), logits, bias=l3.bias.detach())
        ctx.cls_saved = (tf16, h, u0, p)
        return logits

    @staticmethod
    def backward(ctx, dout):
        with torch.cuda.device(dout.device):
            return _TemporalFunction._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        enc, cls, saved = ctx.enc, ctx.cls, ctx.saved
        if saved[0]["gates"] is None:
            raise RuntimeError("backward through a forward that ran in eval mode / without grad")
        B, T, _ = ctx.shape
        dev = dout.device
        cls_grads: List[torch.Tensor] = []
        if cls is None:
            dtf = dout.contiguous().float()
        else:
            tf16, h, u0, p = ctx.cls_saved
            l0, l3 = cls["0"], cls["3"]
            C, Dh = l3.weight.shape
            Cp = 16 * ((C + 15) // 16)
            dl16 = torch.zeros(B, Cp, dtype=BF16, device=dev)
            dl16[:, :C] = dout
            g3 = torch.zeros(C, Dh, dtype=F32, device=dev)
            ops.wgrad(dl16, h, g3, n_store=C)
            b3 = torch.zeros(Cp, dtype=F32, device=dev)
            ops.colsum(dl16, b3)
            w3t = torch.zeros(Dh, Cp, dtype=BF16, device=dev)
            w3t[:, :C] = l3.weight.detach().t().to(BF16)
            dh = torch.empty(B, Dh, dtype=BF16, device=dev)
            ops.gemm(dl16, w3t, dh, gate=u0, gate_mode="gelu_grad", drop=("post_act", p, ctx.cls_seed) if p > 0 else None)
            g0 = torch.zeros(Dh, tf16.shape[1], dtype=F32, device=dev)
            ops.wgrad(dh, tf16, g0)
            b0 = torch.zeros(Dh, dtype=F32, device=dev)
            ops.colsum(dh, b0)
            dtf = torch.empty(B, tf16.shape[1], dtype=F32, device=dev)
            ops.gemm(dh, l0.weight.detach().t().contiguous().to(BF16), dtf)
            cls_grads = [g0, b0, g3, b3[:C].contiguous()]
        dfeats, g = _lstm_backward(enc, saved, _scatter_last(enc, dtf, B, T))
        enc_grads = [g[n] for n, _ in enc.rnn.named_parameters()]
        return (dfeats, None, None, None, *enc_grads, *cls_grads)


class _Classifier(nn.ModuleDict):
    """nn.Sequential(Linear, GELU, Dropout, Linear) parameter holder with the Sequential's key names."""

    def __init__(self, dim: int, num_classes: int, drop_p: float):
        super().__init__({"0": _Linear(dim, dim // 2), "3": _Linear(dim // 2, num_classes)})
        self.drop_p = float(drop_p)


class CNNViTSequence(nn.Module):
    """train_vit.py:232-318 / infer.py:230-265: [B,T,3,224,224] -> logits [B,num_classes]."""

    def __init__(self, num_classes, config):
        super().__init__()
        self.spatial_encoder = CNNViTHybrid(num_classes=num_classes, backbone=config["backbone"], embed_dim=config["embed_dim"],
                                            num_heads=config["num_heads"], num_layers=config["num_layers"],
                                            dropout=config["dropout"], pretrained_backbone=False)
        self.temporal_encoder = TemporalEncoder(input_dim=config["embed_dim"], hidden_dim=config.get("temporal_hidden", 256),
                                                num_layers=config.get("temporal_layers", 2),
                                                dropout=config.get("temporal_dropout", 0.3),
                                                bidirectional=config.get("temporal_bidirectional", True),
                                                model_type=config.get("temporal_model", "lstm"))
        d = self.temporal_encoder.output_dim
        if (d // 2) % 8:
            raise NotImplementedError("classifier width must be a multiple of 8 for the GEMM kernels")
        self.classifier = _Classifier(d, num_classes, config.get("classifier_dropout", 0.4))
        self.config = config
        self.embed_dim = config["embed_dim"]

    def forward(self, x):
        if x.dim() != 5:
            raise ValueError(f"expected [B,T,3,224,224], got {tuple(x.shape)}")
        B, T = x.shape[:2]
        feats = self.spatial_encoder.forward_features(x.reshape(B * T, *x.shape[2:]))
        return self.forward_temporal_only(feats.view(B, T, -1))

    def forward_spatial_only(self, x):
        """Single frames -> spatial features (streaming)."""
        return self.spatial_encoder.forward_features(x)

    def forward_temporal_only(self, spatial_features):
        """[B,T,E] features -> logits (streaming)."""
        enc = self.temporal_encoder
        return _TemporalFunction.apply(spatial_features, enc, self.classifier, self.training and torch.is_grad_enabled(),
                                       *enc.parameters(), *self.classifier.parameters())

    def freeze_backbone(self):
        for p in self.spatial_encoder.cnn_backbone.parameters():
            p.requires_grad = False

    def unfreeze_backbone(self):
        for p in self.spatial_encoder.cnn_backbone.parameters():
            p.requires_grad = True


@dataclass
class StreamPrediction:   # infer.py:58-66
    prediction: str
    confidence: float
    is_reliable: bool
    all_probs: List[Tuple[str, float]]
    frame_count: int
    latency_ms: float


class StreamingClassifier:
    """infer.py:479-630: sliding window of per-frame features; a prediction once `sequence_length` frames are in,
    smoothed over the last `smoothing_window` predictions.  Accepts a checkpoint path (reference layout, read with
    weights_only=True) or a ready (model, classes, config) triple."""

    def __init__(self, model_path=None, device: str = "auto", smoothing_window: int = 3, *, model=None, classes=None,
                 config=None, confidence_threshold: float = 0.5, device_preprocess: bool = False):
        # device_preprocess: uint8 frames given as arrays are resized / normalised on the GPU (vit_amd.pipeline, the
        # same arithmetic as the host transform, bit for bit) instead of by Pillow on the host
        self.device_preprocess = bool(device_preprocess)
        self._pre = {}
        self.device = torch.device("cuda" if device == "auto" else device)
        if self.device.type != "cuda":
            raise RuntimeError("StreamingClassifier runs on the GPU through librvit only")
        if model is None:
            from torch.torch_version import TorchVersion
            with torch.serialization.safe_globals([TorchVersion]):
                ckpt = torch.load(model_path, map_location="cpu", weights_only=True)
            classes = ckpt["classes"]
            config = ckpt.get("model_config", {})
            confidence_threshold = ckpt.get("metadata", {}).get("confidence_threshold", 0.5)
            if config.get("mode", "single") == "sequence":
                model = CNNViTSequence(num_classes=len(classes), config=config)
            else:
                model = CNNViTHybrid(num_classes=len(classes), backbone=config.get("backbone", "resnet18"),
                                     embed_dim=config.get("embed_dim", 256), num_heads=config.get("num_heads", 4),
                                     num_layers=config.get("num_layers", 4), dropout=config.get("dropout", 0.1))
            model.load_state_dict(ckpt["model"])
        self.classes, self.config = list(classes), dict(config or {})
        self.mode = "sequence" if isinstance(model, CNNViTSequence) else "single"
        self.sequence_length = self.config.get("sequence_length", 8) if self.mode == "sequence" else 1
        self.confidence_threshold = confidence_threshold
        self.model = model.to(self.device).eval()
        self.feature_buffer = deque(maxlen=self.sequence_length)
        self.prediction_history = deque(maxlen=smoothing_window)
        self.frame_count = 0

    def _preprocess(self, image_input) -> torch.Tensor:
        from .infer import preprocess
        if isinstance(image_input, torch.Tensor):
            return image_input
        if isinstance(image_input, np.ndarray):
            if self.device_preprocess and image_input.dtype == np.uint8 and image_input.ndim in (2, 3):
                from .pipeline import DevicePreprocessor
                hw = tuple(image_input.shape[:2])
                pre = self._pre.get(hw)
                if pre is None:
                    pre = self._pre[hw] = DevicePreprocessor(hw, train=False, device=self.device)
                frame = torch.from_numpy(np.ascontiguousarray(image_input))[None].to(self.device, non_blocking=True)
                return pre(frame)[0]
            if image_input.ndim == 2:
                image_input = np.stack([image_input] * 3, axis=-1)
            from PIL import Image
            return preprocess(Image.fromarray(image_input.astype(np.uint8)))
        if isinstance(image_input, (str, bytes)) or hasattr(image_input, "__fspath__"):
            from PIL import Image
            return preprocess(Image.open(image_input).convert("RGB"))
        if hasattr(image_input, "convert"):
            return preprocess(image_input)
        raise TypeError(f"Unsupported input type: {type(image_input)}")

    @torch.no_grad()
    def update(self, frame) -> Optional[StreamPrediction]:
        start = time.perf_counter()
        tensor = self._preprocess(frame).unsqueeze(0).to(self.device)
        self.frame_count += 1
        if self.mode == "sequence":
            self.feature_buffer.append(self.model.forward_spatial_only(tensor))
            if len(self.feature_buffer) < self.sequence_length:
                return None
            logits = self.model.forward_temporal_only(torch.cat(list(self.feature_buffer), dim=0).unsqueeze(0))
        else:
            logits = self.model(tensor)
        probs = torch.softmax(logits.float(), dim=1)[0].cpu()
        self.prediction_history.append(probs)
        avg = torch.stack(list(self.prediction_history)).mean(dim=0) if len(self.prediction_history) > 1 else probs
        conf, idx = avg.max(0)
        conf, idx = conf.item(), idx.item()
        all_probs = sorted(((self.classes[i], avg[i].item()) for i in range(len(self.classes))), key=lambda v: v[1], reverse=True)
        return StreamPrediction(prediction=self.classes[idx], confidence=conf, is_reliable=conf >= self.confidence_threshold,
                                all_probs=all_probs, frame_count=self.frame_count,
                                latency_ms=(time.perf_counter() - start) * 1000)

    def reset(self):
        self.feature_buffer.clear()
        self.prediction_history.clear()
        self.frame_count = 0

    def benchmark(self, n_frames: int = 100) -> dict:
        dummy = (np.random.default_rng(0).random((224, 224, 3)) * 255).astype(np.uint8)
        self.reset()
        for _ in range(self.sequence_length + 5):
            self.update(dummy)
        self.reset()
        lat = [r.latency_ms for r in (self.update(dummy) for _ in range(n_frames)) if r]
        return {"mean_latency_ms": float(np.mean(lat)) if lat else 0, "std_latency_ms": float(np.std(lat)) if lat else 0,
                "fps": 1000 / float(np.mean(lat)) if lat else 0, "sequence_length": self.sequence_length, "mode": self.mode,
                "device": str(self.device)}
