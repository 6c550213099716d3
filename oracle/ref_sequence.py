"""ORACLE (test infrastructure, not product code): the reference's sequence model on the CPU in fp32.

`TemporalEncoderRef` / `CNNViTSequenceRef` instantiate the SAME torch.nn modules with the SAME arguments as
train_vit.py:188-318 (nn.LSTM(batch_first, bidirectional, dropout), Linear-GELU-Dropout-Linear classifier) around
`oracle.ref_model.CNNViTHybridRef`; since torch is the reference's own dependency, `nn.LSTM` IS the reference
arithmetic for the temporal part.  `lstm_explicit` restates nn.LSTM as a per-step loop (torch activation.py /
rnn.py cell equations, gate order i,f,g,o) so that the kernels' inter-layer dropout masks can be replayed; it is
pinned against nn.LSTM itself in tests/test_sequence.py (eval mode, to fp32 round-off).
PARITY UNPINNED beyond that: the reference ships no fixtures or tests for this model either.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from .ref_model import CNNViTHybridRef


class TemporalEncoderRef(nn.Module):       # train_vit.py:188-229
    def __init__(self, input_dim, hidden_dim=256, num_layers=2, dropout=0.3, bidirectional=True, model_type="lstm"):
        super().__init__()
        self.bidirectional, self.hidden_dim = bidirectional, hidden_dim
        cls = {"lstm": nn.LSTM, "gru": nn.GRU}[model_type]                       # train_vit.py:202-213
        self.rnn = cls(input_dim, hidden_dim, num_layers=num_layers, batch_first=True,
                       dropout=dropout if num_layers > 1 else 0, bidirectional=bidirectional)
        self.output_dim = hidden_dim * (2 if bidirectional else 1)

    def forward(self, x):
        output, _ = self.rnn(x)
        if self.bidirectional:
            return torch.cat([output[:, -1, :self.hidden_dim], output[:, 0, self.hidden_dim:]], dim=1)
        return output[:, -1, :]


class CNNViTSequenceRef(nn.Module):        # train_vit.py:232-318
    def __init__(self, num_classes, config):
        super().__init__()
        self.spatial_encoder = CNNViTHybridRef(num_classes=num_classes, backbone=config["backbone"], embed_dim=config["embed_dim"],
                                               num_heads=config["num_heads"], num_layers=config["num_layers"],
                                               dropout=config["dropout"])
        self.temporal_encoder = TemporalEncoderRef(config["embed_dim"], config.get("temporal_hidden", 256),
                                                   config.get("temporal_layers", 2), config.get("temporal_dropout", 0.3),
                                                   config.get("temporal_bidirectional", True),
                                                   config.get("temporal_model", "lstm"))
        d = self.temporal_encoder.output_dim
        self.classifier = nn.Sequential(nn.Linear(d, d // 2), nn.GELU(), nn.Dropout(config.get("classifier_dropout", 0.4)),
                                        nn.Linear(d // 2, num_classes))

    def forward(self, x):
        B, T, C, H, W = x.shape
        f = self.spatial_encoder.forward_features(x.view(B * T, C, H, W)).view(B, T, -1)
        return self.classifier(self.temporal_encoder(f))

    def forward_temporal_only(self, f):
        return self.classifier(self.temporal_encoder(f))


def lstm_explicit(rnn: nn.LSTM, x: torch.Tensor, masks: Optional[Dict[int, torch.Tensor]] = None, p: float = 0.0):
    """nn.LSTM (batch_first, zero initial state) as explicit per-step equations.  masks[layer] (bool/0-1, shape
    [B,T,ndir*H]) = keep-mask applied, scaled by 1/(1-p), to that layer's output before the next layer."""
    H, ndir = rnn.hidden_size, 2 if rnn.bidirectional else 1
    B, T, _ = x.shape
    cur = x
    for layer in range(rnn.num_layers):
        outs = []
        for d in range(ndir):
            sfx = "_reverse" if d else ""
            w_ih, w_hh = getattr(rnn, f"weight_ih_l{layer}{sfx}"), getattr(rnn, f"weight_hh_l{layer}{sfx}")
            b_ih, b_hh = getattr(rnn, f"bias_ih_l{layer}{sfx}"), getattr(rnn, f"bias_hh_l{layer}{sfx}")
            h = x.new_zeros(B, H)
            c = x.new_zeros(B, H)
            hs = [None] * T
            for t in (range(T - 1, -1, -1) if d else range(T)):
                pre = cur[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
                i, f, g, o = pre.chunk(4, dim=1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h = torch.sigmoid(o) * torch.tanh(c)
                hs[t] = h
            outs.append(torch.stack(hs, dim=1))
        cur = torch.cat(outs, dim=2)
        if masks is not None and layer in masks:
            cur = cur * masks[layer].to(cur.dtype) / (1.0 - p)
    return cur


def temporal_forward_with_masks(ref: CNNViTSequenceRef, feats: torch.Tensor, lstm_masks: Dict[int, torch.Tensor], p_lstm: float,
                                cls_mask: Optional[torch.Tensor], p_cls: float) -> torch.Tensor:
    """forward_temporal_only in train mode with given dropout keep-masks instead of torch's generator."""
    enc = ref.temporal_encoder
    out = lstm_explicit(enc.rnn, feats, lstm_masks, p_lstm)
    H = enc.hidden_dim
    tf = torch.cat([out[:, -1, :H], out[:, 0, H:]], dim=1) if enc.bidirectional else out[:, -1]
    h = ref.classifier[1](ref.classifier[0](tf))
    if cls_mask is not None:
        h = h * cls_mask.to(h.dtype) / (1.0 - p_cls)
    return ref.classifier[3](h)
