"""ORACLE — test infrastructure only.  Never imported by vit_amd/ (the product path).

CPU restatement (torch fp32) of the reference classifier `CNNViTHybrid`
(/root/reference/train_vit.py:54-185; duplicates infer.py:69-180, deploy/infer_edge.py:77-159).

What is pinned and what is not
  * Tail (patch_embed .. head): built from the SAME torch.nn modules with the SAME constructor
    arguments as the reference (train_vit.py:93-127), so on this torch build it IS the reference
    arithmetic; state_dict keys/shapes equal the reference's (checked in
    tests/test_oracle.py against the shapes the reference source declares).
  * Stem: the reference takes it from `timm.create_model('resnet18', features_only=True)`
    (train_vit.py:87).  timm 1.x is an un-vendored, unpinned dependency (deploy/setup_pi.sh:76-79)
    that is not installed here, and the reference ships no tests, fixtures or checkpoints
    (.MISSING_LARGE_BLOBS).  The stem below restates the published ResNet-18 (He et al. 2015;
    BasicBlock x[2,2,2,2], conv7x7/2 + BN + ReLU, maxpool 3x3/2 pad 1, BN eps 1e-5) under timm's
    FeatureListNet key names.  PARITY UNPINNED at the timm boundary: no reference-held vector
    covers it.
  * Pure-ViT `vit_radar.pt` layout (BASELINE.json configs 1,4,5) has no defining code in the
    reference; `backbone='patch16'` below is this repo's definition.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class _BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = torch.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return torch.relu(y + idn)


class ResNetFeatures(nn.Module):
    """ResNet-18/34 trunk with timm FeatureListNet key names; forward returns the 5 feature maps."""
    DEPTHS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}

    def __init__(self, name: str = "resnet18"):
        super().__init__()
        d = self.DEPTHS[name]
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        chans = (64, 128, 256, 512)
        cin = 64
        for li, (c, n) in enumerate(zip(chans, d), start=1):
            blocks = []
            for bi in range(n):
                blocks.append(_BasicBlock(cin, c, 2 if (bi == 0 and li > 1) else 1))
                cin = c
            setattr(self, f"layer{li}", nn.Sequential(*blocks))

    def forward(self, x):
        f0 = torch.relu(self.bn1(self.conv1(x)))
        x = nn.functional.max_pool2d(f0, 3, 2, 1)
        f1 = self.layer1(x)
        f2 = self.layer2(f1)
        f3 = self.layer3(f2)
        f4 = self.layer4(f3)
        return [f0, f1, f2, f3, f4]


class CNNViTHybridRef(nn.Module):
    """train_vit.py:54-185 with the timm stem replaced by ResNetFeatures (or a 16x16 patch conv)."""
    BACKBONES = {"resnet18": (512, 7), "resnet34": (512, 7), "patch16": (3, 14)}

    def __init__(self, num_classes, embed_dim=384, num_heads=6, num_layers=6, dropout=0.1, backbone="resnet18"):
        super().__init__()
        if backbone not in self.BACKBONES:
            raise ValueError(f"Unknown backbone: {backbone}. Choose from {list(self.BACKBONES.keys())}")
        self.backbone_name = backbone
        cout, feat = self.BACKBONES[backbone]
        self.num_patches = feat * feat
        if backbone == "patch16":
            self.cnn_backbone = None
            self.patch_embed = nn.Conv2d(3, embed_dim, kernel_size=16, stride=16)
        else:
            self.cnn_backbone = ResNetFeatures(backbone)
            self.patch_embed = nn.Conv2d(cout, embed_dim, kernel_size=1)          # train_vit.py:93
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))               # :96
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim))  # :99
        self.pos_drop = nn.Dropout(dropout)
        layer = nn.TransformerEncoderLayer(d_model=embed_dim, nhead=num_heads, dim_feedforward=embed_dim * 4,
                                           dropout=dropout, activation="gelu", batch_first=True,
                                           norm_first=True)                       # :103-111
        self.transformer = nn.TransformerEncoder(layer, num_layers=num_layers, enable_nested_tensor=False)
        self.norm = nn.LayerNorm(embed_dim)                                       # :119
        self.head = nn.Sequential(nn.Linear(embed_dim, embed_dim // 2), nn.GELU(), nn.Dropout(dropout),
                                  nn.Linear(embed_dim // 2, num_classes))         # :122-127
        nn.init.trunc_normal_(self.pos_embed, std=0.02)                           # :132-139
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.head.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def tokens(self, x):
        B = x.shape[0]
        if self.cnn_backbone is not None:
            x = self.cnn_backbone(x)[-1]                                          # :145-146
        x = self.patch_embed(x)                                                   # :149
        x = x.flatten(2).transpose(1, 2)                                          # :152
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], dim=1)               # :155-156
        return self.pos_drop(x + self.pos_embed)                                  # :159-160

    def forward_features(self, x):                                                # :173-185
        return self.norm(self.transformer(self.tokens(x))[:, 0])

    def forward(self, x):                                                         # :141-171
        return self.head(self.forward_features(x))
