import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle.ref_model import CNNViTHybridRef
from vit_amd.model import CNNViTHybrid
from vit_amd.synth import synth_state_dict, micro_doppler_batch
from vit_amd.loss import cross_entropy
torch.set_num_threads(16)
cfg = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="resnet18")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = CNNViTHybrid(**cfg); sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 4); m.load_state_dict(sd)
ref = CNNViTHybridRef(**cfg); ref.load_state_dict(sd)
dev = torch.device("cuda:0"); m = m.to(dev).train(); ref.train()
x = micro_doppler_batch(B, seed=21); y = torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(3))
F.cross_entropy(ref(x), y, label_smoothing=0.05).backward()
cross_entropy(m(x.to(dev)), y.to(dev), 0.05).backward()
gref = dict(ref.named_parameters())
for n in m.train_engine.layout.names:
    p = dict(m.named_parameters())[n]
    g, r = p.grad.float().cpu(), gref[n].grad
    rel = ((g - r).norm() / r.norm().clamp_min(1e-12)).item()
    cos = F.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
    if n.endswith("weight") or "embed" in n or "cls" in n:
        print(f"{n:55s} rel {rel:.4f} cos {cos:.5f} |g| {r.norm():.3e}")
