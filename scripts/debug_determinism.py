"""Run the training-mode forward twice on the same model and input; report which buffers differ run to run."""
import sys, torch
sys.path.insert(0, ".")
from vit_amd.model import CNNViTHybrid
from vit_amd.synth import micro_doppler_batch, synth_state_dict
dev = torch.device("cuda:0")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = CNNViTHybrid(num_classes=11, embed_dim=128, num_heads=2, num_layers=2, dropout=0.0, backbone=sys.argv[1])
m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 9))
m = m.to(dev).train()
x = micro_doppler_batch(B, seed=4).to(dev)
eng = m.train_engine
snaps = []
for r in range(3):
    out = m(x)
    torch.cuda.synchronize()
    b = eng.buffers(B, dev)
    snap = {"logits": out.detach().float().clone()}
    def walk(pref, d):
        for k, v in d.items():
            if torch.is_tensor(v): snap[pref + str(k)] = v.detach().float().clone()
            elif isinstance(v, dict): walk(pref + str(k) + ".", v)
            elif isinstance(v, (list, tuple)):
                for i, e in enumerate(v):
                    if torch.is_tensor(e): snap[f"{pref}{k}[{i}]"] = e.detach().float().clone()
                    elif isinstance(e, dict): walk(f"{pref}{k}[{i}].", e)
    walk("", b)
    snaps.append(snap)
for k in snaps[0]:
    d01 = (snaps[0][k] - snaps[1][k]).abs().max().item()
    d02 = (snaps[0][k] - snaps[2][k]).abs().max().item()
    if d01 > 0 or d02 > 0:
        nz = ((snaps[0][k] - snaps[1][k]) != 0).float().mean().item()
        print(f"{k:40s} max|d| {d01:.3e} {d02:.3e}  frac differing {nz:.2e}  scale {snaps[0][k].abs().max().item():.3e}")
print("done; buffers compared:", len(snaps[0]))
