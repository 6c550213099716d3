"""Where the host time of a step goes (cProfile): python scripts/profile_host.py [train|infer] [batch]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd.model import CNNViTHybrid  # noqa: E402
from vit_amd.synth import micro_doppler_batch, synth_state_dict  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "train"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    m = CNNViTHybrid(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone="resnet18")
    m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=0))
    m = m.to(dev)
    x = micro_doppler_batch(B, seed=1).to(dev)
    if mode == "train":
        from vit_amd.loss import CrossEntropyLoss
        from vit_amd.optim import FusedAdamW
        m.train()
        y = torch.randint(0, 11, (B,)).to(dev)
        opt = FusedAdamW(m, lr=1e-4, weight_decay=0.05, max_grad_norm=1.0)
        crit = CrossEntropyLoss(label_smoothing=0.05)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = crit(m(x), y)
            loss.backward()
            opt.step()
    else:
        m.eval()

        def step():
            with torch.no_grad():
                m(x)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    torch.autograd.set_multithreading_enabled(False)   # run backward on this thread so that it is profiled
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)


if __name__ == "__main__":
    main()
