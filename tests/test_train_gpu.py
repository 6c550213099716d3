"""One training step on librvit (forward in train() mode, loss, backward) against torch autograd on
the fp32 CPU oracle with the same synthetic checkpoint, inputs and labels (north_star: "gradients
stay within tolerance across one training step").  dropout = 0 in both (torch's Philox dropout
stream cannot be reproduced by another kernel).

Stated tolerance: per parameter tensor, relative L2 error <= max(grad_rel, 1.6 x E_bf16), where
E_bf16 is the error the fp32 CPU oracle ITSELF shows for that tensor when only its stem
activations are rounded to bf16 (oracle/ref_bf16_train.py).  On the encoder/head E_bf16 ~ 0 and
the bound is grad_rel = 5-6 %; through the 20-conv BatchNorm/ReLU stem of this random-weight
network at batch 8, bf16 storage alone moves the oracle's conv gradients by 11-26 % (ReLU masks
of near-zero pre-activations flip), so no bf16 implementation can sit closer to fp32 than that.
The loss must agree within LOSS_TOL; BatchNorm running statistics must match the oracle's.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LOSS_TOL = 0.02


def _pair(cfg, seed, dev):
    from oracle.ref_model import CNNViTHybridRef
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import synth_state_dict
    m = CNNViTHybrid(**cfg)
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed)
    m.load_state_dict(sd)
    ref = CNNViTHybridRef(**cfg)
    ref.load_state_dict(sd)
    return m.to(dev).train(), ref.train()


CASES = [
    ("vit-small-patch16", dict(num_classes=11, embed_dim=128, num_heads=2, num_layers=2, dropout=0.0, backbone="patch16"),
     6, 0.05, 0.995),
    ("vit-E256-L4", dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="patch16"),
     3, 0.06, 0.995),
    ("R-hybrid", dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="resnet18"),
     8, 0.06, 0.90),
]


@pytest.mark.parametrize("name,cfg,B,grad_rel,grad_cos", CASES, ids=[c[0] for c in CASES])
def test_one_training_step_matches_autograd(dev, name, cfg, B, grad_rel, grad_cos):
    from vit_amd.loss import cross_entropy
    from vit_amd.synth import micro_doppler_batch
    torch.set_num_threads(16)
    m, ref = _pair(cfg, seed=4, dev=dev)
    x = micro_doppler_batch(B, seed=21)
    y = torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(3))
    loss_ref = F.cross_entropy(ref(x), y, label_smoothing=0.05)
    loss_ref.backward()
    # sensitivity of the oracle's own gradients to bf16 activation storage in the stem
    e_bf16 = {}
    if cfg["backbone"] != "patch16":
        from oracle.ref_bf16_train import attach_bf16_rounding
        _, ref2 = _pair(cfg, seed=4, dev=dev)
        attach_bf16_rounding(ref2)
        F.cross_entropy(ref2(x), y, label_smoothing=0.05).backward()
        g0 = dict(ref.named_parameters())
        for n, p in ref2.named_parameters():
            e_bf16[n] = ((p.grad - g0[n].grad).norm() / g0[n].grad.norm().clamp_min(1e-12)).item()
    logits = m(x.to(dev))
    loss = cross_entropy(logits, y.to(dev), label_smoothing=0.05)
    loss.backward()
    print(f"\n[{name}] loss hip {loss.item():.5f} ref {loss_ref.item():.5f}")
    assert abs(loss.item() - loss_ref.item()) <= LOSS_TOL * max(1.0, abs(loss_ref.item()))
    gref = dict(ref.named_parameters())
    worst = []
    for n, p in m.named_parameters():
        assert p.grad is not None, f"no gradient for {n}"
        g, r = p.grad.float().cpu(), gref[n].grad
        assert g.shape == r.shape
        rel = ((g - r).norm() / r.norm().clamp_min(1e-12)).item()
        cos = F.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
        worst.append((rel, cos, n))
    worst.sort(reverse=True)
    for rel, cos, n in worst[:6]:
        print(f"   {n:55s} rel {rel:.4f} cos {cos:.5f}  (oracle bf16 sensitivity {e_bf16.get(n, 0.0):.4f})")
    bad = [(n, rel, cos) for rel, cos, n in worst
           if rel > max(grad_rel, 1.6 * e_bf16.get(n, 0.0)) or cos < grad_cos]
    assert not bad, f"{name}: gradients out of tolerance: {bad[:5]}"
    if cfg["backbone"] != "patch16":
        sd_h, sd_r = m.state_dict(), ref.state_dict()
        for k in sd_r:
            if k.endswith("running_mean") or k.endswith("running_var"):
                assert torch.allclose(sd_h[k].cpu(), sd_r[k], rtol=0.03, atol=0.02), k
            if k.endswith("num_batches_tracked"):
                assert int(sd_h[k]) == int(sd_r[k])


def test_torch_loss_and_feature_gradients(dev):
    """The model's autograd node also works with a stock torch loss and through forward_features."""
    from vit_amd.synth import micro_doppler_batch
    name, cfg, B, grad_rel, grad_cos = CASES[0]
    m, ref = _pair(cfg, seed=5, dev=dev)
    x = micro_doppler_batch(B, seed=9)
    w = torch.randn(B, cfg["embed_dim"], generator=torch.Generator().manual_seed(1))
    (ref.forward_features(x) * w).sum().backward()
    (m.forward_features(x.to(dev)) * w.to(dev)).sum().backward()
    gref = dict(ref.named_parameters())
    for n, p in m.named_parameters():
        if n.startswith("head."):
            continue
        r = gref[n].grad
        rel = ((p.grad.cpu() - r).norm() / r.norm().clamp_min(1e-12)).item()
        assert rel < 0.06, (n, rel)


def test_sgd_steps_reduce_the_loss(dev):
    """A few optimiser steps through the HIP forward/backward lower the loss (end-to-end sanity)."""
    from vit_amd.loss import CrossEntropyLoss
    from vit_amd.synth import micro_doppler_batch
    name, cfg, B, _, _ = CASES[2]
    m, _ = _pair(cfg, seed=6, dev=dev)
    x = micro_doppler_batch(16, seed=2).to(dev)
    y = torch.randint(0, 11, (16,), generator=torch.Generator().manual_seed(5)).to(dev)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-4, weight_decay=0.05)
    crit = CrossEntropyLoss(label_smoothing=0.05)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x), y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(loss.item())
    print("losses", [round(v, 4) for v in losses])
    assert losses[-1] < losses[0]
