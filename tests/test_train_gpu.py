"""One training step on librvit (forward in train() mode, loss, backward) against torch autograd on
the fp32 CPU oracle with the same synthetic checkpoint, inputs and labels (north_star: "gradients
stay within tolerance across one training step").  dropout = 0 in both (torch's Philox dropout
stream cannot be reproduced by another kernel).

Stated tolerances (fixed; the test prints measured / bound):
  * loss within LOSS_TOL of the fp32 oracle's;
  * encoder / head / patch-embed tensors: relative L2 error against the FP32 oracle <= grad_rel (5-6 %), cosine
    >= 0.995;
  * stem tensors of the hybrid: against the fp32 oracle only cosine >= 0.90 can be asked - through the 20-conv
    BatchNorm/ReLU stem of this random-weight network, rounding the FORWARD maps to bf16 alone moves the fp32
    oracle's own gradients by 15-33 % (measured on the CPU: forward rounding 33 %, gradient rounding 1.7 %, and it
    does not shrink with the batch: 33 / 31 / 27 % at B = 8 / 32 / 64).  The sharp gate is therefore against the
    oracle run with bf16 storage at exactly the points where the HIP path stores bf16
    (oracle/ref_bf16_train.py emulate_hip_training_storage): cosine >= STEM_COS_VS_EMULATION, relative L2 <=
    STEM_REL_VS_EMULATION per tensor.  What is left between the two is the summation order inside the convolutions
    (one-ulp flips of ~0.1 % of the stored activations, tests/test_model_gpu.py) amplified the same way: a ReLU
    whose pre-activation sits within a bf16 ulp of zero (0.3 % of the elements of a zero-mean map) flips, each flip
    switches one gradient path on or off (~5 % relative L2 per layer, incoherent over 17 ReLU layers).  The
    kernel-level backward tests (tests/test_backward_gpu.py: every conv / BatchNorm / pool backward against torch
    autograd on identical saved operands, where no flip can occur) are the tight ones, 1-2 %.
  * the SHARP stem gate (round 3, test_stem_backward_teacher_forced): the bf16-storage oracle teacher-forced to the conv
    outputs the HIP forward stored, so that no ReLU / pool decision can differ: every stem gradient within 3 % relative
    L2 (measured <= 1.94 %, median 1.2 %), cosine >= 0.9995;
  * BatchNorm running statistics match the oracle's.
Since round 2 the BatchNorm reductions are ordered, so the HIP step is run-to-run reproducible and these numbers are
the same on every run.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LOSS_TOL = 0.02
STEM_REL_VS_EMULATION = 0.30   # measured 0.24 (layer1.0.bn2.weight); most stem tensors 0.15-0.20
STEM_COS_VS_EMULATION = 0.97   # measured 0.973 at worst


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
    # BASELINE configs[3] at its own size: ViT-B/16-scale (E768 / H12 / L12, 86 M parameters), one training step
    ("vit-B-scale", dict(num_classes=11, embed_dim=768, num_heads=12, num_layers=12, dropout=0.0, backbone="patch16"),
     2, 0.06, 0.995),
    # head dims other than 64 (csrc/attn_dh.hip forward + backward inside a whole training step)
    ("patch16-H8-dh32", dict(num_classes=11, embed_dim=256, num_heads=8, num_layers=3, dropout=0.0, backbone="patch16"),
     3, 0.06, 0.995),
    ("patch16-H2-dh128", dict(num_classes=11, embed_dim=256, num_heads=2, num_layers=3, dropout=0.0, backbone="patch16"),
     3, 0.06, 0.995),
    ("patch16-H4-dh48", dict(num_classes=11, embed_dim=192, num_heads=4, num_layers=2, dropout=0.0, backbone="patch16"),
     3, 0.06, 0.995),
]


@pytest.mark.parametrize("name,cfg,B,grad_rel,grad_cos", CASES, ids=[c[0] for c in CASES])
def test_one_training_step_matches_autograd(dev, name, cfg, B, grad_rel, grad_cos):
    check_training_step(dev, name, cfg, B, grad_rel, grad_cos)


def check_training_step(dev, name, cfg, B, grad_rel, grad_cos):
    """Shared with tests/test_benched_path_gpu.py (the same comparison at the batch size bench.py times)."""
    from vit_amd.loss import cross_entropy
    from vit_amd.synth import micro_doppler_batch
    torch.set_num_threads(16)
    m, ref = _pair(cfg, seed=4, dev=dev)
    x = micro_doppler_batch(B, seed=21)
    y = torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(3))
    loss_ref = F.cross_entropy(ref(x), y, label_smoothing=0.05)
    loss_ref.backward()
    # the oracle with bf16 storage where the HIP path stores bf16 (stem gate)
    gemu = {}
    if cfg["backbone"] != "patch16":
        from oracle.ref_bf16_train import emulate_hip_training_storage
        _, ref2 = _pair(cfg, seed=4, dev=dev)
        emulate_hip_training_storage(ref2)
        F.cross_entropy(ref2(x), y, label_smoothing=0.05).backward()
        gemu = {n: p.grad for n, p in ref2.named_parameters()}
    logits = m(x.to(dev))
    loss = cross_entropy(logits, y.to(dev), label_smoothing=0.05)
    loss.backward()
    print(f"\n[{name}] loss hip {loss.item():.5f} ref {loss_ref.item():.5f}")
    assert abs(loss.item() - loss_ref.item()) <= LOSS_TOL * max(1.0, abs(loss_ref.item()))
    gref = dict(ref.named_parameters())
    worst, worst_emu, bad = [], [], []
    for n, p in m.named_parameters():
        assert p.grad is not None, f"no gradient for {n}"
        g, r = p.grad.float().cpu(), gref[n].grad
        assert g.shape == r.shape
        rel = ((g - r).norm() / r.norm().clamp_min(1e-12)).item()
        cos = F.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
        worst.append((rel, cos, n))
        if n.startswith("cnn_backbone."):
            e = gemu[n]
            rel_e = ((g - e).norm() / e.norm().clamp_min(1e-12)).item()
            cos_e = F.cosine_similarity(g.flatten(), e.flatten(), dim=0).item()
            worst_emu.append((rel_e, cos_e, n))
            if rel_e > STEM_REL_VS_EMULATION or cos_e < STEM_COS_VS_EMULATION or cos < grad_cos:
                bad.append((n, rel_e, cos_e, cos))
        elif rel > grad_rel or cos < 0.995:
            bad.append((n, rel, cos))
    worst.sort(reverse=True)
    for rel, cos, n in worst[:4]:
        print(f"   vs fp32 oracle      {n:50s} rel {rel:.4f} cos {cos:.5f}")
    if worst_emu:
        worst_emu.sort(reverse=True)
        for rel, cos, n in worst_emu[:4]:
            print(f"   vs bf16-storage oracle {n:47s} rel {rel:.4f} / {STEM_REL_VS_EMULATION}  cos {cos:.5f} / {STEM_COS_VS_EMULATION}")
    enc = [w for w in worst if not w[2].startswith("cnn_backbone.")]
    print(f"   encoder/head worst vs fp32: rel {max(w[0] for w in enc):.4f} / {grad_rel}")
    assert not bad, f"{name}: gradients out of tolerance: {bad[:5]}"
    if cfg["backbone"] != "patch16":
        sd_h, sd_r = m.state_dict(), ref.state_dict()
        for k in sd_r:
            if k.endswith("running_mean") or k.endswith("running_var"):
                assert torch.allclose(sd_h[k].cpu(), sd_r[k], rtol=0.03, atol=0.02), k
            if k.endswith("num_batches_tracked"):
                assert int(sd_h[k]) == int(sd_r[k])


STEM_REL_TEACHER_FORCED = 0.03    # measured 0.0194 at worst (cnn_backbone.bn1.weight), median 0.0118 over the 60 stem tensors
STEM_COS_TEACHER_FORCED = 0.9995  # measured 0.99983 at worst


def test_stem_backward_teacher_forced(dev):
    """The sharp end-to-end gate on the stem's hand-scheduled backward: the bf16-storage oracle is
    teacher-forced to the conv outputs the HIP forward stored (oracle/ref_bf16_train.force_stem_conv_outputs), so its
    BatchNorm statistics, ReLU masks and max-pool argmax taps coincide with the HIP path's and no decision can flip
    between the two.  What is left is the gradient arithmetic: 20 convolution input / weight gradients, 20 BatchNorm
    backward passes, the residual sums and the pool backward, chained as vit_amd/trainer.py schedules them.  Every stem
    parameter gradient must then agree within STEM_REL_TEACHER_FORCED relative L2 (fixed, 1.5 x measured), against
    0.30 for the free-running comparison above."""
    from oracle.ref_bf16_train import emulate_hip_training_storage, force_stem_conv_outputs, stem_convs_in_order
    from vit_amd.loss import cross_entropy
    from vit_amd.synth import micro_doppler_batch
    torch.set_num_threads(16)
    name, cfg, B, _, _ = CASES[2]
    m, ref = _pair(cfg, seed=4, dev=dev)
    x = micro_doppler_batch(B, seed=21)
    y = torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(3))
    loss = cross_entropy(m(x.to(dev)), y.to(dev), label_smoothing=0.05)
    loss.backward()
    b = m.train_engine.buffers(B, dev)
    nchw = lambda t: t.float().permute(0, 3, 1, 2).contiguous().cpu()  # noqa: E731
    saved = [b["c1_y"]]
    for d in b["blocks"]:
        saved.append(d["y1"])
        if "yd" in d:
            saved.append(d["yd"])
        saved.append(d["y2"])
    names = [n for n, _ in stem_convs_in_order(ref)]
    assert len(names) == len(saved) == 20
    forced = {n: nchw(t) for n, t in zip(names, saved)}
    force_stem_conv_outputs(ref, forced)
    emulate_hip_training_storage(ref)
    loss_ref = F.cross_entropy(ref(x), y, label_smoothing=0.05)
    loss_ref.backward()
    print(f"\n[teacher-forced stem] loss hip {loss.item():.5f} oracle {loss_ref.item():.5f}")
    assert abs(loss.item() - loss_ref.item()) <= 0.01 * max(1.0, abs(loss_ref.item()))
    gref = dict(ref.named_parameters())
    rows = []
    for n, p in m.named_parameters():
        if not n.startswith("cnn_backbone."):
            continue
        g, r = p.grad.float().cpu(), gref[n].grad
        rel = ((g - r).norm() / r.norm().clamp_min(1e-12)).item()
        cos = F.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
        rows.append((rel, cos, n))
    rows.sort(reverse=True)
    for rel, cos, n in rows[:6]:
        print(f"   {n:50s} rel {rel:.4f} / {STEM_REL_TEACHER_FORCED}  cos {cos:.6f}")
    print(f"   median rel {rows[len(rows) // 2][0]:.4f} over {len(rows)} stem tensors")
    bad = [(n, rel, cos) for rel, cos, n in rows if rel > STEM_REL_TEACHER_FORCED or cos < STEM_COS_TEACHER_FORCED]
    assert not bad, bad[:5]


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


def test_frozen_backbone_skips_the_stem(dev):
    """freeze_backbone / unfreeze_backbone of the reference (train_vit.py:310-318, 918-923): with every cnn_backbone
    parameter at requires_grad=False the stem backward is skipped - its gradients stay None, every other gradient is
    what the unfrozen step gives - and FusedAdamW steps only the unfrozen range: stem parameters (and their bf16
    operand copies) keep their bits, no weight decay touches them; BatchNorm running statistics still move (train()
    mode, as in the reference).  Unfreezing restores the full step; a partly frozen stem is refused."""
    from vit_amd.loss import cross_entropy
    from vit_amd.optim import FusedAdamW
    from vit_amd.synth import micro_doppler_batch
    name, cfg, B, _, _ = CASES[2]
    m, _ = _pair(cfg, seed=6, dev=dev)
    x = micro_doppler_batch(B, seed=2).to(dev)
    y = torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(5)).to(dev)

    def grads():
        for p in m.parameters():
            p.grad = None
        cross_entropy(m(x), y, label_smoothing=0.05).backward()
        return {n: (None if p.grad is None else p.grad.clone()) for n, p in m.named_parameters()}

    full = grads()
    for p in m.cnn_backbone.parameters():
        p.requires_grad = False
    rm0 = m.cnn_backbone.bn1.running_mean.clone()
    frozen = grads()
    assert not torch.equal(rm0, m.cnn_backbone.bn1.running_mean)
    for n, g in frozen.items():
        if n.startswith("cnn_backbone."):
            assert g is None, n
        else:
            rel = ((g - full[n]).norm() / full[n].norm().clamp_min(1e-20)).item()
            assert rel < 1e-5, (n, rel)
    # optimiser: only the unfrozen range moves
    opt = FusedAdamW(m, lr=1e-3, weight_decay=0.05, max_grad_norm=1.0)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    mirror0 = opt.mirror.bf16.clone()
    grads()
    opt.step()
    start = m.train_engine.stem_start()
    assert torch.equal(opt.mirror.bf16[start:], mirror0[start:]) and not torch.equal(opt.mirror.bf16[:start], mirror0[:start])
    for n, p in m.named_parameters():
        if n.startswith("cnn_backbone."):
            assert torch.equal(p.detach(), before[n]), n
        elif p.numel() > 16:
            assert not torch.equal(p.detach(), before[n]), n
    # grad-norm clipping saw the unfrozen gradients only
    unfrozen_norm = torch.sqrt(sum((p.grad.float() ** 2).sum() for n, p in m.named_parameters()
                                   if not n.startswith("cnn_backbone.")))
    assert torch.allclose(opt.grad_norm(), unfrozen_norm, rtol=1e-4)
    # partly frozen: refused
    next(iter(m.cnn_backbone.parameters())).requires_grad = True
    with pytest.raises(RuntimeError):
        grads()
    # unfreeze: the stem trains again
    for p in m.cnn_backbone.parameters():
        p.requires_grad = True
    again = grads()
    assert all(g is not None for g in again.values())
    opt.step()
    assert not torch.equal(m.cnn_backbone.conv1.weight.detach(), before["cnn_backbone.conv1.weight"])
