"""One training step of the `backbone='mobilenetv3'` classifier on librvit (train-mode forward, loss, hand-scheduled
backward: vit_amd/mobile_trainer.py) against torch autograd on the fp32 CPU oracle with the same synthetic checkpoint,
inputs and labels (train_vit.py:935-939 on the reference's fastest backbone, train_vit.py:69).  dropout = 0 in both.
Parity of the timm stem itself is unpinned (timm is not installed: DESIGN.md section 3); this pins the HIP training
pass against the oracle's restatement of it.

Stated bounds (fixed at about 1.5x what was measured when the test was written; the test prints measured / bound):
loss within 2 %; encoder / head / patch-embed gradients: relative L2 <= 8 %, cosine >= 0.995; stem gradients, whose
BatchNorm + hard-swish / ReLU / hard-sigmoid chain amplifies bf16 storage of the 50 saved maps as the resnet stem's
does (tests/test_train_gpu.py): cosine >= STEM_COS and relative L2 <= STEM_REL per tensor against the fp32 oracle;
BatchNorm running statistics match; two runs of the step give the same bits (every reduction is ordered).
Where the stem distance comes from is measured in the test itself: the oracle with bf16 storage at the HIP path's storage
points (oracle/ref_bf16_train.py) sits as far from the fp32 oracle as the HIP gradients do (median relative L2 0.18
against 0.16), and the HIP gradients are as far from that emulation as from the fp32 oracle (two roundings of the same
50 maps at different points of the arithmetic flip different hard-swish / ReLU / hard-sigmoid branches): bf16 map storage,
not an arithmetic difference, sets these bounds, and a tighter per-tensor gate does not exist for this stem at B = 8."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CFG = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=2, dropout=0.0, backbone="mobilenetv3")
STEM_COS, STEM_REL = 0.90, 0.45
ENC_REL = 0.08
NULL_ABS = 4e-3
EMU_FACTOR = 2.0
EMU_REL, EMU_COS = 0.45, 0.90   # HIP stem gradients against the bf16-storage emulation of the oracle, per tensor


def _pair(dev, seed=4):
    from oracle.ref_model import CNNViTHybridRef
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import synth_state_dict
    m = CNNViTHybrid(**CFG)
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed)
    m.load_state_dict(sd)
    ref = CNNViTHybridRef(**CFG)
    ref.load_state_dict(sd)
    return m.to(dev).train(), ref.train()


def test_mobilenetv3_training_step_matches_autograd(dev):
    from vit_amd.loss import cross_entropy
    from vit_amd.synth import micro_doppler_batch
    torch.set_num_threads(16)
    B = 8
    m, ref = _pair(dev)
    x = micro_doppler_batch(B, seed=21)
    y = torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(3))
    loss_ref = F.cross_entropy(ref(x), y, label_smoothing=0.05)
    loss_ref.backward()
    loss = cross_entropy(m(x.to(dev)), y.to(dev), label_smoothing=0.05)
    loss.backward()
    print(f"\nloss hip {loss.item():.5f} ref {loss_ref.item():.5f}")
    assert abs(loss.item() - loss_ref.item()) <= 0.02 * max(1.0, abs(loss_ref.item()))
    gref = dict(ref.named_parameters())
    rows, bad, null = [], [], []
    # a per-channel constant added in front of a convolution that feeds a train-mode BatchNorm is removed by that
    # BatchNorm's mean: the bias of a block's LAST BatchNorm has an exactly zero gradient wherever every consumer is
    # conv -> BatchNorm (the oracle leaves ~1e-9 of round-off there).  Those tensors are held to "small", not to a ratio.
    gmax = max(gref[n].grad.norm().item() for n, _ in m.named_parameters() if n.startswith("cnn_backbone."))
    for n, p in m.named_parameters():
        assert p.grad is not None, f"no gradient for {n}"
        g, r = p.grad.float().cpu(), gref[n].grad
        assert g.shape == r.shape, n
        if n.startswith("cnn_backbone.") and r.norm().item() < 1e-5 * gmax:
            null.append((g.norm().item() / gmax, n))
            if g.norm().item() > NULL_ABS * gmax:
                bad.append((n, "null gradient", g.norm().item() / gmax))
            continue
        rel = ((g - r).norm() / r.norm().clamp_min(1e-12)).item()
        cos = F.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
        rows.append((rel, cos, n))
        if n.startswith("cnn_backbone."):
            if rel > STEM_REL or cos < STEM_COS:
                bad.append((n, rel, cos))
        elif rel > ENC_REL or cos < 0.995:
            bad.append((n, rel, cos))
    rows.sort(reverse=True)
    for rel, cos, n in rows[:8]:
        print(f"   {n:55s} rel {rel:.4f} cos {cos:.5f}")
    stem = [r for r in rows if r[2].startswith("cnn_backbone.")]
    print(f"   stem: worst rel {max(r[0] for r in stem):.4f} / {STEM_REL}, worst cos {min(r[1] for r in stem):.5f} / {STEM_COS}; "
          f"median rel {sorted(r[0] for r in stem)[len(stem) // 2]:.4f}")
    # Sensitivity probe (not a pass / fail bound by itself): the same oracle with bf16 storage where the HIP path stores
    # bf16 (oracle/ref_bf16_train.py) moves away from the fp32 oracle by about as much as the HIP gradients do - the
    # distance above is what bf16 maps cost on this network, not an arithmetic difference.  Held loosely: the median
    # stem distance of the HIP path stays within EMU_FACTOR x the emulation's.
    from oracle.ref_bf16_train import emulate_hip_training_storage
    from oracle.ref_model import CNNViTHybridRef
    emu = CNNViTHybridRef(**CFG)
    emu.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    # (the step above updated the oracle's running statistics only: train-mode BatchNorm does not read them)
    emulate_hip_training_storage(emu)
    emu.train()
    F.cross_entropy(emu(x), y, label_smoothing=0.05).backward()
    erel, vs_emu = [], []
    ghip = {n: p.grad.float().cpu() for n, p in m.named_parameters()}
    for n, p in emu.named_parameters():
        r = gref[n].grad
        if n.startswith("cnn_backbone.") and r.norm().item() >= 1e-5 * gmax:
            erel.append(((p.grad - r).norm() / r.norm()).item())
            vs_emu.append((((ghip[n] - p.grad).norm() / p.grad.norm().clamp_min(1e-12)).item(),
                           F.cosine_similarity(ghip[n].flatten(), p.grad.flatten(), dim=0).item(), n))
    vs_emu.sort(reverse=True)
    for rel, cos, n in vs_emu[:4]:
        print(f"   vs bf16-storage oracle {n:45s} rel {rel:.4f} / {EMU_REL}  cos {cos:.5f} / {EMU_COS}")
    assert all(rel <= EMU_REL and cos >= EMU_COS for rel, cos, _ in vs_emu), vs_emu[:4]
    med_hip = sorted(r[0] for r in stem)[len(stem) // 2]
    med_emu = sorted(erel)[len(erel) // 2]
    print(f"   bf16-storage emulation of the oracle vs the fp32 oracle: stem median rel {med_emu:.4f}, worst {max(erel):.4f} "
          f"(HIP: median {med_hip:.4f}, worst {max(r[0] for r in stem):.4f}); factor {med_hip / max(med_emu, 1e-9):.2f} / {EMU_FACTOR}")
    assert med_hip <= EMU_FACTOR * med_emu, (med_hip, med_emu)
    null.sort(reverse=True)
    print(f"   {len(null)} tensors with a zero reference gradient: largest HIP norm / largest stem gradient norm "
          f"{null[0][0] if null else 0:.2e} ({null[0][1] if null else '-'}) / {NULL_ABS}")
    assert not bad, bad[:6]
    # BatchNorm running statistics (momentum 0.1, unbiased variance) and step counters
    bufs = dict(ref.named_buffers())
    for n, b in m.named_buffers():
        if n.endswith("running_mean") or n.endswith("running_var"):
            r = bufs[n]
            assert torch.allclose(b.float().cpu(), r, rtol=0.05, atol=0.02 * max(1.0, r.abs().max().item())), n
        elif n.endswith("num_batches_tracked"):
            assert int(b.item()) == int(bufs[n].item()), (n, int(b.item()), int(bufs[n].item()))


def test_mobilenetv3_training_step_is_bit_reproducible(dev):
    from vit_amd.loss import cross_entropy
    from vit_amd.synth import micro_doppler_batch
    x = micro_doppler_batch(6, seed=5).to(dev)
    y = torch.randint(0, 11, (6,), generator=torch.Generator().manual_seed(1)).to(dev)
    outs = []
    for _ in range(2):
        m, _ = _pair(dev, seed=9)
        loss = cross_entropy(m(x), y, label_smoothing=0.05)
        loss.backward()
        outs.append((loss.detach().clone(), [p.grad.clone() for p in m.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


def test_mobilenetv3_trains_with_the_fused_optimizer(dev):
    """A few FusedAdamW steps (flat parameter / gradient buffers, bf16 operand mirror kept by the optimiser kernel) lower
    the loss on a fixed batch: the whole train_vit.py:935-946 loop on this backbone."""
    from vit_amd.loss import cross_entropy
    from vit_amd.optim import FusedAdamW
    from vit_amd.synth import micro_doppler_batch
    m, _ = _pair(dev, seed=2)
    opt = FusedAdamW(m, lr=2e-3, weight_decay=0.01, max_grad_norm=1.0)
    x = micro_doppler_batch(8, seed=8).to(dev)
    y = (torch.arange(8) % 11).to(dev)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = cross_entropy(m(x), y, label_smoothing=0.05)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print("\nlosses", [round(v, 3) for v in losses])
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0] - 0.2
