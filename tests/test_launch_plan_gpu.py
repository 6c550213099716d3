"""Launch plans of the training step (vit_amd/plan.py): the stem's forward and backward passes replayed from recorded
C-ABI calls must leave the same BITS in every parameter, optimiser state and BatchNorm buffer as the same steps through
the op wrappers (the reductions are ordered, so two runs of a step are bit-reproducible and any difference is an error
of the plan: a stale pointer, a missed launch, a missed patch of the input or gradient address).

The schedules below move everything a plan must notice or patch: a fresh input tensor at a fresh address on every
step (earlier batches are kept alive), batch-size changes and returns (new buffer sets), a frozen and unfrozen
backbone (train_vit.py:310-318), an evaluation forward in between, parameters edited behind the optimiser's back
(load_state_dict), the profiler switched on for a step, a BatchNorm momentum / eps changed by hand.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1)


def _run(dev, backbone, schedule, plans: bool):
    from vit_amd import ops
    from vit_amd.loss import CrossEntropyLoss
    from vit_amd.model import CNNViTHybrid
    from vit_amd.optim import FusedAdamW
    from vit_amd.synth import micro_doppler_batch, synth_state_dict
    torch.manual_seed(1234)   # the dropout keys of every step come from torch's CPU generator
    m = CNNViTHybrid(**CFG, backbone=backbone)
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=3)
    m.load_state_dict(sd)
    m = m.to(dev).train()
    m.train_engine.launch_plans = plans
    opt = FusedAdamW(m, lr=3e-4, weight_decay=0.05, max_grad_norm=1.0)
    crit = CrossEntropyLoss(label_smoothing=0.05)
    keep, losses = [], []
    for i, (B, what) in enumerate(schedule):
        if what == "freeze":
            for p in m.cnn_backbone.parameters():
                p.requires_grad = False
        elif what == "unfreeze":
            for p in m.cnn_backbone.parameters():
                p.requires_grad = True
        elif what == "eval":
            m.eval()
            with torch.no_grad():
                m(micro_doppler_batch(2, seed=77).to(dev))
            m.train()
        elif what == "reload":
            with torch.no_grad():
                for p in m.parameters():
                    p.mul_(0.999)
        elif what == "profile":
            ops.set_profiler([])
        elif what == "momentum":   # a scalar that is baked into the recorded arguments
            m.cnn_backbone.bn1.momentum = 0.3
            m.cnn_backbone.layer3["0"].bn2.eps = 2e-5
        x = micro_doppler_batch(B, seed=100 + i).to(dev)
        y = torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(i)).to(dev)
        keep.append(x)        # the next batch cannot land on this one's address
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.detach().clone())
        if what == "profile":
            ops.set_profiler(None)
    torch.cuda.synchronize()
    state = {n: t.detach().clone() for n, t in m.state_dict().items()}
    return state, torch.stack(losses), dict(m.train_engine.plan_replays), m.train_engine


SCHEDULES = {
    "steady": [(6, None)] * 8,
    "batch-changes": [(6, None)] * 5 + [(4, None)] * 5 + [(6, None)] * 5,
    "freeze-unfreeze": [(5, None)] * 4 + [(5, "freeze")] + [(5, None)] * 4 + [(5, "unfreeze")] + [(5, None)] * 5,
    "eval-reload-profile": [(6, None)] * 5 + [(6, "eval")] + [(6, None)] * 2 + [(6, "reload")] + [(6, None)] * 4
                           + [(6, "profile")] + [(6, None)] * 2 + [(6, "momentum")] + [(6, None)] * 4,
}


@pytest.mark.parametrize("backbone,name", [("resnet18", "steady"), ("resnet18", "batch-changes"),
                                           ("resnet18", "freeze-unfreeze"), ("resnet18", "eval-reload-profile"),
                                           ("resnet34", "steady")])
def test_planned_steps_leave_the_same_bits(dev, backbone, name):
    sched = SCHEDULES[name]
    ref_state, ref_losses, ref_replays, _ = _run(dev, backbone, sched, plans=False)
    assert not ref_replays
    state, losses, replays, eng = _run(dev, backbone, sched, plans=True)
    print(name, "replays", replays, "refused", {k: s.refused for k, s in eng._plans.items()})
    for slot in eng._plans.values():
        assert slot.refused is None, slot.refused
    assert replays.get("stem_forward", 0) >= 2 and replays.get("stem_backward", 0) >= 2, replays
    assert torch.equal(losses, ref_losses), (losses - ref_losses).abs().max().item()
    bad = [n for n in ref_state if not torch.equal(state[n], ref_state[n])]
    assert not bad, bad[:8]


def test_plan_lists_the_launches_of_the_pass(dev):
    """The recorded list is what the op wrappers launched: entry points only, every one with the stream it was recorded
    on, the input batch and the gradient buffer found by address range."""
    from vit_amd import ops
    _, _, replays, eng = _run(dev, "resnet18", [(4, None)] * 6, plans=True)
    fwd, bwd = eng._plans["stem_forward"].plan, eng._plans["stem_backward"].plan
    assert fwd is not None and bwd is not None
    names_f = [c[2] for c in fwd.calls]
    names_b = [c[2] for c in bwd.calls]
    print(len(names_f), "forward launches,", len(names_b), "backward launches")
    assert names_f[0] == "rvit_stem_conv1_stats_f32_fwd" and "rvit_bn_apply_pool" in names_f
    assert names_b[-1] == "rvit_stem_conv1_wgrad_f32" and "rvit_bn_bwd_apply_pool" in names_b
    assert 40 <= len(names_f) <= 80 and 60 <= len(names_b) <= 120   # C-ABI calls; several launch more than one kernel
    stream = ops._stream()
    assert all(c[1][-1] == stream for c in fwd.calls + bwd.calls)
    assert len(fwd.patches["xin"]) == 1 and len(bwd.patches["xin"]) == 1
    # every stem parameter's gradient segment: 19 conv weights (the 7x7 conv's goes through the padded dw1 scratch) + 20
    # BatchNorm weight / bias pairs
    assert len(bwd.patches["flat"]) == 59
