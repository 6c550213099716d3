"""Host logic of the launch plans (vit_amd/plan.py, vit_amd/_lib.py CallRecorder) without a GPU: what is recorded, how
the moving operands are rebased, when a pass is recorded / replayed / refused.  The one entry point called here is
rvit_cast_bf16 with n = 0, which checks its arguments and returns before any HIP call (csrc/optim.hip); the bit-equality
of planned training steps is tests/test_launch_plan_gpu.py."""
import ctypes
import threading

import numpy as np
import pytest
import torch

from vit_amd import _lib
from vit_amd.plan import LaunchPlan, run_planned


def _aligned(nbytes=256):
    a = np.zeros(nbytes + 64, dtype=np.uint8)
    p = a.ctypes.data
    off = (-p) % 64
    return a, p + off


def test_recorder_keeps_launches_only_and_only_on_its_thread():
    lib = _lib.load()
    keep, px = _aligned()
    keep2, po = _aligned()
    rec = _lib.record_calls(True)
    try:
        assert _lib.load() is rec
        seen = []
        t = threading.Thread(target=lambda: seen.append(_lib.load()))
        t.start()
        t.join()
        assert seen[0] is lib                       # another thread still gets the library itself
        assert _lib.load().rvit_version() == _lib.ABI_VERSION          # no stream argument: a query, not recorded
        assert _lib.load().rvit_cast_bf16(px, po, 0, None) == 0
        assert _lib.load().rvit_cast_bf16(px + 64, po, 0, None) == 0
    finally:
        _lib.record_calls(False)
    assert _lib.load() is lib
    assert [c[2] for c in rec.calls] == ["rvit_cast_bf16", "rvit_cast_bf16"]
    assert rec.calls[0][1] == [px, po, 0, None] and rec.calls[1][1][0] == px + 64


def test_plan_rebases_by_address_range_and_checks_status():
    log = []

    def fake(*a):
        log.append(a)
        return 0
    base, out = 0x7F0000001000, 0x7F0000800000
    calls = [[fake, [base + 128, out, 7, None], "one"], [fake, [out, base, 3.5, None], "two"], [fake, [base + 4096, 1, 2, None], "three"]]
    plan = LaunchPlan(calls, {"x": (base, 4096)})
    assert sorted((i, off) for _, i, off in plan.patches["x"]) == [(0, 128), (1, 0)]     # base + 4096 is outside [base, base + 4096)
    plan.run({"x": base})
    assert log[0] == (base + 128, out, 7, None)
    nb = 0x7E0000000000
    plan.run({"x": nb})
    plan.run({"x": nb})
    assert log[3] == (nb + 128, out, 7, None) and log[4] == (out, nb, 3.5, None) and log[5][0] == base + 4096
    assert log[6:] == log[3:6]
    # a failing call raises through _lib.check, with the entry point's name
    bad = LaunchPlan([[lambda *a: -1, [1], "rvit_bad"]], {})
    with pytest.raises(_lib.RvitError, match="rvit_bad"):
        bad.run({})


def test_pass_is_recorded_on_second_sighting_then_replayed():
    keep, px = _aligned()
    keep2, po = _aligned()
    slots, stats, ran = {}, {}, []

    def pass_(x):
        ran.append(x)
        _lib.check(_lib.load().rvit_cast_bf16(x, po, 0, None), "rvit_cast_bf16")
        return "result"
    key, refs = ("k", 1), (keep,)
    for i in range(5):
        x = px + 64 * (i % 2)                       # the operand moves inside its range
        assert run_planned(slots, "p", key, refs, {"x": (x, 64)}, lambda: pass_(x), stats) == "result"
    assert len(ran) == 2 and stats == {"p": 3}      # first sighting, recording, then three replays
    plan = slots["p"].plan
    assert plan.n == 1 and plan.calls[0][1][0] == px and plan.bases["x"] == px      # rebased to the last input
    # another key: back to the wrappers (first sighting of the new key), the old plan is dropped
    run_planned(slots, "p", ("k", 2), refs, {"x": (px, 64)}, lambda: pass_(px), stats)
    assert len(ran) == 3 and slots["p"].plan is None and slots["p"].key == ("k", 2)
    # no key: plans are off for the call
    run_planned(slots, "p", None, None, {}, lambda: pass_(px), stats)
    assert len(ran) == 4 and slots["p"].key == ("k", 2)


def test_pass_with_a_torch_kernel_inside_is_refused():
    keep, px = _aligned()
    slots, stats, ran = {}, {}, []
    t = torch.zeros(4)

    def pass_():
        ran.append(1)
        _lib.load().rvit_cast_bf16(px, px, 0, None)
        v = t[1:3].view(2)          # views are fine
        v.add_(1.0)                 # a kernel is not: it would be missing from the replay
    for _ in range(4):
        run_planned(slots, "p", "key", (), {}, pass_, stats)
    assert len(ran) == 4 and not stats
    assert slots["p"].plan is None and "aten.add_.Tensor" in slots["p"].refused
    # ... and one without any launch
    slots2 = {}
    for _ in range(3):
        run_planned(slots2, "q", "key", (), {}, lambda: None, stats)
    assert slots2["q"].refused == "no launches recorded"
