"""Launch plans for the training step: a fixed sequence of C-ABI launches recorded once and replayed.

At the reference's own batch size (32, train_vit.py:505-515) a training step is bound by the host: ~320 launches at
~13 us of Python each (argument checks of the op wrappers, views, slices, module attribute look-ups) against ~5 us for
the ctypes call and hipLaunchKernel themselves.  The stem's forward and backward passes (~250 of those launches) take
no per-step argument at all - no dropout site, every operand in a persistent buffer - so the second time a pass runs
with the same operands its C-ABI calls are recorded (vit_amd/_lib.py CallRecorder: function + argument list), and from
then on the pass is that list, called in order.  Same launches, same arguments, same stream: the results are the same
bits; nothing is skipped, reordered or cached.

What keeps a plan honest:
  * the KEY (TrainEngine._plan_key) holds every pointer the launches can see that is not patched below: the buffer set,
    the optimiser's operand copies, the data pointers of all stem parameters and BatchNorm buffers, the workspaces, the
    stream, the engine's switches.  Any difference -> the pass runs through the op wrappers again (and is re-recorded).
  * two operands change from step to step and are PATCHED by address range: the fp32 input batch and the flat gradient
    buffer.  Every recorded pointer argument inside [base, base + bytes) is rebased when the base moves.
  * recording runs under a TorchDispatchMode that refuses the plan if any torch operator other than a view runs inside
    the recorded region (a torch kernel would be missing from the replay; a temporary would leave a dangling pointer).
  * profiling (ops.set_profiler) and RVIT_LAUNCH_PLAN=0 switch plans off.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _lib

# aten operators that only make views of (or read metadata from) existing tensors
_VIEW_OPS = {"aten.view.default", "aten.slice.Tensor", "aten.detach.default", "aten.select.int", "aten.alias.default",
             "aten._unsafe_view.default", "aten.as_strided.default", "aten.t.default", "aten.transpose.int",
             "aten.permute.default", "aten.reshape.default", "aten.unsqueeze.default", "aten.squeeze.dim",
             "aten.expand.default", "aten.narrow.default", "aten.unbind.int", "aten.split.Tensor",
             "aten.view_as.default", "aten.flatten.using_ints", "aten.unflatten.int", "aten.sym_size.int",
             "aten.sym_stride.int", "aten.sym_numel.default", "aten.sym_storage_offset.default",
             "aten.is_contiguous.default", "aten.is_same_size.default", "aten.lift_fresh.default"}


class _NoKernels(TorchDispatchMode):
    """Notes every torch operator that is not a view while a plan is recorded."""

    def __init__(self):
        super().__init__()
        self.offenders = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name not in _VIEW_OPS:
            self.offenders.append(name)
        return func(*args, **(kwargs or {}))


class LaunchPlan:
    __slots__ = ("calls", "bases", "patches", "n")

    def __init__(self, calls, ranges: Dict[str, tuple]):
        """calls: [[fn, [args], name], ...] from a CallRecorder; ranges: tag -> (base address, bytes) of the operands
        whose address may differ on the next step."""
        self.calls = calls
        self.n = len(calls)
        self.bases = {tag: base for tag, (base, _) in ranges.items()}
        self.patches = {tag: [] for tag in ranges}
        for _, args, _ in calls:
            for i, a in enumerate(args):
                if type(a) is int:
                    for tag, (base, nbytes) in ranges.items():
                        if base <= a < base + nbytes:
                            self.patches[tag].append((args, i, a - base))

    def run(self, bases: Dict[str, int]) -> None:
        for tag, base in bases.items():
            if base != self.bases[tag]:
                for args, i, off in self.patches[tag]:
                    args[i] = base + off
                self.bases[tag] = base
        for fn, args, name in self.calls:
            rc = fn(*args)
            if rc:
                _lib.check(rc, name)


class PlanSlot:
    """One recorded pass of an engine: its key, the objects the key's ids refer to (kept alive so an id cannot be
    reused), the plan or the reason it was refused, and what the pass returned."""
    __slots__ = ("key", "refs", "plan", "refused", "result", "seen")

    def __init__(self, key, refs):
        self.key, self.refs = key, refs
        self.plan: Optional[LaunchPlan] = None
        self.refused: Optional[str] = None
        self.result = None
        self.seen = 1


def run_planned(slots: dict, which: str, key, refs, ranges: Dict[str, tuple], fn: Callable, stats: dict):
    """Run the pass `fn` (C-ABI launches and tensor views only).  First sighting of `key`: through the op wrappers.
    Second: the same, recorded.  From then on: the recorded launches.  `key` None: plans are off for this call."""
    if key is None:
        return fn()
    slot = slots.get(which)
    if slot is None or slot.key != key:
        slots[which] = PlanSlot(key, refs)
        return fn()
    if slot.plan is not None:
        slot.plan.run({tag: base for tag, (base, _) in ranges.items()})
        stats[which] = stats.get(which, 0) + 1
        return slot.result
    if slot.refused is not None:
        return fn()
    slot.seen += 1
    rec = _lib.record_calls(True)
    guard = _NoKernels()
    try:
        with guard:
            result = fn()
    finally:
        _lib.record_calls(False)
    if guard.offenders:
        slot.refused = "torch operators inside the pass: " + ", ".join(sorted(set(guard.offenders)))
    elif not rec.calls:
        slot.refused = "no launches recorded"
    else:
        slot.plan = LaunchPlan(rec.calls, ranges)
        slot.result = result
    return result
