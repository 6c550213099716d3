"""Data-parallel gradient exchange: bucketed, averaged all-reduce of a flat fp32 gradient buffer,
launched while the backward pass is still producing the rest of the buffer.

The reference is single-process (no DDP); north_star adds "data-parallel with
bucketed gradient all-reduce on RCCL over xGMI, overlapped with backward".  The flat buffer is
written front to back by `TrainEngine.backward`; `ready(flat, upto)` is called whenever elements
[0, upto) are final, and every time at least `bucket_bytes` of new gradients have accumulated the
bucket [done, upto) is exchanged; `finish` exchanges the tail and makes the compute stream wait for
every bucket of the step.

How a bucket is issued (`mode`; chosen from the process group's backend unless given or RVIT_DP_MODE is set):
  "async"  backend "nccl" (= RCCL; the product path) and CPU tensors: `all_reduce(chunk, async_op=True)` on the compute
           stream; the Work objects are waited for in `finish`.  With RCCL the collective runs on the process group's
           own (normal-priority) stream behind an event on the compute stream, so it overlaps the rest of the
           backward, and `wait()` is a stream-level dependency: the host never blocks.
  "staged" backend "gloo" with device tensors (only the one-GPU rehearsal, scripts/rehearse_dp.sh): the reducer
           stages the bucket through pinned host memory itself - device->host copy on the compute stream, host
           wait, all-reduce of the HOST tensor, host->device copy.  No overlap.  gloo's OWN device staging must not
           be used while kernels are queued: it waits for the caller's stream on a HIGH-priority pool stream
           (ProcessGroupGloo.cpp initializeStreamsEvents), and on this platform a high-priority stream waiting for an
           event behind queued normal-priority work never gets it (priority inversion; reproduced without
           torch.distributed or any librvit kernel by scripts/rehearse_prio.py).  That was round 1's
           stall.
  "side"   round 1's form: a synchronous `all_reduce` issued inside `torch.cuda.stream(side)` behind an event.
  "inline" synchronous `all_reduce` on the compute stream, no overlap.
           Both kept as diagnosis aids for the rehearsal scripts.
`presync=True` (RVIT_DP_PRESYNC=1) synchronises the compute stream on the host before a bucket is issued
(diagnosis aid: removes every queued-kernel dependency from the collective).

Averaging: the bucket is scaled by 1/world before the sum (sum of pre-scaled addends; identical on every rank).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of b bytes costs about
2*(7/8)*b/153 GB/s, i.e. 0.66 ms for the 58 MB of fp32 gradients of the default hybrid; 8 MB
buckets (0.09 ms each on the ring) keep each collective bandwidth- rather than latency-bound while
the default hybrid still exchanges in 5-6 pieces: after the encoder, then after every stem block
from layer4 down, the last one (layer1 + conv1, 0.6 MB) being all that is left exposed.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import os
import sys

import torch

_VERBOSE = bool(os.environ.get("RVIT_DP_VERBOSE"))
MODES = ("async", "side", "inline", "staged")


class BucketReducer:
    def __init__(self, total: int, process_group=None, bucket_bytes: int = 8 << 20, mode: Optional[str] = None,
                 presync: Optional[bool] = None, exchange_at_world_one: Optional[bool] = None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.dist = dist
        self.group = process_group if process_group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.total = int(total)
        self.bucket_elems = max(1, int(bucket_bytes) // 4)
        self.mode = mode if mode is not None else os.environ.get("RVIT_DP_MODE")   # None: by backend, at first use
        if self.mode is not None and self.mode not in MODES:
            raise ValueError(f"BucketReducer: mode {self.mode!r} not in {MODES}")
        self.presync = bool(int(os.environ.get("RVIT_DP_PRESYNC", "0"))) if presync is None else bool(presync)
        # test-only switch (RVIT_DP_FORCE=1): a world of one normally exchanges nothing; with it every bucket is issued
        # anyway, which puts the product issue path (async all_reduce of flat-buffer slices from the autograd thread,
        # behind queued kernels) through the real process group on ONE GPU (tests/test_dp_gpu.py, backend "nccl")
        force = bool(int(os.environ.get("RVIT_DP_FORCE", "0"))) if exchange_at_world_one is None else bool(exchange_at_world_one)
        self.active = self.world > 1 or force
        self.done = 0
        self.buckets: List[Tuple[int, int]] = []  # (start, end) of every bucket launched in this step
        self._works: list = []
        self._stream: Optional[torch.cuda.Stream] = None

    def _say(self, msg: str) -> None:
        if _VERBOSE:
            print(f"[dp rank {self.dist.get_rank(self.group)}] {msg}", file=sys.stderr, flush=True)

    def _launch(self, flat: torch.Tensor, start: int, end: int) -> None:
        if self.mode is None:
            backend = str(self.dist.get_backend(self.group)).lower()
            self.mode = "staged" if (flat.is_cuda and "nccl" not in backend) else "async"
        chunk = flat[start:end]
        self._say(f"all_reduce [{start}, {end}) mode={self.mode}")
        if flat.is_cuda and self.presync:
            torch.cuda.current_stream(flat.device).synchronize()
        if flat.is_cuda and self.mode == "side":
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(ev)
                chunk.mul_(1.0 / self.world)
                self.dist.all_reduce(chunk, op=self.dist.ReduceOp.SUM, group=self.group)
            chunk.record_stream(self._stream)
        elif flat.is_cuda and self.mode == "staged":
            host = self._pinned(end - start)
            host.copy_(chunk, non_blocking=True)
            torch.cuda.current_stream(flat.device).synchronize()
            host.mul_(1.0 / self.world)
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            chunk.copy_(host, non_blocking=True)
        elif self.mode == "inline":
            chunk.mul_(1.0 / self.world)
            self.dist.all_reduce(chunk, op=self.dist.ReduceOp.SUM, group=self.group)
        else:
            chunk.mul_(1.0 / self.world)
            self._works.append(self.dist.all_reduce(chunk, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._say(f"issued [{start}, {end})")
        self.buckets.append((start, end))
        self.done = end

    def _pinned(self, n: int) -> torch.Tensor:
        """One pinned host buffer per bucket of the step (a bucket's host->device copy may still be in flight
        when the next bucket is staged), kept across steps."""
        k = len(self.buckets)
        pool = self.__dict__.setdefault("_host", [])
        while len(pool) <= k:
            pool.append(None)
        if pool[k] is None or pool[k].numel() < n:
            pool[k] = torch.empty(n, dtype=torch.float32, pin_memory=True)
        return pool[k][:n]

    def ready(self, flat: torch.Tensor, upto: int) -> None:
        """Elements [0, upto) of `flat` are final; launch a bucket if enough has accumulated."""
        if not self.active:
            return
        if upto - self.done >= self.bucket_elems:
            self._launch(flat, self.done, upto)

    def finish(self, flat: torch.Tensor, limit: Optional[int] = None) -> None:
        """Exchange what is left of [0, limit) (default: the whole buffer; a frozen stem leaves its tail out: every
        rank must pass the same limit) and wait for every bucket of the step."""
        if not self.active:
            return
        total = self.total if limit is None else min(int(limit), self.total)
        if self.done < total:
            self._launch(flat, self.done, total)
        for w in self._works:      # nccl: the compute stream waits for the group's stream; gloo: the host waits
            w.wait()
        self._works = []
        if flat.is_cuda and self._stream is not None and self.mode == "side":
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)
        self._say(f"step exchanged in {len(self.buckets)} buckets")
        self.done = 0

    def begin_step(self) -> None:
        self.done = 0
        self.buckets = []
        self._works = []
