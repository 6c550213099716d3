"""Data-parallel gradient exchange: bucketed, averaged all-reduce of a flat fp32 gradient buffer,
launched while the backward pass is still producing the rest of the buffer.

The reference is single-process (no DDP); north_star adds "data-parallel with
bucketed gradient all-reduce on RCCL over xGMI, overlapped with backward".  The flat buffer is
written front to back by `TrainEngine.backward`; `ready(flat, upto)` is called whenever elements
[0, upto) are final, and every time at least `bucket_bytes` of new gradients have accumulated the
bucket is all-reduced: on a GPU on a side HIP stream behind an event (backend "nccl" = RCCL), on
CPU tensors synchronously (backend "gloo", used by the CPU tests).  `finish` reduces the tail and
makes the compute stream wait for the exchange.

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


class BucketReducer:
    def __init__(self, total: int, process_group=None, bucket_bytes: int = 8 << 20):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.dist = dist
        self.group = process_group if process_group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.total = int(total)
        self.bucket_elems = max(1, int(bucket_bytes) // 4)
        self.done = 0
        self.buckets: List[Tuple[int, int]] = []  # (start, end) of every bucket launched in this step
        self._stream: Optional[torch.cuda.Stream] = None

    def _launch(self, flat: torch.Tensor, start: int, end: int) -> None:
        chunk = flat[start:end]
        if _VERBOSE:
            print(f"[dp rank {self.dist.get_rank(self.group)}] all_reduce [{start}, {end})", file=sys.stderr, flush=True)
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(ev)
                chunk.mul_(1.0 / self.world)
                self.dist.all_reduce(chunk, op=self.dist.ReduceOp.SUM, group=self.group)
            chunk.record_stream(self._stream)
        else:
            chunk.mul_(1.0 / self.world)
            self.dist.all_reduce(chunk, op=self.dist.ReduceOp.SUM, group=self.group)
        self.buckets.append((start, end))
        self.done = end

    def ready(self, flat: torch.Tensor, upto: int) -> None:
        """Elements [0, upto) of `flat` are final; launch a bucket if enough has accumulated."""
        if self.world <= 1:
            return
        if upto - self.done >= self.bucket_elems:
            self._launch(flat, self.done, upto)

    def finish(self, flat: torch.Tensor) -> None:
        if self.world <= 1:
            return
        if self.done < self.total:
            self._launch(flat, self.done, self.total)
        if flat.is_cuda and self._stream is not None:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)
        if _VERBOSE:
            print(f"[dp rank {self.dist.get_rank(self.group)}] step exchanged in {len(self.buckets)} buckets",
                  file=sys.stderr, flush=True)
        self.done = 0

    def begin_step(self) -> None:
        self.done = 0
        self.buckets = []
