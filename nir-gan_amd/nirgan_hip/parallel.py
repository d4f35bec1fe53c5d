"""Data parallelism: one process per GPU, tile batches sharded by rank, gradients averaged with RCCL
all-reduces over the flat gradient buffer of each network.

The reference gets this from Lightning's ``strategy: "ddp"`` (train.py:118-120,
configs/config_px2px.yaml:60-63): torch DDP broadcasts rank 0's weights when it wraps the module and
averages gradients over ranks during backward.  InstanceNorm has no cross-sample statistics and every
loss is a mean, so the mean of the per-rank gradients on equal shards equals the gradient on the
concatenated batch.

Two buckets per network (``begin`` / ``finish``): the backward plans call ``begin`` on the tail of the
flat gradient as soon as it is complete (the PatchGAN's two last layers = 8.4 of 11 MB after the first
backward launches; the generator's residual blocks + decoder = 28 of 31 MB before the encoder's backward)
and on the head at the end.  ``torch.distributed`` launches an async collective on RCCL's own stream, which
waits for the launch stream at the call and runs under the kernels issued after it; ``finish`` makes the
launch stream wait for the collectives (no host synchronisation).  Payloads are 11-62 MB of fp32 per
network against a >= 28 ms step: on the 8-GPU xGMI mesh (7 links x ~153 GB/s per GPU) a 31 MB all-reduce
is a few hundred microseconds, of which only the head bucket is exposed.
``backend='nccl'`` IS RCCL on ROCm; the same code runs over gloo on CPU for the tests.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torchrun / init_process_group)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._avg = dist.get_backend(group) == "nccl"          # ncclAvg: the division rides in the collective
        self._pending: List[tuple] = []
        self.exposed_events: Optional[list] = None             # bench.py: [(start, end)] HIP events around every finish()

    # ------------------------------------------------------------------ blocking form (one bucket)
    def all_reduce_mean(self, flat_grad: torch.Tensor) -> None:
        self.begin(flat_grad)
        self.finish()

    # ------------------------------------------------------------------ bucketed form
    def begin(self, part: torch.Tensor) -> None:
        """Start averaging ``part`` (a contiguous slice of a flat gradient) over the ranks; returns at once."""
        if self.world == 1 and not self._avg:
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        work = dist.all_reduce(part, op=op, group=self.group, async_op=True)
        self._pending.append((work, part))

    def finish(self) -> None:
        """Order everything issued after this call behind the collectives started with ``begin``."""
        timed = self.exposed_events is not None and self._pending and self._pending[0][1].is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for work, part in self._pending:
            work.wait()
            if not self._avg and self.world > 1:
                part.mul_(1.0 / self.world)
        self._pending.clear()
        if timed:
            e1.record()
            self.exposed_events.append((e0, e1))

    def broadcast_params(self, flat, src: int = 0) -> None:
        """Make every rank start from rank ``src``'s weights (DDP does this at wrap time).  ``flat``: a FlatParams (its
        version is bumped so that packed / transformed weight caches rebuild) or a plain tensor."""
        t = flat.flat if hasattr(flat, "flat") else flat
        if self.world > 1:
            dist.broadcast(t, src=src, group=self.group)
            if hasattr(flat, "touch"):
                flat.touch()


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous equal shards of the leading (tile) dimension (what DistributedSampler + DataLoader hand each rank)."""
    assert t.shape[0] % world == 0, "global batch must divide the number of ranks"
    n = t.shape[0] // world
    return t[rank * n:(rank + 1) * n].contiguous()
