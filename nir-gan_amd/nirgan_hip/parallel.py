"""Data parallelism: one process per GPU, tile batches sharded by rank, gradients averaged
with ONE RCCL all-reduce per network per optimizer step over the flat gradient buffer.

The reference gets this from Lightning's ``strategy: "ddp"`` (train.py:118-120,
configs/config_px2px.yaml:60-63): torch DDP averages gradients over ranks during backward.
InstanceNorm has no cross-sample statistics and every loss is a mean, so the mean of the
per-rank gradients on equal shards equals the gradient on the concatenated batch.
Payloads are small (11 MB for D, 31-62 MB for G in fp32) against a >= 50 ms step, so one
bucket per network is used; on the 8-GPU xGMI mesh RCCL picks the algorithm.
``backend='nccl'`` IS RCCL on ROCm; the same code runs over gloo on CPU for the tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torchrun / init_process_group)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_reduce_mean(self, flat_grad: torch.Tensor) -> None:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        if self.world > 1:
            flat_grad.mul_(1.0 / self.world)

    def broadcast_params(self, flat_params: torch.Tensor, src: int = 0) -> None:
        """Make every rank start from rank ``src``'s weights (DDP does this at wrap time)."""
        if self.world > 1:
            dist.broadcast(flat_params, src=src, group=self.group)


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous equal shards of the leading (tile) dimension."""
    assert t.shape[0] % world == 0, "global batch must divide the number of ranks"
    n = t.shape[0] // world
    return t[rank * n:(rank + 1) * n].contiguous()
