"""Data-parallel training over the GPUs of one node: one whole assembly graph per GPU, one
process per GPU, a single RCCL all-reduce of one flat fp32 gradient buffer per step.

The reference has no distributed code at all (single device string, hyperparameters.py:25);
its training loop takes one optimizer step per graph (train.py:238-258).  Under DP the W graphs
of a step contribute the MEAN of their gradients.  BatchNorm statistics
are per graph by construction, so there is no data-path collective: the only exchange is the
gradient (826,033 floats = 3.3 MB at H=128/L=8), latency-bound on xGMI."""
from __future__ import annotations

import os
from typing import Iterable

import torch
import torch.distributed as dist

__all__ = ["init_process_group", "FlatGradients", "shard_graphs"]


def init_process_group(backend: str | None = None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run).
    backend 'nccl' is RCCL on ROCm; 'gloo' for the CPU tests."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL needs it)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl" and "GNM_BENCH_DEVICE" not in os.environ:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


class FlatGradients:
    """Makes every parameter's .grad a view into ONE contiguous fp32 buffer, so that the
    gradient exchange is a single all-reduce and zeroing is a single memset."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero_(self):
        self.flat.zero_()
        for p in self.params:       # autograd may have replaced a view; re-bind is cheap
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * 4:
                raise RuntimeError("FlatGradients: a parameter's .grad was re-allocated; use "
                                   "optimizer.zero_grad(set_to_none=False) or FlatGradients.zero_()")

    def all_reduce_mean(self, async_op: bool = False):
        """Average the gradient over all ranks (one collective on the current stream)."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return None
        w = dist.get_world_size()
        if dist.get_backend() == "nccl":
            return dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, async_op=async_op)
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=False)
        self.flat.div_(w)
        return work


def shard_graphs(num_graphs: int, rank: int, world: int, sizes=None):
    """Graph i of the (shuffled) epoch list -> rank i mod W; with `sizes` the list is first
    sorted by size so that the graphs of one step are of similar size (the step time is the
    slowest rank's: SURVEY.md section 8e 'what actually limits scaling')."""
    order = list(range(num_graphs))
    if sizes is not None:
        order.sort(key=lambda i: -sizes[i])
    return [order[i] for i in range(rank, num_graphs, world)]
