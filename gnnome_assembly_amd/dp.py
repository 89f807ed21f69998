"""Data-parallel training over the GPUs of one node: one whole assembly graph per GPU, one
process per GPU, a single RCCL all-reduce of one flat fp32 gradient buffer per step.

The reference has no distributed code at all (single device string, hyperparameters.py:25);
its training loop takes one optimizer step per graph (train.py:238-258).  Under DP the W graphs
of a step contribute the MEAN of their gradients.  BatchNorm statistics
are per graph by construction, so there is no data-path collective: the only exchange is the
gradient (826,033 floats = 3.3 MB at H=128/L=8), latency-bound on xGMI."""
from __future__ import annotations

import os
import weakref
from typing import Iterable, Optional

import torch
import torch.distributed as dist

__all__ = ["init_process_group", "FlatGradients", "shard_graphs", "steps_per_epoch", "fresh_flat_gradients"]

_live = weakref.WeakSet()      # the FlatGradients objects alive in this process
# GNM_FORCE_COLLECTIVE=1: do not skip the collectives at world size 1 (lets a 1-GPU box execute the RCCL path end to end)
FORCE_COLLECTIVE = os.environ.get("GNM_FORCE_COLLECTIVE", "0") == "1"


def fresh_flat_gradients(params) -> "Optional[FlatGradients]":
    """The FlatGradients(direct_write=True) whose buffer holds the .grad of EVERY tensor in `params`, if it has been
    zeroed since its last use (so that writing a gradient into it equals accumulating it) and no parameter carries
    a gradient hook (hooks only see gradients that travel through autograd); None otherwise."""
    if not params:
        return None
    for fg in _live:
        if fg.direct_write and fg.fresh and fg.owns(params):
            if any(getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) for p in params):
                fg.fresh = False        # autograd is about to accumulate into the buffer
                return None
            return fg
    return None


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
    kw = {}
    if backend == "nccl":
        if "GNM_BENCH_DEVICE" not in os.environ:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        # bind the communicator to this rank's device up front: barrier() and the first collective then need not GUESS the
        # device from the rank (torch warns about that guess and, on a node whose ranks are not device-ordered, gets it wrong)
        kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
    try:
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    except TypeError:           # a torch without the device_id keyword
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


class FlatGradients:
    """Makes every parameter's .grad a view into ONE contiguous fp32 buffer, so that the
    gradient exchange is a single all-reduce and zeroing is a single memset.

    `flat` carries one extra trailing element: the number of graphs this rank contributed to the step
    (1, or 0 for a padding step of a rank whose shard is shorter).  It rides in the same collective, and the
    summed gradient is divided by the summed count -- the mean over the graphs that actually took part.

    `direct_write=True` (what train.train and bench.py pass) is an OPT-IN to the engine's fast path: after zero_(), the
    whole-model backward writes every gradient straight into these views and tells autograd "no gradient" (writing
    into zeros == accumulating; ~140 accumulation kernels per step disappear).  The price, hence opt-in: while the
    buffer is fresh, torch.autograd.grad(loss, params) returns None for the parameters (and still fills .grad), and
    parameter hooks would not fire (the fast path stands down when a parameter has hooks).  Anything that writes a
    gradient other than the model's backward -- all_reduce_mean, a manual p.grad.add_ -- must leave `fresh` False;
    all_reduce_mean does.  With the default direct_write=False every gradient takes the ordinary autograd route."""

    def __init__(self, params: Iterable[torch.nn.Parameter], direct_write: bool = False):
        self.direct_write = bool(direct_write)
        self.params = [p for p in params if p.requires_grad]
        if all(hasattr(p, "_gnm_slot") for p in self.params):
            # a model flattened by models.flatten_parameters: follow its layout, so that the five stacked projection
            # gradients of a layer are one [5H,H] block the kernels can write directly
            self.params.sort(key=lambda p: p._gnm_slot)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.grads = self.flat[:n]                # the gradient proper (what an optimizer sees through p.grad)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()
        self.fresh = True          # all zeros: a kernel may WRITE a gradient instead of accumulating it
        self.contributors = None   # after all_reduce_mean: how many ranks contributed a graph to the step (device scalar)
        self._ids = {id(p) for p in self.params}
        _live.add(self)

    def owns(self, params) -> bool:
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + self.grads.numel() * 4
        return all(id(p) in self._ids and p.grad is not None and lo <= p.grad.data_ptr() < hi for p in params)

    def zero_(self):
        self.flat.zero_()
        self.fresh = True
        for p in self.params:       # autograd may have replaced a view; re-bind is cheap
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * 4:
                raise RuntimeError("FlatGradients: a parameter's .grad was re-allocated; use "
                                   "optimizer.zero_grad(set_to_none=False) or FlatGradients.zero_()")

    def all_reduce_mean(self, contributed: bool = True, async_op: bool = False):
        """Average the gradient over the ranks that contributed a graph to this step: ONE all-reduce (sum) of
        the flat buffer on the current stream, then a device-side division by the summed count (no host
        sync).  Every rank must call it the same number of times; a rank without a graph for this step
        calls zero_() and all_reduce_mean(contributed=False).  `async_op` is accepted for callers of the round-1
        signature and ignored: the collective is stream-ordered, there is nothing to wait for on the host."""
        self.fresh = False          # the buffer now holds a gradient: the next backward must accumulate, not overwrite
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not FORCE_COLLECTIVE):
            return None
        self.flat[-1] = 1.0 if contributed else 0.0
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        # a step no rank contributed to leaves a zero gradient, not 0/0 -- steps_per_epoch keeps that unreachable; the summed
        # count stays readable (a device scalar: no host sync here) for a caller or a debug run that wants to assert it
        self.contributors = self.flat[-1].clone()
        self.grads.div_(self.contributors.clamp(min=1.0))
        return None


def make_adam(params, lr: float):
    """torch.optim.Adam(params, lr) as train.py:209 builds it (default betas / eps, same state_dict schema: one state per
    parameter), as ONE fused multi-tensor kernel per step when every parameter lives on a HIP device: the default per-tensor
    ("foreach") implementation costs ~1 ms of host time per step for this model's 138 tensors -- invisible when the host runs
    steps ahead of the device (full-graph training), 5 % of a 21 ms mini-batch step (profiles/r04_minibatch_breakdown.txt)."""
    params = list(params)
    opt, impl = None, "default"
    if params and all(p.is_cuda for p in params):
        try:
            opt, impl = torch.optim.Adam(params, lr=lr, fused=True), "fused"
        except (RuntimeError, TypeError, ValueError) as ex:
            import warnings
            warnings.warn(f"dp.make_adam: fused Adam is not available here ({ex}); using the default implementation")
    if opt is None:
        opt = torch.optim.Adam(params, lr=lr)
    opt.gnm_impl = impl
    # replicas must not drift apart through different optimizer arithmetic: every rank has to have made the same choice
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = params[0].device if params else None
        t = torch.tensor([1.0 if impl == "fused" else 0.0], device=dev if dist.get_backend() == "nccl" else None)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if float(lo) != float(hi):
            raise RuntimeError("dp.make_adam: the ranks chose different Adam implementations (fused on some, default on others)")
    return opt


def steps_per_epoch(local_steps: int, device=None) -> int:
    """The number of optimizer steps EVERY rank takes in an epoch: the maximum of the ranks' local counts
    (shard_graphs gives uneven shards whenever num_graphs % world != 0); shorter ranks pad with
    zero-contribution steps so that the collectives stay matched."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not FORCE_COLLECTIVE):
        return int(local_steps)
    t = torch.tensor([int(local_steps)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def shard_graphs(num_graphs: int, rank: int, world: int, sizes=None):
    """Graph i of the (shuffled) epoch list -> rank i mod W; with `sizes` the list is first sorted by size so that the graphs of
    one step are of similar size (the step time is the slowest rank's: SURVEY.md section 8e 'what actually limits scaling'), and
    dealt in alternating directions (step 0: ranks 0 .. W-1, step 1: W-1 .. 0, ...) so that no rank always holds its step's smallest
    graph: on the chr19 / chr20 / chr21 mix a rank idles < 10 % in any step and ~4 % over the epoch
    (tests/test_dp_gloo.py::test_size_sorted_shards_keep_every_rank_busy_on_the_mixed_chromosome_set)."""
    order = list(range(num_graphs))
    if sizes is None:
        return [order[i] for i in range(rank, num_graphs, world)]
    order.sort(key=lambda i: -sizes[i])
    mine = []
    for step, lo in enumerate(range(0, num_graphs, world)):
        row = order[lo:lo + world]
        pos = rank if step % 2 == 0 else world - 1 - rank
        if pos < len(row):
            mine.append(row[pos])
    return mine
