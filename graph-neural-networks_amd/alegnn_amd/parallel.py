"""Batch-axis data parallelism for the GraphFilter path: one process per GPU, RCCL over xGMI.

The path shards over the batch axis with no data-path communication (every sample's taps and outputs depend only on
(S, h), reference graphML.py:152-171 has no cross-batch term): rank r holds x[r*B/W:(r+1)*B/W], a full replica of the
CSR plan and of the parameters.  The only exchange is ONE all-reduce per step over a single flat gradient bucket
(all GFL.*.weight/bias + MLP.*: ~20-40 KB at the benchmark shapes, latency-bound, so one collective instead of one
per tensor), issued between loss.backward() and optim.step() -- the hook point is reference training.py:248-251.
The reference has no distributed code at all, so there is no call pattern to mirror.

backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

# GFHIP_FORCE_COLLECTIVES=1: issue the collectives even at world size 1 (exercises the RCCL path on a one-GPU box)
_FORCE = os.environ.get("GFHIP_FORCE_COLLECTIVES", "0") == "1"


def _stage_through_host(t, group=None):
    """True when the collective has to go through a host copy: the "gloo" backend with a device tensor (a debugging / test
    configuration -- e.g. two ranks sharing one GPU; RCCL reduces device buffers in place)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(t, group=None):
    if _stage_through_host(t, group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast(t, src=0, group=None):
    if _stage_through_host(t, group):
        h = t.detach().cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


class GradBucket:
    """All parameter gradients as views into one contiguous buffer; ``allreduce_mean()`` is a single collective.

    Usage per step:   bucket.zero_(); loss.backward(); bucket.allreduce_mean(); optim.step()
    (use ``bucket.zero_()`` instead of ``zero_grad(set_to_none=True)`` so the views stay attached).
    """

    def __init__(self, params, process_group=None, extra=0):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params), "one device / dtype per bucket"
        self.numel = sum(p.numel() for p in self.params)
        # ``extra`` trailing slots ride in the same collective (the trainer's per-step loss / cost scalars)
        self.flat = torch.zeros(self.numel + extra, dtype=dt, device=dev)
        self.extra = self.flat[self.numel:]
        self.group = process_group
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def allreduce_mean(self):
        """Sum over ranks then divide by world size == gradient of the global batch-mean loss when every rank holds
        an equal share of the batch (nn.CrossEntropyLoss / SmoothL1Loss default reduction, sourceLocGNN.py:167)."""
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(self.group)
            if world > 1 or _FORCE:
                all_reduce_sum(self.flat, self.group)
                self.flat.div_(world)
        return self.flat

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


def shard_batch(indices, rank=None, world=None):
    """Contiguous equal split of one global batch's sample indices; replaces idxEpoch[batchIndex[b]:batchIndex[b+1]]
    at reference training.py:398-399.  The global batch must divide evenly (equal shares keep the mean exact)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    n = len(indices)
    assert n % world == 0, f"global batch {n} does not divide over {world} ranks"
    per = n // world
    return indices[rank * per:(rank + 1) * per]


def broadcast_parameters(module, src=0):
    """Make every rank start from rank ``src``'s parameters (replicas must be identical for DP to be exact)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE):
        for p in module.parameters():
            broadcast(p.data, src=src)
