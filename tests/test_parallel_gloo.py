"""world_size-2 `gloo` tests (CPU) of the batch-DP plumbing: bucketed gradient all-reduce == full-batch gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from alegnn_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        if rank == 1:                                   # de-synchronise rank 1, then broadcast must repair it
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        parallel.broadcast_parameters(model, src=0)
        bucket = parallel.GradBucket(model.parameters())
        g = torch.Generator().manual_seed(1)
        X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
        idx = parallel.shard_batch(list(range(8)))
        for _ in range(2):                              # two steps: the views must survive zero_() / all-reduce
            bucket.zero_()
            torch.nn.functional.mse_loss(model(X[idx]), Y[idx]).backward()
            flat = bucket.allreduce_mean()
        ret[rank] = (flat.clone(), [p.grad.clone() for p in model.parameters()], idx)
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_equals_full_batch_gradient():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    model = _model()
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    torch.nn.functional.mse_loss(model(X), Y).backward()
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert ret[0][2] == [0, 1, 2, 3] and ret[1][2] == [4, 5, 6, 7]
    for r in range(world):
        flat, grads, _ = ret[r]
        assert torch.allclose(flat, want, atol=1e-6)
        assert torch.allclose(torch.cat([g_.reshape(-1) for g_ in grads]), want, atol=1e-6)   # p.grad are bucket views
    assert torch.equal(ret[0][0], ret[1][0])            # replicas agree bitwise after the collective


def test_shard_batch_requires_even_split():
    assert parallel.shard_batch(list(range(6)), rank=1, world=3) == [2, 3]
    with pytest.raises(AssertionError):
        parallel.shard_batch(list(range(7)), rank=0, world=2)


def test_bucket_single_process_is_identity():
    model = _model()
    bucket = parallel.GradBucket(model.parameters())
    model(torch.ones(2, 6)).sum().backward()
    before = bucket.flat.clone()
    assert torch.equal(bucket.allreduce_mean(), before) and bucket.nbytes() == before.numel() * 4
