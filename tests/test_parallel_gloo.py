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


# ---- the data-parallel Trainer: 2 ranks must retrace the reference's single-process run ------------
def _trainer_worker(rank, world, port, saveDir, ret):
    import ast
    import numpy as np
    from _util import GOLDEN, ArrayData, load
    from alegnn_amd.modules import evaluation, model, training
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = load(os.path.join(GOLDEN, "trainer_mlp.npz"))
        N = d["S"].shape[1]
        net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(N, 16), torch.nn.Tanh(), torch.nn.Linear(16, 5)).double()
        if rank == 0:                                   # only rank 0 starts from the reference's weights: Trainer broadcasts
            net.load_state_dict({k[5:]: torch.tensor(v) for k, v in d.items() if k.startswith("init:")})
        optim = torch.optim.Adam(net.parameters(), lr=0.005, betas=(0.9, 0.999))
        m = model.Model(net, torch.nn.CrossEntropyLoss(), optim, training.Trainer,
                        evaluation.evaluate, 'cpu', 'mlp', saveDir)
        np.random.seed(int(d["seed"]) + 1 if rank == 0 else 12345)      # rank 0's permutation is the one used
        data = ArrayData(d, torch.float64)
        tv = m.train(data, int(d["nEpochs"]), int(d["batchSize"]), printInterval=0, **ast.literal_eval(str(d["trainKw"])))
        ev = m.evaluate(data)
        ret[rank] = (tv, ev, {k: v.clone() for k, v in net.state_dict().items()})
    finally:
        dist.destroy_process_group()


def test_data_parallel_trainer_retraces_reference_run(tmp_path):
    import numpy as np
    from _util import GOLDEN, load
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path), ret), nprocs=world, join=True)
    d = load(os.path.join(GOLDEN, "trainer_mlp.npz"))
    for r in range(world):
        tv, ev, sd = ret[r]
        for k in ("lossTrain", "costTrain", "lossValid", "costValid"):      # batches of 40/40/16 split 20+20 / 20+20 / 8+8
            assert tv[k].shape == d[k].shape and np.allclose(tv[k], d[k], rtol=1e-9, atol=1e-12), (r, k)
        assert ev == {"costBest": float(d["costBest"]), "costLast": float(d["costLast"])}
    for k in ret[0][2]:
        assert torch.equal(ret[0][2][k], ret[1][2][k])                       # replicas never diverge
    best = torch.load(tmp_path / "savedModels" / "mlpArchitBest.ckpt")       # written once, by rank 0
    ref = torch.load(os.path.join(GOLDEN, "ckpt", "mlpArchitBest.ckpt"))
    assert all(torch.allclose(best[k], ref[k], rtol=1e-9, atol=1e-12) for k in ref)


def _uneven_worker(rank, world, port, ret):
    from alegnn_amd.modules import training

    class _M:
        pass
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _model().double()
        g = torch.Generator().manual_seed(1)
        X, Y = torch.randn(7, 6, generator=g).double(), torch.randn(7, 3, generator=g).double()

        class Data:
            nTrain = 7

            def getSamples(self, split, *a):
                return (X[a[0]], Y[a[0]]) if a else (X, Y)

            def evaluate(self, yHat, y):
                return torch.mean((yHat - y) ** 2)
        m = _M()
        m.archit, m.loss, m.device = net, torch.nn.MSELoss(), 'cpu'
        m.optim = torch.optim.SGD(net.parameters(), lr=0.0)
        tr = training.Trainer(m, Data(), 1, 7, printInterval=0)
        out = [tr.trainBatch(idx) for idx in ([0, 1, 2, 3, 4, 5, 6], [3])]    # 7 = 4 + 3 (rounded cut); 1 = 1 + 0: an idle rank
        ret[rank] = (out, tr.bucket.flat[:tr.bucket.numel].clone())
    finally:
        dist.destroy_process_group()


def test_data_parallel_uneven_and_empty_shares_give_the_global_mean_gradient():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_uneven_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    net = _model().double()
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(7, 6, generator=g).double(), torch.randn(7, 3, generator=g).double()
    full = torch.nn.functional.mse_loss(net(X), Y)
    one = torch.nn.functional.mse_loss(net(X[[3]]), Y[[3]])
    one.backward()
    want = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    for r in range(world):
        out, flat = ret[r]
        assert abs(out[0][0] - full.item()) < 1e-12 and abs(out[0][1] - full.item()) < 1e-12
        assert abs(out[1][0] - one.item()) < 1e-12
        assert torch.allclose(flat, want, atol=1e-12)
