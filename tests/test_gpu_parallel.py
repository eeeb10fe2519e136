"""The HIP path under world_size 2: two processes share the one GPU of the test box, `gloo` carries the collectives
(parallel.py stages device tensors through the host for that backend; on a node the backend is RCCL and nothing else changes).
What has to hold is what batch-axis data parallelism promises (hook point reference training.py:248-251):
  * the HIP GraphFilter backward writes into the bucket's views, ONE all-reduce gives the gradient of the global batch mean -- equal to the
    single-process full-batch gradient of the same layer;
  * replicas that start equal stay bit-identical through optimiser steps;
  * the Trainer (hipGraph off) run on two ranks retraces the reference's single-process training run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _filter_setup(dev):
    from alegnn_amd import graphgen
    import alegnn_amd.utils.graphML as gml
    S = graphgen.sbm(600, seed=3) if hasattr(graphgen, "sbm") else graphgen.er(600, avg_degree=8.0, seed=3)
    torch.manual_seed(0)
    layer = gml.GraphFilter(32, 32, 4)
    layer.addGSO(S)
    layer = layer.to(dev)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(16, 32, 600, generator=g)
    T = torch.randn(16, 32, 600, generator=g)
    return layer, X, T


def _filter_worker(rank, world, port, ret):
    from alegnn_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        layer, X, T = _filter_setup(dev)
        if rank == 1:                                   # de-synchronise rank 1, then broadcast must repair it
            with torch.no_grad():
                for p in layer.parameters():
                    p.add_(0.5)
        parallel.broadcast_parameters(layer, src=0)
        bucket = parallel.GradBucket(layer.parameters())
        optim = torch.optim.SGD(layer.parameters(), lr=0.05)
        idx = parallel.shard_batch(list(range(16)))
        x, t = X[idx].to(dev), T[idx].to(dev)
        first = None
        for step in range(3):
            bucket.zero_()
            torch.nn.functional.mse_loss(layer(x), t).backward()          # HIP backward writes into the bucket's views
            flat = bucket.allreduce_mean()
            if first is None:
                first = flat.detach().cpu().clone()
            optim.step()
        ret[rank] = (first, [p.detach().cpu().clone() for p in layer.parameters()], idx)
    finally:
        dist.destroy_process_group()


def test_hip_graph_filter_gradients_reduce_to_the_full_batch_gradient():
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_filter_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    layer, X, T = _filter_setup(dev)
    optim = torch.optim.SGD(layer.parameters(), lr=0.05)
    want_first = None
    for step in range(3):                                                 # the single-process run on the whole batch
        optim.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(layer(X.to(dev)), T.to(dev)).backward()
        if want_first is None:
            want_first = torch.cat([p.grad.reshape(-1) for p in layer.parameters()]).cpu()
        optim.step()
    assert ret[0][2] == list(range(8)) and ret[1][2] == list(range(8, 16))
    scale = float(want_first.abs().max())
    for r in range(world):
        assert float((ret[r][0] - want_first).abs().max()) <= 1e-6 * scale + 1e-9, r
    for a, b in zip(ret[0][1], ret[1][1]):
        assert torch.equal(a, b)                                          # replicas never diverge
    for a, p in zip(ret[0][1], layer.parameters()):                       # and follow the single-process trajectory
        assert float((a - p.detach().cpu()).abs().max()) <= 1e-5 * float(p.detach().abs().max())


def _trainer_worker(rank, world, port, saveDir, ret):
    import ast
    from _util import GOLDEN, ArrayData, load
    import alegnn_amd.utils.graphML as gml
    from alegnn_amd.modules import evaluation, model, training
    from alegnn_amd.modules.architectures import SelectionGNN
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = load(os.path.join(GOLDEN, "trainer_selgnn.npz"))
        net = SelectionGNN([1, 8, 8], [3, 3], True, torch.nn.ReLU, [20, 10], gml.MaxPoolLocal, [2, 2], [5], d["S"][0])
        if rank == 0:                                   # only rank 0 starts from the reference's weights: Trainer broadcasts
            net.load_state_dict({k[5:]: torch.tensor(v) for k, v in d.items() if k.startswith("init:")})
        net = net.float()
        optim = torch.optim.Adam(net.parameters(), lr=0.005, betas=(0.9, 0.999))
        m = model.Model(net, torch.nn.CrossEntropyLoss(), optim, training.Trainer, evaluation.evaluate, torch.device("cuda:0"), "selgnn", saveDir)
        np.random.seed(int(d["seed"]) + 1 if rank == 0 else 999)          # rank 0's permutation is the one used
        tv = m.train(ArrayData(d, torch.float32), int(d["nEpochs"]), int(d["batchSize"]), printInterval=0, **ast.literal_eval(str(d["trainKw"])))
        ret[rank] = ({k: np.asarray(tv[k]) for k in ("lossTrain", "lossValid", "costValid")}, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    finally:
        dist.destroy_process_group()


def test_data_parallel_trainer_on_the_hip_path_retraces_the_reference_run(tmp_path):
    from _util import GOLDEN, load
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path), ret), nprocs=world, join=True)
    d = load(os.path.join(GOLDEN, "trainer_selgnn.npz"))
    for r in range(world):
        tv, _ = ret[r]
        assert np.allclose(tv["lossTrain"], d["lossTrain"], rtol=2e-4), (r, tv["lossTrain"], d["lossTrain"])
        assert np.allclose(tv["lossValid"], d["lossValid"], rtol=2e-4)
        assert np.max(np.abs(tv["costValid"] - d["costValid"])) <= 1.0 / 32 + 1e-6
    for k in ret[0][1]:
        assert torch.equal(ret[0][1][k], ret[1][1][k]), k                 # replicas never diverge


def _chain_worker(rank, world, port, ret):
    """Fused K-hop chains (cooperative, every-CU launches) from TWO processes on one GPU at the same time."""
    import ctypes
    import scipy.sparse as sp
    from alegnn_amd import _lib
    from alegnn_amd.gso import SparseGSO
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        n, B, K, W = 60000, 16, 4, 32
        rng = np.random.RandomState(7)
        r = np.repeat(np.arange(n), 4)
        A = sp.csr_matrix((np.ones(r.size), (r, rng.randint(0, n, size=r.size))), shape=(n, n))
        A = ((A + A.T) > 0).astype(np.float64)
        A.setdiag(0)
        A.eliminate_zeros()
        gso = SparseGSO([sp.csr_matrix(A * 0.0625)])
        plans = gso.plans(dev)
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        assert L.gf_spmm_hop_kernel(plans[0], 0, B, W) == 1
        torch.manual_seed(5)
        Z = torch.full((K, B, n, W), float("nan"), device=dev)
        Z[0].normal_()
        dist.barrier()                                  # both processes start launching together
        outs = []
        for rep in range(40):
            Z[1:].fill_(float("nan"))
            _lib.check(L.gf_khop(plans, 1, 0, Z.data_ptr(), B, W, K, st))
            if rep % 8 == 7:
                torch.cuda.synchronize()
                outs.append(Z.clone())
        torch.cuda.synchronize()
        same = all(bool(torch.equal(o, outs[0])) for o in outs)
        # reference: the per-hop launches of the same kernel (GFHIP_MSWEEP_FUSE semantics: one launch per hop has no hand-over inside a launch)
        f, on = ctypes.c_uint32(0), ctypes.c_int32(0)
        L.gf_msweep_status(ctypes.byref(f), ctypes.byref(on))
        Zr = Z.clone()
        Zr[1:].fill_(float("nan"))
        for k in range(1, K):
            _lib.check(L.gf_spmm_hop(plans[0], 0, Zr[k - 1].data_ptr(), Zr[k].data_ptr(), B, W, st))
        torch.cuda.synchronize()
        ret[rank] = (same, bool(torch.equal(outs[-1], Zr)), int(f.value), int(on.value))
    finally:
        dist.destroy_process_group()


def test_fused_chains_of_two_processes_on_one_gpu_do_not_trap_hang_or_differ():
    """two processes share one GPU (this file's configuration) and both launch the one-launch K-hop chain, a cooperative grid
    that needs every CU.  Whatever the runtime does with the two grids -- serialise them, or interleave them so that neither is whole, in which
    case the census gives up after its time limit and the repair kernel does the work -- every chain must come out with the per-hop launches'
    bits, and neither process may die.  The status flags say which of the two happened (informational)."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_chain_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for rank in range(world):
        same, equal_ref, flags, fusion_on = ret[rank]
        assert same and equal_ref, (rank, same, equal_ref, flags, fusion_on)
    import warnings
    warnings.warn(f"two processes, one GPU: status flags per rank {[ret[r][2] for r in range(world)]}, fusion still on {[ret[r][3] for r in range(world)]}")
