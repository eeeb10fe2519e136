#!/usr/bin/env python3
"""Callers of the K-hop kernel on the HIP path, next to the reference's formulation run with plain
torch ops on the same GPU (dense S; the "no-rewrite" comparator of SURVEY.md 8d).  One JSON line per item:
  nvgf     NodeVariantGF fwd+bwd                    (comparator: tools/_dense_torch.py, graphML.py:341-387 in torch ops)
  grnn     HiddenState / GatedGRNN fwd+bwd, T steps (comparator: the same recursion on lsigf_dense)
  trainer  Model + Trainer epochs on SelectionGNN (samples/s through getSamples -> forward -> loss -> backward -> Adam)
"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch
from alegnn_amd import graphgen
from alegnn_amd.modules import evaluation, model, training
from alegnn_amd.modules.architectures import SelectionGNN
from alegnn_amd.utils import graphML as gml
from _dense_torch import dense_lsigf, dense_nvgf

dev = torch.device("cuda:0")


def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def nvgf(N, B, G, F, K, M):
    A = graphgen.sbm(N, avg_degree=10.0, seed=0)
    layer = gml.NodeVariantGF(G, F, K, M, 1, True)
    layer.addGSO(A)
    layer.to(dev)
    x = torch.randn(B, G, N, device=dev, requires_grad=True)
    def step():
        layer.zero_grad(set_to_none=True); x.grad = None
        layer(x).square().sum().backward()
    ms = timeit(step)
    T = K
    algo = 4.0 * (2 * (T - 1) * B * N * G          # forward hops: read + write per tap
                  + T * B * N * G + T * G * F * N + B * N * F          # contraction: taps + bank in, y out
                  + 3 * (T - 1) * B * N * G                            # Horner adjoint hops: read g, read dZ_k, write
                  + 2 * (T * B * N * G + B * N * F) + 2 * T * G * F * N)   # dZ and dbank kernels
    out = dict(item="nvgf", N=N, nnz=int(A.nnz), B=B, G=G, F=F, K=K, M=M, hip_ms_fwd_bwd=round(ms, 3),
               algorithmic_GBps=round(algo / ms / 1e6, 1))
    if N <= 2000:
        S = torch.tensor(A.toarray(), dtype=torch.float32, device=dev)[None]
        w = layer.weight.detach().clone().requires_grad_(True)
        cn = layer.copyNodes.to(dev)
        def ref():
            w.grad = None; x.grad = None
            dense_nvgf(torch.index_select(w, 4, cn), S, x, layer.bias.detach()).square().sum().backward()
        out["torch_dense_same_gpu_ms"] = round(timeit(ref, n=5, warm=1), 3)
        out["speedup"] = round(out["torch_dense_same_gpu_ms"] / ms, 2)
    print(json.dumps(out), flush=True)


def grnn(N, B, T, F, H, K):
    A = graphgen.sbm(N, avg_degree=10.0, seed=0)
    layer = gml.HiddenState(F, H, K, nonlinearity=torch.tanh, E=1, bias=True)
    layer.addGSO(A)
    layer.to(dev)
    x = torch.randn(B, T, F, N, device=dev, requires_grad=True)
    z0 = torch.randn(B, H, N, device=dev)
    def step():
        layer.zero_grad(set_to_none=True); x.grad = None
        layer(x, z0)[0].square().sum().backward()
    ms = timeit(step)
    out = dict(item="grnn", N=N, nnz=int(A.nnz), B=B, T=T, F=F, H=H, K=K, hip_ms_fwd_bwd=round(ms, 3),
               edges_taps_per_s=round(int(A.nnz) * K * B * T * 2 / ms * 1e3, 1))      # two filters (A(S), B(S)) per time step
    if N <= 2000:
        S = torch.tensor(A.toarray(), dtype=torch.float32, device=dev)[None]
        ps = [p.detach().clone().requires_grad_(True) for p in (layer.aWeights, layer.bWeights, layer.xBias, layer.zBias)]
        def ref():
            for p in ps: p.grad = None
            x.grad = None
            Ax = dense_lsigf(ps[0], S, x.reshape(B * T, F, N), ps[2]).reshape(B, T, H, N)
            zt, acc = z0, 0.0
            for t in range(T):
                zt = torch.tanh(Ax[:, t] + dense_lsigf(ps[1], S, zt, ps[3]))
                acc = acc + zt.square().sum()
            acc.backward()
        out["torch_dense_same_gpu_ms"] = round(timeit(ref, n=5, warm=1), 3)
        out["speedup"] = round(out["torch_dense_same_gpu_ms"] / ms, 2)
    print(json.dumps(out), flush=True)


def trainer(N, nTrain, batchSize, epochs, hipGraph=False, F=(1, 32, 32)):
    class Data:
        def __init__(self):
            g = torch.Generator().manual_seed(0)
            self.x = {"train": torch.randn(nTrain, 1, N, generator=g), "valid": torch.randn(64, 1, N, generator=g)}
            self.y = {"train": torch.randint(0, 5, (nTrain,), generator=g), "valid": torch.randint(0, 5, (64,), generator=g)}
            self.x["test"], self.y["test"] = self.x["valid"], self.y["valid"]
            self.nTrain = nTrain
        def getSamples(self, split, *a):
            return (self.x[split][a[0]], self.y[split][a[0]]) if a else (self.x[split], self.y[split])
        def evaluate(self, yHat, y, tol=1e-9):
            return (torch.argmax(yHat, dim=1) != y).float().mean()
    A = graphgen.sbm(N, avg_degree=10.0, seed=0)
    net = SelectionGNN(list(F), [5, 5], True, torch.nn.ReLU, [N, N], gml.NoPool, [1, 1], [5], A)
    optim = torch.optim.Adam(net.parameters(), lr=1e-3)
    with tempfile.TemporaryDirectory() as tmp:
        m = model.Model(net, torch.nn.CrossEntropyLoss(), optim, training.Trainer, evaluation.evaluate,
                        dev, "bench", tmp)
        np.random.seed(0)
        m.train(Data(), 1, batchSize, printInterval=0, doSaveVars=False, hipGraph=hipGraph)   # warm-up epoch (plans, allocator, capture)
        graphs = m.trainer._graphs
        m.trainer = training.Trainer
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tv = None
        trainer_obj = training.Trainer(m, Data(), epochs, batchSize, printInterval=0, doSaveVars=False, hipGraph=hipGraph)
        trainer_obj._graphs = graphs                                  # keep the captured steps of the warm-up
        m.trainer = trainer_obj
        tv = trainer_obj.train()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps(dict(item="trainer", hipGraph=hipGraph, arch="SelectionGNN F=%s K=[5,5] NoPool MLP[5]" % list(F), N=N, nnz=int(A.nnz), nTrain=nTrain,
                          batchSize=batchSize, epochs=epochs, steps=len(tv["lossTrain"]), wall_s=round(dt, 3),
                          samples_per_s=round(epochs * nTrain / dt, 1), loss_first_last=[round(float(tv["lossTrain"][0]), 4),
                                                                                       round(float(tv["lossTrain"][-1]), 4)])), flush=True)


if __name__ == "__main__":
    nvgf(N=1000, B=64, G=32, F=32, K=5, M=100)
    nvgf(N=10000, B=64, G=32, F=32, K=5, M=1000)
    grnn(N=1000, B=16, T=10, F=8, H=32, K=4)
    grnn(N=10000, B=16, T=10, F=8, H=32, K=4)
    trainer(N=10000, nTrain=2048, batchSize=256, epochs=2)
    trainer(N=100, nTrain=2000, batchSize=20, epochs=2)                       # the reference's own scale (sourceLocGNN.py): launch-bound
    trainer(N=100, nTrain=2000, batchSize=20, epochs=2, hipGraph=True)
