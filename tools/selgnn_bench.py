#!/usr/bin/env python3
"""SelectionGNN forward+backward (filter + fused ReLU + MaxPoolLocal/NoPool + MLP) on the HIP path, BASELINE configs 1 and 3
shapes, and the same architecture in the reference's formulation (dense S, matmul/cat/permute, repeat+gather+max) run on the
same device with plain torch ops -- the "no-rewrite" comparator of SURVEY.md 8d.  One JSON line per configuration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import numpy as np, torch
from alegnn_amd import graphgen
from alegnn_amd.modules.architectures import SelectionGNN
from alegnn_amd.utils import graphML as gml
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _dense_torch import dense_lsigf

dev = torch.device("cuda:0")
CFG = {
    # examples/sourceLocGNN.py:243-260: SBM N=100, F=[1,32,32], K=[5,5], MaxPoolLocal N=[100,10,10], alpha=[6,8], MLP [5]
    "cfg1": dict(N=100, deg=30.0, F=[1, 32, 32], K=[5, 5], sel=[10, 10], pool="MaxPoolLocal", alpha=[6, 8], mlp=[5], B=100),
    # examples/movieGNN.py:259-276 on a MovieLens-100k-sized graph (N=1682, kNN-10-like degree): F=[1,64,32], K=[5,5], NoPool, MLP [1]
    "cfg3": dict(N=1682, deg=15.0, F=[1, 64, 32], K=[5, 5], sel=[1682, 1682], pool="NoPool", alpha=[1, 1], mlp=[1], B=256),
}

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for name in (sys.argv[1:] or list(CFG)):
    c = CFG[name]
    A = graphgen.sbm(c["N"], avg_degree=c["deg"], seed=0)
    torch.manual_seed(0)
    net = SelectionGNN(c["F"], c["K"], True, torch.nn.ReLU, c["sel"], getattr(gml, c["pool"]), c["alpha"], c["mlp"], A).to(dev)
    x = torch.randn(c["B"], c["F"][0], c["N"], device=dev)
    def step():
        net.zero_grad(set_to_none=True)
        net(x).square().sum().backward()
    ms = timeit(step)
    # the same step captured once in a HIP graph and replayed (the library only launches on the stream it is given)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    ms_graph = timeit(graph.replay)
    # reference formulation with torch ops on the same GPU (dense S)
    S = torch.tensor(A.toarray(), dtype=torch.float32, device=dev)[None]
    ws = [(net.GFL[3 * l].weight.detach().clone().requires_grad_(True), net.GFL[3 * l].bias.detach().clone().requires_grad_(True)) for l in range(2)]
    nbhs = [getattr(net.GFL[3 * l + 2], "neighborhood", None) for l in range(2)]
    mlp = net.MLP
    def ref_step():
        for w, b in ws: w.grad = None; b.grad = None
        mlp.zero_grad(set_to_none=True)
        y = x
        Ns = [c["N"]] + c["sel"]
        for l, (w, b) in enumerate(ws):
            Nin = y.shape[2]
            if Nin < c["N"]:
                y = torch.cat((y, torch.zeros(y.shape[0], y.shape[1], c["N"] - Nin, device=dev)), dim=2)
            y = torch.relu(dense_lsigf(w, S, y, b)[:, :, :Nin])
            if nbhs[l] is not None:
                y, _ = torch.max(y[:, :, nbhs[l].long()], dim=3)
            else:
                y = y[:, :, :Ns[l + 1]]
        mlp(y.reshape(y.shape[0], -1)).square().sum().backward()
    ms_ref = timeit(ref_step, n=10, warm=2)
    nnz = int(A.nnz)
    print(json.dumps(dict(workload=name, N=c["N"], nnz=nnz, B=c["B"], F=c["F"], K=c["K"], pool=c["pool"],
                          hip_ms_fwd_bwd=round(ms, 3), hip_graph_replay_ms=round(ms_graph, 3), torch_dense_same_gpu_ms=round(ms_ref, 3), speedup=round(ms_ref / ms, 2))), flush=True)
