#!/usr/bin/env python3
"""Layer-to-layer hand-over on the node-major pipeline: a [GraphFilter, ReLU, NoPool] x L stack on a config-4-class graph (ER N = 1e5,
32 features, K = 5) stepped forward + backward with the inner signals handed over as node-major rows and as separate layers
(reference-layout round trip per boundary).  Prints ms per step for both, the difference per inner boundary and whether outputs and
gradients are bitwise equal.   Usage: handover_bench.py [N] [B] [layers]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import torch
from alegnn_amd import graphgen, functional
from alegnn_amd.modules.architectures import SelectionGNN
from alegnn_amd.utils import graphML as gml

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
Lr = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
A = graphgen.er(N, avg_degree=10.0, seed=0)
torch.manual_seed(0)
net = SelectionGNN([32] * (Lr + 1), [5] * Lr, True, torch.nn.ReLU, [N] * Lr, gml.NoPool, [1] * Lr, [1], A).to(dev)
x = torch.randn(B, 32, N, device=dev)

def run(mode):
    functional._HANDOVER = mode
    xr = x.clone().requires_grad_(True)
    def step():
        net.zero_grad(set_to_none=True)
        xr.grad = None
        net(xr).square().sum().backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    return ms, [xr.grad.clone()] + [p.grad.clone() for p in net.parameters()]

ms_off, g_off = run(False)
ms_on, g_on = run(True)
same = all(torch.equal(a, b) for a, b in zip(g_on, g_off))
print(json.dumps(dict(N=N, B=B, layers=Lr, ms_separate=round(ms_off, 3), ms_handover=round(ms_on, 3),
                      saved_ms_per_inner_boundary=round((ms_off - ms_on) / (Lr - 1), 3), gradients_bitwise_equal=same)))
