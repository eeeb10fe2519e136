#!/usr/bin/env python3
"""Time EdgeVariantGF (per-edge storage, sparse=True) forward+backward at BASELINE config 5:
SBM N=50k nnz~500k, K=3, F=G=32, batch 16.  Prints one JSON line (HIP-event timings on torch's current stream, which is
the stream the C ABI launches on).  Usage: python tools/evgf_bench.py [N] [B]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import numpy as np, torch
from alegnn_amd import graphgen
from alegnn_amd.utils import graphML as gml

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
G = F = 32
K = 3
dev = torch.device("cuda:0")
A = graphgen.sbm(N, seed=0)
layer = gml.EdgeVariantGF(G, F, K, N, N, 1, True, sparse=True)
layer.addGSO(A)
layer.to(dev)
nnzp = layer._patterns[0].nnzp
x = torch.randn(B, G, N, device=dev, requires_grad=True)
dy = torch.randn(B, F, N, device=dev)

def step():
    for p in layer.parameters():
        p.grad = None
    x.grad = None
    y = layer(x)
    y.backward(dy)

for _ in range(2):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); step(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = float(np.median(ts))
C = F * G
state = B * C * N * 4
# algorithmic bytes fwd: V0 write + (K-1) x (weights + state read + state write) + sum read of K states; bwd ~ 2x that + SDDMM
fwd_bytes = state + (K - 1) * (C * nnzp * 4 + 2 * state) + K * state
print(json.dumps(dict(workload="cfg5 EdgeVariantGF sparse", N=N, nnz=int(A.nnz), nnzp=nnzp, B=B, G=G, F=F, K=K, ms_fwd_bwd=round(ms, 3),
                      edges_taps_per_s=B * A.nnz * K / (ms * 1e-3), fwd_algorithmic_GB=round(fwd_bytes / 1e9, 2),
                      param_GB=round(C * (K - 1) * nnzp * 4 / 1e9, 2), state_GB_per_tap=round(state / 1e9, 2))))
