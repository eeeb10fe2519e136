#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r15; rm -rf $O; mkdir -p $O
python - > $O/zigzag.log 2>&1 <<'PY'
import ctypes, os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "graph-neural-networks_amd")]
import numpy as np, torch
from alegnn_amd import _lib, graphgen
from alegnn_amd.utils import graphML as gml
L = _lib.lib(); dev = torch.device("cuda:0")
N, B, G, F, K = 10000, 256, 32, 32, 5
A = graphgen.sbm(N, seed=0)
layer = gml.GraphFilter(G, F, K, 1, True); layer.addGSO(A); layer.to(dev)
x = torch.randn(B, G, N, device=dev, requires_grad=True); dy = torch.randn(B, F, N, device=dev)
def step():
    x.grad = None; layer.zero_grad()
    y = layer(x); y.backward(dy)
for zz in (1, 0, 1, 0, 1, 0):
    assert L.gf_tune(b"panel_fuse_hops", zz) == 0
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"fuse_hops={zz}: {dt*1e3:.3f} ms/step", flush=True)
PY
cat $O/zigzag.log; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lsigf or pipelines or graph_filter or selection or full_size" 2>&1 | tail -3
