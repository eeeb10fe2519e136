#!/usr/bin/env python3
"""Launch the column-panel SpMM hop a few times (for rocprofv3 --pmc / --kernel-trace).  Usage: panel_probe.py [shape] [iters] [key=val ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import torch
from alegnn_amd import SparseGSO, _lib, graphgen
SHAPES = {"cfg2": (10_000, 256, 32), "mid5k": (5_000, 256, 32)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = _lib.lib()
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    assert L.gf_tune(k.encode(), int(v)) == 0, k
N, B, W = SHAPES[name]
dev = torch.device("cuda:0")
gso = SparseGSO([graphgen.sbm(N, seed=0)])
plans = gso.plans(dev)
P = B * W // 4
X = torch.randn(P, N, 4, device=dev); Y = torch.empty_like(X)
for _ in range(iters):
    _lib.check(L.gf_spmm_hop_panel(plans[0], 0, X.data_ptr(), Y.data_ptr(), P, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
