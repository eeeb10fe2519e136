#!/usr/bin/env python3
"""Launch ONLY the SpMM hop kernel a few times on a benchmark shape (for rocprofv3 --pmc passes).
Usage: python tools/spmm_probe.py cfg2|cfg4 [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import torch
from alegnn_amd import SparseGSO, _lib, graphgen
import ctypes
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for kv in sys.argv[3:]:                                  # knob=value ... (gf_tune)
    k, v = kv.split("=")
    assert _lib.lib().gf_tune(k.encode(), int(v)) == 0, kv
model, N, B, W = {"cfg2": ("sbm", 10_000, 256, 32), "cfg4": ("er", 100_000, 128, 32)}[name]
A = (graphgen.sbm if model == "sbm" else graphgen.er)(N, seed=0)
dev = torch.device("cuda:0")
gso = SparseGSO([A]); plans = gso.plans(dev)
X0 = torch.randn(B, N, W, device=dev); X1 = torch.empty_like(X0)
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    _lib.check(L.gf_spmm_hop(plans[0], 0, X0.data_ptr(), X1.data_ptr(), B, W, st))
torch.cuda.synchronize()
print(f"probe {name}: nnz={A.nnz} algorithmic_bytes={2*B*N*W*4 + A.nnz*8 + (N+1)*4}")
