#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r24; rm -rf $O; mkdir -p $O
python - > $O/small.log 2>&1 <<'PY'
import ctypes, os, sys, statistics
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "graph-neural-networks_amd")]
import numpy as np, torch
from alegnn_amd import SparseGSO, _lib, graphgen
L = _lib.lib(); dev = torch.device("cuda:0")
def tune(**kw):
    for k, v in kw.items(): assert L.gf_tune(k.encode(), v) == 0, k
ms = ctypes.c_float()
for name, N, B, W, wt in (("n2k", 2000, 256, 32, 0), ("cfg3", 1682, 256, 64, 0), ("cfg3w", 1682, 256, 64, 1), ("mid5k", 5000, 256, 32, 0), ("n7k", 7000, 256, 32, 0), ("cfg2", 10000, 256, 32, 0), ("cfg2w", 10000, 256, 32, 1), ("cfg2_B1024", 10000, 1024, 32, 0)):
    A = graphgen.sbm(N, seed=0)
    if wt:
        A = A.copy(); A.data = np.random.RandomState(0).uniform(0.1, 1.0, A.nnz) * A.data
    gso = SparseGSO([A]); plans = gso.plans(dev)
    P = B * W // 4
    X = torch.randn(P, N, 4, device=dev); Y = torch.empty_like(X)
    Xn = torch.randn(B, N, W, device=dev); Yn = torch.empty_like(Xn)
    nbytes = 2 * B * N * W * 4 + A.nnz * 8 + (N + 1) * 4
    out = []
    for rot in (1, 0):
        tune(panel_rotate=rot)
        v = []
        for rep in range(5):
            _lib.check(L.gf_time_spmm_hop_panel(plans[0], 0, X.data_ptr(), Y.data_ptr(), P, 20, torch.cuda.current_stream().cuda_stream, ctypes.byref(ms)))
            v.append(ms.value * 1e3)
        out.append(statistics.median(v))
    tune(panel_rotate=1)
    v = []
    for rep in range(3):
        _lib.check(L.gf_time_spmm_hop(plans[0], 0, Xn.data_ptr(), Yn.data_ptr(), B, W, 20, torch.cuda.current_stream().cuda_stream, ctypes.byref(ms)))
        v.append(ms.value * 1e3)
    l2 = statistics.median(v)
    print(f"{name:11s} N={N:6d} B={B:5d} W={W:3d} {'weighted' if wt else 'uniform '}: panel rotate=1 {out[0]:7.1f} us ({100*nbytes/out[0]/8e6:4.1f} %)  rotate=0 {out[1]:7.1f} us ({100*nbytes/out[1]/8e6:4.1f} %)   node-major L2 {l2:7.1f} us ({100*nbytes/l2/8e6:4.1f} %)", flush=True)
PY
cat $O/small.log
