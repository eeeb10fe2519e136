#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r25; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel or pipelines or edge_cases" 2>&1 | tail -4
python - > $O/np.log 2>&1 <<'PY'
import ctypes, os, sys, statistics
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "graph-neural-networks_amd")]
import numpy as np, torch
from alegnn_amd import SparseGSO, _lib, graphgen
L = _lib.lib(); dev = torch.device("cuda:0")
def tune(**kw):
    for k, v in kw.items(): assert L.gf_tune(k.encode(), v) == 0, k
ms = ctypes.c_float()
for name, N, B, W, wt in (("n1k", 1000, 256, 32, 0), ("cfg3", 1682, 256, 64, 0), ("cfg3w", 1682, 256, 64, 1), ("n2k", 2000, 256, 32, 0), ("n2.5k", 2500, 256, 32, 0), ("n4k", 4000, 256, 32, 0), ("mid5k", 5000, 256, 32, 0), ("mid5kw", 5000, 256, 32, 1)):
    A = graphgen.sbm(N, seed=0)
    if wt:
        A = A.copy(); A.data = np.random.RandomState(0).uniform(0.1, 1.0, A.nnz) * A.data
    gso = SparseGSO([A]); plans = gso.plans(dev)
    P = B * W // 4
    X = torch.randn(P, N, 4, device=dev); Y = torch.empty_like(X)
    nbytes = 2 * B * N * W * 4 + A.nnz * 8 + (N + 1) * 4
    out = {}
    for rep in range(5):
        for np_ in (1, 2):
            tune(panel_np=np_)
            _lib.check(L.gf_time_spmm_hop_panel(plans[0], 0, X.data_ptr(), Y.data_ptr(), P, 20, torch.cuda.current_stream().cuda_stream, ctypes.byref(ms)))
            out.setdefault(np_, []).append(ms.value * 1e3)
    a, b = statistics.median(out[1]), statistics.median(out[2])
    print(f"{name:8s} N={N:5d} W={W:3d} {'weighted' if wt else 'uniform '}: 1 panel/pass {a:7.1f} us ({100*nbytes/a/8e6:4.1f} %)   2 panels/pass {b:7.1f} us ({100*nbytes/b/8e6:4.1f} %)", flush=True)
PY
cat $O/np.log
