#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r34; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lsigf or graph_filter or selection or pipelines or panel or grnn" 2>&1 | tail -2
timeout 200 python - 2>&1 <<'PY' | tee gpurun_out/r34/split.log
import ctypes, os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "graph-neural-networks_amd")]
import torch
from alegnn_amd import SparseGSO, _lib, graphgen
L = _lib.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
for N in (10000, 3000):
    gso = SparseGSO([graphgen.sbm(N, seed=0)]); plans = gso.plans(dev)
    for B in (1, 4, 16, 32, 64):
        P = B * 8
        X = torch.randn(P, N, 4, device=dev); Y = {k: torch.empty_like(X) for k in (0, 1)}
        res = {}
        for split in (1, 0, 1, 0):
            assert L.gf_tune(b"panel_split", split) == 0
            ms = ctypes.c_float()
            _lib.check(L.gf_time_spmm_hop_panel(plans[0], 0, X.data_ptr(), Y[split].data_ptr(), P, 50, st, ctypes.byref(ms)))
            res.setdefault(split, []).append(ms.value * 1e3)
        torch.cuda.synchronize()
        print(f"N={N} B={B} panels={P}: one WG per panel {min(res[1]):.1f} us, split {min(res[0]):.1f} us, identical={torch.equal(Y[0], Y[1])}", flush=True)
L.gf_tune(b"panel_split", 0)
PY
timeout 100 python -c "
import sys; sys.argv=['x']; sys.path.insert(0,'tools')
import callers_bench as c
c.grnn(N=10000, B=16, T=10, F=8, H=32, K=4)
" 2>&1 | grep item
