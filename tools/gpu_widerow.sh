#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r37; export TMPDIR=/tmp
timeout 250 python - 2>&1 <<'PY' | tee gpurun_out/r37/widerow.log
import ctypes, os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "graph-neural-networks_amd")]
import torch
from alegnn_amd import SparseGSO, _lib, graphgen
L = _lib.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
N = 100000
gso = SparseGSO([graphgen.er(N, seed=0)]); plans = gso.plans(dev)
for (B, W) in [(128, 32), (64, 64), (32, 128), (256, 16)]:
    X = torch.randn(B, N, W, device=dev); Y = torch.empty_like(X)
    best = 1e9
    for _ in range(2):
        ms = ctypes.c_float()
        _lib.check(L.gf_time_spmm_hop(plans[0], 0, X.data_ptr(), Y.data_ptr(), B, W, 5, st, ctypes.byref(ms)))
        best = min(best, ms.value)
    alg = 2 * B * N * W * 4 / 1e9
    print(f"N={N} B={B} W={W}: {best*1e3:.0f} us  {alg/best*1e3/8000*100:.1f} % of 8 TB/s", flush=True)
PY
