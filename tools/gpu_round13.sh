#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r13; rm -rf $O; mkdir -p $O
python - > $O/phase.log 2>&1 <<'PY'
import ctypes, os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "graph-neural-networks_amd")]
import torch
from alegnn_amd import SparseGSO, _lib, graphgen
L = _lib.lib(); dev = torch.device("cuda:0")
def tune(**kw):
    for k, v in kw.items(): assert L.gf_tune(k.encode(), v) == 0, k
N, B, W = 10000, 256, 32
gso = SparseGSO([graphgen.sbm(N, seed=0)]); plans = gso.plans(dev)
P = B * W // 4
X = torch.randn(P, N, 4, device=dev); Y = torch.empty_like(X)
ms = ctypes.c_float()
import statistics
res = {}
plans_by = {}
for even in (0, 1):
    tune(panel_even=even)
    g = SparseGSO([graphgen.sbm(N, seed=0)]); plans_by[even] = (g, g.plans(dev))
for rep in range(7):
  for even in (0, 1):
        tune(panel_debug=0, panel_uniform=1, panel_grid=0, panel_stagger=20, panel_rotate=1, spmm_store=2)
        _lib.check(L.gf_time_spmm_hop_panel(plans_by[even][1][0], 0, X.data_ptr(), Y.data_ptr(), P, 30, torch.cuda.current_stream().cuda_stream, ctypes.byref(ms)))
        res.setdefault(even, []).append(ms.value * 1e3)
for k, v in sorted(res.items()):
    print(f"even-padded group-rows={k}: median {statistics.median(v):7.1f} us  min {min(v):7.1f}   {[round(x) for x in v]}", flush=True)
PY
cat $O/phase.log; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k 'panel or pipelines or edge_cases' 2>&1 | tail -3
