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
for rep in range(3):
    for uni in (1, 0):
        tune(panel_debug=0, panel_uniform=uni, spmm_store=0, panel_stagger=1)
        _lib.check(L.gf_time_spmm_hop_panel(plans[0], 0, X.data_ptr(), Y.data_ptr(), P, 20, torch.cuda.current_stream().cuda_stream, ctypes.byref(ms)))
        print(f"{ms.value*1e3:8.1f} us  uniform={uni}", flush=True)
for dbg in (1, 2, 3, 4):
    tune(panel_debug=dbg, panel_uniform=1)
    _lib.check(L.gf_time_spmm_hop_panel(plans[0], 0, X.data_ptr(), Y.data_ptr(), P, 20, torch.cuda.current_stream().cuda_stream, ctypes.byref(ms)))
    print(f"{ms.value*1e3:8.1f} us  uniform=1 debug={dbg} (1 = load only, 2 = compute only, 3 = compute only with L1-resident entry loads, 4 = compute only without stores)", flush=True)
PY
cat $O/phase.log; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k 'panel or pipelines or edge_cases' 2>&1 | tail -3
