#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r18; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel or pipelines or edge_cases or golden" 2>&1 | tail -4
python - <<'PY'
import ctypes, os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "graph-neural-networks_amd")]
import numpy as np, torch
from alegnn_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
B, N, G, F, E, K = 256, 10000, 32, 32, 1, 5
Z = torch.randn(K, B * G // 4, N, 4, device=dev); P = torch.randn(B * F // 4, N, 4, device=dev)
dh = torch.empty(F, E, K, G, device=dev); db = torch.empty(F, device=dev)
nb = L.gf_grad_taps_workspace_bytes(B, N, G, F, E, K); ws = torch.empty(nb // 4 + 1, device=dev)
st = torch.cuda.current_stream().cuda_stream
for lds in (1, 0, 1, 0):
    assert L.gf_tune(b"gradw_lds", lds) == 0
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(L.gf_grad_taps_panel(Z.data_ptr(), P.data_ptr(), dh.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, B, N, G, F, E, K, st)); e1.record(); e1.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1))
    print(f"grad_taps_panel gradw_lds={lds}: {np.median(ts):.4f} ms", flush=True)
PY
