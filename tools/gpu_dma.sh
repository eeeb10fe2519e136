#!/bin/bash
# A/B of the panel load phase: registers vs LDS-DMA.  Hop timing via gf_time_spmm_hop_panel at several sizes + correctness vs knob 0.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r32; export TMPDIR=/tmp
timeout 300 python - 2>&1 <<'PY' | tee gpurun_out/r32/dma.log
import ctypes, os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "graph-neural-networks_amd")]
import torch
from alegnn_amd import SparseGSO, _lib, graphgen
L = _lib.lib(); dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for (N, B, W, weighted) in [(10000, 256, 32, False), (10000, 256, 32, True), (5000, 256, 32, False), (2000, 256, 32, False), (7777, 64, 32, False), (10239, 64, 32, True)]:
    A = graphgen.sbm(N, seed=0, normalize=not weighted) if not weighted else graphgen.sbm(N, seed=0, directed=True)
    if weighted:
        import numpy as np
        A = A.copy(); A.data = np.random.RandomState(0).uniform(0.5, 1.5, A.nnz)
    gso = SparseGSO([A]); plans = gso.plans(dev)
    P = B * W // 4
    X = torch.randn(P, N, 4, device=dev); Y0 = torch.empty_like(X); Y1 = torch.empty_like(X)
    res = {}
    for dma in (0, 1, 0, 1):
        assert L.gf_tune(b"panel_dma", dma) == 0
        ms = ctypes.c_float()
        _lib.check(L.gf_time_spmm_hop_panel(plans[0], 0, X.data_ptr(), (Y1 if dma else Y0).data_ptr(), P, 30, st, ctypes.byref(ms)))
        res.setdefault(dma, []).append(ms.value * 1e3)
    torch.cuda.synchronize()
    same = torch.equal(Y0, Y1)
    alg = 2 * P * N * 16 / 1e9
    print(f"N={N} B={B} weighted={weighted}: regs {min(res[0]):.1f} us ({alg/min(res[0])*1e6/8000*100:.1f}%)  dma {min(res[1]):.1f} us ({alg/min(res[1])*1e6/8000*100:.1f}%)  identical={same}", flush=True)
PY
