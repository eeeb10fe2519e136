#!/usr/bin/env python3
"""Time one SpMM hop (gf_time_spmm_hop, HIP events) over the tuning knobs of libgfhip, for the benchmark shapes.
Usage: python tools/spmm_sweep.py [cfg2 cfg4 ...]   -> table on stdout"""
import ctypes, itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import numpy as np, torch
from alegnn_amd import SparseGSO, _lib, graphgen

SHAPES = {"cfg2": ("sbm", 10_000, 256, 32), "cfg4": ("er", 100_000, 128, 32), "cfg2w64": ("sbm", 10_000, 128, 64),
          "mid": ("sbm", 30_000, 256, 32)}
L = _lib.lib()
L.gf_tune.restype = ctypes.c_int
L.gf_tune.argtypes = [ctypes.c_char_p, ctypes.c_int32]
dev = torch.device("cuda:0")

def tune(**kw):
    for k, v in kw.items():
        assert L.gf_tune(k.encode(), v) == 0, k

def time_hop(plan, X0, X1, B, W, iters=20):
    ms = ctypes.c_float()
    _lib.check(L.gf_time_spmm_hop(plan, 0, X0.data_ptr(), X1.data_ptr(), B, W, iters, torch.cuda.current_stream().cuda_stream, ctypes.byref(ms)))
    return ms.value

for name in (sys.argv[1:] or ["cfg2", "cfg4"]):
    model, N, B, W = SHAPES[name]
    A = (graphgen.sbm if model == "sbm" else graphgen.er)(N, seed=0)
    gso = SparseGSO([A]); plans = gso.plans(dev)
    X0 = torch.randn(B, N, W, device=dev); X1 = torch.empty_like(X0)
    nbytes = 2 * B * N * W * 4 + A.nnz * 8 + (N + 1) * 4
    print(f"== {name}: N={N} nnz={A.nnz} B={B} W={W}  algorithmic MB/hop={nbytes/1e6:.1f}  roof@8TB/s={nbytes/8e12*1e6:.1f} us")
    rows = []
    for bt, store in itertools.product((1, 2), (1,)):
        tune(spmm_algo=1, spmm_bt=bt, spmm_xcd=1, spmm_store=store)
        rows.append((time_hop(plans[0], X0, X1, B, W), f"algo=csr  bt={bt} store={store}"))
    for bt, spw, store, pf, ucap in itertools.product((1, 2), (1, 2), (1, 2), (0, 6, 10, 16, 24), (8, 16)):
        tune(spmm_algo=0, spmm_bt=bt, spmm_spw=spw, spmm_xcd=1, spmm_store=store, spmm_load=0, spmm_ucap=ucap, spmm_pf=pf)
        rows.append((time_hop(plans[0], X0, X1, B, W), f"algo=sell bt={bt} spw={spw} store={store} pf={pf} ucap={ucap}"))
    tune(spmm_algo=0, spmm_bt=0, spmm_spw=0, spmm_xcd=1, spmm_store=2, spmm_load=0, spmm_ucap=0, spmm_pf=-1)
    rows.append((time_hop(plans[0], X0, X1, B, W), "DEFAULT (heuristics)"))
    for ms, label in sorted(rows):
        print(f"  {ms*1e3:9.1f} us  {nbytes/ms/1e6:8.1f} GB/s  {100*nbytes/ms/1e6/8000:5.1f}%  {label}")
