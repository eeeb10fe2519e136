#!/usr/bin/env python3
"""Time one column-panel (LDS) SpMM hop against the node-major (L2) hop on the benchmark shapes, with the panel knobs:
value-free stream on/off, bank-aware neighbour order on/off.  Usage: python tools/panel_sweep.py [cfg2 mid5k ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import numpy as np, torch
from alegnn_amd import SparseGSO, _lib, graphgen

SHAPES = {"cfg2": ("sbm", 10_000, 256, 32), "mid5k": ("sbm", 5_000, 256, 32), "n2k": ("sbm", 2_000, 256, 32),
          "cfg3": ("sbm", 1_682, 256, 64), "cfg2w": ("sbmw", 10_000, 256, 32)}
L = _lib.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream

def tune(**kw):
    for k, v in kw.items():
        assert L.gf_tune(k.encode(), v) == 0, k

for name in (sys.argv[1:] or ["cfg2", "cfg2w", "mid5k", "n2k"]):
    model, N, B, W = SHAPES[name]
    A = graphgen.sbm(N, seed=0)
    if model == "sbmw":                                   # same pattern, random weights: the general (6 bytes / edge) stream
        A = A.copy(); A.data = np.random.RandomState(0).uniform(0.1, 1.0, A.nnz) * A.data
    P = B * W // 4
    Xn = torch.randn(B, N, W, device=dev); Yn = torch.empty_like(Xn)
    Xp = torch.randn(P, N, 4, device=dev); Yp = torch.empty_like(Xp)
    nbytes = 2 * B * N * W * 4 + A.nnz * 8 + (N + 1) * 4
    print(f"== {name}: N={N} nnz={A.nnz} B={B} W={W} panels={P} algorithmic MB/hop={nbytes/1e6:.1f} roof@8TB/s={nbytes/8e12*1e6:.1f} us", flush=True)
    ms = ctypes.c_float()
    for order, srt in ((1, 1), (0, 1), (1, 0)):
        tune(panel_order=order, panel_sort=srt)
        gso = SparseGSO([A]); plans = gso.plans(dev)   # gso owns the plans
        ns, uni, cyc, fill = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
        _lib.check(L.gf_plan_panel_info(plans[0], 0, ctypes.byref(ns), ctypes.byref(uni), ctypes.byref(cyc), ctypes.byref(fill)))
        for useu in ((1, 0) if uni.value else (0,)):
            for stag in (0, 1, 2, 3):
                tune(panel_uniform=useu, spmm_store=0, panel_stagger=stag)
                _lib.check(L.gf_time_spmm_hop_panel(plans[0], 0, Xp.data_ptr(), Yp.data_ptr(), P, 20, st(), ctypes.byref(ms)))
                print(f"  panel  {ms.value*1e3:8.1f} us  {nbytes/ms.value/1e6:8.1f} GB/s  {100*nbytes/ms.value/1e6/8000:5.1f}%  order={order} "
                      f"sort={srt} (model {cyc.value:.2f} LDS cyc/step, ELL fill {fill.value:.2f}) uniform_stream={useu} stagger={stag}", flush=True)
    tune(panel_order=1, panel_sort=1, panel_uniform=1, spmm_store=2, panel_stagger=1)
    _lib.check(L.gf_time_spmm_hop(plans[0], 0, Xn.data_ptr(), Yn.data_ptr(), B, W, 20, st(), ctypes.byref(ms)))
    print(f"  L2     {ms.value*1e3:8.1f} us  {nbytes/ms.value/1e6:8.1f} GB/s  {100*nbytes/ms.value/1e6/8000:5.1f}%  node-major gather kernel (defaults)", flush=True)
