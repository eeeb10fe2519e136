#!/usr/bin/env python3
"""Config-3-class panel hop (N = 1682 kNN-10 weighted GSO, signal width 64, batch 256): where does the hop's time go?
Times one hop (gf_time_spmm_hop_panel, HIP events inside the library) for the weighted image, the same pattern with equal weights
(value-free stream), an empty graph (load + store skeleton) and the knob variants given as v:key=val+key=val; prints the plan's
modelled LDS cycles per gather step.   Usage: panel_w_probe.py [N] [W] [B] [v:variant ...]   (PROBE_ONLY=1: one launch of the default
weighted hop, for counter passes)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
os.environ.setdefault("GFHIP_EXPERIMENTS", "1")
import numpy as np, scipy.sparse as sp, torch
from alegnn_amd import SparseGSO, _lib, graphgen
args = [a for a in sys.argv[1:] if not a.startswith("v:")]
variants = [a[2:] for a in sys.argv[1:] if a.startswith("v:")]
N = int(args[0]) if args else 1682
W = int(args[1]) if len(args) > 1 else 64
B = int(args[2]) if len(args) > 2 else 256
L = _lib.lib()
dev = torch.device("cuda:0")
P = B * W // 4
st = torch.cuda.current_stream().cuda_stream
ms = ctypes.c_float()
A = graphgen.knn_weighted(N, k=10, seed=0)
Au = A.copy(); Au.data[:] = 0.37
graphs = [("weighted", A)] if os.environ.get("PROBE_ONLY") else [("weighted", A), ("equal-w", Au), ("empty", sp.csr_matrix((N, N), dtype=np.float32))]
KH = os.environ.get("PROBE_KHOP") == "1"
X = torch.randn(P, N, 4, device=dev)
Z = torch.randn(5, P, N, 4, device=dev) if KH else None
Y = torch.empty_like(X)
alg = 2 * P * N * 16
print(f"N={N} W={W} B={B} panels={P} nnz={A.nnz} algorithmic bytes/hop={alg}")
DEFAULTS = dict(panel_np=0, panel_chain=1, panel_rotate=1, panel_split=0, panel_grid=0, panel_db=0, panel_thr=0, panel_loaders=0)
for name, M in graphs:
    gso = SparseGSO([M]); plans = gso.plans(dev)
    ns, uni, cyc, fill = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
    _lib.check(L.gf_plan_panel_info(plans[0], 0, ctypes.byref(ns), ctypes.byref(uni), ctypes.byref(cyc), ctypes.byref(fill)))
    print(f"{name:9s} slices={ns.value} uniform={uni.value} modelled LDS cycles/step={cyc.value:.3f} fill={fill.value:.3f}")
    for v in [""] + variants:
        kv = dict(DEFAULTS)
        for t in filter(None, v.split("+")):
            k, x = t.split("="); kv[k] = int(x)
        for k, x in kv.items():
            _lib.check(L.gf_tune(k.encode(), x), k)
        it = 1 if os.environ.get("PROBE_ONLY") else 20
        if KH:      # the bench's measure: the K - 1 = 4 per-hop launches of a tap stack back to back (every hop reads what the last one wrote)
            _lib.check(L.gf_tune(b"panel_chain", 0))
            _lib.check(L.gf_time_khop_panel(plans, 1, 0, Z.data_ptr(), B, W, 5, it, st, ctypes.byref(ms)))
            ms.value /= 4
            Y.copy_(Z[4])
        else:
            _lib.check(L.gf_time_spmm_hop_panel(plans[0], 0, X.data_ptr(), Y.data_ptr(), P, it, st, ctypes.byref(ms)))
        torch.cuda.synchronize()
        if not v:
            Yref = Y.clone()
        same = "bitwise = default" if torch.equal(Y, Yref) else f"DIFFERS from default (max abs {float((Y - Yref).abs().max()):.3g})"
        print(f"   {v or 'default':28s} {ms.value * 1e3:7.1f} us/hop   {alg / ms.value / 1e6:7.0f} GB/s = {alg / ms.value / 8e9 * 100:5.1f} %   {same}", flush=True)
        Y.fill_(float("nan"))
    for k, x in DEFAULTS.items():
        _lib.check(L.gf_tune(k.encode(), x), k)
