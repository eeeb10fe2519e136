#!/usr/bin/env python3
"""SELL-8 against the MFMA sweep on ER graphs of several sizes / degrees / batch sizes, uniform and weighted: one hop and the K-1 = 4 hop
chain of gf_khop, ms per hop, bitwise comparison.  usage: GFHIP_EXPERIMENTS=1 python tools/msweep_sizes.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import numpy as np, scipy.sparse as sp, torch
from alegnn_amd import _lib, graphgen
from alegnn_amd.gso import SparseGSO
L = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def tune(**kw):
    for k, v in kw.items():
        assert L.gf_tune(k.encode(), int(v)) == 0, k
cases = [(33000, 10, False, 128), (40000, 10, False, 128), (60000, 10, False, 128), (80000, 10, False, 128), (100000, 10, False, 128),
         (100000, 10, True, 128), (60000, 10, True, 128), (100000, 10, False, 100), (100000, 10, False, 16), (100000, 10, False, 8),
         (100000, 20, False, 64), (100000, 4, False, 128), (50000, 10, False, 256)]
args = [a for a in sys.argv[1:] if "=" not in a]
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); tune(**{k: int(v)})
if args:
    cases = [tuple(int(x) for x in a.split(",")) for a in args]   # n,deg,weighted,B[,W]
K = 5
for case in cases:
    n, deg, weighted, B = case[:4]
    W = case[4] if len(case) > 4 else 32
    A = graphgen.er(n, avg_degree=float(deg), seed=1)
    if weighted:
        A = sp.csr_matrix(A); A.data = np.random.RandomState(2).uniform(0.1, 1.0, A.data.size)
    gso = SparseGSO([sp.csr_matrix(A)])
    plans = gso.plans(dev)
    Z = torch.empty(K, B, n, W, device=dev); Z[0].normal_()
    ms = ctypes.c_float(); out = []
    ref = None
    for name, kw in (("sell", dict(spmm_algo=3)), ("msweep", dict(spmm_algo=5))):
        tune(**kw)
        Z[1:].fill_(float("nan"))
        rc = L.gf_time_khop(plans, 1, 0, Z.data_ptr(), B, W, K, 5, st, ctypes.byref(ms))
        torch.cuda.synchronize()
        if rc != 0:
            out.append(f"{name}: n/a"); continue
        same = "" if ref is None else (" bitwise" if torch.equal(ref, Z[1:]) else " DIFFERENT")
        if ref is None: ref = Z[1:].clone()
        out.append(f"{name}: {ms.value / (K - 1):.4f} ms/hop{same}")
    print(f"N={n} deg={deg} {'weighted' if weighted else 'uniform'} B={B} W={W} nnz={A.nnz}: " + "  ".join(out), flush=True)
    del Z, ref, gso, plans
