#!/usr/bin/env python3
"""Where does the chain kernel's time go?  Times gf_khop_panel (HIP events inside the library) for the config-2 panel count over
graphs of different density (0 edges = the load / rewrite / store skeleton alone) and chain lengths; knobs as key=value.
Usage: chain_probe.py [N] [B] [key=val ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
os.environ.setdefault("GFHIP_EXPERIMENTS", "1")
import numpy as np, scipy.sparse as sp, torch
from alegnn_amd import SparseGSO, _lib, graphgen
args = [a for a in sys.argv[1:] if "=" not in a]
N = int(args[0]) if args else 10_000
B = int(args[1]) if len(args) > 1 else 256
L = _lib.lib()
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        assert L.gf_tune(k.encode(), int(v)) == 0, k
dev = torch.device("cuda:0")
W = 32
P = B * W // 4
ms = ctypes.c_float()
st = torch.cuda.current_stream().cuda_stream
print(f"N={N} panels={P}")
for deg in [float(d) for d in os.environ.get('PROBE_DEGS', '0,2.5,5,10,20').split(',')]:
    A = graphgen.sbm(N, avg_degree=deg, seed=0) if deg > 0 else sp.csr_matrix((N, N), dtype=np.float32)
    if os.environ.get('PROBE_WEIGHTED') == '1' and A.nnz:
        A = A.copy(); A.data = A.data * np.random.RandomState(1).uniform(0.5, 1.5, A.nnz)
    gso = SparseGSO([A])
    plans = gso.plans(dev)
    row = [f"deg {deg:4.1f} nnz {A.nnz:7d}"]
    for K in (2, 3, 5):
        Z = torch.randn(K, P, N, 4, device=dev)
        for chain in (2, 0):
            assert L.gf_tune(b"panel_chain", chain) == 0
            _lib.check(L.gf_time_khop_panel(plans, 1, 0, Z.data_ptr(), B, W, K, 10, st, ctypes.byref(ms)))
            row.append(f"K={K} {'chain' if chain else 'perhop'} {ms.value * 1e3:7.1f} us ({ms.value * 1e3 / (K - 1):6.1f}/hop)")
        del Z
    print("  ".join(row), flush=True)
