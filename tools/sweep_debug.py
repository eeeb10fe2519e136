#!/usr/bin/env python3
"""Where does the sweep kernel (a variant) differ from SELL-8?  usage: sweep_debug.py [n] [deg] [B] key=val..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd"), os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp, torch
from alegnn_amd import _lib
from alegnn_amd.gso import SparseGSO
args = [a for a in sys.argv[1:] if "=" not in a]
n, deg, B = (int(args[0]) if args else 12000), (int(args[1]) if len(args) > 1 else 4), (int(args[2]) if len(args) > 2 else 9)
L = _lib.lib()
rng = np.random.RandomState(0)
r = np.repeat(np.arange(n), deg); c = rng.randint(0, n, size=r.size)
A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(n, n)); A = ((A + A.T) > 0).astype(np.float64); A.setdiag(0); A.eliminate_zeros(); A = sp.csr_matrix(A * 0.0625)
gso = SparseGSO([A]); plans = gso.plans(torch.device("cuda:0"))
X = torch.randn(B, n, 32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def hop(**kw):
    for k, v in kw.items():
        assert L.gf_tune(k.encode(), int(v)) == 0
    out = torch.full((B, n, 32), float("nan"), device="cuda")
    _lib.check(L.gf_spmm_hop(plans[0], 0, X.data_ptr(), out.data_ptr(), B, 32, st)); torch.cuda.synchronize()
    return out
ref = hop(spmm_algo=3)
kw = dict(kv.split("=") for kv in sys.argv[1:] if "=" in kv)
got = hop(spmm_algo=4, **{k: int(v) for k, v in kw.items()})
bad = (got != ref) | torch.isnan(got)
print("mismatching elements:", int(bad.sum()), "of", bad.numel(), " rows:", int(bad.any(dim=2).sum()), "of", B * n, " nan:", int(torch.isnan(got).sum()))
rows = bad.any(dim=2).nonzero()
print("first bad (b, row):", rows[:10].tolist())
if len(rows):
    b, i = rows[0].tolist()
    print("got", got[b, i, :6].tolist(), "\nref", ref[b, i, :6].tolist(), "\ndeg", A.T.tocsr()[i].nnz)
    d = (got - ref)[b, i]
    print("diff lanes nonzero:", int((d != 0).sum()), " per-batch bad rows:", bad.any(dim=2).sum(dim=1).tolist())
    deg_all = np.asarray(A.T.tocsr().getnnz(axis=1))
    br = bad.any(dim=2)[b].cpu().numpy()
    print("mean degree of bad rows", deg_all[br].mean(), "of all", deg_all.mean())
if len(rows):
    At = A.T.tocsr()
    for (b, i) in rows[:3].tolist():
        cols = At[i].indices
        d = ((got - ref)[b, i] / 0.0625).cpu().numpy()
        print(f"--- bad (b={b}, row={i}) cols={cols.tolist()}")
        print("diff/uval [0:8]:", np.round(d[:8], 5).tolist())
        Xb = X[b].cpu().numpy()
        for cc in cols:
            print(f"   X[{cc}][0:8] =", np.round(Xb[cc][:8], 5).tolist())
        # does the diff equal  X[c'] - X[c]  for a neighbour c and some other row c'?
        for cc in cols:
            cand = d + Xb[cc]
            hit = np.where(np.abs(Xb - cand[None, :]).max(axis=1) < 1e-4)[0]
            if len(hit):
                print(f"   explained: neighbour {cc} was replaced by row {hit.tolist()}")
            # shifted-lane hypothesis: lane l got element 2l, or element l of another row
        print("   got/uval [0:8]", np.round((got[b, i] / 0.0625).cpu().numpy()[:8], 5).tolist())
import ctypes
ms = ctypes.c_float()
out = torch.empty_like(X)
for tag, kws in (("SELL-8", dict(spmm_algo=3)), ("sweep", dict(spmm_algo=4))):
    for k, v in kws.items():
        assert L.gf_tune(k.encode(), int(v)) == 0
    _lib.check(L.gf_time_spmm_hop(plans[0], 0, X.data_ptr(), out.data_ptr(), B, 32, 10, st, ctypes.byref(ms)))
    print(f"timing n={n} nnz={A.nnz} B={B}: {tag}: {ms.value:.4f} ms per hop")
