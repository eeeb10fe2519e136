#!/usr/bin/env python3
"""Random shapes through the default node-major K-hop path (the MFMA sweep where it applies: rows of 32-128 columns, hub rows, one or two passes, fused /
layout pre-phase) against SELL-8: bit for bit, except rows the image splits (1e-6).  usage: GFHIP_EXPERIMENTS=1 python tools/msweep_fuzz.py [cases] [seed]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import numpy as np, scipy.sparse as sp, torch
from alegnn_amd import _lib
from alegnn_amd.gso import SparseGSO
L = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def tune(**kw):
    for k, v in kw.items():
        assert L.gf_tune(k.encode(), int(v)) == 0, k
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    n = int(rng.choice([33000, 41000, 50000, 65536, 77777, 100000, 102400, 131000, 180000, 204800]))
    kind = rng.choice(["er", "er", "powerlaw", "directed"])
    deg = int(rng.choice([2, 4, 6, 10, 16, 24]))
    weighted = bool(rng.randint(2))
    W = int(rng.choice([32, 32, 64, 96, 128]))
    K = int(rng.choice([2, 3, 5]))
    B = int(rng.choice([3, 8, 9, 16, 33]))
    B = max(B, -(-8 // (W // 32)))
    while K * B * n * W * 4 > 6e9: B = max(1, B // 2)
    if B * (W // 32) < 5: continue
    if kind == "powerlaw":
        d = np.minimum(n // 8, (0.5 * deg / np.sqrt(np.maximum(rng.uniform(size=n), 1e-9))).astype(np.int64))
    else:
        d = rng.poisson(deg, size=n)
    r = np.repeat(np.arange(n), d)
    A = sp.csr_matrix((np.ones(r.size), (r, rng.randint(0, n, size=r.size))), shape=(n, n))
    if kind == "er": A = A + A.T
    A = sp.csr_matrix((A > 0).astype(np.float64))
    A.data[:] = rng.uniform(-1, 1, size=A.data.size) if weighted else 1.0 / 32
    gso = SparseGSO([A])
    plans = gso.plans(dev)
    x0 = torch.randn(B, n, W, device=dev)
    for op in (0, 1):
        info = (ctypes.c_int32 * 8)()
        L.gf_debug_msweep_info(plans[0], op, info)
        uses = L.gf_spmm_hop_kernel(plans[0], op, B, W)
        outs = []
        for algo in (3, 0):
            tune(spmm_algo=algo)
            Z = torch.full((K, B, n, W), float("nan"), device=dev)
            Z[0].copy_(x0)
            _lib.check(L.gf_khop(plans, 1, op, Z.data_ptr(), B, W, K, st))
            torch.cuda.synchronize()
            outs.append(Z)
        ref, got = outs
        exact = bool(torch.equal(ref, got))
        err = float((ref - got).abs().max() / ref.abs().max())
        ok = exact or (info[5] > 0 and err < 2e-5)
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} n={n} {kind} deg={deg} {'w' if weighted else 'u'} W={W} K={K} B={B} op={op} sweep={uses} image={list(info)} exact={exact} relerr={err:.1e}", flush=True)
    del gso, plans, x0, outs, ref, got
print("fuzz:", "all ok" if bad == 0 else f"{bad} BAD")
sys.exit(1 if bad else 0)
