#!/usr/bin/env python3
"""Random EdgeVariantGF shapes (per-edge taps on the pattern of S, graphML.py:389-488 / :2511-2712) through EVGF_edges, forward + backward, against the
float64 CPU oracle (oracle/evgf_oracle.py): y, dx, the node-tap and edge-tap gradients, db; run-to-run bitwise.  Varies N, degree, directed / hub rows,
B (row widths of 4 .. 528 bytes: the tap kernels' lane groupings), G, F, K, the hybrid limit M < N.
usage: python tools/evgf_fuzz.py [cases] [seed]      (test infrastructure: the oracle is the checker)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd"), os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp, torch
from _util import FWD_RTOL, GRAD_RTOL, relerr
from alegnn_amd import EVGF_edges, EdgePattern, graphgen
from oracle import evgf_oracle as evo
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cu = lambda a, g=False: torch.tensor(a, device=dev, requires_grad=g)
bad = 0
for it in range(cases):
    N = int(rng.choice([37, 100, 333, 700, 1500, 4000, 9000]))
    B = int(rng.choice([1, 3, 4, 7, 12, 16, 33, 64, 70, 132]))
    G = int(rng.choice([1, 2, 3, 4, 8, 32]))
    F = int(rng.choice([1, 3, 4, 8, 12, 32]))
    K = int(rng.choice([1, 2, 3, 4]))
    M = N if rng.randint(2) else int(rng.randint(1, N))
    directed = bool(rng.randint(2))
    hub = N <= 1500 and rng.randint(4) == 0
    while F * G * K * N * 12 * 8 * max(1, B // 16) > 3e9:      # keep the oracle's states small
        B = max(1, B // 2)
    A = graphgen.sbm(N, seed=100 + it, directed=directed, avg_degree=float(rng.choice([3, 10, 20]))) if N >= 100 else graphgen.sbm(N, seed=100 + it, directed=directed, avg_degree=4.0)
    if hub:
        A = A.tolil(); A[5, :] = 1.0 / N; A[:, 5] = 1.0 / N; A = A.tocsr()
    pat = EdgePattern.from_gso(A, M)
    P = evo.ev_pattern(A, M)
    assert np.array_equal(pat.indices, P.indices)
    r = np.random.RandomState(it)
    wdiag = (r.uniform(-1, 1, (F, G, N)) * (np.arange(N) < M)).astype(np.float32)
    wedge = (r.uniform(-1, 1, (F, max(K - 1, 0), G, pat.nnzp)) * 0.3).astype(np.float32)
    x = r.randn(B, G, N).astype(np.float32)
    b = r.uniform(-1, 1, (F, 1)).astype(np.float32)
    dy = (r.randn(B, F, N) / np.sqrt(G * K)).astype(np.float32)
    outs = []
    for rep in range(2):
        wd, we, xt, bt = cu(wdiag, True), cu(wedge, True), cu(x, True), cu(b, True)
        y = EVGF_edges(pat, wd, we, xt, bt)
        y.backward(cu(dy))
        torch.cuda.synchronize()
        outs.append((y.detach(), xt.grad, wd.grad, we.grad, bt.grad))
    same = all(torch.equal(a, c) for a, c in zip(*outs))
    want = evo.evgf_sparse(P, wdiag, wedge, x, b)
    dx, dwd, dwe, db = evo.evgf_sparse_grads(P, wdiag, wedge, x, dy)
    y, gx, gwd, gwe, gb = [t.cpu().numpy() for t in outs[0]]
    e = [relerr(y, want), relerr(gx, dx), relerr(gwd, dwd), relerr(gwe, dwe) if K > 1 else 0.0, relerr(gb, db)]
    ok = same and e[0] < FWD_RTOL and all(v < GRAD_RTOL for v in e[1:])
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} N={N} B={B} G={G} F={F} K={K} M={M} directed={int(directed)} hub={int(hub)} nnz'={pat.nnzp}: y {e[0]:.1e} dx {e[1]:.1e} dwdiag {e[2]:.1e} dwedge {e[3]:.1e} db {e[4]:.1e} bitwise={same}", flush=True)
print("evgf fuzz: all ok" if not bad else f"evgf fuzz: {bad} BAD")
sys.exit(1 if bad else 0)
