#!/usr/bin/env python3
"""Random GraphFilter layers on LARGE graphs (33 000 .. 204 800 nodes: the sizes where the node-major hops run as the MFMA sweep, with hub rows, wide
rows, the layout pass inside the fused launch) through the host layer, forward + backward, against the float64 CPU oracle (oracle/lsigf_oracle.py,
graphML.py:152-175 and :2125-2144 restated with sparse S).  Varies: graph kind, weights, G / F (also widths the sweep does not take), K (also 1),
E (1, 2), bias, Nin < N (the layer pads, graphML.py:2131-2135), fused ReLU (SelectionGNN's layers), batch sizes that are no multiple of 8.
y / dx on two batch entries, dh / db on the whole batch; tolerances of tests/_util.py.
usage: [FUZZ_N=100,1682,5000,...] [FUZZ_B=1,3,20,...] python tools/layer_fuzz.py [cases] [seed]   (other graph sizes / batch sizes: the panel, chain and
SELL-8 pipelines; test infrastructure: the oracle is the checker, never the path measured)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd"), os.path.join(ROOT, "tests")]
import numpy as np, scipy.sparse as sp, torch
from _util import FWD_RTOL, GRAD_RTOL, relerr
from alegnn_amd import _lib
from alegnn_amd.utils import graphML as gml
from oracle import lsigf_oracle as orc
L = _lib.lib()
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def graph(kind, n, deg, weighted, seed):
    r = np.random.RandomState(seed)
    if kind == "powerlaw":
        d = np.minimum(n // 8, (0.5 * deg / np.sqrt(np.maximum(r.uniform(size=n), 1e-9))).astype(np.int64))
        rows = np.repeat(np.arange(n), d)
        A = sp.csr_matrix((np.ones(rows.size), (rows, r.randint(0, n, size=rows.size))), shape=(n, n))
    else:
        m = n * deg // (1 if kind == "directed" else 2)
        A = sp.csr_matrix((np.ones(m), (r.randint(0, n, size=m), r.randint(0, n, size=m))), shape=(n, n))
        if kind == "er":
            A = A + A.T
    A = sp.csr_matrix(A)
    A.sum_duplicates()
    A.data[:] = r.uniform(0.2, 1.0, A.data.size) if weighted else 1.0
    A.data /= np.sqrt(max(1.0, A.nnz / n))                    # keeps S^k x of order one for a random x in the ordinary rows (hub rows stand out)
    return A


bad = 0
for it in range(cases):
    n = int(rng.choice([int(v) for v in os.environ["FUZZ_N"].split(",")] if os.environ.get("FUZZ_N") else [33000, 50000, 65536, 100000, 131000, 204800]))
    kind = str(rng.choice(["er", "er", "powerlaw", "directed"]))
    deg = int(rng.choice([4, 10, 16]))
    weighted = bool(rng.randint(2))
    G = int(rng.choice([32, 32, 64, 128, 16, 8, 96]))
    F = int(rng.choice([32, 32, 64, 16, 128]))
    K = int(rng.choice([1, 2, 3, 5]))
    E = int(rng.choice([1, 1, 1, 2]))
    B = int(rng.choice([int(v) for v in os.environ["FUZZ_B"].split(",")] if os.environ.get("FUZZ_B") else [5, 6, 8, 9, 12, 16]))
    bias = bool(rng.randint(4))
    Nin = n if rng.randint(3) else int(n * rng.uniform(0.5, 0.99))
    act = "relu" if rng.randint(3) == 0 else None
    while E * K * B * n * max(G, F) * 4 > 5e9:
        B = max(2, B // 2)
    S = [graph(kind, n, deg, weighted, 100 * it + e) for e in range(E)]
    torch.manual_seed(it)
    layer = gml.GraphFilter(G, F, K, E, bias)
    layer.addGSO(S if E > 1 else S[0])
    layer.fused_activation = act
    layer.to(dev)
    plans = layer._gso.plans(dev)
    sweep = [int(L.gf_spmm_hop_kernel(plans[0], op, B, w)) for op, w in ((0, G), (1, F))]
    x = torch.randn(B, G, Nin, device=dev, requires_grad=True)
    t0 = time.time()
    y = layer(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    torch.cuda.synchronize()
    w = layer.weight.detach().cpu().numpy()
    b = layer.bias.detach().cpu().numpy() if bias else None
    xs, dys, ys = x.detach().cpu().numpy(), dy.cpu().numpy(), y.detach().cpu().numpy()
    xp = np.zeros((B, G, n)); xp[:, :, :Nin] = xs
    sl = [0, B - 1]
    pre = orc.lsigf_sparse(w, S, xp[sl], b)[:, :, :Nin]
    ref = np.maximum(pre, 0.0) if act else pre
    e_y = relerr(ys[sl], ref)
    # the gradient that reaches the filter: dy where the output is positive (ReLU), zero on the padded nodes
    dfull = np.zeros((B, F, n))
    dfull[:, :, :Nin] = dys * (ys > 0) if act else dys     # (the mask from the layer's own output: a pre-activation of +-1e-8 may round either way)
    dx, _, _ = orc.lsigf_sparse_grads(w, S, xp[sl], b, dfull[sl])
    e_dx = relerr(x.grad[sl].cpu().numpy(), dx[:, :, :Nin])
    dh = np.zeros((F, E, K, G)); db = np.zeros((F, 1))
    for b0 in range(0, B, 4):
        z = orc.lsigf_taps_sparse(S, xp[b0:b0 + 4], K)
        dh += np.einsum("bfn,bekgn->fekg", dfull[b0:b0 + 4], z, optimize=True)
        db += dfull[b0:b0 + 4].sum(axis=(0, 2)).reshape(F, 1)
    e_dh = relerr(layer.weight.grad.cpu().numpy(), dh)
    e_db = relerr(layer.bias.grad.cpu().numpy(), db) if bias else 0.0
    ok = e_y < FWD_RTOL and e_dx < GRAD_RTOL and e_dh < GRAD_RTOL and e_db < GRAD_RTOL
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} n={n} {kind} deg={deg} {'w' if weighted else 'u'} G={G} F={F} K={K} E={E} B={B} bias={int(bias)} Nin={Nin} act={act} "
          f"sweep(fwd,bwd)={sweep} pipeline={int(L.gf_lsigf_pipeline(plans, E, G, F, K))}: y {e_y:.1e} dx {e_dx:.1e} dh {e_dh:.1e} db {e_db:.1e}", flush=True)
    del layer, x, y, dy, plans
    torch.cuda.empty_cache()
print("layer fuzz: all ok" if not bad else f"layer fuzz: {bad} BAD")
sys.exit(1 if bad else 0)
