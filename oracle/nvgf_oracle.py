"""CPU oracle for the node-variant graph filter (NVGF / NodeVariantGF).  TEST INFRASTRUCTURE ONLY.

Restates, on the CPU (numpy / scipy / torch-CPU), what the reference computes in

    alegnn/utils/graphML.py:293-387     NVGF(h, S, x, b)
    alegnn/utils/graphML.py:2402-2498   NodeVariantGF.addGSO (copyNodes) and .forward (expand taps, zero-pad, keep Nin nodes)

Only tests/ may import it (see oracle/lsigf_oracle.py for the rules).  Parity pin: tests/golden/nvgf_*.npz, produced by
tests/golden/make_golden.py from the real reference (outputs and torch-autograd gradients, float64);
tests/test_oracle_golden.py checks both functions below against them.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch


def nvgf_dense(h: torch.Tensor, S: torch.Tensor, x: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
    """Literal restatement of graphML.py:341-387 in torch (autograd gives the reference's backward).
    h [F,E,K,G,N], S [E,N,N], x [B,G,N], b [F,1]|[F,N]|None -> y [B,F,N]."""
    F_, E, K, G, N = h.shape
    assert S.shape[0] == E and S.shape[1] == S.shape[2] == N       # :346-347
    B = x.shape[0]
    assert x.shape[1] == G and x.shape[2] == N                     # :350-351
    cur = x.reshape(B, 1, G, N)                                    # :356
    Sb = S.reshape(1, E, N, N)
    taps = [x.reshape(B, 1, 1, G, N).repeat(1, E, 1, 1, 1)]        # :359  k = 0, the same x for every e
    for _ in range(1, K):
        cur = torch.matmul(cur, Sb)                                # :363  row-vector product
        taps.append(cur.reshape(B, E, 1, G, N))
    z = torch.cat(taps, dim=2).reshape(B, 1, E, K, G, N)           # :366-371
    y = (z * h.reshape(1, F_, E, K, G, N)).sum(dim=4).sum(dim=3).sum(dim=2)   # :376-381  sum over g, k, e
    if b is not None:
        y = y + b                                                  # :383-384
    return y


def nvgf_sparse(h, mats, x, b=None):
    """The same arithmetic on scipy CSR (one matrix per edge feature) in numpy -- any N.  x may have Nin <= N nodes
    (NodeVariantGF.forward's zero padding, :2487-2489); the output keeps Nin nodes (:2496-2497)."""
    F_, E, K, G, N = h.shape
    B, _, Nin = x.shape
    xp = np.zeros((B, G, N), dtype=x.dtype)
    xp[:, :, :Nin] = x
    y = np.zeros((B, F_, N), dtype=np.result_type(h.dtype, x.dtype))
    for e in range(E):
        St = sp.csr_matrix(mats[e]).T.tocsr()                      # (x @ S)[b,g,:] = S^T x[b,g,:]
        cur = xp.reshape(B * G, N).T                               # [N, B*G]
        for k in range(K):
            if k > 0:
                cur = St @ cur
            zk = np.ascontiguousarray(cur.T).reshape(B, G, N)
            y += np.einsum("bgn,fgn->bfn", zk, h[:, e, k])
    if b is not None:
        y = y + b
    return y[:, :, :Nin]


def copy_nodes(mats, M):
    """NodeVariantGF.addGSO's tap assignment (:2411-2468) by plain BFS on the union pattern (|S_e| summed over e, self
    included): node n >= M copies the smallest node < M within the smallest number of hops that reaches any."""
    N = mats[0].shape[0]
    if M >= N:
        return list(range(N))
    pat = sum(abs(sp.csr_matrix(m)) for m in mats)
    pat = sp.csr_matrix(pat)
    pat.data = (pat.data > 1e-9).astype(np.float64)                # zeroTolerance, graphTools.py:36
    pat.eliminate_zeros()
    pat = sp.csr_matrix(pat + sp.identity(N))
    out = list(range(M))
    for n in range(M, N):
        reach = np.zeros(N, dtype=bool)
        reach[n] = True
        for _ in range(N):
            reach = (pat[reach].sum(axis=0).A1 > 0) | reach
            hits = np.flatnonzero(reach[:M])
            if hits.size:
                break
        out.append(int(hits.min()))
    return out
