"""CPU oracle for the edge-variant graph filter (EVGF / EdgeVariantGF).  TEST INFRASTRUCTURE ONLY.

Restates, with the filter matrices stored PER EDGE (only on-pattern entries exist), the arithmetic of

    alegnn/utils/graphML.py:389-488    EVGF(S=Phi, x, b)         v_0 = Phi_0 x, v_k = Phi_k v_{k-1}, y = sum_{e,k,g} v_k + b
    alegnn/utils/graphML.py:2608-2668  EdgeVariantGF.addGSO      pattern = (|S| + I > 1e-9) & hybrid mask; tap 0 = identity & mask
    alegnn/utils/graphML.py:2670-2698  EdgeVariantGF.forward     Phi = weightEV * pattern; EVGF (+ LSIGF with the same bias if M < N)

The reference holds Phi as a dense [F,E,K,G,N,N] tensor (3e13 bytes at BASELINE config 5); here tap 0 is the diagonal
wdiag[F,G,N] and taps k >= 1 are wedge[F,K-1,G,nnzp] on the CSR pattern (one edge feature per call, column convention
Phi @ x as in graphML.py:464, 475).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.

Parity pin: tests/golden/evgf_*.npz are outputs (and autograd gradients) of the real reference EdgeVariantGF, produced
by tests/golden/make_golden.py; tests/test_oracle_golden.py checks every function below against them.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

ZERO_TOLERANCE = 1e-9          # graphML.py:27 (zeroTolerance)


def ev_pattern(S2d, M: int) -> sp.csr_matrix:
    """Sparsity pattern of taps k >= 1 for one edge feature (graphML.py:2617-2643):
    (|S| + I > zeroTolerance) restricted to entries with row < M or col < M.  Returns a CSR matrix of ones with
    sorted column indices; its entry order defines the value index p of wedge[..., p]."""
    S2d = sp.csr_matrix(S2d)
    N = S2d.shape[0]
    P = (abs(S2d) + sp.identity(N, format="csr")) > ZERO_TOLERANCE
    P = sp.coo_matrix(P)
    keep = ((P.row < M) | (P.col < M)) & (P.data != 0)
    P = sp.csr_matrix((np.ones(int(keep.sum()), dtype=np.float64), (P.row[keep], P.col[keep])), shape=(N, N))
    P.sum_duplicates()
    P.sort_indices()
    return P


def ev_split_dense_weight(weightEV_e, P: sp.csr_matrix, M: int):
    """weightEV_e [F,K,G,N,N] (one edge feature of the reference parameter) -> (wdiag [F,G,N], wedge [F,K-1,G,nnzp]):
    the entries the reference's mask keeps (graphML.py:2653-2676).  Diagonal entries n >= M are masked to zero."""
    F_, K, G, N, _ = weightEV_e.shape
    idx = np.arange(N)
    wdiag = weightEV_e[:, 0][:, :, idx, idx] * (idx < M)
    coo = P.tocoo()                                  # CSR order == entry order p
    wedge = weightEV_e[:, 1:][:, :, :, coo.row, coo.col]
    return np.ascontiguousarray(wdiag), np.ascontiguousarray(wedge)


def evgf_sparse(P: sp.csr_matrix, wdiag, wedge, x, b=None, dtype=np.float64, return_states=False):
    """One edge feature.  wdiag [F,G,N], wedge [F,K-1,G,nnzp], x [B,G,N], b [F,1]|None -> y [B,F,N]."""
    wdiag = np.asarray(wdiag, dtype=dtype)
    wedge = np.asarray(wedge, dtype=dtype)
    x = np.asarray(x, dtype=dtype)
    F_, G, N = wdiag.shape
    K = wedge.shape[1] + 1
    B = x.shape[0]
    y = np.zeros((B, F_, N), dtype=dtype)
    V = np.zeros((K, F_, G, B, N), dtype=dtype) if return_states else None
    for f in range(F_):
        for g in range(G):
            v = wdiag[f, g][None, :] * x[:, g, :]                               # Phi_0 x_g   (:464, diagonal)
            y[:, f] += v
            if return_states:
                V[0, f, g] = v
            for k in range(1, K):
                Phi = sp.csr_matrix((wedge[f, k - 1, g], P.indices, P.indptr), shape=(N, N))
                v = np.ascontiguousarray((Phi @ v.T).T)                         # Phi_k v_{k-1}   (:475)
                y[:, f] += v                                                    # sums over k and g (:481-485)
                if return_states:
                    V[k, f, g] = v
    if b is not None:
        y = y + np.asarray(b, dtype=dtype).reshape(1, F_, -1)
    return (y, V) if return_states else y


def evgf_sparse_grads(P: sp.csr_matrix, wdiag, wedge, x, dy, dtype=np.float64):
    """Analytic backward of evgf_sparse.  Returns (dx, dwdiag, dwedge, db[F,1])."""
    wdiag = np.asarray(wdiag, dtype=dtype)
    wedge = np.asarray(wedge, dtype=dtype)
    x = np.asarray(x, dtype=dtype)
    dy = np.asarray(dy, dtype=dtype)
    F_, G, N = wdiag.shape
    K = wedge.shape[1] + 1
    _, V = evgf_sparse(P, wdiag, wedge, x, None, dtype, return_states=True)
    coo = P.tocoo()
    rows, cols = coo.row, coo.col
    dx = np.zeros_like(x)
    dwdiag = np.zeros_like(wdiag)
    dwedge = np.zeros_like(wedge)
    for f in range(F_):
        for g in range(G):
            u = dy[:, f, :]                                                     # u_{K-1} = dy_f
            for k in range(K - 1, 0, -1):
                dwedge[f, k - 1, g] = np.einsum("bp,bp->p", u[:, rows], V[k - 1, f, g][:, cols])   # dPhi_k = u_k v_{k-1}^T on P
                Phi = sp.csr_matrix((wedge[f, k - 1, g], P.indices, P.indptr), shape=(N, N))
                u = np.ascontiguousarray((Phi.T @ u.T).T) + dy[:, f, :]         # u_{k-1} = Phi_k^T u_k + dy_f
            dwdiag[f, g] = np.einsum("bn,bn->n", u, x[:, g, :])
            dx[:, g, :] += wdiag[f, g][None, :] * u
    db = dy.sum(axis=(0, 2)).reshape(F_, 1)
    return dx, dwdiag, dwedge, db


def edge_variant_gf_forward(S, weightEV, weightLSI, bias, M, x, dtype=np.float64):
    """EdgeVariantGF.forward (graphML.py:2670-2698) from the reference's DENSE parameters, evaluated sparsely.
    S [E,N,N]; weightEV [F,E,K,G,N,N]; weightLSI [F,E,K,G]|None; bias [F,1]|None; x [B,G,Nin]."""
    from .lsigf_oracle import lsigf_sparse
    S = np.asarray(S, dtype=dtype)
    E, N, _ = S.shape
    B, G, Nin = x.shape
    xp = np.zeros((B, G, N), dtype=dtype)
    xp[:, :, :Nin] = x                                                          # :2678-2680
    y = 0.0
    for e in range(E):
        P = ev_pattern(S[e], M)
        wdiag, wedge = ev_split_dense_weight(np.asarray(weightEV, dtype=dtype)[:, e], P, M)
        y = y + evgf_sparse(P, wdiag, wedge, xp, None, dtype)
    if bias is not None:
        y = y + np.asarray(bias, dtype=dtype).reshape(1, -1, 1)                 # EVGF adds b once (:486-487)
    if M < N:
        y = y + lsigf_sparse(weightLSI, S, xp, bias, dtype)                     # bias a second time (:2686)
    return y[:, :, :Nin]                                                        # :2696-2697
