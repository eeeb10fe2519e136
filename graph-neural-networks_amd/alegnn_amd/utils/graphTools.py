"""Host-side graph utilities the GraphFilter/SelectionGNN path needs (numpy / scipy, CPU).

Mirrors the parts of the reference's alegnn/utils/graphTools.py that SelectionGNN touches:
  permIdentity / permDegree / permSpectralProxies / permEDS   (graphTools.py:990-1161)   node orderings
  computeNeighborhood                                         (graphTools.py:378-527)    K-hop neighbourhoods for MaxPoolLocal
re-written on sparse boolean reachability instead of Python list BFS.  Everything else in that file (graph
generation, GFT, Graclus coarsening, plotting) is outside the hot path.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

zeroTolerance = 1e-9   # graphTools.py:36: absolute values below this are "no edge"


def _as3d(S):
    S = np.asarray(S)
    assert len(S.shape) == 2 or len(S.shape) == 3
    if len(S.shape) == 2:
        assert S.shape[0] == S.shape[1]
        return S.reshape([1, S.shape[0], S.shape[1]]), True
    assert S.shape[1] == S.shape[2]
    return S, False


def _apply_order(S3, order, scalar):
    S3 = S3[:, order, :][:, :, order]
    return S3.reshape(S3.shape[1], S3.shape[2]) if scalar else S3


def permIdentity(S):
    """(S, order) with the identity ordering -- graphTools.py:990-1018."""
    S3, scalar = _as3d(S)
    N = S3.shape[1]
    return (S3.reshape(N, N) if scalar else S3), np.arange(N).tolist()


def permDegree(S):
    """Nodes from highest to lowest degree (column sums over all edge features) -- graphTools.py:1020-1050."""
    S3, scalar = _as3d(S)
    d = np.sum(np.sum(S3, axis=1), axis=0)
    order = np.flip(np.argsort(d), 0)
    return _apply_order(S3, order, scalar), order.tolist()


def permSpectralProxies(S):
    """Greedy spectral-proxies ordering with k = 8 -- graphTools.py:1052-1110."""
    S3, scalar = _as3d(S)
    simple = S3[0].copy() if scalar else np.mean(S3, axis=0)
    k = 8
    N = simple.shape[0]
    Sk = np.linalg.matrix_power(simple, k)
    STk = np.linalg.matrix_power(simple.conj().T, k)
    M = STk @ Sk
    nodes = []
    while len(nodes) < N:
        remaining = [n for n in range(N) if n not in nodes]
        w, V = np.linalg.eig(M[remaining][:, remaining])
        v = V[:, np.argmin(w.real)]
        nodes.append(remaining[int(np.argmax(np.square(np.absolute(v))))])
    return _apply_order(S3, np.array(nodes), scalar), nodes


def permEDS(S):
    """Experimentally-designed-sampling score ordering -- graphTools.py:1112-1161."""
    S3, scalar = _as3d(S)
    simple = S3[0].copy() if scalar else np.mean(S3, axis=0)
    _, V = np.linalg.eig(simple)
    kappa2 = np.square(np.max(np.absolute(V), axis=1))
    order = np.flip(np.argsort(kappa2), 0)
    return _apply_order(S3, order, scalar), order.tolist()


def edge_pattern(S) -> sp.csr_matrix:
    """Boolean N x N CSR with an entry wherever sum_e |S_e| > zeroTolerance (graphTools.py:424-432).
    S: dense [E,N,N] / [N,N] array, scipy sparse, or a list of scipy sparse (one per edge feature)."""
    if sp.issparse(S):
        S = [S]
    if isinstance(S, (list, tuple)):
        acc = None
        for m in S:
            a = abs(sp.csr_matrix(m))
            acc = a if acc is None else acc + a
        acc = sp.csr_matrix(acc)
        acc.data = (acc.data > zeroTolerance).astype(np.int8)
        acc.eliminate_zeros()
        return acc.astype(bool)
    S3, _ = _as3d(S)
    return sp.csr_matrix(np.sum(np.abs(S3), axis=0) > zeroTolerance)


def computeNeighborhood(S, K, N='all', nb='all', outputType='list'):
    """K-hop neighbourhoods (node itself included) of the first N nodes, restricted to nodes < nb.
    Same contract as graphTools.py:378-527; 'matrix' output pads each row with the node's own index."""
    assert outputType == 'list' or outputType == 'matrix'
    A = edge_pattern(S)
    n = A.shape[0]
    N = n if N == 'all' else N
    nb = n if nb == 'all' else nb
    assert K >= 0 and 0 <= N <= n and 0 <= nb <= n
    hop = (A + sp.identity(n, dtype=bool, format='csr')).astype(np.int32)
    reach = sp.identity(n, dtype=np.int32, format='csr')[:N]
    for _ in range(K):
        reach = reach @ hop
        reach.data[:] = 1
    reach = sp.csr_matrix(reach)
    reach.sort_indices()
    neighbors = []
    for i in range(N):
        idx = reach.indices[reach.indptr[i]:reach.indptr[i + 1]]
        neighbors.append([int(j) for j in idx if j < nb])
    if outputType == 'matrix':
        width = max(len(v) for v in neighbors)
        out = np.empty((N, width), dtype=np.int64)
        for i, v in enumerate(neighbors):
            out[i, :len(v)] = v
            out[i, len(v):] = i
        return out
    return neighbors
