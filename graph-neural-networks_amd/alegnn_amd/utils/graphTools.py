"""Host-side graph utilities the GraphFilter/SelectionGNN path needs (numpy / scipy, CPU).

Mirrors the parts of the reference's alegnn/utils/graphTools.py that SelectionGNN touches:
  permIdentity / permDegree / permSpectralProxies / permEDS   (graphTools.py:990-1161)   node orderings
  computeNeighborhood                                         (graphTools.py:378-527)    K-hop neighbourhoods for MaxPoolLocal
  coarsen / metis / metis_one_level / compute_perm / perm_adjacency / permCoarsening
                                                              (graphTools.py:1337-1614)  Graclus multilevel coarsening
re-written on sparse boolean reachability instead of Python list BFS, and on CSR segments instead of per-entry Python
loops.  Everything else in that file (graph generation, GFT, plotting) is outside the hot path (SURVEY.md section 2,
item 17).
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


# ---- the same orderings for GSOs that only exist sparse (superset: the reference's functions take a dense array) ------------------
kDenseOrderingMaxNodes = 4096   # EDS / SpectralProxies need the eigendecomposition of the dense N x N matrix: O(N^3) host work


def perm_sparse(mats, name):
    """(list of reordered CSR matrices, order) for a list of scipy sparse N x N matrices (one per edge feature) and an ordering name
    (None | 'Degree' | 'EDS' | 'SpectralProxies': graphTools.py:990-1161, as architectures.py:203-256 applies them).
    'Degree' works on the sparse matrices (column sums over all edge features, highest first: O(nnz)).  Where several nodes have the
    SAME degree their relative order is whatever numpy's argsort makes of the sums, as in the reference; a sum accumulated over a sparse
    column can differ from the dense one in the last bit, so such ties may come out in another (equally valid) order than on the
    dense copy of the same graph.  'EDS' / 'SpectralProxies' need the dense eigendecomposition: up to kDenseOrderingMaxNodes nodes the
    matrices are densified for the computation of the ORDER only; beyond that they are refused."""
    mats = [sp.csr_matrix(m) for m in mats]
    n = mats[0].shape[0]
    if name is None:
        return mats, list(range(n))
    if name == 'Degree':
        d = np.zeros(n)
        for m in mats:
            d = d + np.asarray(m.sum(axis=0)).ravel()                     # column sums (np.sum(S, axis=1) of the [E,N,N] array), summed over e
        order = np.flip(np.argsort(d), 0)
    elif name in ('EDS', 'SpectralProxies'):
        if n > kDenseOrderingMaxNodes:
            raise NotImplementedError(f"order='{name}' needs the eigendecomposition of the dense {n} x {n} GSO (O(N^3) host work); it is "
                                      f"available up to {kDenseOrderingMaxNodes} nodes -- use order='Degree' / None, or reorder the graph beforehand")
        dense = np.stack([np.asarray(m.todense()) for m in mats])
        _, order = (permEDS if name == 'EDS' else permSpectralProxies)(dense)
        order = np.asarray(order)
    else:
        raise ValueError(f"unknown ordering '{name}'")
    out = [sp.csr_matrix(m[order][:, order]) for m in mats]
    return out, [int(i) for i in order]


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


# ---------------------------------------------------------------------------------------------------------------
# Graclus multilevel coarsening (SelectionGNN(coarsening=True)), graphTools.py:1337-1614
# ---------------------------------------------------------------------------------------------------------------
def metis_one_level(rr, cc, vv, rid, weights):
    """One greedy Graclus pairing pass -- graphTools.py:1453-1499.

    ``rr`` (sorted), ``cc``, ``vv``: the graph's entries by row; ``rid``: visiting order; ``weights``: node degrees.
    Every unmarked node, in visiting order, is merged with the unmarked neighbour that maximises
    w_ij (1/d_i + 1/d_j) (strictly positive, first maximum wins) or stays alone.  Returns cluster_id[N].

    The reference's row bookkeeping (:1467-1472) is reproduced as is, because it decides the clusters: row lengths are
    accumulated one entry late, so the first row's candidate list also holds the first entry of the second row and
    the last row's list misses its last entry; rows are numbered by order of appearance in ``rr`` (a node without
    entries shifts the numbering of the rows after it)."""
    rr = np.asarray(rr)
    nnz = rr.shape[0]
    N = int(rr[nnz - 1]) + 1
    first = np.flatnonzero(np.concatenate(([True], rr[1:] != rr[:-1])))     # first entry of every row present
    m = first.shape[0]
    rowstart = np.zeros(N, dtype=np.int64)
    rowlength = np.zeros(N, dtype=np.int64)
    rowstart[:m] = first
    rowlength[:m] = np.diff(np.concatenate((first, [nnz])))
    if m > 1:
        rowlength[0] += 1
        rowlength[m - 1] -= 1
    inv = 1.0 / np.asarray(weights, dtype=np.float64)
    marked = np.zeros(N, dtype=bool)
    cluster_id = np.zeros(N, dtype=np.int32)
    clusters = 0
    for t in (int(v) for v in rid[:N]):
        if marked[t]:
            continue
        marked[t] = True
        seg = slice(rowstart[t], rowstart[t] + rowlength[t])
        nb = cc[seg]
        score = np.where(marked[nb], 0.0, vv[seg] * (inv[t] + inv[nb]))
        score = np.where(np.isnan(score), 0.0, score)                       # "nan > wmax" is False in the reference loop
        cluster_id[t] = clusters
        if score.shape[0] > 0:
            j = int(np.argmax(score))
            if score[j] > 0.0:
                cluster_id[nb[j]] = clusters
                marked[nb[j]] = True
        clusters += 1
    return cluster_id


def metis(W, levels, rid=None):
    """``levels`` successive pairings of a symmetric sparse W -- graphTools.py:1369-1450.
    Returns (graphs[levels+1], parents[levels]); parents[i][n] = cluster of node n of graphs[i] in graphs[i+1].
    The first visiting order is ``np.random.permutation(range(N))`` (seed numpy's global RNG for reproducibility),
    later ones sort the coarse nodes by degree."""
    N = W.shape[0]
    if rid is None:
        rid = np.random.permutation(range(N))
    degree = W.sum(axis=0) - W.diagonal()                                   # self loops left out at the finest level only
    graphs, parents = [W], []
    for _ in range(levels):
        weights = np.array(degree).squeeze()
        rows, cols, vals = sp.find(W)
        by_row = np.argsort(rows)                                           # same (unstable) sort as :1413: it fixes tie order
        rr, cc, vv = rows[by_row], cols[by_row], vals[by_row]
        cluster_id = metis_one_level(rr, cc, vv, rid, weights)
        parents.append(cluster_id)
        Nc = int(cluster_id.max()) + 1
        W = sp.csr_matrix((vv, (cluster_id[rr], cluster_id[cc])), shape=(Nc, Nc))   # duplicates add up
        W.eliminate_zeros()
        graphs.append(W)
        degree = W.sum(axis=0)
        rid = np.argsort(np.array(degree).squeeze())
    return graphs, parents


def compute_perm(parents):
    """Node orderings, finest level first, under which clusters are consecutive pairs (a binary tree) --
    graphTools.py:1501-1547.  Single nodes get a fake sibling, fake nodes two fake children; fake ids start at the
    level's real node count."""
    if len(parents) == 0:
        return []
    layers = [list(range(int(max(parents[-1])) + 1))]
    for parent in parents[::-1]:
        parent = np.asarray(parent)
        by_cluster = np.argsort(parent, kind="stable")                      # children of a cluster in increasing id
        lo = np.searchsorted(parent[by_cluster], np.arange(len(layers[-1]) + 1))
        fake = len(parent)
        layer = []
        for i in layers[-1]:
            pair = [int(c) for c in by_cluster[lo[i]:lo[i + 1]]] if i < len(lo) - 1 else []
            assert len(pair) <= 2
            while len(pair) < 2:
                pair.append(fake)
                fake += 1
            layer.extend(pair)
        layers.append(layer)
    for i, layer in enumerate(layers):
        assert sorted(layer) == list(range(len(layers[0]) * 2 ** i))
    return layers[::-1]


def perm_adjacency(A, indices):
    """Append len(indices) - M isolated nodes to A and renumber node ``indices[i]`` as ``i`` -- graphTools.py:1549-1579."""
    if indices is None:
        return A
    A = sp.coo_matrix(A)
    M, Mnew = A.shape[0], len(indices)
    assert Mnew >= M
    rank = np.argsort(indices)
    return sp.coo_matrix((A.data, (rank[A.row], rank[A.col])), shape=(Mnew, Mnew))


def coarsen(A, levels, self_connections=False):
    """Multilevel Graclus coarsening -- graphTools.py:1337-1367.
    Returns (graphs, perm): levels+1 CSR matrices, each but the coarsest padded with fake nodes and reordered so that
    pooling pairs of consecutive nodes walks the cluster tree; perm = the ordering of the finest level (entries >= N
    are fake nodes), or None when levels == 0."""
    graphs, parents = metis(A, levels)
    perms = compute_perm(parents)
    for i, g in enumerate(graphs):
        g = sp.coo_matrix(g)
        if not self_connections:
            g.setdiag(0)
        if i < levels:
            g = perm_adjacency(g, perms[i])
        g = sp.csr_matrix(g)
        g.eliminate_zeros()
        graphs[i] = g
    return graphs, (perms[0] if levels > 0 else None)


def permCoarsening(x, indices):
    """Reorder the node axis of x [B, F, N] by ``indices``; fake nodes (index >= N) are zero signals, so that max
    pooling keeps their real sibling -- graphTools.py:1581-1614.  numpy in, numpy out (x's dtype is kept)."""
    if indices is None:
        return x
    B, F, N = x.shape
    idx = np.asarray(indices, dtype=np.int64)
    assert idx.shape[0] >= N
    padded = np.concatenate((x, np.zeros((B, F, 1), dtype=x.dtype)), axis=2)
    return padded[:, :, np.minimum(idx, N)]
