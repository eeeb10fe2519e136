"""Sparse synthetic graphs for the benchmark configurations.

The reference's generator (graphTools.createGraph 'SBM', graphTools.py:747-800) builds a dense N x N adjacency and
normalises by a full eigendecomposition (graphTools.py:562-589, examples/sourceLocGNN.py:752): O(N^2) memory, O(N^3)
time, unusable past N ~ 1e4.  These samplers draw the same random-graph models edge by edge and normalise by the
largest eigenvalue from ARPACK, returning scipy CSR in float64.

  sbm(N, ...)  : C balanced communities, p_intra : p_inter = 4 : 1 (examples/sourceLocGNN.py:128-130 uses 0.8 : 0.2),
                 scaled to a target average degree; undirected, no self-loops, unit weights / lambda_max.
  er(N, ...)   : Erdos-Renyi G(N, p).
  directed=True drops the symmetrisation (asymmetric S: catches S vs S^T mistakes).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _sample_pairs(n_rows, n_cols, p, rng, upper_only):
    """Bernoulli(p) over an n_rows x n_cols block (strict upper triangle if upper_only) -> (rows, cols)."""
    if upper_only:
        M = n_rows * (n_rows - 1) // 2
    else:
        M = n_rows * n_cols
    m = rng.binomial(M, p) if M > 0 else 0
    if m == 0:
        return np.empty(0, np.int64), np.empty(0, np.int64)
    lin = np.unique(rng.randint(0, M, size=int(m * 1.05) + 8, dtype=np.int64))
    while lin.size < m:                                    # collisions are rare in the sparse regime; top up
        lin = np.unique(np.concatenate([lin, rng.randint(0, M, size=m, dtype=np.int64)]))
    lin = rng.permutation(lin)[:m]
    if upper_only:                                          # unrank k -> (i, j), i < j, row-major strict upper triangle
        i = (n_rows - 2 - np.floor(np.sqrt(-8.0 * lin + 4.0 * n_rows * (n_rows - 1) - 7) / 2.0 - 0.5)).astype(np.int64)
        j = lin + i + 1 - n_rows * (n_rows - 1) // 2 + (n_rows - i) * ((n_rows - i) - 1) // 2
        return i, j
    return lin // n_cols, lin % n_cols


def _normalise(A: sp.csr_matrix, normalize: bool) -> sp.csr_matrix:
    A = sp.csr_matrix(A, dtype=np.float64)
    A.sum_duplicates()
    A.sort_indices()
    if normalize and A.nnz > 0:
        if A.shape[0] <= 64:
            lam = np.max(np.abs(np.linalg.eigvals(A.toarray())))
        else:
            try:
                lam = abs(spla.eigs(A, k=1, which="LM", return_eigenvectors=False, maxiter=5000, tol=1e-8)[0])
            except Exception:                               # ARPACK non-convergence: fall back to power iteration
                v = np.ones(A.shape[0]) / np.sqrt(A.shape[0])
                for _ in range(200):
                    w = A @ v
                    lam = np.linalg.norm(w)
                    v = w / max(lam, 1e-300)
        if lam > 0:
            A = A / lam
    return sp.csr_matrix(A)


def sbm(N, avg_degree=10.0, n_communities=5, ratio=4.0, seed=0, directed=False, normalize=True) -> sp.csr_matrix:
    rng = np.random.RandomState(seed)
    C = n_communities
    sizes = [N // C + (1 if c < N % C else 0) for c in range(C)]
    starts = np.concatenate([[0], np.cumsum(sizes)])
    # expected degree = p_in * (n_c - 1) + p_out * (N - n_c), p_in = ratio * p_out
    nc = N / C
    p_out = avg_degree / (ratio * (nc - 1) + (N - nc))
    p_in = min(1.0, ratio * p_out)
    rows, cols = [], []
    for a in range(C):
        for b in range(C):
            if directed:
                if a == b:
                    i, j = _sample_pairs(sizes[a], sizes[a], p_in, rng, False)
                    keep = i != j
                    i, j = i[keep], j[keep]
                else:
                    i, j = _sample_pairs(sizes[a], sizes[b], p_out, rng, False)
            else:
                if b < a:
                    continue
                if a == b:
                    i, j = _sample_pairs(sizes[a], sizes[a], p_in, rng, True)
                else:
                    i, j = _sample_pairs(sizes[a], sizes[b], p_out, rng, False)
            rows.append(i + starts[a])
            cols.append(j + starts[b])
    r = np.concatenate(rows)
    c = np.concatenate(cols)
    if not directed:
        r, c = np.concatenate([r, c]), np.concatenate([c, r])
    A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(N, N))
    A.data[:] = 1.0
    return _normalise(A, normalize)


def er(N, avg_degree=10.0, seed=0, directed=False, normalize=True) -> sp.csr_matrix:
    rng = np.random.RandomState(seed)
    p = avg_degree / (N - 1)
    if directed:
        i, j = _sample_pairs(N, N, p, rng, False)
        keep = i != j
        r, c = i[keep], j[keep]
    else:
        i, j = _sample_pairs(N, N, p, rng, True)
        r, c = np.concatenate([i, j]), np.concatenate([j, i])
    A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(N, N))
    A.data[:] = 1.0
    return _normalise(A, normalize)


def knn_weighted(N, k=10, seed=0, normalize=True) -> sp.csr_matrix:
    """MovieLens-shaped GSO (BASELINE configs[2]): every node keeps k neighbours with similarity weights in (0, 1], the graph is
    symmetrised (an edge survives if either end kept it, as the reference's kNN sparsification does, graphTools.py:650-678 on the
    Pearson matrix of dataTools.py:1814-1871) and divided by its largest eigenvalue.  Synthetic stand-in: the data set itself needs
    the network."""
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(N), k)
    cols = rng.randint(0, N - 1, size=N * k)
    cols = cols + (cols >= rows)                       # no self loops
    w = rng.uniform(0.05, 1.0, size=N * k)
    A = sp.csr_matrix((w, (rows, cols)), shape=(N, N))
    A.sum_duplicates()
    A = A.maximum(A.T)
    return _normalise(A, normalize)
