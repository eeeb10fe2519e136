"""CPU oracle for the LSIGF / GraphFilter hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain CPU restatement (numpy / scipy / torch-CPU) of the arithmetic the
reference performs in

    alegnn/utils/graphML.py:83-176    LSIGF(h, S, x, b)            (the K-hop loop + filter-bank contraction)
    alegnn/utils/graphML.py:2125-2144 GraphFilter.forward          (zero-pad to N nodes, call LSIGF, keep Nin nodes)

It exists so the HIP path can be checked on any box (the GPU box has no /root/reference).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the
product package (graph-neural-networks_amd/) never does, and has no CPU fallback.

Parity pin: the reference ships no tests / golden vectors, so the oracle
is pinned against outputs of the reference itself: tests/golden/*.npz were produced by
tests/golden/make_golden.py, which imports the real reference from /root/reference in the
build container; tests/test_oracle_golden.py checks every function below against them.

Conventions restated from the reference:
  * row-vector convention:  x_k = x_{k-1} @ S_e           (graphML.py:159)
  * tap k=0 is x itself, replicated over the E edge features (graphML.py:154)
  * z is [B, E, K, G, N]; flattened (e, k, g) order against h.reshape(F, E*K*G) (graphML.py:170-171)
  * bias [F, 1] (or [F, N]) is broadcast-added last        (graphML.py:174-175)
  * S receives no gradient (plain attribute, graphML.py:2099)
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch


# --------------------------------------------------------------------------------------
# (1) literal dense restatement -- follows graphML.py:152-175 step for step, in torch so
#     that autograd gives the reference's own backward.  Any dtype.
# --------------------------------------------------------------------------------------
def lsigf_dense(h: torch.Tensor, S: torch.Tensor, x: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
    """h [F,E,K,G], S [E,N,N] dense, x [B,G,N], b [F,1]|[F,N]|None  ->  y [B,F,N]."""
    F_, E, K, G = h.shape
    assert S.shape[0] == E and S.shape[1] == S.shape[2]            # graphML.py:135-137
    N = S.shape[1]
    B = x.shape[0]
    assert x.shape[1] == G and x.shape[2] == N                     # graphML.py:139-140
    cur = x.reshape(B, 1, G, N)                                    # :152
    Sb = S.reshape(1, E, N, N)                                     # :153
    taps = [x.reshape(B, 1, 1, G, N).repeat(1, E, 1, 1, 1)]        # :154  (k = 0)
    for _ in range(1, K):                                          # :158
        cur = torch.matmul(cur, Sb)                                # :159  row-vector product
        taps.append(cur.reshape(B, E, 1, G, N))                    # :160
    z = torch.cat(taps, dim=2)                                     # :161  [B,E,K,G,N]
    zf = z.permute(0, 4, 1, 2, 3).reshape(B, N, E * K * G)         # :170
    y = torch.matmul(zf, h.reshape(F_, E * K * G).permute(1, 0)).permute(0, 2, 1)  # :170-171
    if b is not None:
        y = y + b                                                  # :174-175
    return y


def graph_filter_forward_dense(weight, bias, S, x):
    """GraphFilter.forward semantics (graphML.py:2125-2144): zero-pad nodes Nin..N-1, filter, keep first Nin."""
    B, G, Nin = x.shape
    N = S.shape[1]
    if Nin < N:
        x = torch.cat((x, torch.zeros(B, G, N - Nin, dtype=x.dtype)), dim=2)   # :2131-2135
    u = lsigf_dense(weight, S, x, bias)                                      # :2137
    if Nin < N:
        u = u[:, :, :Nin]                                                    # :2142-2143
    return u


# --------------------------------------------------------------------------------------
# (2) sparse restatement -- same arithmetic with S_e held as scipy CSR, for sizes where the
#     dense [E,N,N] tensor cannot exist (N = 1e5 -> 40 GB).  numpy, any float dtype.
#     x @ S  ==  (S^T @ x^T)^T, so each hop is one CSR(S^T) x dense product.
# --------------------------------------------------------------------------------------
def _as_csr_list(S, dtype):
    if isinstance(S, (list, tuple)):
        return [sp.csr_matrix(s, dtype=dtype) for s in S]
    if sp.issparse(S):
        return [sp.csr_matrix(S, dtype=dtype)]
    S = np.asarray(S)
    if S.ndim == 2:
        S = S[None]
    return [sp.csr_matrix(S[e], dtype=dtype) for e in range(S.shape[0])]


def lsigf_taps_sparse(S, x: np.ndarray, K: int, dtype=np.float64) -> np.ndarray:
    """Return z [B,E,K,G,N] (graphML.py:152-161) with S given sparse. x [B,G,N]."""
    Ss = _as_csr_list(S, dtype)
    E = len(Ss)
    B, G, N = x.shape
    x = np.asarray(x, dtype=dtype)
    z = np.empty((B, E, K, G, N), dtype=dtype)
    for e, Se in enumerate(Ss):
        St = Se.T.tocsr()
        cur = x.reshape(B * G, N)
        z[:, e, 0] = x
        for k in range(1, K):
            cur = np.ascontiguousarray((St @ cur.T).T)             # == cur @ S_e  (:159)
            z[:, e, k] = cur.reshape(B, G, N)
    return z


def lsigf_sparse(h: np.ndarray, S, x: np.ndarray, b: np.ndarray | None = None, dtype=np.float64) -> np.ndarray:
    """Sparse-S LSIGF.  h [F,E,K,G], x [B,G,N] -> y [B,F,N]."""
    h = np.asarray(h, dtype=dtype)
    F_, E, K, G = h.shape
    z = lsigf_taps_sparse(S, x, K, dtype)                          # [B,E,K,G,N]
    B, N = z.shape[0], z.shape[-1]
    zf = z.transpose(0, 4, 1, 2, 3).reshape(B, N, E * K * G)       # :170
    y = (zf @ h.reshape(F_, E * K * G).T).transpose(0, 2, 1)       # :170-171
    if b is not None:
        y = y + np.asarray(b, dtype=dtype)                         # :174-175
    return y


def lsigf_sparse_grads(h, S, x, b, dy, dtype=np.float64):
    """Analytic backward of lsigf_sparse, validated against
    torch.autograd of lsigf_dense in tests/test_oracle_golden.py.
    Returns (dx [B,G,N], dh [F,E,K,G], db [F,1] or None)."""
    h = np.asarray(h, dtype=dtype)
    dy = np.asarray(dy, dtype=dtype)
    F_, E, K, G = h.shape
    Ss = _as_csr_list(S, dtype)
    B, _, N = dy.shape
    z = lsigf_taps_sparse(Ss, x, K, dtype)                         # [B,E,K,G,N]
    # dh[f,e,k,g] = sum_{b,n} dy[b,f,n] z[b,e,k,g,n]
    dh = np.einsum("bfn,bekgn->fekg", dy, z, optimize=True)
    db = dy.sum(axis=(0, 2)).reshape(F_, 1) if b is not None else None
    # dz[b,e,k,g,n] = sum_f h[f,e,k,g] dy[b,f,n];  dx = sum_e sum_k dz_{e,k} @ (S_e^T)^k  (Horner)
    dz = np.einsum("fekg,bfn->bekgn", h, dy, optimize=True)
    dx = np.zeros((B, G, N), dtype=dtype)
    for e, Se in enumerate(Ss):
        g = dz[:, e, K - 1].reshape(B * G, N)
        for k in range(K - 2, -1, -1):
            g = np.ascontiguousarray((Se @ g.T).T) + dz[:, e, k].reshape(B * G, N)   # g @ S_e^T + dz_k
        dx += g.reshape(B, G, N)
    return dx, dh, db


def graph_filter_forward_sparse(weight, bias, S, x, dtype=np.float64):
    """GraphFilter.forward (graphML.py:2125-2144) with sparse S."""
    Ss = _as_csr_list(S, dtype)
    N = Ss[0].shape[0]
    B, G, Nin = x.shape
    xp = np.zeros((B, G, N), dtype=dtype)
    xp[:, :, :Nin] = x
    u = lsigf_sparse(weight, Ss, xp, bias, dtype)
    return u[:, :, :Nin]


# --------------------------------------------------------------------------------------
# (3) fwd+bwd of one GraphFilter layer on CPU in the reference's own form (dense torch
#     matmul loop).  Used by bench.py's cpu_baseline leg: this is what the reference runs.
# --------------------------------------------------------------------------------------
def graph_filter_step_dense(weight, bias, S, x):
    """One forward+backward of the literal dense path with loss = y.sum(). Returns y (detached)."""
    weight = weight.detach().clone().requires_grad_(True)
    bias_ = bias.detach().clone().requires_grad_(True) if bias is not None else None
    xx = x.detach().clone().requires_grad_(True)
    y = graph_filter_forward_dense(weight, bias_, S, xx)
    y.sum().backward()
    return y.detach(), xx.grad, weight.grad, (bias_.grad if bias_ is not None else None)


def graph_filter_step_sparse_torch(weight, bias, St_csr: torch.Tensor, x):
    """Same step with S held as torch.sparse_csr(S^T) (E = 1): the 'sparse CPU restatement'
    of BASELINE.md section 3.  weight [F,1,K,G], x [B,G,N]."""
    F_, E, K, G = weight.shape
    assert E == 1
    B, _, N = x.shape
    weight = weight.detach().clone().requires_grad_(True)
    bias_ = bias.detach().clone().requires_grad_(True) if bias is not None else None
    xx = x.detach().clone().requires_grad_(True)
    cur = xx.permute(2, 0, 1).reshape(N, B * G)                    # node-major view of x
    taps = [cur]
    for _ in range(1, K):
        cur = torch.sparse.mm(St_csr, cur)                         # (x @ S)^T = S^T @ x^T
        taps.append(cur)
    z = torch.stack(taps, 0).reshape(K, N, B, G).permute(2, 1, 0, 3).reshape(B, N, K * G)
    y = torch.matmul(z, weight.reshape(F_, K * G).t()).permute(0, 2, 1)
    if bias_ is not None:
        y = y + bias_
    y.sum().backward()
    return y.detach(), xx.grad, weight.grad, (bias_.grad if bias_ is not None else None)


# --------------------------------------------------------------------------------------
# (4) EVGF / EdgeVariantGF restatement (graphML.py:389-488, 2608-2698), per-edge storage.
#     Column convention (Phi @ x).  Dense literal version for small N.
# --------------------------------------------------------------------------------------
def evgf_dense(Phi: torch.Tensor, x: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
    """Phi [F,E,K,G,N,N], x [B,G,N] -> y [B,F,N]   (graphML.py:457-488)."""
    F_, E, K, G, N, _ = Phi.shape
    B = x.shape[0]
    v = x.reshape(B, 1, 1, G, N, 1)
    acc = None
    for k in range(K):
        v = torch.matmul(Phi[:, :, k].unsqueeze(0), v)             # :464 / :475  Phi_k @ v   [B,F,E,G,N,1]
        term = v.squeeze(-1).sum(dim=(2, 3))                       # sum over E and G (:481-485)
        acc = term if acc is None else acc + term
    if b is not None:
        acc = acc + b
    return acc
