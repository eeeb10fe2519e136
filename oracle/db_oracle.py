"""CPU oracle for the per-sample / per-time-step GSO filters.  TEST INFRASTRUCTURE ONLY.

Plain torch (CPU, any float dtype; autograd gives the gradients) restatements of what the reference computes in

    alegnn/utils/graphML.py:977-1094    LSIGF_DB(h, S, x, b)                 delayed filter, S [B,T,E,N,N]
    alegnn/utils/graphML.py:1096-1290   GRNN_DB(a, b, S, x, z0, sigma, ...)  recurrent hidden state on delayed, per-sample operators
    alegnn/utils/graphML.py:1292-1527   GatedGRNN(...) incl. EDGE gates      (:1394-1419 input gate, :1434-1456 forget gate)
    alegnn/utils/graphML.py:640-737     learnAttentionGSO(x, a, W, S)        attention coefficients used as edge gates

written as explicit loops over time steps and taps (the reference shifts / concatenates / takes diagonals of all-pairs products).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product package never does.

Parity pin: tests/golden/{lsigfdb,grnndb,edgegrnn,edgehs,attention}_*.npz were produced by tests/golden/make_golden.py from the real
reference (/root/reference) in the build container; tests/test_oracle_golden.py checks every function below against them.
"""
from __future__ import annotations

import torch


def lsigf_db(h, S, x, b=None):
    """h [F,E,K,G], S [B,T,E,N,N], x [B,T,G,N], b [F,1]|[F,N]|None -> y [B,T,F,N]     (graphML.py:977-1094)

    z_0(t) = x(t);  z_k(t) = z_{k-1}(t-1) @ S_e(t) for t >= 1, zero at t = 0  (the shift + zero row of :1062-1067, the product :1069);
    y(t) = sum_{e,k} h[:,e,k,:] z^e_k(t) + b  (:1073-1092)."""
    F_, E, K, G = h.shape
    B, T, _, N, _ = S.shape
    y = torch.zeros(B, T, F_, N, dtype=x.dtype)
    for e in range(E):
        z = x                                                         # k = 0 (replicated over e, :1049-1052)
        for k in range(K):
            if k > 0:
                shifted = torch.cat((torch.zeros(B, 1, G, N, dtype=x.dtype), z[:, :T - 1]), dim=1)     # z_{k-1}(t-1)
                z = torch.matmul(shifted, S[:, :, e])                  # row-vector product with S_e(t)
            y = y + torch.einsum("fg,btgn->btfn", h[:, e, k, :], z)
    if b is not None:
        y = y + b                                                     # :1091-1092
    return y


def filter_per_sample(h, S5, x, b=None):
    """Per-(b, t) operator without delay: y(b,t) = sum_{e,k} h[:,e,k,:] (x(b,t) S5(b,t,e)^k) + b -- the filter GatedGRNN applies when
    a gate multiplies the GSO entrywise (graphML.py:1400-1419 / :1444-1459)."""
    F_, E, K, G = h.shape
    y = 0.0
    for e in range(E):
        z = x
        for k in range(K):
            if k > 0:
                z = torch.matmul(z, S5[:, :, e])
            y = y + torch.einsum("fg,btgn->btfn", h[:, e, k, :], z)
    if b is not None:
        y = y + b.reshape(1, 1, F_, -1)
    return y


def grnn_db(a, b, S, x, z0, sigma, xBias=None, zBias=None):
    """z [B,T,H,N]: z_t = sigma(A(S) x_t + B(S) z_{t-1})                                  (graphML.py:1096-1290)

    A(S)x = lsigf_db(a, S, x, xBias) (:1164).  The taps of B(S) at time t: W_0(t) = z_{t-1} (z_{-1} = z0), W_k(t) = W_{k-1}(t-1) S_e(t)
    (:1224-1262: the kept products are multiplied by the newest operator, the newest state is prepended, the oldest dropped); taps that
    do not exist yet are zero (:1189-1203, :1243-1245)."""
    H, E, K, F = a.shape
    B, T, _, N, _ = S.shape
    Ax = lsigf_db(a, S, x, None if xBias is None else xBias.reshape(H, 1))
    zb = None if zBias is None else zBias.reshape(1, H, 1)
    W = [[None] * K for _ in range(E)]                                # W[e][k]: [B,H,N] or None (= zero)
    zt = z0
    out = []
    for t in range(T):
        for e in range(E):
            new = [zt] + [None if (W[e][k - 1] is None or t == 0) else torch.matmul(W[e][k - 1], S[:, t, e]) for k in range(1, K)]
            W[e] = new
        Bz = 0.0
        for e in range(E):
            for k in range(K):
                if W[e][k] is not None:
                    Bz = Bz + torch.einsum("hg,bgn->bhn", b[:, e, k, :], W[e][k])
        if zb is not None:
            Bz = Bz + zb
        zt = sigma(Ax[:, t] + Bz)                                     # :1208 / :1281
        out.append(zt)
    return torch.stack(out, dim=1)


def gated_grnn(a, b, S, x, z0, sigma, q_hat=None, q_check=None, xBias=None, zBias=None):
    """GatedGRNN with any of the reference's gate shapes, edge gates included                 (graphML.py:1292-1527)

    S dense [E,N,N].  q of 5 dims ([B,T,1,N,N]) multiplies the GSO entrywise per (b, t) (:1397-1399, :1441-1443); the filters then run
    on the gated operators without delay."""
    H, E, K, F = a.shape
    N = S.shape[1]
    B, T = x.shape[0], x.shape[1]
    xb = None if xBias is None else xBias.reshape(H, 1)
    zb = None if zBias is None else zBias.reshape(H, 1)
    if q_hat is not None and q_hat.dim() > 4:
        Ax = filter_per_sample(a, q_hat.reshape(B, T, E, N, N) * S, x, xb)
    else:
        Ax = filter_per_sample(a, S.reshape(1, 1, E, N, N).expand(B, T, E, N, N), x, xb)           # the static filter, :1389-1391
        if q_hat is not None:
            Ax = q_hat * Ax
    zt = z0
    out = []
    for t in range(T):
        if q_check is not None and q_check.dim() > 4:
            Bz = filter_per_sample(b, (q_check[:, t] * S).reshape(B, 1, E, N, N), zt.reshape(B, 1, H, N), zb).reshape(B, H, N)
        else:
            Bz = filter_per_sample(b, S.reshape(1, 1, E, N, N).expand(B, 1, E, N, N), zt.reshape(B, 1, H, N), zb).reshape(B, H, N)
            if q_check is not None:
                Bz = (q_check[:, t] if q_check.dim() > 1 else q_check) * Bz
        zt = sigma(Ax[:, t] + Bz)
        out.append(zt)
    return torch.stack(out, dim=1)


def learn_attention_gso(x, a, W, S, negative_slope=0.2):
    """alpha [B,P,E,N,N]: alpha_ij = softmax_j(LeakyReLU(a2 . W x_i + a1 . W x_j)) over the support of S + I      (graphML.py:640-737;
    the code at :706-712 puts the FIRST half of the mixing vector on the neighbour j, the second on node i)."""
    B, G, N = x.shape
    P, E, F2 = a.shape
    F = F2 // 2
    mask = ((S + torch.eye(N, dtype=S.dtype)).abs().sum(dim=0) > 1e-9).to(x.dtype)               # :692, :726-728
    out = torch.zeros(B, P, E, N, N, dtype=x.dtype)
    for p in range(P):
        for e in range(E):
            Wx = torch.einsum("fg,bgn->bfn", W[p, e], x)                                          # :703
            sj = torch.einsum("f,bfn->bn", a[p, e, :F], Wx)                                       # a1 . W x_j  (row vector 1 x N, :708)
            si = torch.einsum("f,bfn->bn", a[p, e, F:], Wx)                                       # a2 . W x_i  (transposed to N x 1, :712)
            eij = torch.nn.functional.leaky_relu(si[:, :, None] + sj[:, None, :], negative_slope)
            out[:, p, e] = torch.softmax(eij * mask - (1 - mask) * 1e12, dim=2) * mask            # :733-737
    return out
