"""The reference's dense formulation of the filters in plain torch ops, for the same-GPU comparators of the tools/*_bench.py
scripts (SURVEY.md 8d's "no-rewrite" leg): dense S, one matmul per tap, the taps collected and contracted in one matmul.
Written out here so that the tools do not touch oracle/ (which is reserved for tests, smoke() and bench.py's cpu_baseline)."""
import torch


def _taps(S, x, K):
    """z = [x, xS, xS^2, ...] per edge feature, grown by torch.cat as the reference does: B x E x K x G x N
    (graphML.py:152-161, 356-371)."""
    B, G, N = x.shape
    E = S.shape[0]
    cur = x.reshape(B, 1, G, N)
    z = x.reshape(B, 1, 1, G, N).repeat(1, E, 1, 1, 1)
    for _ in range(1, K):
        cur = torch.matmul(cur, S.reshape(1, E, N, N))
        z = torch.cat((z, cur.reshape(B, E, 1, G, N)), dim=2)
    return z


def dense_lsigf(h, S, x, b=None):
    """h [F,E,K,G], S [E,N,N], x [B,G,N] -> [B,F,N]: permute + one matmul + permute (graphML.py:170-175)."""
    F, E, K, G = h.shape
    B, _, N = x.shape
    z = _taps(S, x, K)
    y = torch.matmul(z.permute(0, 4, 1, 2, 3).reshape(B, N, E * K * G), h.reshape(F, E * K * G).permute(1, 0)).permute(0, 2, 1)
    return y if b is None else y + b


def dense_nvgf(h, S, x, b=None):
    """h [F,E,K,G,N] -> [B,F,N]: broadcast product and three sums (graphML.py:372-384)."""
    F, E, K, G, N = h.shape
    B = x.shape[0]
    z = _taps(S, x, K).reshape(B, 1, E, K, G, N)
    y = (z * h.reshape(1, F, E, K, G, N)).sum(dim=4).sum(dim=3).sum(dim=2)
    return y if b is None else y + b
