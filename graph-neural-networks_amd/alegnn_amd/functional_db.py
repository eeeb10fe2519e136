"""Autograd boundary of the per-sample / per-time-step GSO filters: the reference's "_DB" (batch + delay) family and edge gating.

    LSIGF_DB(h, S, x, b)                     alegnn/utils/graphML.py:977-1094     S [B,T,E,N,N], x [B,T,G,N] -> y [B,T,F,N]
    GRNN_DB(a, b, S, x, z0, sigma, ...)      alegnn/utils/graphML.py:1096-1290    hidden states z [B,T,H,N]
    filter_per_sample(h, S5, x, b)           the edge-gated filters inside GatedGRNN (:1394-1419, :1434-1456): no delay, S5 = q * S

The arithmetic is in libgfhip.so (csrc/gf_db.hip + the contraction / tap-gradient / layout kernels of the static path, run with batch
B*T); torch supplies device memory, streams and the autograd graph that strings the time steps and the nonlinearity together.
S gets a gradient only where the reference's does (edge gating: the gated operator is a function of learnable gate networks).
"""
from __future__ import annotations

import torch

from . import _lib
from .functional import _padded_width, _ptr, _require_f32_cuda


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _LSIGFDBFunction(torch.autograd.Function):
    """y = filter with a per-(b, t) operator: gf_lsigf_db_forward / gf_lsigf_db_backward (include/gfhip.h)."""

    @staticmethod
    def forward(ctx, x, h, bias, S, shift):
        L = _lib.lib()
        B, T, G, N = x.shape
        F_, E, K, _ = h.shape
        taps = 1 + E * (K - 1)
        x, h, S = x.contiguous(), h.contiguous(), S.contiguous()
        bias_c = None if bias is None else bias.contiguous()
        dev = x.device
        with torch.cuda.device(dev):
            Z = torch.empty((taps, B * T, N, G), dtype=torch.float32, device=dev)
            y = torch.empty((B, T, F_, N), dtype=torch.float32, device=dev)
            _lib.check(L.gf_lsigf_db_forward(S.data_ptr(), x.data_ptr(), h.data_ptr(), _ptr(bias_c), Z.data_ptr(), y.data_ptr(),
                                             B, T, G, F_, E, K, N, shift, _stream()), "gf_lsigf_db_forward")
        ctx.dims = (B, T, G, F_, E, K, N, taps, shift)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(h, S, Z)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        h, S, Z = ctx.saved_tensors
        B, T, G, F_, E, K, N, taps, shift = ctx.dims
        need_dx, need_dh = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        need_dS = ctx.needs_input_grad[3]
        dy = dy.contiguous()
        dev = dy.device
        with torch.cuda.device(dev):
            P0 = torch.empty((B * T, N, F_), dtype=torch.float32, device=dev)
            chain = need_dx or (need_dS and K > 1)
            dZ = torch.empty((taps, B * T, N, G), dtype=torch.float32, device=dev) if chain else None
            hop = torch.empty((B * T, N, G), dtype=torch.float32, device=dev) if chain else None
            dx = torch.empty((B, T, G, N), dtype=torch.float32, device=dev) if need_dx else None
            dh = torch.empty_like(h) if need_dh else None
            db = torch.empty((F_, 1), dtype=torch.float32, device=dev) if need_db else None
            dS = None
            if need_dS:                                   # written by the kernels for every (b, t, e) when K > 1
                dS = torch.empty_like(S) if K > 1 else torch.zeros_like(S)
            ws, ws_bytes = None, 0
            if need_dh or need_db:
                ws_bytes = L.gf_grad_taps_workspace_bytes(B * T, N, G, F_, E, K)
                ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
            _lib.check(L.gf_lsigf_db_backward(S.data_ptr(), dy.data_ptr(), Z.data_ptr(), h.data_ptr(), P0.data_ptr(), _ptr(dZ), _ptr(hop),
                                              _ptr(dx), _ptr(dh), _ptr(db), _ptr(dS) if K > 1 else None, _ptr(ws), ws_bytes,
                                              B, T, G, F_, E, K, N, shift, _stream()), "gf_lsigf_db_backward")
        return dx, dh, db, dS, None


def _check_db_shapes(h, S, x):
    assert h.dim() == 4                                           # graphML.py:1029
    F_, E, K, G = h.shape
    assert S.dim() == 5                                           # :1034
    B, T = S.shape[0], S.shape[1]
    assert S.shape[2] == E                                        # :1037
    N = S.shape[3]
    assert S.shape[4] == N                                        # :1039
    assert x.dim() == 4                                           # :1040
    assert x.shape[0] == B and x.shape[1] == T and x.shape[2] == G and x.shape[3] == N   # :1041-1044
    return B, T, G, F_, E, K, N


def _filter_db(h, S, x, b, shift):
    B, T, G, F_, E, K, N = _check_db_shapes(h, S, x)
    for name, t in (("x", x), ("h", h), ("S", S)):
        _require_f32_cuda(name, t)
    fused = late = None
    if b is not None:
        _require_f32_cuda("b", b)
        b2 = b.reshape(F_, -1)                                    # [F,1] | [F,N] (graphML.py:1008-1010); GRNN_DB hands 1 x 1 x H x 1 (:1123)
        fused, late = (b2, None) if b2.shape[1] == 1 else (None, b2)
    Gp, Fp = _padded_width(G), _padded_width(F_)                  # zero channels through zero taps: exact, and every row is 16-byte wide
    if Gp != G or Fp != F_:
        if Gp != G:
            x = torch.nn.functional.pad(x, (0, 0, 0, Gp - G))
        h = torch.nn.functional.pad(h, (0, Gp - G, 0, 0, 0, 0, 0, Fp - F_))
        if fused is not None and Fp != F_:
            fused = torch.nn.functional.pad(fused, (0, 0, 0, Fp - F_))
    y = _LSIGFDBFunction.apply(x, h, fused, S, shift)
    if Fp != F_:
        y = y[:, :, :F_].contiguous()
    if late is not None:
        y = y + late
    return y


def LSIGF_DB(h, S, x, b=None):
    """Linear shift-invariant graph filter on delayed inputs, reference signature and semantics (graphML.py:977-1094):

        y_f(t) = sum_{e,k,g} h[f,e,k,g] * ( x_g(t-k) S_e(t-k+1) ... S_e(t) ) + b_f,       terms with t - k < 0 are zero

    h [F,E,K,G], S [B,T,E,N,N] (dense, on the device: every (b, t) has its own operator), x [B,T,G,N], b [F,1]|[F,N]|None."""
    return _filter_db(h, S, x, b, 1)


def filter_per_sample(h, S5, x, b=None):
    """y(b,t) = sum_{e,k} x(b,t) S5(b,t,e)^k h[:,e,k,:]^T + b -- a per-(b, t) operator WITHOUT delay: what GatedGRNN computes when a
    gate multiplies the GSO entrywise (edgeS = q * S, graphML.py:1397-1419 and :1441-1456; the reference multiplies all B*T signals with
    all B*T operators and keeps the diagonal, :1407-1413).  S5 [B,T,E,N,N] may require grad (it is a function of the gates)."""
    return _filter_db(h, S5, x, b, 0)


class _DBStepFunction(torch.autograd.Function):
    """One time step of GRNN_DB's hidden-to-hidden recursion (graphML.py:1224-1283).

    The K taps of B(S) at time t are  W_0(t) = z_{t-1},  W_k(t) = W_{k-1}(t-1) S_e(t)  -- the previous step's taps moved one hop by the
    CURRENT operator (:1235 / :1262) -- and Bz_t = sum_{e,k} W^e_k(t) b[:,e,k,:]^T + zBias (:1272-1276).
    Inputs: zprev [B,H,N] (reference layout), Wprev [taps,B,N,H] (node-major stack of step t-1; ignored when first), S5, t.
    Outputs: Bz [B,H,N], W [taps,B,N,H]."""

    @staticmethod
    def forward(ctx, zprev, Wprev, h, bias, S, t, first):
        L = _lib.lib()
        B, H, N = zprev.shape
        F_, E, K, _ = h.shape
        T = S.shape[1]
        taps = 1 + E * (K - 1)
        zprev, h, S = zprev.contiguous(), h.contiguous(), S.contiguous()
        bias_c = None if bias is None else bias.contiguous()
        dev = zprev.device
        sb, NN = T * E * N * N, N * N
        with torch.cuda.device(dev):
            W = torch.empty((taps, B, N, H), dtype=torch.float32, device=dev)
            Bz = torch.empty((B, F_, N), dtype=torch.float32, device=dev)
            st = _stream()
            _lib.check(L.gf_layout_bgn_to_bng(zprev.data_ptr(), W.data_ptr(), B, H, N, N, st), "gf_layout_bgn_to_bng")
            tapsz = B * N * H * 4
            for e in range(E):
                for k in range(1, K):
                    dst = W.data_ptr() + (1 + e * (K - 1) + (k - 1)) * tapsz
                    if first:                                    # t = 0: only b(0) z0 exists (:1189-1203)
                        W[1 + e * (K - 1) + (k - 1)].zero_()
                        continue
                    src = Wprev.data_ptr() + ((0 if k == 1 else 1 + e * (K - 1) + (k - 2)) * tapsz)
                    _lib.check(L.gf_db_hop(S.data_ptr() + (t * E + e) * NN * 4, sb, E * NN, src, dst, B, 1, N, H, _lib.GF_OP_FWD, 0, st),
                               "gf_db_hop")
            _lib.check(L.gf_contract(W.data_ptr(), h.data_ptr(), _ptr(bias_c), Bz.data_ptr(), B, N, N, H, F_, E, K, 0, st), "gf_contract")
        ctx.dims = (B, H, N, F_, E, K, T, taps, int(t), bool(first))
        ctx.has_bias = bias is not None
        ctx.save_for_backward(h, S, W)
        ctx.set_materialize_grads(False)
        return Bz, W

    @staticmethod
    def backward(ctx, dBz, dW):
        L = _lib.lib()
        h, S, W = ctx.saved_tensors
        B, H, N, F_, E, K, T, taps, t, first = ctx.dims
        dev = h.device
        dBz = torch.zeros((B, F_, N), dtype=torch.float32, device=dev) if dBz is None else dBz.contiguous()
        sb, NN = T * E * N * N, N * N
        need_dz, need_dWprev = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and not first
        need_dh, need_db = ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3]
        with torch.cuda.device(dev):
            st = _stream()
            P0 = torch.empty((B, N, F_), dtype=torch.float32, device=dev)
            _lib.check(L.gf_layout_bgn_to_bng(dBz.data_ptr(), P0.data_ptr(), B, F_, N, N, st), "gf_layout_bgn_to_bng")
            dh = db = None
            if need_dh or need_db:
                dh = torch.empty_like(h) if need_dh else None
                db = torch.empty((F_, 1), dtype=torch.float32, device=dev) if need_db else None
                nb = L.gf_grad_taps_workspace_bytes(B, N, H, F_, E, K)
                ws = torch.empty((nb + 3) // 4, dtype=torch.float32, device=dev)
                _lib.check(L.gf_grad_taps(W.data_ptr(), P0.data_ptr(), _ptr(dh), _ptr(db), ws.data_ptr(), nb, B, N, H, F_, E, K, st), "gf_grad_taps")
            dz = dWprev = None
            if need_dz or need_dWprev:
                g = torch.empty((taps, B, N, H), dtype=torch.float32, device=dev)     # gradient of every tap of this step
                _lib.check(L.gf_stack_adjoint(P0.data_ptr(), h.data_ptr(), g.data_ptr(), B * N, H, F_, E, K, st), "gf_stack_adjoint")
                if dW is not None:
                    g = g + dW.contiguous()                       # the next step moved these taps one more hop
                if need_dz:
                    dz = torch.empty((B, H, N), dtype=torch.float32, device=dev)
                    _lib.check(L.gf_layout_bng_to_bgn(g.data_ptr(), dz.data_ptr(), B, H, N, N, st), "gf_layout_bng_to_bgn")
                if need_dWprev:
                    dWprev = torch.zeros((taps, B, N, H), dtype=torch.float32, device=dev)
                    tapsz = B * N * H * 4
                    hop = torch.empty((B, N, H), dtype=torch.float32, device=dev)
                    for e in range(E):
                        for k in range(1, K):                     # W^e_k(t) = W^e_{k-1}(t-1) S_e(t): its adjoint feeds tap k-1 of step t-1
                            src = g.data_ptr() + (1 + e * (K - 1) + (k - 1)) * tapsz
                            _lib.check(L.gf_db_hop(S.data_ptr() + (t * E + e) * NN * 4, sb, E * NN, src, hop.data_ptr(), B, 1, N, H,
                                                   _lib.GF_OP_BWD, 0, st), "gf_db_hop")
                            dWprev[0 if k == 1 else 1 + e * (K - 1) + (k - 2)] += hop
        return dz, dWprev, dh, db, None, None, None


def GRNN_DB(a, b, S, x, z0, sigma, xBias=None, zBias=None):
    """Hidden states z_t = sigma(A(S) x_t + B(S) z_{t-1}) with delayed, per-sample operators -- graphML.py:1096-1290.

    a [H,E,K,F], b [H,E,K,H], S [B,T,E,N,N], x [B,T,F,N], z0 [B,H,N], xBias / zBias [H,1] (any shape with H elements) -> z [B,T,H,N].
    A(S)x for the whole sequence is ONE delayed filter (:1164); the recursion is one HIP step per time instant (K-1 per-sample hops of
    the previous taps by S(t), the filter bank, the bias), the nonlinearity and the sum are elementwise torch ops."""
    H, E, K, F = a.shape
    assert b.shape[0] == H and b.shape[1] == E and b.shape[2] == K and b.shape[3] == H       # :1146-1149
    B, T = S.shape[0], S.shape[1]
    assert S.shape[2] == E                                                                  # :1152
    N = S.shape[3]
    assert S.shape[4] == N
    assert x.shape[0] == B and x.shape[1] == T and x.shape[2] == F and x.shape[3] == N       # :1155-1158
    assert z0.shape[0] == B and z0.shape[1] == H and z0.shape[2] == N                       # :1159-1161
    for name, t_ in (("a", a), ("b", b), ("S", S), ("x", x), ("z0", z0)):
        _require_f32_cuda(name, t_)
    if S.requires_grad:
        # the hidden-to-hidden recursion's step function does not differentiate with respect to the operator (no reference architecture
        # trains S here: the flocking GSOs are data; edge gating goes through GatedGRNN): say so instead of returning a partial S.grad
        raise NotImplementedError("GRNN_DB: S.requires_grad is not supported (the recursion B(S) z_{t-1} has no gradient with respect to S); "
                                  "detach S, or use GatedGRNN for learnable edge gates")
    Ax = LSIGF_DB(a, S, x, xBias)                                                           # :1164   B x T x H x N
    Hp = _padded_width(H)
    bb, zb, zt = b, (None if zBias is None else zBias.reshape(H, 1)), z0
    if Hp != H:                                  # hidden width padded with zero channels (zero taps in and out): exact
        bb = torch.nn.functional.pad(b, (0, Hp - H, 0, 0, 0, 0, 0, Hp - H))
        zb = None if zb is None else torch.nn.functional.pad(zb, (0, 0, 0, Hp - H))
        zt = torch.nn.functional.pad(z0, (0, 0, 0, Hp - H))
    Wstack = torch.zeros((1 + E * (K - 1), B, N, Hp), dtype=torch.float32, device=x.device)
    states = []
    for t in range(T):
        # t = 0: Bz = b(0) z0 (:1189-1205); t >= 1: the taps of step t-1 move one hop by S(t) (:1224-1262)
        Bz, Wstack = _DBStepFunction.apply(zt, Wstack, bb, zb, S, t, t == 0)
        znew = sigma(Ax[:, t] + (Bz[:, :H] if Hp != H else Bz))                             # :1208 / :1281
        states.append(znew)
        zt = znew if Hp == H else torch.nn.functional.pad(znew, (0, 0, 0, Hp - H))
    return torch.stack(states, dim=1)                                                       # B x T x H x N
