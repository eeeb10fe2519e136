"""GPU parity of the per-sample / delayed GSO filters (csrc/gf_db.hip) and of edge gating (run with -m gpu on an MI355X):
  (1) every new C-ABI entry point (gf_db_hop, gf_db_grad_gso, gf_stack_adjoint) against numpy on ragged shapes,
  (2) LSIGF_DB / HiddenState_DB / edge-gated GatedGRNN / EdgeGatedHiddenState against the goldens produced by the real reference
      (tests/golden/{lsigfdb,grnndb,edgegrnn,edgehs}_*.npz) and against the pinned oracle (oracle/db_oracle.py) on seeded random inputs.
Tolerances: forward 1e-5 * max|ref|, gradients 1e-4 * max|ref|, against float64 references.
"""
import numpy as np
import pytest
import torch

from _util import FWD_RTOL, GRAD_RTOL, case_id, golden_files, load, relerr
from oracle import db_oracle as dbo

from alegnn_amd import _lib
from alegnn_amd.utils import graphML as gml

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def cu(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
    return t.requires_grad_(True) if grad else t


def stream():
    return torch.cuda.current_stream().cuda_stream


def raw(path):
    return dict(np.load(path, allow_pickle=False))


# ---------------------------------------------------------------------------------------------------------------
# unit level
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nb,nt,E,N,W,shift", [(2, 5, 1, 20, 8, 1), (3, 4, 2, 33, 32, 1), (1, 1, 1, 7, 4, 0), (2, 3, 1, 100, 64, 0),
                                              (2, 2, 2, 65, 24, 1), (1, 6, 1, 129, 32, 0), (4, 1, 1, 50, 128, 0), (1, 3, 1, 1, 8, 1)])
def test_db_hop_and_its_adjoint(nb, nt, E, N, W, shift):
    """X_out[b,t] = X_in[b,t-shift] @ S[b,t] (op 0) and its adjoint (op 1), edge feature e of a [nb,nt,E,N,N] tensor, rows of W floats."""
    L = _lib.lib()
    rng = np.random.RandomState(nb * 100 + N)
    S = (rng.rand(nb, nt, E, N, N) < 0.3) * rng.randn(nb, nt, E, N, N)
    X = rng.randn(nb, nt, N, W)
    St, Xt = cu(S), cu(X)
    for e in range(E):
        for op in (0, 1):
            out = torch.full((nb, nt, N, W), float("nan"), device=DEV)
            _lib.check(L.gf_db_hop(St.data_ptr() + e * N * N * 4, nt * E * N * N, E * N * N, Xt.data_ptr(), out.data_ptr(), nb, nt, N, W, op,
                                   shift, stream()), "gf_db_hop")
            want = np.zeros((nb, nt, N, W))
            for t in range(nt):
                if op == 0 and t - shift >= 0:
                    want[:, t] = np.einsum("bmn,bmw->bnw", S[:, t, e], X[:, t - shift])
                if op == 1 and t + shift < nt:
                    want[:, t] = np.einsum("bmn,bnw->bmw", S[:, t + shift, e], X[:, t + shift])
            assert relerr(out.cpu().numpy(), want) < FWD_RTOL
    # <u, hop(v)> == <adjoint(u), v>: the two ops are adjoint to each other (what autograd relies on)
    U = cu(rng.randn(nb, nt, N, W))
    a = torch.empty_like(Xt)
    bq = torch.empty_like(Xt)
    _lib.check(L.gf_db_hop(St.data_ptr(), nt * E * N * N, E * N * N, Xt.data_ptr(), a.data_ptr(), nb, nt, N, W, 0, shift, stream()))
    _lib.check(L.gf_db_hop(St.data_ptr(), nt * E * N * N, E * N * N, U.data_ptr(), bq.data_ptr(), nb, nt, N, W, 1, shift, stream()))
    lhs, rhs = float((U.double() * a.double()).sum()), float((bq.double() * Xt.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


@pytest.mark.parametrize("nb,nt,N,W,shift", [(2, 3, 20, 8, 0), (1, 4, 33, 32, 1), (2, 1, 70, 64, 0), (1, 2, 5, 4, 1)])
def test_db_grad_gso(nb, nt, N, W, shift):
    """dS[b,t,m,n] = sum_w X[b,t-shift,m,w] dOut[b,t,n,w], written or accumulated."""
    L = _lib.lib()
    rng = np.random.RandomState(7 + N)
    X, D = rng.randn(nb, nt, N, W), rng.randn(nb, nt, N, W)
    want = np.zeros((nb, nt, N, N))
    for t in range(nt):
        if t - shift >= 0:
            want[:, t] = np.einsum("bmw,bnw->bmn", X[:, t - shift], D[:, t])
    dS = torch.full((nb, nt, 1, N, N), float("nan"), device=DEV)
    Xt, Dt = cu(X), cu(D)
    _lib.check(L.gf_db_grad_gso(Xt.data_ptr(), Dt.data_ptr(), dS.data_ptr(), nt * N * N, N * N, nb, nt, N, W, shift, 0, stream()))
    assert relerr(dS[:, :, 0].cpu().numpy(), want) < FWD_RTOL
    _lib.check(L.gf_db_grad_gso(Xt.data_ptr(), Dt.data_ptr(), dS.data_ptr(), nt * N * N, N * N, nb, nt, N, W, shift, 1, stream()))
    assert relerr(dS[:, :, 0].cpu().numpy(), 2 * want) < FWD_RTOL


@pytest.mark.parametrize("BN,G,F,E,K", [(50, 8, 8, 1, 3), (129, 32, 16, 2, 4), (7, 4, 40, 1, 1), (300, 64, 32, 1, 5)])
def test_stack_adjoint(BN, G, F, E, K):
    L = _lib.lib()
    rng = np.random.RandomState(BN)
    P0, h = rng.randn(BN, F), rng.randn(F, E, K, G)
    T = 1 + E * (K - 1)
    dZ = torch.full((T, BN, G), float("nan"), device=DEV)
    Pt, ht = cu(P0), cu(h)                                       # (keep the tensors alive across the call)
    _lib.check(L.gf_stack_adjoint(Pt.data_ptr(), ht.data_ptr(), dZ.data_ptr(), BN, G, F, E, K, stream()))
    want = np.zeros((T, BN, G))
    want[0] = P0 @ h[:, :, 0, :].sum(axis=1)
    for e in range(E):
        for k in range(1, K):
            want[1 + e * (K - 1) + (k - 1)] = P0 @ h[:, e, k, :]
    assert relerr(dZ.cpu().numpy(), want) < FWD_RTOL


def test_db_shape_errors():
    L = _lib.lib()
    t = torch.zeros(64, device=DEV)
    assert L.gf_db_hop(t.data_ptr(), 1, 1, t.data_ptr(), t.data_ptr(), 1, 1, 2, 6, 0, 0, stream()) == _lib.GF_ERR_SHAPE   # W % 4
    assert L.gf_db_hop(None, 1, 1, t.data_ptr(), t.data_ptr(), 1, 1, 2, 4, 0, 0, stream()) != 0
    assert L.gf_db_hop(t.data_ptr(), 1, 1, t.data_ptr(), t.data_ptr(), 1, 1, 2, 4, 0, 2, stream()) != 0                    # shift in {0, 1}


# ---------------------------------------------------------------------------------------------------------------
# against the reference's own outputs
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("lsigfdb"), ids=case_id)
def test_lsigf_db_matches_reference(path):
    """LSIGF_DB (graphML.py:977-1094) and GraphFilter_DB (:3278-3393): delayed filter on per-sample operators, all gradients."""
    d = raw(path)
    h, x = cu(d["h"], True), cu(d["x"], True)
    b = cu(d["b"], True) if "b" in d else None
    y = gml.LSIGF_DB(h, cu(d["S"]), x, b)
    y.backward(cu(d["dy"]))
    assert relerr(y.detach().cpu().numpy(), d["y"]) < FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(h.grad.cpu().numpy(), d["dh"]) < GRAD_RTOL
    if b is not None:
        assert relerr(b.grad.cpu().numpy(), d["db"]) < GRAD_RTOL
    F_, E, K, G = d["h"].shape
    layer = gml.GraphFilter_DB(G, F_, K, E, bias="b" in d)
    layer.load_state_dict({"weight": torch.tensor(d["h"]), **({"bias": torch.tensor(d["b"])} if "b" in d else {})})
    layer = layer.float().to(DEV)
    layer.addGSO(cu(d["S"]))
    assert relerr(layer(cu(d["x"])).detach().cpu().numpy(), d["y"]) < FWD_RTOL
    assert "GSO stored" in layer.extra_repr()


@pytest.mark.parametrize("path", golden_files("grnndb"), ids=case_id)
def test_hidden_state_db_matches_reference(path):
    """HiddenState_DB / GRNN_DB (graphML.py:1096-1290, 3395-3538): the reference's state_dict loads; states and every gradient,
    including the ones that reach z0 and the taps through the whole recursion."""
    d = raw(path)
    F, H, K, E = (int(v) for v in d["dims"])
    layer = gml.HiddenState_DB(F, H, K, nonlinearity=torch.tanh, E=E, bias=True)
    layer.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    layer = layer.float().to(DEV)
    layer.addGSO(cu(d["S"]))
    x, z0 = cu(d["x"], True), cu(d["z0"], True)
    z, zT = layer(x, z0)
    assert list(zT.shape) == d["zT_shape"].tolist()
    (z * cu(d["dz"])).sum().backward()
    assert relerr(z.detach().cpu().numpy(), d["z"]) < 2 * FWD_RTOL          # T chained steps + tanh
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(z0.grad.cpu().numpy(), d["dz0"]) < GRAD_RTOL
    for k, p in layer.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k


@pytest.mark.parametrize("path", golden_files("edgegrnn"), ids=case_id)
def test_edge_gated_grnn_matches_reference(path):
    """GatedGRNN with edge gates (graphML.py:1394-1419, :1434-1456), gate gradients included."""
    d = load(path)
    H, E, K, F = d["sd:aWeights"].shape
    layer = gml.HiddenState(F, H, K, nonlinearity=torch.tanh, E=E, bias=True)
    layer.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    layer.addGSO(torch.tensor(d["S"]))
    layer = layer.float().to(DEV)
    x, z0 = cu(d["x"], True), cu(d["z0"], True)
    gates = {k: cu(d[k], True) for k in ("q_hat", "q_check") if k in d}
    z = gml.GatedGRNN(layer.aWeights, layer.bWeights, layer._gso, x, z0, torch.tanh, xBias=layer.xBias, zBias=layer.zBias, **gates)
    (z * cu(d["dz"])).sum().backward()
    assert relerr(z.detach().cpu().numpy(), d["z"]) < 2 * FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(z0.grad.cpu().numpy(), d["dz0"]) < GRAD_RTOL
    for k, p in layer.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k
    for k, g in gates.items():
        assert relerr(g.grad.cpu().numpy(), d["d" + k]) < GRAD_RTOL, k


@pytest.mark.parametrize("path", golden_files("edgehs"), ids=case_id)
def test_edge_gated_hidden_state_matches_reference(path):
    """EdgeGatedHiddenState (graphML.py:4033-4208): the reference's state_dict (attention gate networks included) loads."""
    d = load(path)
    F, H, K, E = (int(v) for v in d["dims"])
    layer = gml.EdgeGatedHiddenState(F, H, K, nonlinearity=torch.tanh, E=E, bias=True)
    layer.addGSO(torch.tensor(d["S"]))
    layer.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    layer = layer.float().to(DEV)
    x, z0 = cu(d["x"], True), cu(d["z0"], True)
    z, zT = layer(x, z0)
    assert list(zT.shape) == d["zT_shape"].tolist()
    (z * cu(d["dz"])).sum().backward()
    assert relerr(z.detach().cpu().numpy(), d["z"]) < 3 * FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(z0.grad.cpu().numpy(), d["dz0"]) < GRAD_RTOL
    for k, p in layer.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < 2 * GRAD_RTOL, k


# ---------------------------------------------------------------------------------------------------------------
# against the pinned oracle, flocking-sized and ragged
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,E,N,G,F,K", [(4, 12, 1, 50, 6, 32, 3), (2, 7, 2, 77, 32, 32, 4), (3, 3, 1, 130, 1, 5, 5), (1, 1, 1, 9, 3, 3, 2)])
def test_lsigf_db_against_oracle(B, T, E, N, G, F, K):
    rng = np.random.RandomState(B * 10 + N)
    S = (rng.rand(B, T, E, N, N) < 0.15) * rng.rand(B, T, E, N, N) / 3.0
    h, x, b, dy = rng.randn(F, E, K, G) / np.sqrt(G * K), rng.randn(B, T, G, N), rng.randn(F, 1), rng.randn(B, T, F, N)
    ref = [torch.tensor(v, requires_grad=True) for v in (h, x, b)]
    yr = dbo.lsigf_db(ref[0], torch.tensor(S), ref[1], ref[2])
    yr.backward(torch.tensor(dy))
    got = [cu(v, True) for v in (h, x, b)]
    y = gml.LSIGF_DB(got[0], cu(S), got[1], got[2])
    y.backward(cu(dy))
    assert relerr(y.detach().cpu().numpy(), yr.detach().numpy()) < FWD_RTOL
    for g, r in zip(got, ref):
        assert relerr(g.grad.cpu().numpy(), r.grad.numpy()) < GRAD_RTOL
    y2 = gml.LSIGF_DB(got[0].detach(), cu(S), got[1].detach(), got[2].detach())
    assert torch.equal(y2, y.detach())                                   # run-to-run bitwise determinism


def test_grnn_db_against_oracle_flocking_size():
    B, T, E, N, F, H, K = 4, 10, 1, 50, 6, 16, 3
    rng = np.random.RandomState(5)
    S = (rng.rand(B, T, E, N, N) < 0.15) * rng.rand(B, T, E, N, N) / 4.0
    arrs = dict(a=rng.randn(H, E, K, F) / np.sqrt(F * K), b=rng.randn(H, E, K, H) / np.sqrt(H * K), x=rng.randn(B, T, F, N), z0=rng.randn(B, H, N),
                xb=rng.randn(H, 1) * 0.1, zb=rng.randn(H, 1) * 0.1)
    dz = rng.randn(B, T, H, N)
    ref = {k: torch.tensor(v, requires_grad=True) for k, v in arrs.items()}
    zr = dbo.grnn_db(ref["a"], ref["b"], torch.tensor(S), ref["x"], ref["z0"], torch.tanh, ref["xb"], ref["zb"])
    (zr * torch.tensor(dz)).sum().backward()
    got = {k: cu(v, True) for k, v in arrs.items()}
    z = gml.GRNN_DB(got["a"], got["b"], cu(S), got["x"], got["z0"], torch.tanh, got["xb"], got["zb"])
    (z * cu(dz)).sum().backward()
    assert relerr(z.detach().cpu().numpy(), zr.detach().numpy()) < 2 * FWD_RTOL
    for k in arrs:
        assert relerr(got[k].grad.cpu().numpy(), ref[k].grad.numpy()) < GRAD_RTOL, k


# ---------------------------------------------------------------------------------------------------------------
# edge cases of the host layer: missing biases, per-node bias, one tap, a single time step, gates of mixed kinds
# ---------------------------------------------------------------------------------------------------------------
def test_lsigf_db_per_node_bias_and_single_step():
    """bias [F,N] (graphML.py:1008-1010: added after the filter, not fused) and T = 1 (every delayed tap is zero)."""
    rng = np.random.RandomState(11)
    B, T, E, N, G, F, K = 3, 1, 1, 21, 5, 7, 3
    S = (rng.rand(B, T, E, N, N) < 0.3) * rng.randn(B, T, E, N, N)
    h, x, b, dy = rng.randn(F, E, K, G), rng.randn(B, T, G, N), rng.randn(F, N), rng.randn(B, T, F, N)
    ref = [torch.tensor(v, requires_grad=True) for v in (h, x, b)]
    yr = dbo.lsigf_db(ref[0], torch.tensor(S), ref[1], ref[2])
    yr.backward(torch.tensor(dy))
    got = [cu(v, True) for v in (h, x, b)]
    y = gml.LSIGF_DB(got[0], cu(S), got[1], got[2])
    y.backward(cu(dy))
    assert relerr(y.detach().cpu().numpy(), yr.detach().numpy()) < FWD_RTOL
    for g, r in zip(got, ref):
        assert relerr(g.grad.cpu().numpy(), r.grad.numpy()) < GRAD_RTOL
    # with T = 1 only tap 0 contributes: the taps k >= 1 get exactly zero gradient
    assert float(got[0].grad[:, :, 1:].abs().max()) == 0.0


def test_hidden_state_db_without_bias_and_one_tap():
    rng = np.random.RandomState(12)
    B, T, E, N, F, H, K = 2, 4, 1, 18, 3, 8, 1
    S = (rng.rand(B, T, E, N, N) < 0.3) * rng.rand(B, T, E, N, N)
    layer = gml.HiddenState_DB(F, H, K, nonlinearity=torch.tanh, E=E, bias=False).to(DEV)
    layer.addGSO(cu(S))
    x, z0 = rng.randn(B, T, F, N), rng.randn(B, H, N)
    xt, z0t = cu(x, True), cu(z0, True)
    z, zT = layer(xt, z0t)
    assert tuple(zT.shape) == (B, 1, 1, H, N)
    z.sum().backward()
    a, b_ = (p.detach().cpu().double().requires_grad_(True) for p in (layer.aWeights, layer.bWeights))
    xr, z0r = torch.tensor(x, requires_grad=True), torch.tensor(z0, requires_grad=True)
    zr = dbo.grnn_db(a, b_, torch.tensor(S), xr, z0r, torch.tanh, None, None)
    zr.sum().backward()
    assert relerr(z.detach().cpu().numpy(), zr.detach().numpy()) < 2 * FWD_RTOL
    assert relerr(xt.grad.cpu().numpy(), xr.grad.numpy()) < GRAD_RTOL
    assert relerr(z0t.grad.cpu().numpy(), z0r.grad.numpy()) < GRAD_RTOL
    assert relerr(layer.aWeights.grad.cpu().numpy(), a.grad.numpy()) < GRAD_RTOL
    assert relerr(layer.bWeights.grad.cpu().numpy(), b_.grad.numpy()) < GRAD_RTOL


def test_per_sample_filter_one_tap_gives_zero_gate_gradient():
    """K = 1: the operator is never applied, so an edge gate gets a zero gradient (and the kernels are not asked for one)."""
    from alegnn_amd.functional_db import filter_per_sample
    rng = np.random.RandomState(13)
    B, T, N, G, F = 2, 3, 10, 8, 8
    S5 = cu(rng.rand(B, T, 1, N, N), True)
    h, x = cu(rng.randn(F, 1, 1, G), True), cu(rng.randn(B, T, G, N), True)
    y = filter_per_sample(h, S5, x, None)
    y.sum().backward()
    assert float(S5.grad.abs().max()) == 0.0
    want = torch.einsum("fg,btgn->btfn", h.detach()[:, 0, 0], x.detach())
    assert relerr(y.detach().cpu().numpy(), want.cpu().numpy()) < FWD_RTOL


def test_mixed_gates_edge_input_gate_with_node_forget_gate():
    """q_hat an edge gate, q_check a node gate: each branch of GatedGRNN is independent (graphML.py:1385, :1421)."""
    rng = np.random.RandomState(14)
    B, T, N, F, H, K = 2, 3, 16, 3, 8, 3
    S = ((rng.rand(1, N, N) < 0.3) * rng.randn(1, N, N)).astype(np.float64)
    arrs = dict(a=rng.randn(H, 1, K, F) / 3, b=rng.randn(H, 1, K, H) / 5, x=rng.randn(B, T, F, N), z0=rng.randn(B, H, N), qh=rng.rand(B, T, 1, N, N),
                qc=rng.rand(B, T, 1, N), xb=rng.randn(H, 1) * 0.1, zb=rng.randn(H, 1) * 0.1)
    ref = {k: torch.tensor(v, requires_grad=True) for k, v in arrs.items()}
    zr = dbo.gated_grnn(ref["a"], ref["b"], torch.tensor(S), ref["x"], ref["z0"], torch.tanh, ref["qh"], ref["qc"], ref["xb"], ref["zb"])
    zr.sum().backward()
    got = {k: cu(v, True) for k, v in arrs.items()}
    z = gml.GatedGRNN(got["a"], got["b"], torch.tensor(S), got["x"], got["z0"], torch.tanh, got["qh"], got["qc"], got["xb"], got["zb"])
    z.sum().backward()
    assert relerr(z.detach().cpu().numpy(), zr.detach().numpy()) < 2 * FWD_RTOL
    for k in arrs:
        assert relerr(got[k].grad.cpu().numpy(), ref[k].grad.numpy()) < GRAD_RTOL, k
