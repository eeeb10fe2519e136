"""spmm_stream_kernel (round 4: long-lived waves on the STREAM image, gf_stream_image.h) against scipy in float64 and -- bit for bit --
against the SELL-8 kernel it replaces on large graphs (same per-row summation order: ascending columns, one lane group per row).
Reference lines: the hop `x = torch.matmul(x, S)` of graphML.py:158-161."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from _util import relerr
from alegnn_amd import _lib
from alegnn_amd.gso import SparseGSO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def stream():
    return torch.cuda.current_stream().cuda_stream


def tune(**kw):
    L = _lib.lib()
    for k, v in kw.items():
        _lib.check(L.gf_tune(k.encode(), int(v)), "gf_tune " + k)


@pytest.fixture
def knobs():
    yield tune
    tune(spmm_algo=0, spmm_sd=0, spmm_wps=0, spmm_tk=-1, spmm_nc=0, panel_uniform=1, spmm_group=1)


def hop(plans, op, Xt, algo, **kw):
    tune(spmm_algo=algo, **kw)
    B, n, W = Xt.shape
    out = torch.full((B, n, W), float("nan"), device=DEV)
    _lib.check(_lib.lib().gf_spmm_hop(plans[0], op, Xt.data_ptr(), out.data_ptr(), B, W, stream()))
    torch.cuda.synchronize()
    return out


def graph(n, density, seed, uniform, hubs=True):
    rng = np.random.RandomState(seed)
    A = sp.random(n, n, density=density, random_state=rng, data_rvs=rng.randn, format="lil")
    if hubs and n > 200:
        A[5, :] = 1.0                      # a hub row and a hub column: slices longer than a run -> residual SELL image
        A[:, 9] = 1.0
        A[7, : n // 3] = 0.5               # a medium hub
    if n > 4:
        A[3, :] = 0                        # empty row / column
        A[:, 2] = 0
    A = sp.csr_matrix(A)
    if uniform:
        A.data[:] = 0.37
    return A


@pytest.mark.parametrize("variant", [dict(spmm_wps=4, spmm_tk=2), dict(spmm_wps=5, spmm_tk=0), dict(spmm_wps=8, spmm_tk=2, spmm_nc=1),
                                     dict(spmm_wps=4, spmm_tk=2, spmm_nc=3)])
@pytest.mark.parametrize("uniform", [0, 1])
@pytest.mark.parametrize("n,B", [(1003, 40), (203, 5), (4099, 17), (8, 1), (61, 9)])
def test_stream_hop_against_scipy_and_bitwise_against_sell(n, B, uniform, variant, knobs):
    A = graph(n, 0.01 if n > 500 else 0.05, seed=n + B, uniform=uniform)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    rng = np.random.RandomState(1)
    X = rng.randn(B, n, 32).astype(np.float32)
    Xt = torch.tensor(X, device=DEV)
    for op, M in ((0, A.T.tocsr()), (1, A)):
        got = hop(plans, op, Xt, 2, **variant)
        want = np.stack([M.astype(np.float64) @ X[b].astype(np.float64) for b in range(B)])
        assert relerr(got.cpu().numpy(), want) < 5e-6, (op, n, B, uniform, variant)      # (hub rows of ~n terms in fp32)
        ref = hop(plans, op, Xt, 3)
        assert torch.equal(got, ref), (op, n, B, uniform, variant)


def test_stream_hop_large_graph_with_locality_groups_and_repeated_launches(knobs):
    """N >= 32768: the schedule walks locality groups (the stream image packs each group on its own); 70 launches in a row rotate through
    every ticket-counter slot and must all give the same bits."""
    rng = np.random.RandomState(3)
    n, deg = 40000, 8
    r = np.repeat(np.arange(n), deg // 2)
    c = rng.randint(0, n, size=r.size)
    A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(n, n))
    A = ((A + A.T) > 0).astype(np.float64)
    A.setdiag(0)
    A.eliminate_zeros()
    A = sp.csr_matrix(A / 17.0)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    B = 11
    X = rng.randn(B, n, 32).astype(np.float32)
    Xt = torch.tensor(X, device=DEV)
    ref = hop(plans, 0, Xt, 3)
    want = np.stack([A.T.tocsr().astype(np.float64) @ X[b].astype(np.float64) for b in range(B)])
    assert relerr(ref.cpu().numpy(), want) < 2e-6
    for i in range(70):
        got = hop(plans, 0, Xt, 2)
        assert torch.equal(got, ref), i


def test_stream_hop_weighted_power_law(knobs):
    rng = np.random.RandomState(7)
    n = 6000
    rows = [np.full(5000, 0), np.full(2500, 17)]
    cols = [rng.choice(n, 5000, replace=False), rng.choice(n, 2500, replace=False)]
    deg = np.minimum((rng.pareto(1.5, n) * 3).astype(int), 300)
    for i in range(100, n):
        if deg[i]:
            rows.append(np.full(deg[i], i))
            cols.append(rng.choice(n, deg[i], replace=False))
    r, c = np.concatenate(rows), np.concatenate(cols)
    A = sp.csr_matrix((rng.randn(r.size), (r, c)), shape=(n, n))
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    X = rng.randn(3, n, 32).astype(np.float32)
    Xt = torch.tensor(X, device=DEV)
    for op, M in ((0, A.T.tocsr()), (1, A)):
        got = hop(plans, op, Xt, 2)
        want = np.stack([M @ X[b].astype(np.float64) for b in range(3)])
        assert relerr(got.cpu().numpy(), want) < 5e-6
        assert torch.equal(got, hop(plans, op, Xt, 3))
