"""spmm_sweep_kernel (round 4: source sweep with the partial sums of two batch entries in the XCD's vector registers, gf_sweep_image.h)
against scipy in float64 and bit for bit against the SELL-8 kernel (same per-row summation order: ascending columns).
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
    tune(spmm_algo=0, spmm_lag=1, spmm_group=1)


def hop(plans, op, Xt, algo, **kw):
    tune(spmm_algo=algo, **kw)
    B, n, W = Xt.shape
    out = torch.full((B, n, W), float("nan"), device=DEV)
    _lib.check(_lib.lib().gf_spmm_hop(plans[0], op, Xt.data_ptr(), out.data_ptr(), B, W, stream()))
    torch.cuda.synchronize()
    return out


def er(n, deg, seed, directed=False):
    rng = np.random.RandomState(seed)
    r = np.repeat(np.arange(n), deg)
    c = rng.randint(0, n, size=r.size)
    A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(n, n))
    if not directed:
        A = A + A.T
    A = (A > 0).astype(np.float64)
    A.setdiag(0)
    A.eliminate_zeros()
    A[3, :] = 0                                 # an empty row (and, undirected, a sparse column)
    A = sp.csr_matrix(A)
    A.eliminate_zeros()
    return sp.csr_matrix(A * 0.0625)


@pytest.mark.parametrize("lag", [1, 2])
@pytest.mark.parametrize("n,deg,B,directed", [(12000, 4, 9, False), (40000, 5, 17, True), (100000, 5, 24, False), (102000, 3, 3, False), (20011, 6, 1, False), (60000, 4, 40, False)])
def test_sweep_hop_against_scipy_and_bitwise_against_sell(n, deg, B, directed, lag, knobs):
    A = er(n, deg, seed=n + B, directed=directed)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    rng = np.random.RandomState(1)
    X = rng.randn(B, n, 32).astype(np.float32)
    Xt = torch.tensor(X, device=DEV)
    for op, M in ((0, A.T.tocsr()), (1, A)):
        ref = hop(plans, op, Xt, 3)
        want = np.stack([M.astype(np.float64) @ X[b].astype(np.float64) for b in range(B)])
        assert relerr(ref.cpu().numpy(), want) < 2e-6
        for rep in range(3):                    # (the barrier counters live in the plan: launches in a row must all agree)
            got = hop(plans, op, Xt, 4, spmm_lag=lag)
            assert torch.equal(got, ref), (op, n, B, lag, rep, float((got - ref).abs().max()))


def test_sweep_hop_soak_at_config4_size(knobs):
    """The kernel addresses its accumulators relatively and keeps gathers in flight in registers it names by hand: 20 launches at the
    bench size (N = 1e5, ~1e6 entries, 64 batch entries: 8 per XCD back to back, barriers on) must all reproduce SELL-8's bits."""
    A = er(100000, 5, seed=11)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    Xt = torch.randn(64, 100000, 32, device=DEV)
    ref = hop(plans, 0, Xt, 3)
    for rep in range(20):
        got = hop(plans, 0, Xt, 4, spmm_lag=1)
        bad = int((got != ref).sum())
        assert bad == 0, (rep, bad)
