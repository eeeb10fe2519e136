"""Pin oracle/lsigf_oracle.py against outputs of the real reference (tests/golden/*.npz).

CPU-only.  If these fail the oracle is wrong and no GPU parity claim means anything.
"""
import numpy as np
import pytest
import torch

from _util import case_id, golden_files, load, relerr
from oracle import lsigf_oracle as orc

LSIGF = golden_files("lsigf")
GFILT = golden_files("gfilter")


def test_fixtures_present():
    assert len(LSIGF) >= 9 and len(GFILT) >= 2


@pytest.mark.parametrize("path", LSIGF, ids=case_id)
def test_dense_restatement_matches_reference(path):
    d = load(path)
    h = torch.tensor(d["h"], requires_grad=True)
    x = torch.tensor(d["x"], requires_grad=True)
    b = torch.tensor(d["b"], requires_grad=True) if "b" in d else None
    y = orc.lsigf_dense(h, torch.tensor(d["S"]), x, b)
    y.backward(torch.tensor(d["dy"]))
    assert relerr(y.detach().numpy(), d["y"]) < 1e-13
    assert relerr(x.grad.numpy(), d["dx"]) < 1e-12
    assert relerr(h.grad.numpy(), d["dh"]) < 1e-12
    if b is not None:
        assert relerr(b.grad.numpy(), d["db"]) < 1e-12


@pytest.mark.parametrize("path", LSIGF, ids=case_id)
def test_sparse_restatement_matches_reference(path):
    d = load(path)
    b = d.get("b")
    y = orc.lsigf_sparse(d["h"], d["S"], d["x"], b)
    assert relerr(y, d["y"]) < 1e-12
    dx, dh, db = orc.lsigf_sparse_grads(d["h"], d["S"], d["x"], b, d["dy"])
    assert relerr(dx, d["dx"]) < 1e-12
    assert relerr(dh, d["dh"]) < 1e-12
    if b is not None:
        assert relerr(db, d["db"]) < 1e-12


@pytest.mark.parametrize("path", GFILT, ids=case_id)
def test_graph_filter_padding_semantics(path):
    d = load(path)
    y = orc.graph_filter_forward_sparse(d["weight"], d["bias"], d["S"], d["x"])
    assert y.shape == d["y"].shape
    assert relerr(y, d["y"]) < 1e-12
    yd = orc.graph_filter_forward_dense(torch.tensor(d["weight"]), torch.tensor(d["bias"]), torch.tensor(d["S"]),
                                        torch.tensor(d["x"]))
    assert relerr(yd.numpy(), d["y"]) < 1e-13


def test_step_helpers_agree():
    """The two cpu_baseline step functions (dense literal, torch sparse CSR) compute the same thing."""
    d = load([p for p in LSIGF if "fbego_G32" in p][0])
    w = torch.tensor(d["h"], dtype=torch.float32)
    b = torch.tensor(d["b"], dtype=torch.float32)
    x = torch.tensor(d["x"], dtype=torch.float32)
    S = torch.tensor(d["S"], dtype=torch.float32)
    y1, dx1, dw1, db1 = orc.graph_filter_step_dense(w, b, S, x)
    St = S[0].t().contiguous().to_sparse_csr()
    y2, dx2, dw2, db2 = orc.graph_filter_step_sparse_torch(w, b, St, x)
    assert relerr(y1.numpy(), d["y"]) < 1e-5
    assert relerr(y2.numpy(), d["y"]) < 1e-5
    assert relerr(dx2.numpy(), dx1.numpy()) < 1e-5
    assert relerr(dw2.numpy(), dw1.numpy()) < 1e-5
    assert relerr(db2.numpy(), db1.numpy()) < 1e-5


def test_evgf_dense_shape():
    rng = np.random.RandomState(0)
    Phi = torch.tensor(rng.randn(3, 1, 2, 2, 5, 5))
    x = torch.tensor(rng.randn(2, 2, 5))
    y = orc.evgf_dense(Phi, x, torch.tensor(rng.randn(3, 1)))
    assert y.shape == (2, 3, 5)
