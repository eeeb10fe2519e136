"""GPU parity at the FULL sizes of BASELINE.json's configs (the sizes bench.py times), against the CPU oracle.

The oracle cannot run a whole full-size step in seconds, so each check uses what the mathematics offers:
  * forward and dx are independent per batch entry          -> a few entries of the full batch;
  * dh / db are sums over the batch                          -> the oracle is streamed over the batch in float64;
  * EVGF outputs are sums of independent (f, g) chains, its weight gradients are per chain
                                                             -> a few output features f (all g), incl. the LAST one: its per-edge
                                                                taps sit past 2^31 bytes in the 4.5 GB parameter tensor.
Tolerances: tests/_util.py (forward 1e-5, gradients 1e-4, relative to the largest reference entry, reference in float64).
Reference lines: graphML.py:152-175 (LSIGF), :2125-2144 (GraphFilter.forward), :457-488 and :2670-2698 (EVGF / EdgeVariantGF).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from _util import FWD_RTOL, GRAD_RTOL, relerr

from alegnn_amd import _lib, graphgen
from alegnn_amd.utils import graphML as gml
from oracle import evgf_oracle as evo
from oracle import lsigf_oracle as orc

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def streamed_tap_grads(h, A, x, dy, chunk):
    """dh [F,1,K,G], db [F,1] of LSIGF with the oracle's taps (lsigf_taps_sparse, float64), `chunk` batch entries at a time."""
    F_, E, K, G = h.shape
    dh = np.zeros((F_, K, G))
    db = np.zeros(F_)
    for b0 in range(0, x.shape[0], chunk):
        z = orc.lsigf_taps_sparse(A, x[b0:b0 + chunk], K)[:, 0]               # [b, K, G, N]
        d = dy[b0:b0 + chunk].astype(np.float64)
        dh += np.einsum("bkgn,bfn->fkg", z, d, optimize=True)
        db += d.sum(axis=(0, 2))
    return dh[:, None], db[:, None]


def test_config4_full_batch_against_oracle():
    """BASELINE configs[3] per-GPU shard exactly as bench.py runs it: ER N = 1e5, nnz ~ 1e6, B = 128, 32 -> 32, K = 5 (node-major
    pipeline, the batch-tile / prefetch heuristics of that batch size).  y and dx on three batch entries, dh and db on the whole
    batch."""
    N, B, G, F, K = 100_000, 128, 32, 32, 5
    A = graphgen.er(N, seed=0)
    torch.manual_seed(1)
    layer = gml.GraphFilter(G, F, K, 1, True)
    layer.addGSO(A)
    layer.to(DEV)
    assert _lib.lib().gf_lsigf_pipeline(layer._gso.plans(DEV), 1, G, F, K) == 1
    x = torch.randn(B, G, N, device=DEV, requires_grad=True)
    y = layer(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    w, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    sl = [0, 64, 127]
    xs, dys = x.detach()[sl].cpu().numpy(), dy[sl].cpu().numpy()
    assert relerr(y.detach()[sl].cpu().numpy(), orc.lsigf_sparse(w, A, xs, b)) < FWD_RTOL
    dx, _, _ = orc.lsigf_sparse_grads(w, A, xs, b, dys)
    assert relerr(x.grad[sl].cpu().numpy(), dx) < GRAD_RTOL
    dh, db = streamed_tap_grads(w, A, x.detach().cpu().numpy(), dy.cpu().numpy(), chunk=16)
    assert relerr(layer.weight.grad.cpu().numpy(), dh) < GRAD_RTOL
    assert relerr(layer.bias.grad.cpu().numpy(), db) < GRAD_RTOL


def test_config2_full_batch_tap_gradients_against_oracle():
    """BASELINE configs[1] at full size (N = 1e4, B = 256): dh and db of the panel pipeline's one-pass backward -- a reduction over
    2.56 M rows -- against the oracle streamed over the whole batch; y and dx on three entries."""
    N, B, G, F, K = 10_000, 256, 32, 32, 5
    A = graphgen.sbm(N, seed=0)
    torch.manual_seed(0)
    layer = gml.GraphFilter(G, F, K, 1, True)
    layer.addGSO(A)
    layer.to(DEV)
    assert _lib.lib().gf_lsigf_pipeline(layer._gso.plans(DEV), 1, G, F, K) == 2
    x = torch.randn(B, G, N, device=DEV, requires_grad=True)
    y = layer(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    w, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    sl = [0, 101, 255]
    xs, dys = x.detach()[sl].cpu().numpy(), dy[sl].cpu().numpy()
    assert relerr(y.detach()[sl].cpu().numpy(), orc.lsigf_sparse(w, A, xs, b)) < FWD_RTOL
    dx, _, _ = orc.lsigf_sparse_grads(w, A, xs, b, dys)
    assert relerr(x.grad[sl].cpu().numpy(), dx) < GRAD_RTOL
    dh, db = streamed_tap_grads(w, A, x.detach().cpu().numpy(), dy.cpu().numpy(), chunk=64)
    assert relerr(layer.weight.grad.cpu().numpy(), dh) < GRAD_RTOL
    assert relerr(layer.bias.grad.cpu().numpy(), db) < GRAD_RTOL


def test_config3_two_layers_against_oracle():
    """BASELINE configs[2] shapes: MovieLens-100k-sized weighted kNN graph (N = 1682), batch 256, GraphFilter 1 -> 64 -> 32, K = 5 with
    a ReLU in between (architectures.py:286-289) -- both layers, forward and every gradient, whole batch."""
    N, B, K = 1682, 256, 5
    A = graphgen.knn_weighted(N, k=10, seed=0)
    torch.manual_seed(3)
    l1, l2 = gml.GraphFilter(1, 64, K, 1, True), gml.GraphFilter(64, 32, K, 1, True)
    for l in (l1, l2):
        l.addGSO(A)
        l.to(DEV)
    x = torch.randn(B, 1, N, device=DEV, requires_grad=True)
    h1 = l1(x)
    a1 = torch.relu(h1)
    a1.retain_grad()
    y = l2(a1)
    dy = torch.randn_like(y)
    y.backward(dy)
    w1, b1 = l1.weight.detach().cpu().numpy(), l1.bias.detach().cpu().numpy()
    w2, b2 = l2.weight.detach().cpu().numpy(), l2.bias.detach().cpu().numpy()
    xn = x.detach().cpu().numpy()
    h1o = orc.lsigf_sparse(w1, A, xn, b1)
    assert relerr(h1.detach().cpu().numpy(), h1o) < FWD_RTOL
    a1o = np.maximum(h1o, 0.0)
    assert relerr(y.detach().cpu().numpy(), orc.lsigf_sparse(w2, A, a1o, b2)) < FWD_RTOL
    dyn = dy.cpu().numpy()
    da1, dh2, db2 = orc.lsigf_sparse_grads(w2, A, a1o, b2, dyn)
    assert relerr(a1.grad.cpu().numpy(), da1) < GRAD_RTOL
    assert relerr(l2.weight.grad.cpu().numpy(), dh2) < GRAD_RTOL
    assert relerr(l2.bias.grad.cpu().numpy(), db2) < GRAD_RTOL
    dh1o = da1 * (h1o > 0)                                                    # through the ReLU
    dx, dh1, db1 = orc.lsigf_sparse_grads(w1, A, xn, b1, dh1o)
    assert relerr(x.grad.cpu().numpy(), dx) < GRAD_RTOL
    assert relerr(l1.weight.grad.cpu().numpy(), dh1) < GRAD_RTOL
    assert relerr(l1.bias.grad.cpu().numpy(), db1) < GRAD_RTOL


def test_config5_full_size_against_oracle():
    """BASELINE configs[4] at full size: EdgeVariantGF with per-edge parameters (sparse=True), SBM N = 5e4, nnz' ~ 5.5e5, F = G = 32,
    K = 3, batch 16.  The per-edge tap tensor is F*(K-1)*G*nnz' = 1.1e9 floats (4.5 GB): the last output feature lives past byte
    offset 2^31 and past float index 2^30."""
    N, B, G, F, K = 50_000, 16, 32, 32, 3
    A = graphgen.sbm(N, seed=0)
    torch.manual_seed(5)
    layer = gml.EdgeVariantGF(G, F, K, N, N, 1, True, sparse=True)
    layer.addGSO(A)
    layer.to(DEV)
    pat = layer._patterns[0]
    assert F * (K - 1) * G * pat.nnzp * 4 > 2 ** 32
    P = sp.csr_matrix((np.ones(pat.nnzp), pat.indices, pat.indptr), shape=(N, N))
    x = torch.randn(B, G, N, device=DEV, requires_grad=True)
    y = layer(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    xn, dyn = x.detach().cpu().numpy(), dy.cpu().numpy()
    wdg, wed = layer.weightEVdiag, layer.weightEVedges[0]
    bias = layer.bias.detach().cpu().numpy()
    fsel = [0, 13, 31]
    wd = wdg.detach()[fsel, 0].cpu().numpy()                                   # [3, G, N]
    we = wed.detach()[fsel].cpu().numpy()                                      # [3, K-1, G, nnzp]
    # forward: y_f = sum over the 32 chains (f, g) (+ bias) -- the selected features, whole batch
    want = evo.evgf_sparse(P, wd, we, xn, bias[fsel])
    assert relerr(y.detach()[:, fsel].cpu().numpy(), want) < FWD_RTOL
    # parameter gradients are per chain: the chains of the selected features, whole batch
    _, dwd, dwe, dbo = evo.evgf_sparse_grads(P, wd, we, xn, dyn[:, fsel])
    assert relerr(wdg.grad[fsel, 0].cpu().numpy(), dwd) < GRAD_RTOL
    assert relerr(wed.grad[fsel].cpu().numpy(), dwe) < GRAD_RTOL
    assert relerr(layer.bias.grad[fsel].cpu().numpy(), dbo) < GRAD_RTOL
    # dx sums over all F*G chains but is independent per batch entry: one entry, every chain (SURVEY.md A.2:
    # u_{K-1} = dy_f, u_k = Phi_{k+1}^T u_{k+1} + dy_f, dx_g = sum_f Phi_0 u_0)
    bsel = 11
    wd_all = wdg.detach()[:, 0].cpu().numpy().astype(np.float64)
    dxo = np.zeros((G, N))
    for f in range(F):
        we_f = wed.detach()[f].cpu().numpy().astype(np.float64)               # [K-1, G, nnzp]
        d = dyn[bsel, f].astype(np.float64)
        for g in range(G):
            u = d
            for k in range(K - 1, 0, -1):
                Phi = sp.csr_matrix((we_f[k - 1, g], pat.indices, pat.indptr), shape=(N, N))
                u = Phi.T @ u + d
            dxo[g] += wd_all[f, g] * u
    assert relerr(x.grad[bsel].cpu().numpy(), dxo) < GRAD_RTOL


def test_wide_layer_at_config4_size_against_oracle():
    """The widths of BASELINE configs[2]'s second layer (64 -> 32, movieGNN's F = [1, 64, 32], reference examples/movieGNN.py:259-276) on
    config 4's graph: the forward hops move 64-column rows (round 6: two slabs of the sweep's image), the adjoint hops 32-column rows.
    The DEFAULT path (this test also runs in the product configuration), y / dx on two batch entries, dh / db on the whole batch."""
    N, B, G, F, K = 100_000, 8, 64, 32, 3
    A = graphgen.er(N, seed=0)
    torch.manual_seed(4)
    layer = gml.GraphFilter(G, F, K, 1, True)
    layer.addGSO(A)
    layer.to(DEV)
    plans = layer._gso.plans(DEV)
    assert _lib.lib().gf_lsigf_pipeline(plans, 1, G, F, K) == 1
    assert _lib.lib().gf_spmm_hop_kernel(plans[0], 0, B, G) == 1 and _lib.lib().gf_spmm_hop_kernel(plans[0], 1, B, F) == 1   # the sweep, both ways
    x = torch.randn(B, G, N, device=DEV, requires_grad=True)
    y = layer(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    w, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    sl = [0, 7]
    xs, dys = x.detach()[sl].cpu().numpy(), dy[sl].cpu().numpy()
    assert relerr(y.detach()[sl].cpu().numpy(), orc.lsigf_sparse(w, A, xs, b)) < FWD_RTOL
    dx, _, _ = orc.lsigf_sparse_grads(w, A, xs, b, dys)
    assert relerr(x.grad[sl].cpu().numpy(), dx) < GRAD_RTOL
    dh, db = streamed_tap_grads(w, A, x.detach().cpu().numpy(), dy.cpu().numpy(), chunk=4)
    assert relerr(layer.weight.grad.cpu().numpy(), dh) < GRAD_RTOL
    assert relerr(layer.bias.grad.cpu().numpy(), db) < GRAD_RTOL


def test_power_law_graph_at_config4_size_against_oracle():
    """A directed graph whose out-degrees have a Barabasi-Albert tail (rows of hundreds of entries among rows of five): the sweep's image leaves
    the long rows out of its groups and the waves sum them separately (round 6, hub rows; the longest ones as 32 partial chains).  Default path,
    forward and backward (the two orientations: one with hub rows, one without), against the float64 oracle."""
    N, B, G, F, K = 100_000, 8, 32, 32, 3
    rng = np.random.RandomState(12)
    deg = np.minimum(N // 8, (5.0 / np.sqrt(np.maximum(rng.uniform(size=N), 1e-9))).astype(np.int64))
    r = np.repeat(np.arange(N), deg)
    A = sp.csr_matrix((np.ones(r.size), (r, rng.randint(0, N, size=r.size))), shape=(N, N))
    A.sum_duplicates()
    A.data[:] = 1.0 / 64.0
    A = sp.csr_matrix(A)
    torch.manual_seed(5)
    layer = gml.GraphFilter(G, F, K, 1, True)
    layer.addGSO(A)
    layer.to(DEV)
    plans = layer._gso.plans(DEV)
    assert _lib.lib().gf_spmm_hop_kernel(plans[0], 0, B, G) == 1 and _lib.lib().gf_spmm_hop_kernel(plans[0], 1, B, F) == 1
    x = torch.randn(B, G, N, device=DEV, requires_grad=True)
    y = layer(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    w, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    sl = [0, 5]
    xs, dys = x.detach()[sl].cpu().numpy(), dy[sl].cpu().numpy()
    assert relerr(y.detach()[sl].cpu().numpy(), orc.lsigf_sparse(w, A, xs, b)) < FWD_RTOL
    dx, _, _ = orc.lsigf_sparse_grads(w, A, xs, b, dys)
    assert relerr(x.grad[sl].cpu().numpy(), dx) < GRAD_RTOL
    dh, db = streamed_tap_grads(w, A, x.detach().cpu().numpy(), dy.cpu().numpy(), chunk=4)
    assert relerr(layer.weight.grad.cpu().numpy(), dh) < GRAD_RTOL
    assert relerr(layer.bias.grad.cpu().numpy(), db) < GRAD_RTOL


def _large_files():
    import glob
    import os
    from _util import GOLDEN
    return sorted(glob.glob(os.path.join(GOLDEN, "large", "gfilter_*.npz")))


@pytest.mark.parametrize("path", _large_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_sweep_size_layer_against_the_literal_reference(path):
    """tests/golden/large/ (round 6): the reference's own GraphFilter run with the dense 49 152 x 49 152 GSO in float64 -- forward + autograd -- at the
    smallest size where the node-major hops are the MFMA source sweep: an undirected weighted graph, 32 -> 32, Nin < N (layout pass inside the fused
    launch, both orientations); a directed power-law graph, 64 -> 32 (wide rows forward, hub rows in the adjoint orientation); and the same literal run at config 2's size
    and taps (N = 1e4, K = 5: the LDS panel pipeline).  Default path; y and dx
    at the fixture's 1024 nodes, their sums of squares over all nodes, dweight and dbias in full."""
    from _util import large_gfilter_inputs
    d = dict(np.load(path, allow_pickle=False))
    N, B, G, F, K, Nin, seed = (int(v) for v in d["cfg"][:7])
    kind = int(d["cfg"][8]) if len(d["cfg"]) > 8 else 0
    A, x = large_gfilter_inputs(N, B, G, Nin, seed, kind)
    dy = np.random.RandomState(seed + 1).randn(B, F, Nin)
    assert A.nnz == int(d["check"][0]) and abs(x.sum() - d["check"][2]) < 1e-6
    layer = gml.GraphFilter(G, F, K, 1, True)
    layer.load_state_dict({"weight": torch.tensor(d["weight"], dtype=torch.float32), "bias": torch.tensor(d["bias"], dtype=torch.float32)})
    layer.addGSO(A)
    layer.to(DEV)
    plans = layer._gso.plans(DEV)
    if N >= 49152:
        assert _lib.lib().gf_spmm_hop_kernel(plans[0], 0, B, G) == 1 and _lib.lib().gf_spmm_hop_kernel(plans[0], 1, B, F) == 1
    else:                                                  # config 2's size: the LDS panel pipeline (chain kernel, one-pass backward)
        assert _lib.lib().gf_lsigf_pipeline(plans, 1, G, F, K) == 2
    xt = torch.tensor(x, dtype=torch.float32, device=DEV, requires_grad=True)
    y = layer(xt)
    assert tuple(y.shape) == (B, F, Nin)
    y.backward(torch.tensor(dy, dtype=torch.float32, device=DEV))
    idx = d["idx"]
    yh, dxh = y.detach().double().cpu().numpy(), xt.grad.double().cpu().numpy()
    assert relerr(yh[:, :, idx], d["y_idx"]) < FWD_RTOL and relerr((yh * yh).sum(-1), d["y_sq"]) < FWD_RTOL
    assert relerr(dxh[:, :, idx], d["dx_idx"]) < GRAD_RTOL and relerr((dxh * dxh).sum(-1), d["dx_sq"]) < GRAD_RTOL
    assert relerr(layer.weight.grad.cpu().numpy(), d["dweight"]) < GRAD_RTOL
    assert relerr(layer.bias.grad.cpu().numpy(), d["dbias"]) < GRAD_RTOL
