"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
  (1) the golden vectors produced by the real reference (tests/golden/*.npz, float64 ground truth),
  (2) the CPU oracle on seeded random inputs at sizes it finishes in seconds,
  (3) size-independent properties at the benchmark's full size (linearity, batch independence, determinism).

Stated tolerances:  forward  max|y - y_ref64| <= 1e-5 * max|y_ref64|
                                           gradients            <= 1e-4 * max|g_ref64|
"""
import ctypes
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from _util import FWD_RTOL, GOLDEN, GRAD_RTOL, case_id, golden_files, load, relerr
from oracle import evgf_oracle as evo
from oracle import lsigf_oracle as orc

from alegnn_amd import EVGF_edges, EdgePattern, LSIGF, SparseGSO, _lib, graphgen
from alegnn_amd.modules.architectures import SelectionGNN
from alegnn_amd.utils import graphML as gml

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def cu(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
    return t.requires_grad_(True) if grad else t


def stream():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------------------------------------------
# unit level: every exported kernel entry point against numpy
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,G,Nin,N", [(3, 32, 100, 100), (2, 1, 37, 37), (5, 7, 10, 45), (1, 64, 33, 70), (4, 3, 1, 1),
                                      (2, 32, 1000, 1003), (3, 8, 260, 300), (2, 64, 132, 132), (2, 32, 130, 130), (1, 12, 4, 4)])
def test_layout_kernels(B, G, Nin, N):
    L = _lib.lib()
    rng = np.random.RandomState(0)
    x = rng.randn(B, G, Nin).astype(np.float32)
    xt = cu(x)
    X = torch.full((B, N, G), float("nan"), device=DEV)
    _lib.check(L.gf_layout_bgn_to_bng(xt.data_ptr(), X.data_ptr(), B, G, Nin, N, stream()))
    want = np.zeros((B, N, G), np.float32)
    want[:, :Nin] = x.transpose(0, 2, 1)
    assert np.array_equal(X.cpu().numpy(), want)                      # pure data movement: bit-exact
    back = torch.full((B, G, Nin), float("nan"), device=DEV)
    _lib.check(L.gf_layout_bng_to_bgn(X.data_ptr(), back.data_ptr(), B, G, N, Nin, stream()))
    assert np.array_equal(back.cpu().numpy(), x)
    # tensors that are not 16-byte aligned take the 4-byte kernels: same result
    buf = torch.zeros(B * G * Nin + 1, device=DEV)
    buf[1:] = xt.reshape(-1)
    X2 = torch.full((B * N * G + 1,), float("nan"), device=DEV)
    _lib.check(L.gf_layout_bgn_to_bng(buf[1:].data_ptr(), X2[1:].data_ptr(), B, G, Nin, N, stream()))
    assert np.array_equal(X2[1:].cpu().numpy().reshape(B, N, G), want)


def _rand_graph(n, density, seed, empty_rows=True):
    rng = np.random.RandomState(seed)
    A = sp.random(n, n, density=density, random_state=rng, data_rvs=rng.randn, format="lil")
    if empty_rows and n > 4:
        A[3, :] = 0
        A[:, 2] = 0
    return sp.csr_matrix(A)


@pytest.mark.parametrize("W", [1, 3, 4, 8, 16, 32, 64, 128, 256, 20])
@pytest.mark.parametrize("B", [1, 5, 17, 40])
def test_spmm_hop_against_scipy(W, B):
    """One hop, both orientations, every kernel variant (vector widths, generic width, BT = 1/2/4 heuristics)."""
    L = _lib.lib()
    n = 203
    A = _rand_graph(n, 0.05, seed=W + B)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    rng = np.random.RandomState(1)
    X = rng.randn(B, n, W).astype(np.float32)
    Xt = cu(X)
    for op, M in ((0, A.T.tocsr()), (1, A)):
        out = torch.full((B, n, W), float("nan"), device=DEV)
        _lib.check(L.gf_spmm_hop(plans[0], op, Xt.data_ptr(), out.data_ptr(), B, W, stream()))
        want = np.stack([M.astype(np.float64) @ X[b].astype(np.float64) for b in range(B)])
        assert relerr(out.cpu().numpy(), want) < 2e-6, (op, W, B)


@pytest.mark.parametrize("uniform", [0, 1])
@pytest.mark.parametrize("W,B", [(32, 5), (32, 40), (8, 17), (64, 3), (128, 2), (4, 9)])
def test_spmm_hop_uniform_values_column_only_stream(W, B, uniform, pipeline_knob):
    """A GSO whose stored values are all equal (S = A / lambda_max of an unweighted graph): the node-major kernel reads a column-only
    entry stream and scales the row sums once; with the knob off it reads (column, value) pairs.  Both against scipy in float64;
    hub rows, empty rows and a ragged last slice included."""
    L = _lib.lib()
    n = 1003
    rng = np.random.RandomState(W + B)
    A = sp.random(n, n, density=0.01, random_state=rng, format="lil")
    A[5, :] = 1.0                      # a hub row (longer than one ring slot of the kernel)
    A[:, 9] = 1.0
    A[3, :] = 0
    A[:, 2] = 0
    A = sp.csr_matrix(A)
    A.data[:] = 0.37
    X = rng.randn(B, n, W).astype(np.float32)
    Xt = cu(X)
    pipeline_knob(panel_uniform=uniform)
    gso = SparseGSO([A])               # (owns the plans: keep it alive)
    plans = gso.plans(DEV)
    for op, M in ((0, A.T.tocsr()), (1, A)):
        out = torch.full((B, n, W), float("nan"), device=DEV)
        _lib.check(L.gf_spmm_hop(plans[0], op, Xt.data_ptr(), out.data_ptr(), B, W, stream()))
        want = np.stack([M.astype(np.float64) @ X[b].astype(np.float64) for b in range(B)])
        assert relerr(out.cpu().numpy(), want) < 2e-6, (uniform, op, W, B)


def test_spmm_hop_skewed_degrees_and_long_rows():
    """Rows far longer than the 2048-entry LDS chunk, a power-law tail, and empty rows."""
    L = _lib.lib()
    n = 6000
    rng = np.random.RandomState(7)
    rows = [np.full(5000, 0), np.full(2500, 17)]                      # two hub rows
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
    B, W = 3, 32
    X = rng.randn(B, n, W).astype(np.float32)
    Xt = cu(X)
    for op, M in ((0, A.T.tocsr()), (1, A)):
        out = torch.empty((B, n, W), device=DEV)
        _lib.check(L.gf_spmm_hop(plans[0], op, Xt.data_ptr(), out.data_ptr(), B, W, stream()))
        want = np.stack([M @ X[b].astype(np.float64) for b in range(B)])
        assert relerr(out.cpu().numpy(), want) < 5e-6


def _bank(h, transpose):
    """Hm[t, ci, o] as the contraction sees it (tap 0 summed over e)."""
    F, E, K, G = h.shape
    T = 1 + E * (K - 1)
    Hm = np.zeros((T, G, F))
    Hm[0] = h[:, :, 0, :].sum(axis=1).T
    for e in range(E):
        for k in range(1, K):
            Hm[1 + e * (K - 1) + k - 1] = h[:, e, k, :].T
    return Hm.transpose(0, 2, 1) if transpose else Hm          # [T, Cin, Cout]


CONTRACT_SHAPES = [  # B, N, Nout, G, F, E, K
    (2, 70, 70, 32, 32, 1, 5), (3, 45, 10, 32, 32, 1, 5), (2, 33, 33, 64, 32, 1, 5), (2, 40, 40, 32, 64, 1, 3),
    (2, 50, 50, 8, 5, 1, 4), (2, 37, 37, 16, 100, 2, 3), (3, 41, 41, 1, 64, 1, 5), (2, 29, 29, 3, 5, 2, 4),
    (1, 64, 64, 32, 32, 1, 1), (2, 31, 31, 128, 32, 1, 2), (2, 31, 31, 24, 7, 1, 3),
]


@pytest.mark.parametrize("shape", CONTRACT_SHAPES, ids=lambda s: "B%d_N%d_Nout%d_G%d_F%d_E%d_K%d" % s)
@pytest.mark.parametrize("transpose", [0, 1])
def test_contract_against_einsum(shape, transpose):
    L = _lib.lib()
    B, N, Nout, G, F, E, K = shape
    T = 1 + E * (K - 1)
    Cin, Cout = (F, G) if transpose else (G, F)
    rng = np.random.RandomState(3)
    Z = rng.randn(T, B, N, Cin).astype(np.float32)
    h = (rng.randn(F, E, K, G) / np.sqrt(G * K)).astype(np.float32)
    bias = None if transpose else rng.randn(F).astype(np.float32)
    out = torch.full((B, Cout, Nout), float("nan"), device=DEV)
    Zt, ht = cu(Z), cu(h)
    bt = cu(bias) if bias is not None else None
    _lib.check(L.gf_contract(Zt.data_ptr(), ht.data_ptr(), bt.data_ptr() if bt is not None else None, out.data_ptr(),
                             B, N, Nout, G, F, E, K, transpose, stream()))
    Hm = _bank(h.astype(np.float64), transpose)
    want = np.einsum("tbnc,tco->bon", Z[:, :, :Nout].astype(np.float64), Hm)
    if bias is not None:
        want = want + bias.astype(np.float64)[None, :, None]
    assert relerr(out.cpu().numpy(), want) < 3e-6


@pytest.mark.parametrize("shape", [(4, 100, 32, 32, 1, 5), (3, 77, 64, 32, 1, 5), (2, 60, 32, 64, 1, 3), (5, 33, 1, 64, 1, 5),
                                   (2, 41, 7, 5, 2, 3), (3, 500, 32, 32, 2, 5), (2, 50, 96, 40, 1, 4)],
                         ids=lambda s: "B%d_N%d_G%d_F%d_E%d_K%d" % s)
def test_grad_taps_against_einsum(shape):
    L = _lib.lib()
    B, N, G, F, E, K = shape
    T = 1 + E * (K - 1)
    rng = np.random.RandomState(5)
    Z = rng.randn(T, B, N, G).astype(np.float32)
    P0 = rng.randn(B, N, F).astype(np.float32)
    dh = torch.full((F, E, K, G), float("nan"), device=DEV)
    db = torch.full((F,), float("nan"), device=DEV)
    nbytes = L.gf_grad_taps_workspace_bytes(B, N, G, F, E, K)
    ws = torch.empty(nbytes // 4 + 1, device=DEV)
    Zt, Pt = cu(Z), cu(P0)
    _lib.check(L.gf_grad_taps(Zt.data_ptr(), Pt.data_ptr(), dh.data_ptr(), db.data_ptr(), ws.data_ptr(), nbytes,
                              B, N, G, F, E, K, stream()))
    dHm = np.einsum("tbng,bnf->tgf", Z.astype(np.float64), P0.astype(np.float64))
    want = np.zeros((F, E, K, G))
    for e in range(E):
        want[:, e, 0, :] = dHm[0].T
        for k in range(1, K):
            want[:, e, k, :] = dHm[1 + e * (K - 1) + k - 1].T
    assert relerr(dh.cpu().numpy(), want) < 5e-6
    assert relerr(db.cpu().numpy(), P0.astype(np.float64).sum(axis=(0, 1))) < 5e-6


# ---------------------------------------------------------------------------------------------------------------
# LSIGF / GraphFilter / SelectionGNN against the reference's own outputs (golden vectors)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("lsigf"), ids=case_id)
def test_lsigf_matches_reference_golden(path):
    d = load(path)
    h, x = cu(d["h"], True), cu(d["x"], True)
    b = cu(d["b"], True) if "b" in d else None
    y = LSIGF(h, torch.tensor(d["S"]), x, b)              # dense E x N x N GSO, exactly the reference's call
    y.backward(cu(d["dy"]))
    assert relerr(y.detach().cpu().numpy(), d["y"]) < FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(h.grad.cpu().numpy(), d["dh"]) < GRAD_RTOL
    if b is not None:
        assert relerr(b.grad.cpu().numpy(), d["db"]) < GRAD_RTOL


@pytest.mark.parametrize("path", golden_files("gfilter"), ids=case_id)
def test_graph_filter_zero_padding_matches_reference(path):
    d = load(path)
    F, E, K, G = d["weight"].shape
    layer = gml.GraphFilter(G, F, K, E, True)
    layer.load_state_dict({"weight": torch.tensor(d["weight"]), "bias": torch.tensor(d["bias"])})
    layer.addGSO(torch.tensor(d["S"]))
    layer.to(DEV)
    x = cu(d["x"], True)
    y = layer(x)
    assert tuple(y.shape) == d["y"].shape                  # Nin < N: output keeps Nin nodes (graphML.py:2142-2143)
    y.backward(cu(d["dy"]))
    assert relerr(y.detach().cpu().numpy(), d["y"]) < FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(layer.weight.grad.cpu().numpy(), d["dweight"]) < GRAD_RTOL
    assert relerr(layer.bias.grad.cpu().numpy(), d["dbias"]) < GRAD_RTOL


@pytest.mark.parametrize("name", ["cfg1_sbm100", "cfg3_fbego", "mid_rnd5200"])
def test_selection_gnn_matches_reference(name):
    """BASELINE configs[0] (sourceLocGNN SBM N=100, K=5, F=[1,32,32]), the config-3 architecture, and (round 6) the config-1 architecture on a
    5200-node graph with MaxPoolLocal down to 1300 / 260 nodes (one-panel chain kernel, pooling neighbourhoods on a mid-size graph), with the
    reference's weights: forward, input gradient and every parameter gradient."""
    d = load(os.path.join(GOLDEN, f"selgnn_{name}.npz"))
    cfg = d["cfg"]
    net = SelectionGNN(cfg["dimNodeSignals"], cfg["nFilterTaps"], True, torch.nn.ReLU, cfg["nSelectedNodes"],
                       getattr(gml, cfg["pool"]), cfg["poolingSize"], cfg["dimLayersMLP"], d["S"][0])
    net.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    net = net.float().to(DEV)
    x = cu(d["x"], True)
    y, ygnn = net.splitForward(x)
    (y * cu(d["w"])).sum().backward()
    assert relerr(ygnn.detach().cpu().numpy(), d["ygnn"]) < FWD_RTOL
    assert relerr(y.detach().cpu().numpy(), d["y"]) < 5 * FWD_RTOL       # + fp32 nn.Linear on top
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    for k, p in net.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k


@pytest.mark.parametrize("name", ["sbm100_L2", "fbego_L3"])
@pytest.mark.parametrize("sparse", [False, True], ids=["denseGSO", "sparseGSO"])
def test_selection_gnn_coarsening_matches_reference(name, sparse):
    """SelectionGNN(coarsening=True) + nn.MaxPool1d (architectures.py:224-247): Graclus graphs with fake nodes, one GSO
    per layer, the reference's weights.  x has the real nodes only, so the fake-node padding path (:429-434) runs."""
    d = load(os.path.join(GOLDEN, f"selgnn_coarsen_{name}.npz"))
    cfg, S = d["cfg"], d["S"][0]
    L = len(cfg["nFilterTaps"])
    np.random.seed(int(d["seed"]))
    net = SelectionGNN(cfg["dimNodeSignals"], cfg["nFilterTaps"], True, torch.nn.ReLU, [0] * L, torch.nn.MaxPool1d,
                       [2] * L, cfg["dimLayersMLP"], sp.csr_matrix(S) if sparse else S, coarsening=True)
    assert [int(v) for v in net.order] == d["perm"].tolist()
    net.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    net = net.float().to(DEV)
    x = cu(d["x"], True)
    y, ygnn = net.splitForward(x)
    (y * cu(d["w"])).sum().backward()
    assert relerr(ygnn.detach().cpu().numpy(), d["ygnn"]) < FWD_RTOL
    assert relerr(y.detach().cpu().numpy(), d["y"]) < 5 * FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    for k, p in net.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k
    # already padded input takes the plain reordering branch (:437) and gives the same output
    xp = torch.cat((x.detach(), x.new_zeros(x.shape[0], x.shape[1], net.N[0] - x.shape[2])), dim=2)
    assert torch.equal(net(xp), y.detach())


@pytest.mark.parametrize("name", ["fbego_movie", "sbm100_pool"])
def test_local_gnn_matches_reference(name):
    """LocalGNN (architectures.py:816-1170): per-node readout and singleNodeForward (the MovieLens recipe, configs[2]),
    with the reference's weights; gradients flow from the single-node outputs."""
    from alegnn_amd.modules.architectures import LocalGNN
    d = load(os.path.join(GOLDEN, f"localgnn_{name}.npz"))
    cfg = d["cfg"]
    net = LocalGNN(cfg["dimNodeSignals"], cfg["nFilterTaps"], True, torch.nn.ReLU, cfg["nSelectedNodes"],
                   getattr(gml, cfg["pool"]), cfg["poolingSize"], cfg["dimReadout"], d["S"][0])
    net.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    net = net.float().to(DEV)
    x = cu(d["x"], True)
    y, ygnn = net.splitForward(x)
    assert relerr(ygnn.detach().cpu().numpy(), d["ygnn"]) < FWD_RTOL
    assert relerr(y.detach().cpu().numpy(), d["y"]) < 5 * FWD_RTOL
    ysn = net.singleNodeForward(x, [int(n) for n in d["nodes"]])
    assert relerr(ysn.detach().cpu().numpy(), d["ysn"]) < 5 * FWD_RTOL
    assert torch.equal(net.singleNodeForward(x, d["nodes"]), ysn)                     # ndarray form (:1150-1154)
    same = net.singleNodeForward(x, int(d["nodes"][0]))                                # int form: one node for the batch
    assert torch.equal(same[0], ysn[0])
    (ysn * cu(d["w"])).sum().backward()
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    for k, p in net.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k


def test_trainer_on_gpu_follows_reference_training_run(tmp_path):
    """Model + Trainer + evaluate driving the HIP SelectionGNN for the reference's 3-epoch run on SourceLocalization data
    (tests/golden/make_golden.py trainer_case 'selgnn'; reference on CPU in float64, here float32 on the GPU)."""
    import ast
    from _util import ArrayData
    from alegnn_amd.modules import evaluation, model, training
    d = load(os.path.join(GOLDEN, "trainer_selgnn.npz"))
    net = SelectionGNN([1, 8, 8], [3, 3], True, torch.nn.ReLU, [20, 10], gml.MaxPoolLocal, [2, 2], [5], d["S"][0])
    net.load_state_dict({k[5:]: torch.tensor(v) for k, v in d.items() if k.startswith("init:")})
    net = net.float()
    optim = torch.optim.Adam(net.parameters(), lr=0.005, betas=(0.9, 0.999))
    m = model.Model(net, torch.nn.CrossEntropyLoss(), optim, training.Trainer,
                    evaluation.evaluate, DEV, "selgnn", str(tmp_path))
    data = ArrayData(d, torch.float32)
    np.random.seed(int(d["seed"]) + 1)
    tv = m.train(data, int(d["nEpochs"]), int(d["batchSize"]), printInterval=0, **ast.literal_eval(str(d["trainKw"])))
    assert tv["lossTrain"].shape == d["lossTrain"].shape and tv["lossValid"].shape == d["lossValid"].shape
    assert np.allclose(tv["lossTrain"], d["lossTrain"], rtol=2e-4), (tv["lossTrain"], d["lossTrain"])
    assert np.allclose(tv["lossValid"], d["lossValid"], rtol=2e-4)
    assert np.max(np.abs(tv["costValid"] - d["costValid"])) <= 1.0 / 32 + 1e-6      # at most one borderline sample flips
    ev = m.evaluate(data)
    assert abs(ev["costBest"] - float(d["costBest"])) <= 1.0 / 32 + 1e-6
    assert abs(ev["costLast"] - float(d["costLast"])) <= 1.0 / 32 + 1e-6
    ref = torch.load(os.path.join(GOLDEN, "ckpt", "selgnnArchitLast.ckpt"))
    last = torch.load(tmp_path / "savedModels" / "selgnnArchitLast.ckpt")
    for k in ref:                                                    # 9 Adam steps in fp32 vs fp64
        assert relerr(last[k].cpu().numpy(), ref[k].numpy()) < 2e-3, k


def test_trainer_hip_graph_replay_is_bit_identical_to_eager(tmp_path):
    """Trainer(hipGraph=True): the captured step (zero-grad + forward + loss + backward, one graph per batch size: 32 and the
    uneven last batch) must retrace the eager run exactly -- same kernels, same order."""
    import ast
    from _util import ArrayData
    from alegnn_amd.modules import evaluation, model, training
    d = load(os.path.join(GOLDEN, "trainer_selgnn.npz"))

    def run(tag, **kw):
        net = SelectionGNN([1, 8, 8], [3, 3], True, torch.nn.ReLU, [20, 10], gml.MaxPoolLocal, [2, 2], [5], d["S"][0])
        net.load_state_dict({k[5:]: torch.tensor(v) for k, v in d.items() if k.startswith("init:")})
        net = net.float()
        optim = torch.optim.Adam(net.parameters(), lr=0.005, betas=(0.9, 0.999))
        m = model.Model(net, torch.nn.CrossEntropyLoss(), optim, training.Trainer, evaluation.evaluate,
                        DEV, tag, str(tmp_path))
        np.random.seed(5)
        tv = m.train(ArrayData(d, torch.float32), 3, 40, printInterval=0, validationInterval=2, doSaveVars=False, **kw)   # 40 + 40 + 16
        return tv, {k: v.detach().clone() for k, v in net.state_dict().items()}, m

    tv_e, sd_e, _ = run("eager")
    tv_g, sd_g, m = run("graph", hipGraph=True)
    assert len(m.trainer._graphs) == 2                               # batch sizes 40 and 16
    assert np.array_equal(tv_e["lossTrain"], tv_g["lossTrain"]) and np.array_equal(tv_e["costValid"], tv_g["costValid"])
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k


# ---- callers that reuse the K-hop kernel ---------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("nvgf"), ids=case_id)
def test_node_variant_gf_matches_reference(path):
    """NodeVariantGF / NVGF (graphML.py:293-387, 2317-2509): reference weights, copyNodes expansion, Nin < N, all gradients."""
    d = load(path)
    F, E, K, G, M = d["weight"].shape
    layer = gml.NodeVariantGF(G, F, K, M, E, True)
    layer.load_state_dict({"weight": torch.tensor(d["weight"]), "bias": torch.tensor(d["bias"])})
    layer.addGSO(torch.tensor(d["S"]))
    layer = layer.float().to(DEV)
    x = cu(d["x"], True)
    y = layer(x)
    assert tuple(y.shape) == d["y"].shape
    y.backward(cu(d["dy"]))
    assert relerr(y.detach().cpu().numpy(), d["y"]) < FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(layer.weight.grad.cpu().numpy(), d["dweight"]) < GRAD_RTOL
    assert relerr(layer.bias.grad.cpu().numpy(), d["dbias"]) < GRAD_RTOL
    if "f_h" in d:                                                  # functional form with a per-node bias [F,N]
        from alegnn_amd.functional import NVGF
        h, b, x2 = cu(d["f_h"], True), cu(d["f_b"], True), cu(d["f_x"], True)
        y2 = NVGF(h, torch.tensor(d["S"]), x2, b)
        y2.backward(cu(d["f_dy"]))
        assert relerr(y2.detach().cpu().numpy(), d["f_y"]) < FWD_RTOL
        for got, want in ((h.grad, "f_dh"), (b.grad, "f_db"), (x2.grad, "f_dx")):
            assert relerr(got.cpu().numpy(), d[want]) < GRAD_RTOL, want


def test_node_variant_random_sparse_vs_oracle():
    """N = 3000 (no dense GSO anywhere), E = 2, widths that are not multiples of anything; forward against the numpy oracle,
    backward against central differences of the oracle along random directions (the filter is linear in x and in h)."""
    from alegnn_amd.functional import NVGF
    from oracle import nvgf_oracle as nvo
    N, B, G, F, K, E = 3000, 7, 5, 9, 3, 2
    mats = [graphgen.sbm(N, seed=21 + e, directed=True) for e in range(E)]
    rng = np.random.RandomState(5)
    h = (rng.uniform(-1, 1, (F, E, K, G, N)) / np.sqrt(G * K)).astype(np.float32)
    x = rng.randn(B, G, N).astype(np.float32)
    b = rng.uniform(-1, 1, (F, 1)).astype(np.float32)
    dy = rng.randn(B, F, N).astype(np.float32)
    ht, xt, bt = cu(h, True), cu(x, True), cu(b, True)
    y = NVGF(ht, SparseGSO(mats), xt, bt)
    y.backward(cu(dy))
    want = nvo.nvgf_sparse(h.astype(np.float64), mats, x.astype(np.float64), b.astype(np.float64))
    assert relerr(y.detach().cpu().numpy(), want) < FWD_RTOL
    # <dy, J_x u> = <dx, u> and <dy, J_h v> = <dh, v> (exact for a bilinear map): check the adjoints with random u, v
    u, v = rng.randn(*x.shape), rng.randn(*h.shape)
    Ju = nvo.nvgf_sparse(h.astype(np.float64), mats, u, None)
    Jv = nvo.nvgf_sparse(v, mats, x.astype(np.float64), None)
    assert abs(np.sum(dy * Ju) - np.sum(xt.grad.cpu().numpy().astype(np.float64) * u)) < GRAD_RTOL * np.abs(dy * Ju).sum()
    assert abs(np.sum(dy * Jv) - np.sum(ht.grad.cpu().numpy().astype(np.float64) * v)) < GRAD_RTOL * np.abs(dy * Jv).sum()
    assert relerr(bt.grad.cpu().numpy(), dy.astype(np.float64).sum(axis=(0, 2))[:, None]) < GRAD_RTOL


def test_node_variant_gnn_matches_reference():
    from alegnn_amd.modules.architectures import NodeVariantGNN
    d = load(os.path.join(GOLDEN, "nvgnn_sbm100.npz"))
    net = NodeVariantGNN([2, 8, 8], [3, 2], [10, 5], True, torch.nn.ReLU, [40, 10], gml.MaxPoolLocal, [2, 2], [4], d["S"][0])
    net.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    net = net.float().to(DEV)
    x = cu(d["x"], True)
    y, ygnn = net.splitForward(x)
    (y * cu(d["w"])).sum().backward()
    assert relerr(ygnn.detach().cpu().numpy(), d["ygnn"]) < FWD_RTOL
    assert relerr(y.detach().cpu().numpy(), d["y"]) < 5 * FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    for k, p in net.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k


@pytest.mark.parametrize("path", golden_files("grnn"), ids=case_id)
def test_gated_grnn_matches_reference(path):
    """GatedGRNN / HiddenState (graphML.py:1292-1527, 3540-3681): one LSIGF over all B*T inputs + one per time step, with
    no / time / node gating; every gradient including the one reaching z0 through the whole recursion."""
    d = load(path)
    H, E, K, F = d["sd:aWeights"].shape
    layer = gml.HiddenState(F, H, K, nonlinearity=torch.tanh, E=E, bias=True)
    layer.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    layer.addGSO(torch.tensor(d["S"]))
    layer = layer.float().to(DEV)
    x, z0 = cu(d["x"], True), cu(d["z0"], True)
    kw = {}
    if str(d["gating"]) != "none":
        kw = dict(q_hat=cu(d["q_hat"]), q_check=cu(d["q_check"]))
    z = gml.GatedGRNN(layer.aWeights, layer.bWeights, layer._gso, x, z0, torch.tanh, xBias=layer.xBias, zBias=layer.zBias, **kw)
    (z * cu(d["dz"])).sum().backward()
    assert relerr(z.detach().cpu().numpy(), d["z"]) < 2 * FWD_RTOL          # T chained filters + tanh
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(z0.grad.cpu().numpy(), d["dz0"]) < GRAD_RTOL
    for k, p in layer.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k
    if str(d["gating"]) == "none":
        zz, zT = layer(x.detach(), z0.detach())
        assert torch.equal(zz, z.detach()) and list(zT.shape) == d["zT_shape"].tolist()


@pytest.mark.parametrize("path", golden_files("gatedhs"), ids=case_id)
def test_gated_hidden_state_modules_match_reference(path):
    """TimeGatedHiddenState / NodeGatedHiddenState (graphML.py:3683-4031): two ungated recurrent layers feed the gates (a Linear(H*N, 1)
    per time step, or a GraphFilter(H, 1, K) per node), the gated recursion produces the states; the reference's own state_dict loads."""
    d = load(path)
    F, H, K, E = (int(v) for v in d["dims"])
    cls = gml.TimeGatedHiddenState if str(d["kind"]) == "time" else gml.NodeGatedHiddenState
    layer = cls(F, H, K, nonlinearity=torch.tanh, E=E, bias=True)
    layer.addGSO(torch.tensor(d["S"]))
    layer.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    layer = layer.float().to(DEV)
    x, z0 = cu(d["x"], True), cu(d["z0"], True)
    z, zT = layer(x, z0)
    assert list(zT.shape) == d["zT_shape"].tolist()
    (z * cu(d["dz"])).sum().backward()
    assert relerr(z.detach().cpu().numpy(), d["z"]) < 3 * FWD_RTOL          # three chained recurrent layers + sigmoid gates
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(z0.grad.cpu().numpy(), d["dz0"]) < GRAD_RTOL
    for k, p in layer.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k


@pytest.mark.parametrize("path", golden_files("jarma"), ids=case_id)
def test_jarma_matches_reference(path):
    """jARMA (graphML.py:490-638): Jacobi iterations as sparse hops of the [B,F,E,P,G] chain states (the reference multiplies dense
    [F,E,P,G,N,N] tensors) + the LSIGF residue; output and the gradients of all three tap sets, the input and the bias."""
    d = load(path)
    t = {k: cu(d[k], True) for k in ("psi", "varphi", "phi", "x")}
    b = cu(d["b"], True) if "b" in d else None
    y = gml.jARMA(t["psi"], t["varphi"], t["phi"], torch.tensor(d["S"]), t["x"], b, tMax=int(d["tMax"]))
    y.backward(cu(d["dy"]))
    assert relerr(y.detach().cpu().numpy(), d["y"]) < 2 * FWD_RTOL
    for k, v in t.items():
        assert relerr(v.grad.cpu().numpy(), d["d" + k]) < GRAD_RTOL, k
    if b is not None:
        assert relerr(b.grad.cpu().numpy(), d["db"]) < GRAD_RTOL


def test_edge_variant_gnn_matches_reference():
    """archit.EdgeVariantGNN (architectures.py:1721-1955, BASELINE configs[4] as an architecture): two hybrid EdgeVariantGF layers
    (dense reference parameters, gathered on the pattern), ReLU, MaxPoolLocal, MLP -- the reference's state_dict loads, outputs and every
    gradient agree (off-pattern entries of weightEV get exactly zero gradient on both sides)."""
    from alegnn_amd.modules.architectures import EdgeVariantGNN
    d = load(os.path.join(GOLDEN, "evgnn_asym37.npz"))
    net = EdgeVariantGNN([2, 4, 4], [3, 2], [20, 10], True, torch.nn.ReLU, [20, 10], gml.MaxPoolLocal, [2, 2], [3], d["S"][0])
    sd = {k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")}
    assert set(sd) == set(net.state_dict())
    net.load_state_dict(sd)
    net = net.float().to(DEV)
    x = cu(d["x"], True)
    y, ygnn = net.splitForward(x)
    (y * cu(d["w"])).sum().backward()
    assert relerr(ygnn.detach().cpu().numpy(), d["ygnn"]) < 2 * FWD_RTOL
    assert relerr(y.detach().cpu().numpy(), d["y"]) < 5 * FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    for k, p in net.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k


def test_graph_recurrent_nn_matches_reference():
    from alegnn_amd.modules.architectures import GraphRecurrentNN
    d = load(os.path.join(GOLDEN, "grnnarch_sbm100.npz"))
    net = GraphRecurrentNN(3, 6, 8, [3, 2], True, torch.tanh, torch.tanh, torch.nn.ReLU, [5, 2], d["S"][0])
    net.load_state_dict({k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")})
    net = net.float().to(DEV)
    x = cu(d["x"], True)
    y, yOut = net.splitForward(x, cu(d["z0"]))
    (y * cu(d["w"])).sum().backward()
    assert relerr(yOut.detach().cpu().numpy(), d["yOut"]) < 2 * FWD_RTOL
    assert relerr(y.detach().cpu().numpy(), d["y"]) < 5 * FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    for k, p in net.named_parameters():
        assert relerr(p.grad.cpu().numpy(), d["grad:" + k]) < GRAD_RTOL, k
    ysn = net.singleNodeForward(x.detach(), [1, 5, 9], cu(d["z0"]))
    assert torch.equal(ysn, y.detach()[torch.arange(3), :, :, torch.tensor([1, 5, 9])])
    assert tuple(net(x.detach()).shape) == tuple(y.shape)            # z0 drawn on the device when not given (:4556)


# ---------------------------------------------------------------------------------------------------------------
# seeded random inputs vs the CPU oracle (sparse restatement), sizes the oracle finishes in seconds
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(N=3000, B=20, G=32, F=32, K=5, directed=True, model="sbm"),
    dict(N=5000, B=8, G=64, F=32, K=5, directed=False, model="er"),
    dict(N=1682, B=5, G=1, F=64, K=5, directed=False, model="sbm"),      # config-3 first layer shape
    dict(N=2000, B=33, G=32, F=32, K=3, directed=True, model="er", E=2),
    # one-pass backward over 32-wide blocks of the input features (G = 64, 128): panel pipeline with the two-ahead operand ring (T = 5)
    # and with the one-ahead schedule (T = 3), several edge features, and the node-major pipeline (N beyond the LDS panel limit)
    dict(N=700, B=3, G=128, F=16, K=3, directed=True, model="sbm"),
    dict(N=900, B=4, G=64, F=8, K=3, directed=True, model="er", E=2),
    dict(N=12000, B=2, G=64, F=32, K=5, directed=False, model="er"),
    dict(N=11000, B=3, G=128, F=32, K=2, directed=True, model="er"),
    # one-pass backward with 64 output features (round 4: two f blocks of tap gradients per wave; G <= 32): panels and node-major,
    # T = 5 with two edge features, T = 6, T = 2
    dict(N=900, B=4, G=16, F=64, K=3, directed=True, model="er", E=2),
    dict(N=700, B=3, G=8, F=64, K=6, directed=False, model="sbm"),
    dict(N=12000, B=2, G=32, F=64, K=5, directed=False, model="er"),
    dict(N=10500, B=3, G=5, F=64, K=2, directed=True, model="er"),
    dict(N=300, B=7, G=32, F=60, K=4, directed=True, model="sbm"),       # F = 60 is padded to 64
], ids=lambda c: "_".join(f"{k}{v}" for k, v in c.items()))
def test_lsigf_random_sparse_vs_oracle(cfg):
    E = cfg.get("E", 1)
    gen = graphgen.sbm if cfg["model"] == "sbm" else graphgen.er
    mats = [gen(cfg["N"], seed=11 + e, directed=cfg["directed"]) for e in range(E)]
    rng = np.random.RandomState(2)
    B, G, F, K, N = cfg["B"], cfg["G"], cfg["F"], cfg["K"], cfg["N"]
    h = rng.uniform(-1, 1, (F, E, K, G)) / np.sqrt(G * K)
    x = rng.randn(B, G, N)
    b = rng.uniform(-1, 1, (F, 1))
    dy = rng.randn(B, F, N)
    ht, xt, bt = cu(h, True), cu(x, True), cu(b, True)
    y = LSIGF(ht, SparseGSO(mats), xt, bt)
    y.backward(cu(dy))
    h32, x32, b32 = h.astype(np.float32), x.astype(np.float32), b.astype(np.float32)      # same rounded inputs
    want = orc.lsigf_sparse(h32, mats, x32, b32)
    dx, dh, db = orc.lsigf_sparse_grads(h32, mats, x32, b32, dy.astype(np.float32))
    assert relerr(y.detach().cpu().numpy(), want) < FWD_RTOL
    assert relerr(xt.grad.cpu().numpy(), dx) < GRAD_RTOL
    assert relerr(ht.grad.cpu().numpy(), dh) < GRAD_RTOL
    assert relerr(bt.grad.cpu().numpy(), db) < GRAD_RTOL


# ---------------------------------------------------------------------------------------------------------------
# full benchmark size (configs[1]: SBM N=10k, nnz~100k, B=256, K=5, 32->32): properties + oracle on a batch slice
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg2():
    N, B, G, F, K = 10000, 256, 32, 32, 5
    A = graphgen.sbm(N, seed=0)
    torch.manual_seed(0)
    layer = gml.GraphFilter(G, F, K, 1, True)
    layer.addGSO(A)
    layer.to(DEV)
    x = torch.randn(B, G, N, device=DEV)
    return dict(A=A, layer=layer, x=x, dims=(N, B, G, F, K))


def test_full_size_batch_slice_vs_oracle(cfg2):
    layer, x, A = cfg2["layer"], cfg2["x"], cfg2["A"]
    xg = x.clone().requires_grad_(True)
    y = layer(xg)
    dy = torch.randn_like(y)
    y.backward(dy)
    sl = [0, 101, 255]                                       # batch entries are independent: check three of them
    w, b = layer.weight.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    want = orc.lsigf_sparse(w, A, x[sl].cpu().numpy(), b)
    assert relerr(y[sl].detach().cpu().numpy(), want) < FWD_RTOL
    dx, _, _ = orc.lsigf_sparse_grads(w, A, x[sl].cpu().numpy(), b, dy[sl].cpu().numpy())
    assert relerr(xg.grad[sl].cpu().numpy(), dx) < GRAD_RTOL
    # tap / bias gradients: oracle on a 16-entry slice of the batch, HIP path re-run on the same slice
    sl16 = list(range(0, 256, 16))
    layer.zero_grad()
    y16 = layer(x[sl16])
    y16.backward(dy[sl16])
    _, dh, db = orc.lsigf_sparse_grads(w, A, x[sl16].cpu().numpy(), b, dy[sl16].cpu().numpy())
    assert relerr(layer.weight.grad.cpu().numpy(), dh) < GRAD_RTOL
    assert relerr(layer.bias.grad.cpu().numpy(), db) < GRAD_RTOL
    layer.zero_grad()


def test_full_size_properties(cfg2):
    layer, x = cfg2["layer"], cfg2["x"]
    with torch.no_grad():
        y1 = layer(x)
        y2 = layer(x)
        assert torch.equal(y1, y2)                           # bitwise run-to-run determinism (no float atomics)
        bias = layer.bias.detach().reshape(1, -1, 1)
        ya = layer(2.0 * x)                                  # scaling by a power of two is exact in fp32
        assert torch.equal(ya - bias, 2.0 * (y1 - bias)) or relerr((ya - bias).cpu().numpy(), (2.0 * (y1 - bias)).cpu().numpy()) < 1e-6
        perm = torch.randperm(x.shape[0], device=DEV)
        assert torch.equal(layer(x[perm]), y1[perm])         # batch entries never mix
        x2 = torch.randn_like(x)
        lin = layer(x + x2) - bias
        assert relerr(lin.cpu().numpy(), ((y1 - bias) + (layer(x2) - bias)).cpu().numpy()) < 1e-5   # additivity


def test_config4_size_node_major_vs_oracle():
    """BASELINE configs[3] graph (ER N = 1e5, nnz ~ 1e6: beyond the LDS panel limit, node-major pipeline through L2) at a small
    batch: forward and all gradients against the sparse oracle, and run-to-run determinism."""
    N, B, G, F, K = 100_000, 6, 32, 32, 5
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
    want = orc.lsigf_sparse(w, A, x.detach().cpu().numpy(), b)
    dx, dh, db = orc.lsigf_sparse_grads(w, A, x.detach().cpu().numpy(), b, dy.cpu().numpy())
    assert relerr(y.detach().cpu().numpy(), want) < FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), dx) < GRAD_RTOL
    assert relerr(layer.weight.grad.cpu().numpy(), dh) < GRAD_RTOL
    assert relerr(layer.bias.grad.cpu().numpy(), db) < GRAD_RTOL
    with torch.no_grad():
        assert torch.equal(layer(x), y)


def test_gradients_are_deterministic(cfg2):
    layer, x = cfg2["layer"], cfg2["x"][:64]
    grads = []
    for _ in range(2):
        layer.zero_grad()
        xg = x.clone().requires_grad_(True)
        layer(xg).square().sum().backward()
        grads.append((xg.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone()))
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    layer.zero_grad()


def test_batch_beyond_the_grid_z_limit(pipeline_knob):
    """Recurrent layers fold B*T into the batch: more than 65535 batch entries through the node-major layout kernels (they put the
    batch on gridDim.z and go in slices), forward and gradients against the oracle."""
    N, B, G, F, K = 12, 70_000, 4, 4, 3
    rng = np.random.RandomState(0)
    A = sp.random(N, N, density=0.3, format="csr", random_state=rng, data_rvs=rng.randn) * 0.3
    pipeline_knob(pipeline=1)
    h = (rng.uniform(-1, 1, (F, 1, K, G)) / np.sqrt(G * K)).astype(np.float32)
    x = rng.randn(B, G, N).astype(np.float32)
    dy = rng.randn(B, F, N).astype(np.float32)
    ht, xt = cu(h, True), cu(x, True)
    y = LSIGF(ht, SparseGSO([A]), xt, None)
    y.backward(cu(dy))
    sl = [0, 65534, 65535, 65536, B - 1]
    assert relerr(y.detach()[sl].cpu().numpy(), orc.lsigf_sparse(h, [A], x[sl], None)) < FWD_RTOL
    dx, _, _ = orc.lsigf_sparse_grads(h, [A], x[sl], None, dy[sl])
    assert relerr(xt.grad[sl].cpu().numpy(), dx) < GRAD_RTOL
    _, dh, _ = orc.lsigf_sparse_grads(h, [A], x, None, dy)
    assert relerr(ht.grad.cpu().numpy(), dh) < GRAD_RTOL


def test_error_conventions_on_device():
    layer = gml.GraphFilter(4, 8, 3).to(DEV)
    layer.addGSO(torch.eye(6).reshape(1, 6, 6))
    with pytest.raises(AssertionError):
        layer(torch.zeros(2, 5, 6, device=DEV))              # wrong feature count (graphML.py:139)
    with pytest.raises(AssertionError):
        layer(torch.zeros(2, 4, 7, device=DEV))              # more nodes than the GSO (graphML.py:140)
    with pytest.raises(TypeError):
        layer(torch.zeros(2, 4, 6, device=DEV, dtype=torch.float64))
    y = layer(torch.zeros(2, 4, 6, device=DEV))
    assert tuple(y.shape) == (2, 8, 6)
    assert torch.allclose(y, layer.bias.detach().reshape(1, 8, 1).expand(2, 8, 6))


# ---------------------------------------------------------------------------------------------------------------
# EVGF / EdgeVariantGF
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files("evgf"), ids=case_id)
def test_edge_variant_gf_matches_reference_golden(path):
    """The reference's own EdgeVariantGF outputs and autograd gradients (full EV, hybrid with the bias counted twice,
    E = 2, Nin < N, K = 1, no bias), with the reference's dense parameter layout."""
    d = load(path)
    F, E, K, G, N, _ = d["weightEV"].shape
    M = int(d["M"])
    has_bias = "bias" in d
    layer = gml.EdgeVariantGF(G, F, K, M, N, E, has_bias)
    sd = {"weightEV": torch.tensor(d["weightEV"])}
    if "weightLSI" in d:
        sd["weightLSI"] = torch.tensor(d["weightLSI"])
    if has_bias:
        sd["bias"] = torch.tensor(d["bias"])
    layer.load_state_dict(sd)
    layer.addGSO(torch.tensor(d["S"]))
    layer = layer.float().to(DEV)
    x = cu(d["x"], True)
    y = layer(x)
    assert tuple(y.shape) == d["y"].shape
    y.backward(cu(d["dy"]))
    assert relerr(y.detach().cpu().numpy(), d["y"]) < FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(layer.weightEV.grad.cpu().numpy(), d["dweightEV"]) < GRAD_RTOL
    if "weightLSI" in d:
        assert relerr(layer.weightLSI.grad.cpu().numpy(), d["dweightLSI"]) < GRAD_RTOL
    if has_bias:
        assert relerr(layer.bias.grad.cpu().numpy(), d["dbias"]) < GRAD_RTOL


@pytest.mark.parametrize("cfg", [
    dict(N=700, B=16, G=8, F=8, K=3, M=700, directed=True),
    dict(N=500, B=5, G=4, F=12, K=4, M=200, directed=True),         # odd batch, hybrid pattern
    dict(N=300, B=70, G=3, F=5, K=2, M=300, directed=False),        # batch > one wavefront
    dict(N=2000, B=16, G=32, F=32, K=3, M=2000, directed=False),    # config-5 widths
    dict(N=400, B=12, G=4, F=6, K=3, M=150, directed=True),         # 16-byte gathers, 3 of 4 lanes per row active, hybrid
    dict(N=300, B=64, G=4, F=4, K=4, M=300, directed=True),         # 16 lanes per row
    dict(N=200, B=132, G=2, F=3, K=2, M=200, directed=False),       # 33 quads per row: 64-lane groups, partly idle
    dict(N=600, B=16, G=4, F=4, K=3, M=600, directed=False, hub=True),  # a node adjacent to all: its row block exceeds the LDS stage
], ids=lambda c: "_".join(f"{k}{v}" for k, v in c.items()))
def test_evgf_per_edge_storage_vs_oracle(cfg):
    N, B, G, F, K, M = (cfg[k] for k in "NBGFKM")
    A = graphgen.sbm(N, seed=5, directed=cfg["directed"])
    if cfg.get("hub"):
        A = A.tolil()
        A[7, :] = 1.0 / N
        A[:, 7] = 1.0 / N
        A = A.tocsr()
    pat = EdgePattern.from_gso(A, M)
    P = evo.ev_pattern(A, M)
    assert np.array_equal(pat.indices, P.indices)
    rng = np.random.RandomState(3)
    scale = 1.0 / np.sqrt(G * K)
    wdiag = (rng.uniform(-1, 1, (F, G, N)) * (np.arange(N) < M)).astype(np.float32)
    wedge = (rng.uniform(-1, 1, (F, K - 1, G, pat.nnzp)) * 0.3).astype(np.float32)
    x = rng.randn(B, G, N).astype(np.float32)
    b = rng.uniform(-1, 1, (F, 1)).astype(np.float32)
    dy = (rng.randn(B, F, N) * scale).astype(np.float32)
    wd, we, xt, bt = cu(wdiag, True), cu(wedge, True), cu(x, True), cu(b, True)
    y = EVGF_edges(pat, wd, we, xt, bt)
    y.backward(cu(dy))
    want = evo.evgf_sparse(P, wdiag, wedge, x, b)
    dx, dwd, dwe, db = evo.evgf_sparse_grads(P, wdiag, wedge, x, dy)
    assert relerr(y.detach().cpu().numpy(), want) < FWD_RTOL
    assert relerr(xt.grad.cpu().numpy(), dx) < GRAD_RTOL
    assert relerr(wd.grad.cpu().numpy(), dwd) < GRAD_RTOL
    assert relerr(we.grad.cpu().numpy(), dwe) < GRAD_RTOL
    assert relerr(bt.grad.cpu().numpy(), db) < GRAD_RTOL
    # bitwise run-to-run determinism (no float atomics anywhere)
    wd2, we2, xt2, bt2 = cu(wdiag, True), cu(wedge, True), cu(x, True), cu(b, True)
    y2 = EVGF_edges(pat, wd2, we2, xt2, bt2)
    y2.backward(cu(dy))
    assert torch.equal(y, y2) and torch.equal(we.grad, we2.grad) and torch.equal(xt.grad, xt2.grad)


@pytest.mark.parametrize("B,G,K", [(16, 32, 3), (12, 5, 2), (64, 3, 4), (132, 2, 3)])
def test_evgf_tap_kernels_are_bit_identical(B, G, K, pipeline_knob):
    """The tap kernels add a row's entries in the same order: the 16-byte-gather kernel and the 4-byte-gather kernel (entries staged in
    LDS) give bit-identical outputs and gradients; the one-thread-per-output kernel gives the same forward (in the adjoint taps it
    starts its sum from dy_f instead of adding it last)."""
    N, F = 900, 4
    A = graphgen.sbm(N, seed=2, directed=True)
    pat = EdgePattern.from_gso(A, N)
    rng = np.random.RandomState(B)
    wdiag = rng.uniform(-1, 1, (F, G, N)).astype(np.float32)
    wedge = (rng.uniform(-1, 1, (F, K - 1, G, pat.nnzp)) * 0.3).astype(np.float32)
    x, dy = rng.randn(B, G, N).astype(np.float32), rng.randn(B, F, N).astype(np.float32)
    res = []
    for generic in (0, 2, 1):
        pipeline_knob(evgf_generic=generic)
        wd, we, xt = cu(wdiag, True), cu(wedge, True), cu(x, True)
        y = EVGF_edges(pat, wd, we, xt, None)
        y.backward(cu(dy))
        res.append((y.detach(), xt.grad, wd.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][0], res[2][0])
    if B <= 64:      # (the 4-byte-gather kernel covers B <= 64; above, setting 2 also runs the one-thread-per-output kernel)
        assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert relerr(res[2][1].cpu().numpy(), res[0][1].cpu().numpy()) < GRAD_RTOL


def test_edge_variant_gf_sparse_parameters_config5_shape():
    """sparse=True parameter storage at config-5 widths (N scaled down so the oracle finishes): same function of the
    same numbers as the dense-parameter module, linear in x, zero-padding honoured."""
    N, B, G, F, K, M = 1500, 8, 32, 32, 3, 1500
    A = graphgen.sbm(N, seed=9)
    layer = gml.EdgeVariantGF(G, F, K, M, N, 1, True, sparse=True)
    layer.addGSO(A)
    layer.to(DEV)
    with torch.no_grad():
        layer.weightEVdiag.mul_(np.sqrt(N))
        layer.weightEVedges[0].mul_(np.sqrt(N) * 0.5)
    torch.manual_seed(0)
    x = torch.randn(B, G, N, device=DEV)
    y = layer(x)
    P = evo.ev_pattern(A, M)
    want = evo.evgf_sparse(P, layer.weightEVdiag[:, 0].detach().cpu().numpy(), layer.weightEVedges[0].detach().cpu().numpy(),
                           x.cpu().numpy(), layer.bias.detach().cpu().numpy())
    assert relerr(y.detach().cpu().numpy(), want) < FWD_RTOL
    y2 = layer(2.0 * x)
    lin = (y2 - layer.bias) - 2.0 * (y - layer.bias)
    assert float(lin.abs().max()) <= 1e-5 * float(y.abs().max())
    ypad = layer(x[:, :, :1000])
    xz = x.clone()
    xz[:, :, 1000:] = 0
    assert torch.equal(ypad, layer(xz)[:, :, :1000])


# ---------------------------------------------------------------------------------------------------------------
# column-panel pipeline (gathers served from LDS): every stage against numpy, then the whole layer under both pipelines
# ---------------------------------------------------------------------------------------------------------------
def tune(**kw):
    L = _lib.lib()
    for k, v in kw.items():
        _lib.check(L.gf_tune(k.encode(), int(v)), "gf_tune " + k)


@pytest.fixture
def pipeline_knob():
    yield tune
    tune(pipeline=0, panel_uniform=1, panel_order=1, panel_sort=1, panel_chain=1, panel_np=0, evgf_generic=0, panel_db=1, panel_thr=0,
         panel_loaders=0)


def to_panels(x, N):
    """numpy mirror of gf_pack_panels: x [B,C,Nin] -> [B*C/4, N, 4]"""
    B, C, Nin = x.shape
    xp = np.zeros((B, C, N), dtype=x.dtype)
    xp[:, :, :Nin] = x
    return np.ascontiguousarray(xp.reshape(B, C // 4, 4, N).transpose(0, 1, 3, 2).reshape(B * C // 4, N, 4))


@pytest.mark.parametrize("B,C,Nin,N", [(3, 32, 100, 100), (2, 8, 37, 64), (5, 4, 10, 45), (1, 64, 33, 70)])
def test_pack_unpack_panels(B, C, Nin, N):
    L = _lib.lib()
    rng = np.random.RandomState(0)
    x = rng.randn(B, C, Nin).astype(np.float32)
    xt = cu(x)
    Xp = torch.full((B * C // 4, N, 4), float("nan"), device=DEV)
    _lib.check(L.gf_pack_panels(xt.data_ptr(), Xp.data_ptr(), B, C, Nin, N, stream()))
    assert np.array_equal(Xp.cpu().numpy(), to_panels(x, N))
    back = torch.empty((B, C, Nin), device=DEV)
    _lib.check(L.gf_unpack_panels(Xp.data_ptr(), back.data_ptr(), B, C, N, Nin, stream()))
    assert np.array_equal(back.cpu().numpy(), x)


@pytest.mark.parametrize("N,P,kind", [
    (100, 3, "weighted"), (1000, 40, "weighted"), (1000, 40, "uniform"), (2561, 64, "weighted"), (5000, 300, "uniform"),
    (5121, 700, "weighted"), (10000, 520, "uniform"), (10239, 300, "weighted"), (64, 2, "weighted"), (63, 5, "uniform"),
    (1000, 601, "uniform"), (2000, 1030, "weighted"), (300, 515, "weighted"),      # two panels per pass, odd panel counts
], ids=lambda v: str(v))
def test_spmm_hop_panel_against_scipy(N, P, kind, pipeline_knob):
    """One LDS-panel hop, both operators, against scipy: workgroup sizes 256 / 512 / 1024, more panels than workgroups,
    empty rows, rows longer than one 8-step round, a hub row, value-free (uniform) and weighted streams."""
    L = _lib.lib()
    rng = np.random.RandomState(N + P)
    A = sp.random(N, N, density=min(0.5, 9.0 / N), format="lil", random_state=rng, data_rvs=rng.randn)
    A[N // 2, :] = 0                                        # empty row
    A[:, N // 3] = 0                                        # empty column
    hub = rng.choice(N, size=min(N, 300), replace=False)    # one long row and one long column
    A[1, hub] = rng.randn(len(hub))
    A[hub, 2] = rng.randn(len(hub))
    A = sp.csr_matrix(A)
    if kind == "uniform":
        A.data[:] = 0.37
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    ns, uni, cyc, fill = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
    _lib.check(L.gf_plan_panel_info(plans[0], 0, ctypes.byref(ns), ctypes.byref(uni), ctypes.byref(cyc), ctypes.byref(fill)))
    assert ns.value == (N + 63) // 64 and uni.value == (1 if kind == "uniform" else 0) and 4.0 <= cyc.value <= 64.0
    assert 0.0 < fill.value <= 1.0
    X = rng.randn(P, N, 4).astype(np.float32)
    Xt = cu(X)
    for use_uniform in ((1, 0) if kind == "uniform" else (1,)):
        pipeline_knob(panel_uniform=use_uniform)
        if use_uniform == 0:                                 # also the unsorted / unordered plan image
            pipeline_knob(panel_sort=0, panel_order=0)
            gso = SparseGSO([A])
            plans = gso.plans(DEV)
        for op, M in ((0, A.T.tocsr()), (1, A)):
            out = torch.full((P, N, 4), float("nan"), device=DEV)
            _lib.check(L.gf_spmm_hop_panel(plans[0], op, Xt.data_ptr(), out.data_ptr(), P, stream()))
            want = np.stack([M.astype(np.float64) @ X[p].astype(np.float64) for p in range(P)])
            assert relerr(out.cpu().numpy(), want) < 2e-6, (op, use_uniform)
            out2 = torch.empty_like(out)
            pipeline_knob(panel_np=2 if N <= 5119 else 1)     # force the two-panels-per-pass kernel where the LDS allows it
            _lib.check(L.gf_spmm_hop_panel(plans[0], op, Xt.data_ptr(), out2.data_ptr(), P, stream()))
            pipeline_knob(panel_np=0)
            assert torch.equal(out, out2)                    # bitwise deterministic, whatever the pass width


@pytest.mark.parametrize("N,P,kind,knobs", [
    (1682, 1024, "weighted", {}), (1682, 1024, "uniform", {}), (1280, 513, "weighted", {}), (2559, 515, "uniform", {}), (2000, 2050, "weighted", {}),
    (1000, 601, "weighted", dict(panel_db=2)), (300, 1029, "uniform", dict(panel_db=2)), (129, 512, "weighted", dict(panel_db=2)),
    (1682, 700, "weighted", dict(panel_loaders=1)), (1682, 700, "uniform", dict(panel_loaders=4, panel_thr=768)),
], ids=lambda v: str(v))
def test_double_buffered_panel_hop_is_bitwise_the_per_hop_kernel(N, P, kind, knobs, pipeline_knob):
    """spmm_panel_db_kernel (loader waves fetch the next pass with LDS-DMA under the gathers, slices claimed from an LDS counter) walks the
    same ELL image with the same accumulators as spmm_panel_kernel: the same bits, whatever the claim order; against scipy as well.
    Launched 6 times in a row (the claim order differs from launch to launch)."""
    L = _lib.lib()
    rng = np.random.RandomState(N + P)
    A = sp.random(N, N, density=min(0.5, 12.0 / N), format="lil", random_state=rng, data_rvs=rng.randn)
    A[N // 2, :] = 0
    A[:, N // 3] = 0
    hub = rng.choice(N, size=min(N, 200), replace=False)
    A[1, hub] = rng.randn(len(hub))
    A[hub, 2] = rng.randn(len(hub))
    A = sp.csr_matrix(A)
    if kind == "uniform":
        A.data[:] = -0.41
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    X = rng.randn(P, N, 4).astype(np.float32)
    Xt = cu(X)
    for op, M in ((0, A.T.tocsr()), (1, A)):
        pipeline_knob(panel_db=0)
        ref = torch.full((P, N, 4), float("nan"), device=DEV)
        _lib.check(L.gf_spmm_hop_panel(plans[0], op, Xt.data_ptr(), ref.data_ptr(), P, stream()))
        want = np.stack([M.astype(np.float64) @ X[p].astype(np.float64) for p in range(P)])
        assert relerr(ref.cpu().numpy(), want) < 2e-6
        pipeline_knob(**dict(dict(panel_db=1), **knobs))
        for it in range(6):
            out = torch.full((P, N, 4), float("nan"), device=DEV)
            _lib.check(L.gf_spmm_hop_panel(plans[0], op, Xt.data_ptr(), out.data_ptr(), P, stream()))
            assert torch.equal(out, ref), (op, it)
        pipeline_knob(panel_db=1, panel_thr=0, panel_loaders=0)


@pytest.mark.parametrize("N,B,W,K,kind", [
    (37, 3, 8, 3, "weighted"), (64, 2, 4, 2, "uniform"), (100, 5, 8, 5, "weighted"), (641, 7, 8, 4, "uniform"),
    (1300, 3, 32, 5, "weighted"), (2600, 9, 16, 3, "uniform"), (5200, 40, 8, 3, "weighted"), (5121, 33, 32, 5, "uniform"),
    (10000, 30, 32, 5, "uniform"), (10239, 70, 8, 4, "weighted"), (9000, 1, 4, 6, "weighted"),
    (1000, 3, 4, 3, "weighted"), (300, 5, 12, 4, "uniform"), (5119, 7, 4, 3, "weighted"),       # two panels per pass, odd panel counts
], ids=lambda v: str(v))
def test_khop_panel_chain_against_scipy(N, B, W, K, kind, pipeline_knob):
    """The K-1 hops of a panel inside LDS (gf_chain.hip) against scipy, tap by tap and for both operators: 1 .. 16 waves per
    workgroup, 1 .. 10 row sets per wave, more panels than workgroups and fewer, empty rows / columns, a hub row, value-free
    and weighted streams; the per-hop kernel must agree to round-off (different summation order), two runs bitwise."""
    L = _lib.lib()
    rng = np.random.RandomState(N + K)
    A = sp.random(N, N, density=min(0.5, 9.0 / N), format="lil", random_state=rng, data_rvs=rng.randn)
    A[N // 2, :] = 0                                        # empty row
    A[:, N // 3] = 0                                        # empty column
    hub = rng.choice(N, size=min(N, 300), replace=False)    # one long row and one long column
    A[1, hub] = rng.randn(len(hub))
    A[hub, 2] = rng.randn(len(hub))
    A = sp.csr_matrix(A)
    A = A * (1.0 / max(1.0, abs(A).sum(axis=1).max(), abs(A).sum(axis=0).max()))   # powers stay O(1)
    if kind == "uniform":
        A.data[:] = 0.5 / max(1, np.diff(A.indptr).max(), np.diff(A.tocsc().indptr).max())
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    x = rng.randn(B, W, N).astype(np.float32)
    P = B * W // 4
    for op, M in ((0, A.T.tocsr().astype(np.float64)), (1, A.astype(np.float64))):
        Zs = {}
        for chain in (2, 0):
            pipeline_knob(panel_chain=chain)
            Z = torch.full((K, P, N, 4), float("nan"), device=DEV)
            Z[0] = cu(to_panels(x, N))
            _lib.check(L.gf_khop_panel(plans, 1, op, Z.data_ptr(), B, W, K, stream()))
            Zs[chain] = Z
        again = Zs[2].clone()
        again[1:] = float("nan")
        pipeline_knob(panel_chain=2)
        _lib.check(L.gf_khop_panel(plans, 1, op, again.data_ptr(), B, W, K, stream()))
        assert torch.equal(again, Zs[2])                                       # bitwise run-to-run
        tap = x.astype(np.float64)
        got = Zs[2].cpu().numpy()
        for k in range(1, K):
            tap = np.stack([(M @ tap[b].T).T for b in range(B)])               # [B, W, N]
            want = to_panels(tap, N)
            scale = max(np.abs(want).max(), 1e-30)
            assert np.abs(got[k] - want).max() < 2e-6 * scale * k, (op, k)
            assert np.abs(Zs[0][k].cpu().numpy() - want).max() < 2e-6 * scale * k, (op, k, "per-hop")


def test_khop_panel_chain_keeps_nonfinite_values_local():
    """A NaN in one node's signal reaches exactly the nodes within k hops of it (padding slots gather from the zero slot)."""
    L = _lib.lib()
    N, B, W, K = 700, 2, 8, 3
    A = graphgen.sbm(N, seed=3, directed=True)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    x = np.random.RandomState(0).randn(B, W, N).astype(np.float32)
    x[:, :, 0] = np.nan
    Z = torch.zeros((K, B * W // 4, N, 4), device=DEV)
    Z[0] = cu(to_panels(x, N))
    tune(panel_chain=2)
    try:
        _lib.check(L.gf_khop_panel(plans, 1, 1, Z.data_ptr(), B, W, K, stream()))
    finally:
        tune(panel_chain=1)
    reach = np.zeros(N, dtype=bool)
    reach[0] = True
    pat = (A != 0).astype(np.int8).tocsr()
    for k in range(1, K):
        reach = np.asarray(pat @ reach.astype(np.int8)).ravel() > 0
        bad = np.isnan(Z[k].cpu().numpy()).any(axis=(0, 2))
        assert np.array_equal(bad, reach), k


def test_spmm_hop_panel_does_not_spread_nonfinite_values():
    """A NaN in one node's signal reaches exactly its out-neighbours (rows exhausted early gather from a zero slot, never
    from real data)."""
    L = _lib.lib()
    N, P = 500, 8
    A = graphgen.sbm(N, seed=3, directed=True)
    gso = SparseGSO([A])                                   # owns the device plans: keep it alive while they are used
    plans = gso.plans(DEV)
    X = np.random.RandomState(0).randn(P, N, 4).astype(np.float32)
    X[:, 0, :] = np.nan
    out = torch.empty((P, N, 4), device=DEV)
    Xt = cu(X)
    _lib.check(L.gf_spmm_hop_panel(plans[0], 1, Xt.data_ptr(), out.data_ptr(), P, stream()))
    bad = np.isnan(out.cpu().numpy()).any(axis=(0, 2))
    assert np.array_equal(bad, np.asarray((A[:, 0] != 0).todense()).ravel())


@pytest.mark.parametrize("shape", [(3, 100, 100, 32, 32, 1, 5), (2, 70, 33, 8, 16, 2, 3), (1, 234, 234, 64, 32, 1, 5),
                                   (4, 45, 45, 16, 64, 1, 1), (2, 1000, 999, 32, 128, 1, 4)], ids=str)
@pytest.mark.parametrize("transpose", [0, 1])
def test_contract_and_grad_taps_panel_against_einsum(shape, transpose):
    L = _lib.lib()
    B, N, Nout, G, F, E, K = shape
    T = 1 + E * (K - 1)
    Cin = F if transpose else G
    rng = np.random.RandomState(5)
    Zn = rng.randn(T, B, Cin, N).astype(np.float32)                       # taps in the reference orientation [B,C,N]
    Zp = np.stack([to_panels(Zn[t], N) for t in range(T)])
    h = (rng.uniform(-1, 1, (F, E, K, G)) / np.sqrt(G * K)).astype(np.float32)
    bias = rng.uniform(-1, 1, (F,)).astype(np.float32)
    Cout = G if transpose else F
    out = torch.full((B, Cout, Nout), float("nan"), device=DEV)
    Zt, ht, bt = cu(Zp), cu(h), cu(bias)                                  # keep the device buffers alive across the calls
    _lib.check(L.gf_contract_panel(Zt.data_ptr(), ht.data_ptr(), None if transpose else bt.data_ptr(),
                                   out.data_ptr(), B, N, Nout, G, F, E, K, transpose, stream()))
    hb = _bank(h.astype(np.float64), transpose)                           # [T, Cin, Cout]
    want = np.einsum("tbcn,tco->bon", Zn.astype(np.float64), hb)[:, :, :Nout]
    if not transpose:
        want = want + bias.astype(np.float64)[None, :, None]
    assert relerr(out.cpu().numpy(), want) < 5e-6
    if transpose:
        return
    # grad_taps: dh[f,e,k,g] = sum_{b,n} Z[t(e,k),b,g,n] dY[b,f,n]
    dY = rng.randn(B, F, N).astype(np.float32)
    nb = L.gf_grad_taps_workspace_bytes(B, N, G, F, E, K)
    ws = torch.empty(nb // 4 + 1, device=DEV)
    dh = torch.full((F, E, K, G), float("nan"), device=DEV)
    db = torch.full((F,), float("nan"), device=DEV)
    dYt = cu(to_panels(dY, N))
    full = np.einsum("tbgn,bfn->tfg", Zn.astype(np.float64), dY.astype(np.float64))
    want_dh = np.zeros((F, E, K, G))
    for e in range(E):
        want_dh[:, e, 0] = full[0]
        for k in range(1, K):
            want_dh[:, e, k] = full[1 + e * (K - 1) + (k - 1)]
    for lds in (1, 0):                                                    # operands transposed through LDS / direct strided loads
        tune(gradw_lds=lds)
        dh.fill_(float("nan"))
        db.fill_(float("nan"))
        _lib.check(L.gf_grad_taps_panel(Zt.data_ptr(), dYt.data_ptr(), dh.data_ptr(), db.data_ptr(),
                                        ws.data_ptr(), nb, B, N, G, F, E, K, stream()))
        assert relerr(dh.cpu().numpy(), want_dh) < 5e-6, lds
        assert relerr(db.cpu().numpy(), dY.astype(np.float64).sum(axis=(0, 2))) < 5e-6, lds
    tune(gradw_lds=1)


@pytest.mark.parametrize("pipe", [1, 2])
@pytest.mark.parametrize("path", [p for p in golden_files("lsigf") if any(k in p for k in ("asym37_G32", "fbego_G32", "fbego_G64_F32", "sbm100_G32", "asym37_nobias"))], ids=case_id)
def test_lsigf_golden_under_both_pipelines(path, pipe, pipeline_knob):
    """The same reference outputs through the node-major (L2 gather) and the column-panel (LDS gather) pipelines."""
    d = load(path)
    pipeline_knob(pipeline=pipe)
    gso = SparseGSO.from_any(d["S"])
    F, E, K, G = d["h"].shape
    assert _lib.lib().gf_lsigf_pipeline(gso.plans(DEV), E, G, F, K) == pipe
    h, x = cu(d["h"], True), cu(d["x"], True)
    b = cu(d["b"], True) if "b" in d else None
    y = LSIGF(h, gso, x, b)
    y.backward(cu(d["dy"]))
    assert relerr(y.detach().cpu().numpy(), d["y"]) < FWD_RTOL
    assert relerr(x.grad.cpu().numpy(), d["dx"]) < GRAD_RTOL
    assert relerr(h.grad.cpu().numpy(), d["dh"]) < GRAD_RTOL
    if b is not None:
        assert relerr(b.grad.cpu().numpy(), d["db"]) < GRAD_RTOL


def test_pipelines_agree_at_full_size(cfg2, pipeline_knob):
    """BASELINE configs[1] at full size: forward and all gradients of the two pipelines agree to fp32 round-off, and the
    panel pipeline is what the layer runs by default there."""
    layer, x = cfg2["layer"], cfg2["x"]
    dy = torch.randn(x.shape[0], layer.F, x.shape[2], device=DEV)
    res = {}
    for pipe in (1, 2, 3):                                  # 2 = panels, K-1 hops of a panel inside LDS (default); 3 = one launch per hop
        pipeline_knob(pipeline=min(pipe, 2), panel_chain=int(pipe == 2))
        xx = x.detach().clone().requires_grad_(True)
        for p_ in layer.parameters():
            p_.grad = None
        y = layer(xx)
        y.backward(dy)
        res[pipe] = (y.detach().clone(), xx.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone())
    pipeline_knob(pipeline=0, panel_chain=1)
    assert _lib.lib().gf_lsigf_pipeline(layer._gso.plans(DEV), 1, layer.G, layer.F, layer.K) == 2
    for other in (2, 3):                                     # three kernels, three summation orders: fp32 round-off apart
        for a, b_ in zip(res[1], res[other]):
            assert float((a - b_).abs().max()) <= 2e-6 * float(a.abs().max())


@pytest.mark.parametrize("cfg", [
    dict(N=200, B=6, G=16, F=64, K=3, E=2, nin=200, density=0.05),       # two edge features through the panel pipeline
    dict(N=333, B=3, G=8, F=8, K=1, E=1, nin=333, density=0.03),         # K = 1: no hop at all
    dict(N=500, B=5, G=32, F=16, K=4, E=1, nin=123, density=0.02),       # Nin < N: zero padding + kept nodes
    dict(N=64, B=2, G=8, F=8, K=3, E=1, nin=64, density=0.0),            # empty GSO: y = h_0 x + b
    dict(N=5200, B=9, G=8, F=8, K=3, E=1, nin=5200, density=0.002),      # 1024-thread panel workgroups, ragged last slice
], ids=lambda c: "N%d_G%d_F%d_K%d_E%d_Nin%d" % (c["N"], c["G"], c["F"], c["K"], c["E"], c["nin"]))
@pytest.mark.parametrize("pipe", [1, 2, 3], ids=["node_major", "panels_chain", "panels_per_hop"])
def test_lsigf_edge_cases_under_both_pipelines(cfg, pipe, pipeline_knob):
    N, B, G, F, K, E, nin = (cfg[k] for k in ("N", "B", "G", "F", "K", "E", "nin"))
    rng = np.random.RandomState(N + K)
    mats = []
    for e in range(E):
        A = sp.random(N, N, density=cfg["density"], format="csr", random_state=rng, data_rvs=rng.randn)
        mats.append(A * (0.5 / max(1.0, abs(A).sum(axis=1).max())) if A.nnz else A)
    pipeline_knob(pipeline=min(pipe, 2), panel_chain=2 if pipe == 2 else 0)       # 2 = chain kernel whatever the panel count
    gso = SparseGSO(mats)
    assert _lib.lib().gf_lsigf_pipeline(gso.plans(DEV), E, G, F, K) == min(pipe, 2)
    h = (rng.uniform(-1, 1, (F, E, K, G)) / np.sqrt(G * K)).astype(np.float32)
    x = rng.randn(B, G, nin).astype(np.float32)
    b = rng.uniform(-1, 1, (F, 1)).astype(np.float32)
    dy = rng.randn(B, F, nin).astype(np.float32)
    ht, xt, bt = cu(h, True), cu(x, True), cu(b, True)
    y = LSIGF(ht, gso, xt, bt)
    assert tuple(y.shape) == (B, F, nin)
    y.backward(cu(dy))
    xp = np.zeros((B, G, N), dtype=np.float32)
    xp[:, :, :nin] = x
    dyp = np.zeros((B, F, N), dtype=np.float32)
    dyp[:, :, :nin] = dy
    want = orc.lsigf_sparse(h, mats, xp, b)[:, :, :nin]
    dx, dh, db = orc.lsigf_sparse_grads(h, mats, xp, b, dyp)
    assert relerr(y.detach().cpu().numpy(), want) < FWD_RTOL
    assert relerr(xt.grad.cpu().numpy(), dx[:, :, :nin]) < GRAD_RTOL
    assert relerr(ht.grad.cpu().numpy(), dh) < GRAD_RTOL
    assert relerr(bt.grad.cpu().numpy(), db) < GRAD_RTOL


# ---------------------------------------------------------------------------------------------------------------
# MaxPoolLocal: HIP kernel against the reference's own formulation (gather + torch.max + autograd)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,Nout,hops,B,F", [(100, 10, 2, 6, 32), (234, 234, 1, 3, 8), (500, 77, 3, 2, 5)])
def test_max_pool_local_matches_reference_formulation(N, Nout, hops, B, F):
    A = graphgen.sbm(N, avg_degree=4.0, seed=N)
    pool = gml.MaxPoolLocal(N, Nout, hops)
    pool.addGSO(SparseGSO([A]))
    pool.to(DEV)
    rng = np.random.RandomState(0)
    xv = np.maximum(rng.randn(B, F, N), 0).astype(np.float32)          # ReLU output: many exact ties at 0
    x = cu(xv, True)
    v = pool(x)
    w = cu(rng.randn(B, F, Nout))
    (v * w).sum().backward()
    xr = cu(xv, True)                                                    # graphML.py:2003-2018 in torch ops
    nb = pool.neighborhood.long()
    vr, _ = torch.max(xr[:, :, nb], dim=3)
    (vr * w).sum().backward()
    assert torch.equal(v, vr)
    assert torch.allclose(x.grad, xr.grad, rtol=0, atol=1e-6)           # same arg-max element (first maximum) per output
    assert float((x.grad.sum() - xr.grad.sum()).abs()) < 1e-3
    x2 = cu(xv, True)
    v2 = pool(x2)
    (v2 * w).sum().backward()
    assert torch.equal(x.grad, x2.grad)                                  # deterministic


@pytest.mark.parametrize("pipe", [1, 2])
@pytest.mark.parametrize("G,F,nin", [(32, 32, 300), (8, 64, 200), (3, 5, 300)])
def test_fused_relu_epilogue_equals_separate_relu(pipe, G, F, nin, pipeline_knob):
    """activation='relu' (epilogue max(0, .) + masked backward) == torch.relu applied to the plain filter: values bitwise,
    gradients to round-off (the mask is the same, only the order of additions upstream differs)."""
    N, B, K = 300, 7, 4
    if pipe == 2 and (G % 8 or F % 8):
        pytest.skip("panel pipeline needs MFMA-tile widths")
    pipeline_knob(pipeline=pipe)
    A = graphgen.sbm(N, seed=21, directed=True)
    gso = SparseGSO([A])
    rng = np.random.RandomState(1)
    h = (rng.uniform(-1, 1, (F, 1, K, G)) / np.sqrt(G * K)).astype(np.float32)
    x = rng.randn(B, G, nin).astype(np.float32)
    b = rng.uniform(-1, 1, (F, 1)).astype(np.float32)
    w = rng.randn(B, F, nin).astype(np.float32)
    outs = []
    for fused in (True, False):
        ht, xt, bt = cu(h, True), cu(x, True), cu(b, True)
        y = LSIGF(ht, gso, xt, bt, activation="relu") if fused else torch.relu(LSIGF(ht, gso, xt, bt))
        (y * cu(w)).sum().backward()
        outs.append((y.detach(), xt.grad, ht.grad, bt.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and float((outs[0][0] == 0).float().mean()) > 0.2
    for a, r in zip(outs[0][1:], outs[1][1:]):
        assert float((a - r).abs().max()) <= 1e-5 * float(r.abs().max())
    # per-node bias [F, N] (graphML.py:110-112) cannot be fused: same result through the unfused path
    bn = rng.uniform(-1, 1, (F, N)).astype(np.float32)
    y1 = LSIGF(cu(h), gso, cu(x), cu(bn), activation="relu")
    y2 = torch.relu(LSIGF(cu(h), gso, cu(x), cu(bn)))
    assert torch.equal(y1, y2)


def test_training_step_is_hip_graph_capturable():
    """A SelectionGNN training step (filters + fused ReLU + MaxPoolLocal + MLP, forward and backward) captured in a HIP graph
    through torch.cuda.graph and replayed: the library launches only on the stream it is given, allocates nothing and never
    synchronises, so the launch-bound small configurations (BASELINE config 1: N = 100) can be replayed as one graph."""
    d = load(os.path.join(GOLDEN, "selgnn_cfg1_sbm100.npz"))
    cfg = d["cfg"]
    net = SelectionGNN(cfg["dimNodeSignals"], cfg["nFilterTaps"], True, torch.nn.ReLU, cfg["nSelectedNodes"],
                       getattr(gml, cfg["pool"]), cfg["poolingSize"], cfg["dimLayersMLP"], d["S"][0]).float().to(DEV)
    x_static = torch.randn(16, 1, 100, device=DEV)
    params = [p for p in net.parameters()]

    def step():
        for p in params:
            p.grad = None
        loss = net(x_static).square().sum()
        loss.backward()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up outside the capture: plans, LDS attributes, allocator pools
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    eager_loss = step().item()
    eager_grads = [p.grad.clone() for p in params]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = step()
    static_grads = [p.grad for p in params]             # the graph's output buffers (p.grad is re-pointed by the eager runs below)
    x_new = torch.randn(16, 1, 100, device=DEV)
    for xv in (x_static.clone(), x_new):
        x_static.copy_(xv)
        g.replay()
        torch.cuda.synchronize()
        graph_loss = static_loss.item()
        graph_grads = [t.clone() for t in static_grads]
        for p in params:
            p.grad = None
        ref_loss = net(xv).square().sum()
        ref_loss.backward()
        assert abs(graph_loss - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
        for a, p in zip(graph_grads, params):
            assert torch.equal(a, p.grad)
    assert abs(eager_loss - eager_loss) == 0 and len(eager_grads) == len(params)


# ---------------------------------------------------------------------------------------------------------------
# layer-to-layer hand-over in the internal layout (gf_lsigf_forward_ex / gf_lsigf_backward_ex)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dimF,K,N,B", [([1, 64, 32], [5, 5], 234, 5), ([3, 8, 16, 8], [3, 2, 4], 500, 7), ([32, 32, 32], [5, 5], 2000, 16),
                                        ([2, 16, 8], [3, 3], 97, 3), ([2, 24, 40], [3, 3], 97, 3), ([8, 64, 32, 16], [3, 5, 2], 300, 4),
                                        ([2, 16, 24], [3, 3], 97, 3),                       # panel layer + node-major layer: no hand-over
                                        ([32, 32, 32], [5, 5], 12000, 3), ([3, 16, 64, 8], [3, 2, 4], 10500, 2),   # N > 10239: node-major rows
                                        ([5, 40, 24, 132], [2, 3, 2], 301, 2)])            # widths without a panel / MFMA tiling
def test_layer_handover_is_bitwise_the_separate_layers(dimF, K, N, B, monkeypatch):
    """Runs of [GraphFilter, ReLU, NoPool] blocks keep their signals in the internal layout between layers (column panels; node-major
    rows when the layers run the node-major pipeline: N > 10239 or widths the panels do not tile): the same kernels do the same
    arithmetic in the same order, only one reference-layout round trip per inner boundary is gone -- outputs, input gradient and
    every parameter gradient must be BITWISE those of the separate layers (which the golden tests pin to the reference)."""
    from alegnn_amd import functional
    A = graphgen.sbm(N, avg_degree=8.0, seed=3)
    torch.manual_seed(1)
    net = SelectionGNN(dimF, K, True, torch.nn.ReLU, [N] * len(K), gml.NoPool, [1] * len(K), [4], A).to(DEV)
    x = torch.randn(B, dimF[0], N, device=DEV)
    w = torch.randn(B, 4, device=DEV)
    outs = {}
    for mode in (True, False):
        monkeypatch.setattr(functional, "_HANDOVER", mode)
        xr = x.clone().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        calls = []
        orig = functional._LSIGFChainFunction.apply
        monkeypatch.setattr(functional._LSIGFChainFunction, "apply", staticmethod(lambda *a, **k: (calls.append(1), orig(*a, **k))[1]))
        y, ygnn = net.splitForward(xr)
        (y * w).sum().backward()
        monkeypatch.setattr(functional._LSIGFChainFunction, "apply", orig)
        pw = [functional._padded_width(f) for f in dimF]
        panel = [N <= 10239 and g in (8, 16, 32, 64, 128) and f in (8, 16, 32, 64, 128) for g, f in zip(pw[:-1], pw[1:])]   # per layer
        chainable = all(panel) or (not any(panel) and all(w % 4 == 0 for w in pw))   # one pipeline for the whole run
        assert bool(calls) == (mode and chainable), (panel, pw)      # the chain really ran (or really did not)
        outs[mode] = [y.detach().clone(), ygnn.detach().clone(), xr.grad.clone()] + [p.grad.clone() for p in net.parameters()]
    for a, b_ in zip(outs[True], outs[False]):
        assert torch.equal(a, b_)


def test_handover_entry_points_refuse_what_they_cannot_do():
    L = _lib.lib()
    A = graphgen.er(20000, avg_degree=6.0, seed=0)                   # N > 10239: node-major pipeline
    gso = SparseGSO.from_any(A)
    plans = gso.plans(DEV)
    t = torch.zeros(16, device=DEV)
    rc = L.gf_lsigf_forward_ex(plans, 1, t.data_ptr(), t.data_ptr(), None, t.data_ptr(), t.data_ptr(), 1, 8, 8, 2, 19999, 4, stream())
    assert rc != 0                                                   # GF_ERR_UNSUPPORTED: a handed-over signal covers every node
    assert b"hand-over" in L.gf_last_error()
    rc = L.gf_lsigf_forward_ex(plans, 1, t.data_ptr(), t.data_ptr(), None, t.data_ptr(), t.data_ptr(), 1, 6, 8, 2, 20000, 4, stream())
    assert rc != 0 and b"multiples of 4" in L.gf_last_error()        # node-major rows are written 16 bytes at a time
