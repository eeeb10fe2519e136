"""CPU-only tests: host-side logic of the package and the C-ABI surface (no compute calls -- no GPU here)."""
import ctypes
import os
import sys
import re
import types

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from _util import GOLDEN, case_id, golden_files, load, relerr

from alegnn_amd import SparseGSO, _lib, graphgen
from alegnn_amd.modules.architectures import SelectionGNN
from alegnn_amd.utils import graphML as gml
from alegnn_amd.utils import graphTools as gt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ------------------------------------------------------------------------------------------------
def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gfhip.h")).read()
    declared = sorted(set(re.findall(r"\b(gf_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 15
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in gfhip.h but not exported by libgfhip.so"
    assert sorted(_lib.exported_symbols()) == declared, "ctypes signature table out of sync with gfhip.h"
    m = re.search(r"#define GFHIP_VERSION (\d+)", hdr)
    assert L.gf_version() == int(m.group(1))


def test_c_abi_error_convention_without_gpu():
    """Argument validation happens before any HIP call: status codes + thread-local message, no exceptions in C."""
    L = _lib.lib()
    assert L.gf_plan_info(None, None, None, None) == -2
    assert b"plan is NULL" in L.gf_last_error()
    rowptr = np.array([0, 2, 1], dtype=np.int32)           # not monotone
    col = np.array([0, 1], dtype=np.int32)
    val = np.array([1.0, 2.0], dtype=np.float32)
    out = ctypes.c_void_p()
    rc = L.gf_plan_create(2, 1, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, 0, 0, ctypes.byref(out))
    assert rc == _lib.GF_ERR_SHAPE
    with pytest.raises(AssertionError):                     # the reference raises AssertionError for shape violations
        _lib.check(rc, "gf_plan_create")
    rowptr = np.array([0, 1, 2], dtype=np.int32)
    col = np.array([0, 5], dtype=np.int32)                  # column out of range
    rc = L.gf_plan_create(2, 2, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, 0, 0, ctypes.byref(out))
    assert rc == _lib.GF_ERR_SHAPE and b"column index" in L.gf_last_error()
    assert L.gf_grad_taps_workspace_bytes(256, 10000, 32, 32, 1, 5) > 0
    assert L.gf_spmm_hop(None, 0, None, None, 1, 32, None) == -2
    # node-variant entry points: scratch sizes follow the header's formula, NULL arguments are refused before any launch
    B, N, G, F, E, K = 7, 100, 3, 5, 2, 4
    T = 1 + E * (K - 1)
    assert L.gf_nvgf_scratch_floats(B, N, G, F, E, K, 0) == N * T * G * F + B * N * F
    assert L.gf_nvgf_scratch_floats(B, N, G, F, E, K, 1) == 2 * N * T * G * F + B * N * F + T * B * N * G + B * N * G
    assert L.gf_nvgf_scratch_floats(0, N, G, F, E, K, 0) == 0
    assert L.gf_nvgf_forward(None, 1, None, None, None, None, None, None, 0, B, G, F, K, N, None) == -2
    assert L.gf_nvgf_backward(None, 1, None, None, None, None, None, None, 0, B, G, F, K, N, None) == -2
    assert L.gf_nvgf_fold_taps(None, None, None, None, 10, 5, 2, None) == -2
    assert L.gf_tune(b"bwd_fuse", 1) == 0 and L.gf_tune(b"panel_split", 0) == 0 and L.gf_tune(b"no_such_knob", 1) != 0


def test_tuning_knobs_are_refused_outside_experiment_processes():
    """gf_tune is process-global state; a process that did not set GFHIP_EXPERIMENTS=1 before loading the library cannot change it
    (GF_ERR_UNSUPPORTED = -4), so the product path never reads mutable global state."""
    import subprocess
    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); L.gf_last_error.restype = ctypes.c_char_p; "
            "rc = L.gf_tune(b'pipeline', 1); print(rc, L.gf_last_error().decode())")
    env = {k: v for k, v in os.environ.items() if k != "GFHIP_EXPERIMENTS"}
    out = subprocess.run([sys.executable, "-c", code, _lib.LIB_PATH], env=env, capture_output=True, text=True, check=True).stdout
    assert out.startswith("-4 ") and "GFHIP_EXPERIMENTS" in out
    env["GFHIP_EXPERIMENTS"] = "1"
    out = subprocess.run([sys.executable, "-c", code, _lib.LIB_PATH], env=env, capture_output=True, text=True, check=True).stdout
    assert out.startswith("0 ")


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgfhip.so")
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        _lib.lib()


def test_no_cpu_fallback():
    layer = gml.GraphFilter(4, 8, 3)
    layer.addGSO(torch.eye(5).reshape(1, 5, 5))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(torch.zeros(2, 4, 5))


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under the product package may import, call or link it."""
    pkg = os.path.join(ROOT, "graph-neural-networks_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "lsigf_oracle", "oracle/_ref", "oracle."):
                    assert needle not in src, f"{os.path.join(dirpath, f)} references the oracle ({needle!r})"


# ---- GSO ingest --------------------------------------------------------------------------------------------
def test_sparse_gso_from_any_forms_agree():
    rng = np.random.RandomState(0)
    S = (rng.rand(2, 9, 9) < 0.3) * rng.randn(2, 9, 9)
    ref = SparseGSO.from_any(S)
    assert ref.shape == (2, 9, 9) and ref.nnz == [int((S[e] != 0).sum()) for e in range(2)]
    forms = [torch.tensor(S), [sp.csr_matrix(S[0]), sp.coo_matrix(S[1])], [S[0], S[1]], ref]
    for f in forms:
        g = SparseGSO.from_any(f)
        for e in range(2):
            assert (g.mats[e] != ref.mats[e]).nnz == 0
    one = SparseGSO.from_any(torch.tensor(S[0]).to_sparse_csr())
    assert (one.mats[0] != ref.mats[0]).nnz == 0
    assert torch.equal(ref.to_dense(torch.float64), torch.tensor(S))
    with pytest.raises(AssertionError):
        SparseGSO.from_any(np.zeros((3, 4)))


def test_graphgen_models():
    A = graphgen.sbm(2000, avg_degree=10, seed=3)
    assert A.shape == (2000, 2000) and abs(A.nnz / 2000 - 10) < 1.0
    assert (abs(A - A.T)).nnz == 0 and A.diagonal().sum() == 0
    B = graphgen.er(3000, avg_degree=8, seed=1, directed=True)
    assert (abs(B - B.T)).nnz > 0 and abs(B.nnz / 3000 - 8) < 1.0
    lam = np.max(np.abs(np.linalg.eigvals(graphgen.sbm(300, seed=2).toarray())))
    assert abs(lam - 1.0) < 1e-6                              # S = A / lambda_max (examples/sourceLocGNN.py:752)
    intra = sum(A[i * 400:(i + 1) * 400, i * 400:(i + 1) * 400].nnz for i in range(5))
    assert 0.4 < intra / A.nnz < 0.6                          # 4:1 probabilities, 5 communities -> ~50 % intra edges


# ---- graphTools mirror vs the reference's outputs ------------------------------------------------------------------
def test_graphtools_matches_reference_outputs():
    d = dict(np.load(os.path.join(GOLDEN, "graphtools_sbm100.npz")))
    S = d["S"]
    for key in [k for k in d if k.startswith("nbh_")]:
        K, N, nb = (int(v) for v in key.split("_")[1:])
        src = d["asym37"] if nb == 20 else S[None]
        got = np.sort(gt.computeNeighborhood(src, K, N, nb, "matrix"), axis=1)
        assert got.shape == d[key].shape and np.array_equal(got, d[key]), key
        sparse_src = [sp.csr_matrix(m) for m in (src if src.ndim == 3 else src[None])]
        got2 = np.sort(gt.computeNeighborhood(sparse_src, K, N, nb, "matrix"), axis=1)
        assert np.array_equal(got2, d[key]), key + " (sparse input)"
    for name in ("Degree", "EDS", "SpectralProxies"):
        Sp, order = getattr(gt, "perm" + name)(S)
        assert list(order) == list(d["order_" + name]), name
        assert np.array_equal(Sp, d["S_" + name])
    Sp, order = gt.permDegree(d["asym_E2"])
    assert list(order) == list(d["order_Degree_E2"]) and np.array_equal(Sp, d["S_Degree_E2"])
    Si, oi = gt.permIdentity(S)
    assert oi == list(range(100)) and Si.shape == (100, 100)


# ---- module surface ----------------------------------------------------------------------------------------------
def test_graph_filter_surface_matches_reference():
    torch.manual_seed(0)
    layer = gml.GraphFilter(3, 7, 4, E=2, bias=True)
    assert (layer.G, layer.F, layer.K, layer.E, layer.S) == (3, 7, 4, 2, None)
    assert tuple(layer.weight.shape) == (7, 2, 4, 3) and tuple(layer.bias.shape) == (7, 1)     # graphML.py:2101-2103
    assert list(layer.state_dict().keys()) == ["weight", "bias"]                                # S is not in the state_dict
    bound = 1.0 / np.sqrt(3 * 4)                                                                # graphML.py:2111
    assert float(layer.weight.abs().max()) <= bound and float(layer.bias.abs().max()) <= bound
    assert "no GSO stored" in layer.extra_repr()
    nob = gml.GraphFilter(3, 7, 4, bias=False)
    assert nob.bias is None and list(nob.state_dict().keys()) == ["weight"]
    with pytest.raises(AssertionError):
        layer.addGSO(torch.zeros(5, 5))                   # needs 3 dims (graphML.py:2118)
    with pytest.raises(AssertionError):
        layer.addGSO(torch.zeros(1, 5, 5))                # E mismatch (graphML.py:2120)
    with pytest.raises(AssertionError):
        layer.addGSO(torch.zeros(2, 5, 6))                # not square (graphML.py:2122)
    layer.addGSO(torch.zeros(2, 5, 5))
    assert layer.N == 5 and "GSO stored" in layer.extra_repr()


def _build(d, pool):
    cfg = d["cfg"]
    return SelectionGNN(cfg["dimNodeSignals"], cfg["nFilterTaps"], True, torch.nn.ReLU, cfg["nSelectedNodes"],
                        getattr(gml, pool), cfg["poolingSize"], cfg["dimLayersMLP"], d["S"][0])


@pytest.mark.parametrize("name", ["cfg1_sbm100", "cfg3_fbego"])
def test_selection_gnn_checkpoint_compatibility(name):
    d = load(os.path.join(GOLDEN, f"selgnn_{name}.npz"))
    net = _build(d, d["cfg"]["pool"])
    ref_sd = {k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")}
    assert list(net.state_dict().keys()) == list(ref_sd.keys())                 # same names, same order
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(ref_sd[k].shape), k
    net.load_state_dict(ref_sd, strict=True)                                    # reference checkpoint loads
    assert net.N == [d["S"].shape[1]] + d["cfg"]["nSelectedNodes"]
    # sigma = ReLU is fused into the filter's epilogue; a parameter-free placeholder keeps the GFL indices (state_dict keys)
    assert isinstance(net.GFL[0], gml.GraphFilter) and isinstance(net.GFL[1], gml.FusedReLU)
    assert net.GFL[0].fused_activation == "relu"


def test_selection_gnn_ctor_options():
    d = load(os.path.join(GOLDEN, "selgnn_cfg1_sbm100.npz"))
    S = d["S"][0]
    net = SelectionGNN([1, 8], [3], True, torch.nn.ReLU, [10], gml.MaxPoolLocal, [2], [4], S, order="Degree")
    g = dict(np.load(os.path.join(GOLDEN, "graphtools_sbm100.npz")))
    assert list(net.order) == list(g["order_Degree"])                           # the reference's NameError path, fixed
    assert np.array_equal(net.S.numpy()[0], g["S_Degree"])          # S is kept E x N x N (architectures.py:195)
    with pytest.raises(AssertionError):
        SelectionGNN([1, 8, 8], [3], True, torch.nn.ReLU, [10], gml.NoPool, [1], [], S)
    sparse_net = SelectionGNN([1, 8], [3], True, torch.nn.ReLU, [100], gml.NoPool, [1], [], sp.csr_matrix(S))
    assert sparse_net.E == 1 and sparse_net.N == [100, 100] and len(sparse_net.MLP) == 0
    sparse_net.changeGSO(sp.csr_matrix(S[:50, :50]), nSelectedNodes=[50])
    assert sparse_net.N == [50, 50] and sparse_net.GFL[0].N == 50


def test_node_orderings_on_sparse_gsos():
    """order= with a GSO that only exists sparse (round 5; the reference's permFunction takes the dense array, architectures.py:203-256):
    'Degree' on CSR gives the dense result whenever the degrees are distinct up to rounding -- on the reference's own SBM graph the
    degree SEQUENCE is identical and the reordered matrix is a simultaneous row / column permutation of S; 'EDS' and 'SpectralProxies'
    densify for the order (small graphs) and are refused beyond graphTools.kDenseOrderingMaxNodes nodes."""
    from alegnn_amd.utils import graphTools
    d = load(os.path.join(GOLDEN, "selgnn_cfg1_sbm100.npz"))
    S = d["S"][0]
    g = dict(np.load(os.path.join(GOLDEN, "graphtools_sbm100.npz")))
    for name in ("Degree", "EDS", "SpectralProxies"):
        net = SelectionGNN([1, 8], [3], True, torch.nn.ReLU, [100], gml.NoPool, [1], [], sp.csr_matrix(S), order=name)
        order = np.asarray(net.order)
        assert sorted(order.tolist()) == list(range(100))
        assert np.array_equal(net.S.mats[0].toarray(), S[order][:, order])          # S_e[order][:, order], as graphTools.py:1046-1048
        ref_order = np.asarray(g["order_" + name])
        if name == "Degree":                                                         # same degrees position by position (ties may swap)
            deg = S.sum(axis=0)
            assert np.allclose(deg[order], deg[ref_order], rtol=0, atol=1e-12)
        else:
            assert order.tolist() == ref_order.tolist()
    big = sp.identity(graphTools.kDenseOrderingMaxNodes + 1, format="csr")
    with pytest.raises(NotImplementedError):
        SelectionGNN([1, 8], [3], True, torch.nn.ReLU, [big.shape[0]], gml.NoPool, [1], [], big, order="EDS")
    net = SelectionGNN([1, 8], [3], True, torch.nn.ReLU, [big.shape[0]], gml.NoPool, [1], [], big, order="Degree")   # O(nnz): any size
    assert len(net.order) == big.shape[0]


@pytest.mark.parametrize("name", ["sbm100_L2", "fbego_L3"])
def test_graclus_coarsening_matches_reference(name):
    """coarsen() against the reference's own outputs (graphs level by level, fake nodes, node order), and the
    SelectionGNN(coarsening=True) bookkeeping built on it (architectures.py:224-247, :282-290)."""
    d = load(os.path.join(GOLDEN, f"selgnn_coarsen_{name}.npz"))
    cfg, S = d["cfg"], d["S"][0]
    L = len(cfg["nFilterTaps"])
    np.random.seed(int(d["seed"]))
    graphs, perm = gt.coarsen(sp.csr_matrix(S), levels=L, self_connections=False)
    assert [int(v) for v in perm] == d["perm"].tolist()
    for l, g in enumerate(graphs):
        n = int(d[f"graph{l}_n"])
        want = sp.coo_matrix((d[f"graph{l}_v"], (d[f"graph{l}_r"], d[f"graph{l}_c"])), shape=(n, n)).toarray()
        assert g.shape == (n, n) and np.array_equal(g.toarray(), want)
    x = np.random.RandomState(0).randn(2, 3, S.shape[0]).astype(np.float32)
    xp = gt.permCoarsening(x, perm)
    assert xp.dtype == np.float32 and xp.shape[2] == len(perm)
    for i, j in enumerate(perm):
        assert np.array_equal(xp[:, :, i], x[:, :, j] if j < S.shape[0] else np.zeros((2, 3), np.float32))

    def build(gso):
        np.random.seed(int(d["seed"]))
        return SelectionGNN(cfg["dimNodeSignals"], cfg["nFilterTaps"], True, torch.nn.ReLU, [0] * L, torch.nn.MaxPool1d,
                            [7] * L, cfg["dimLayersMLP"], gso, order="Degree", coarsening=True)   # order / alpha are overridden
    for net in (build(S), build(sp.csr_matrix(S))):
        assert net.coarsening and net.alpha == [2] * L and net.E == 1
        assert net.N == [int(d[f"graph{l}_n"]) for l in range(L + 1)]
        assert [int(v) for v in net.order] == d["perm"].tolist()
        for l in range(L):
            assert isinstance(net.GFL[3 * l + 2], torch.nn.MaxPool1d) and net.GFL[3 * l].N == net.N[l]
        ref_sd = {k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")}
        assert list(net.state_dict().keys()) == list(ref_sd.keys())
        net.load_state_dict(ref_sd, strict=True)
    dense = build(S)
    assert [tuple(t.shape) for t in dense.S] == [(1, n, n) for n in dense.N]       # reference attribute (:232-245)
    two = SelectionGNN([1, 4], [2], True, torch.nn.ReLU, [10], gml.NoPool, [1], [], np.stack([S, S]), coarsening=True)
    assert not two.coarsening and two.N == [S.shape[0], 10]                      # E = 2 falls back to selection (:224, :257)
    np.random.seed(0)
    dense.changeGSO(S[:60, :60])
    assert dense.N[0] >= 60 and dense.GFL[0].N == dense.N[0] and dense.GFL[3].N == dense.N[1]


def test_graclus_edge_cases():
    g, perm = gt.coarsen(sp.csr_matrix(np.ones((4, 4)) - np.eye(4)), levels=0)
    assert perm is None and len(g) == 1 and g[0].nnz == 12
    path = sp.diags([np.ones(6), np.ones(6)], [1, -1]).tocsr()                   # 7-node path: odd count forces a fake node
    np.random.seed(0)
    g, perm = gt.coarsen(path, levels=2)
    assert g[0].shape[0] == 2 * g[1].shape[0] == 4 * g[2].shape[0] and sorted(perm) == list(range(g[0].shape[0]))
    assert g[0].shape[0] >= 8 and all(abs(m - m.T).nnz == 0 for m in g)
    assert gt.compute_perm([]) == []
    assert gt.compute_perm([np.array([0, 1, 0, 2])]) == [[0, 2, 1, 4, 3, 5], [0, 1, 2]]


def test_max_pool_local_semantics():
    """Host side of MaxPoolLocal: neighbourhood lists == the reference's, reverse lists consistent; the compute is HIP-only."""
    d = load(os.path.join(GOLDEN, "selgnn_cfg1_sbm100.npz"))
    S = d["S"]
    pool = gml.MaxPoolLocal(100, 10, 2)
    pool.addGSO(torch.tensor(S))
    nbh = gt.computeNeighborhood(S, 2, 10, 100, "list")
    got = pool.neighborhood.numpy()
    assert got.shape == (10, max(len(nb) for nb in nbh)) and pool.maxNeighborhoodSize == got.shape[1]
    for i, nb in enumerate(nbh):
        assert set(got[i].tolist()) == set(nb)               # padded with the node itself (graphTools 'matrix' output)
    rp, ri, rpos = pool._rev_ptr.numpy(), pool._rev_i.numpy(), pool._rev_p.numpy()
    assert rp[0] == 0 and rp[-1] == len(ri) == sum(len(set(nb)) for nb in nbh)
    for j in range(100):
        for q in range(rp[j], rp[j + 1]):
            assert got[ri[q], rpos[q]] == j and j not in got[ri[q], :rpos[q]]     # FIRST position of j in that list
    assert "neighborhood" not in pool.state_dict() and "_rev_ptr" not in pool.state_dict()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pool(torch.randn(3, 4, 100))


# ---- EdgeVariantGF host logic (pattern construction, parameter surface) -------------------------------------------
def test_edge_pattern_matches_reference_mask():
    """EdgePattern.from_gso == the oracle's pattern == the support of the reference's masked weights / gradients."""
    from alegnn_amd import EdgePattern
    from oracle import evgf_oracle as evo
    for name in ("evgf_asym37_hybrid", "evgf_asym_E2_hybrid", "evgf_asym37_full"):
        d = load(os.path.join(GOLDEN, name + ".npz"))
        S, M = d["S"], int(d["M"])
        for e in range(S.shape[0]):
            p = EdgePattern.from_gso(S[e], M)
            want = evo.ev_pattern(S[e], M)
            assert np.array_equal(p.indptr, want.indptr) and np.array_equal(p.indices, want.indices)
            support = np.any(d["dweightEV"][:, e, 1:] != 0, axis=(0, 1, 2))      # entries the reference trains
            mine = np.zeros_like(support)
            mine[p.rows, p.cols] = True
            assert not np.any(support & ~mine), "the reference trains an entry outside our pattern"
    L = _lib.lib()
    out = ctypes.c_void_p()
    rowptr = np.array([0, 2, 3], dtype=np.int32)
    col = np.array([1, 0, 1], dtype=np.int32)               # row 0 not ascending
    assert L.gf_ev_plan_create(2, 3, rowptr.ctypes.data, col.ctypes.data, ctypes.byref(out)) == _lib.GF_ERR_SHAPE
    assert b"ascending" in L.gf_last_error()
    assert L.gf_evgf_scratch_floats(16, 32, 32, 50000, 1) == 50000 * 16 * (64 + 32 + 2 * 1024)


def test_edge_variant_gf_surface_matches_reference():
    d = load(os.path.join(GOLDEN, "evgf_asym37_hybrid.npz"))
    F, E, K, G, N, _ = d["weightEV"].shape
    M = int(d["M"])
    layer = gml.EdgeVariantGF(G, F, K, M, N, E, True)
    assert {k: tuple(v.shape) for k, v in layer.state_dict().items()} == {
        "weightEV": (F, E, K, G, N, N), "weightLSI": (F, E, K, G), "bias": (F, 1)}    # graphML.py:2585-2595
    full = gml.EdgeVariantGF(G, F, K, N, N, E, False)
    assert full.weightLSI is None and full.bias is None
    assert "GSO stored" not in repr(layer).replace("no GSO stored", "")
    layer.addGSO(torch.tensor(d["S"]))
    assert "selected_nodes=%d" % M in repr(layer) and repr(layer).endswith("GSO stored)")
    with pytest.raises(AssertionError):
        layer.addGSO(torch.zeros(E + 1, N, N))              # graphML.py:2612
    # per-edge gather of the dense parameter == what the oracle extracts == what the reference's mask keeps
    from oracle import evgf_oracle as evo
    layer.load_state_dict({"weightEV": torch.tensor(d["weightEV"]), "weightLSI": torch.tensor(d["weightLSI"]),
                           "bias": torch.tensor(d["bias"])})
    ar, rows, cols, dmask = layer._indices(0, torch.device("cpu"))
    w = layer.weightEV[:, 0]
    wdiag, wedge = evo.ev_split_dense_weight(d["weightEV"][:, 0], evo.ev_pattern(d["S"][0], M), M)
    assert np.array_equal((w[:, 0][:, :, ar, ar] * dmask).detach().numpy(), wdiag.astype(np.float32))
    assert np.array_equal(w[:, 1:][:, :, :, rows, cols].detach().numpy(), wedge.astype(np.float32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(torch.zeros(2, G, N))
    sparse = gml.EdgeVariantGF(G, F, K, M, N, E, True, sparse=True)
    sparse.addGSO(torch.tensor(d["S"]))
    assert tuple(sparse.weightEVedges[0].shape) == (F, K - 1, G, layer._patterns[0].nnzp)


# ---- Model / Trainer / evaluate ----------------------------------------------------
def _mlp(N, nClasses=5):
    return torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(N, 16), torch.nn.Tanh(), torch.nn.Linear(16, nClasses))


def _trainer_model(d, archit, saveDir, name, trainer=None):
    from alegnn_amd.modules import evaluation, model, training
    archit.load_state_dict({k[5:]: torch.tensor(v) for k, v in d.items() if k.startswith("init:")})
    optim = torch.optim.Adam(archit.parameters(), lr=0.005, betas=(0.9, 0.999))
    return model.Model(archit, torch.nn.CrossEntropyLoss(), optim,
                       trainer or training.Trainer, evaluation.evaluate, 'cpu', name, saveDir)


def test_trainer_reproduces_reference_training_run(tmp_path):
    """Our Model + Trainer + evaluate against the reference's own run (tests/golden/make_golden.py trainer_case 'mlp'):
    same epoch permutations, uneven last batch, LR decay, early stopping, Best / Last checkpoints, evaluation."""
    import ast
    from _util import ArrayData
    d = load(os.path.join(GOLDEN, "trainer_mlp.npz"))
    m = _trainer_model(d, _mlp(d["S"].shape[1]).double(), str(tmp_path), "mlp")
    np.random.seed(int(d["seed"]) + 1)
    tv = m.train(ArrayData(d, torch.float64), int(d["nEpochs"]), int(d["batchSize"]), doSaveVars=True, printInterval=0,
                 **ast.literal_eval(str(d["trainKw"])))
    for k in ("lossTrain", "costTrain", "lossValid", "costValid"):
        assert tv[k].shape == d[k].shape, k                                       # early stopping ended at the same step
        assert np.allclose(tv[k], d[k], rtol=1e-9, atol=1e-12), k
    assert tv["batchSize"].tolist() == [40, 40, 16] and tv["batchIndex"].tolist() == [0, 40, 80, 96]
    ev = m.evaluate(ArrayData(d, torch.float64), doSaveVars=True)
    assert ev == {"costBest": float(d["costBest"]), "costLast": float(d["costLast"])}
    assert os.path.exists(tmp_path / "trainVars" / "mlptrainVars.pkl") and os.path.exists(tmp_path / "evalVars" / "mlpevalVars.pkl")
    assert m.getTrainingOptions()["nBatches"] == 3
    for label in ("Best", "Last"):                                   # the files we wrote == the files the reference wrote
        for part in ("Archit", "Optim"):
            ours = torch.load(tmp_path / "savedModels" / f"mlp{part}{label}.ckpt")
            ref = torch.load(os.path.join(GOLDEN, "ckpt", f"mlp{part}{label}.ckpt"))
            if part == "Archit":
                assert list(ours) == list(ref)
                for k in ref:
                    assert torch.allclose(ours[k], ref[k], rtol=1e-9, atol=1e-12), (label, k)
            else:
                assert ours["param_groups"] == ref["param_groups"] and list(ours["state"]) == list(ref["state"])
                for i in ref["state"]:
                    for k in ref["state"][i]:
                        assert torch.allclose(ours["state"][i][k].double(), ref["state"][i][k].double(), rtol=1e-9, atol=1e-14), (label, i, k)


def test_reference_checkpoints_load_into_rebuilt_selection_gnn(tmp_path):
    """Model.save files written by the reference (model.py:106-117) for its SelectionGNN load into ours via Model.load,
    and what Model.save writes back is readable with the reference's keys and values (round trip)."""
    d = load(os.path.join(GOLDEN, "trainer_selgnn.npz"))
    net = SelectionGNN([1, 8, 8], [3, 3], True, torch.nn.ReLU, [20, 10], gml.MaxPoolLocal, [2, 2], [5], d["S"][0]).double()
    m = _trainer_model(d, net, str(tmp_path), "selgnn")
    stem = os.path.join(GOLDEN, "ckpt", "selgnn")
    m.load(loadFiles=(stem + "ArchitBest.ckpt", stem + "OptimBest.ckpt"))
    ref_a, ref_o = torch.load(stem + "ArchitBest.ckpt"), torch.load(stem + "OptimBest.ckpt")
    for k, v in net.state_dict().items():
        assert torch.equal(v, ref_a[k]), k
    assert m.optim.state_dict()["param_groups"] == ref_o["param_groups"]
    m.save(label="RoundTrip")
    back_a = torch.load(tmp_path / "savedModels" / "selgnnArchitRoundTrip.ckpt")
    back_o = torch.load(tmp_path / "savedModels" / "selgnnOptimRoundTrip.ckpt")
    assert list(back_a) == list(ref_a) and all(torch.equal(back_a[k], ref_a[k]) for k in ref_a)
    assert all(torch.equal(back_o["state"][i][k], ref_o["state"][i][k]) for i in ref_o["state"] for k in ref_o["state"][i])
    m.load(label="RoundTrip")                                        # default path: <saveDir>/savedModels/<name>Archit<label>.ckpt


@pytest.mark.skipif(not os.path.isdir("/root/reference/alegnn"), reason="needs the reference checkout (build container only)")
def test_our_checkpoints_load_into_the_reference(tmp_path):
    """The other direction, run where the reference is present: the reference's own SelectionGNN + Model.load read files
    written by our Model.save."""
    import subprocess
    import sys
    d = load(os.path.join(GOLDEN, "trainer_selgnn.npz"))
    net = SelectionGNN([1, 8, 8], [3, 3], True, torch.nn.ReLU, [20, 10], gml.MaxPoolLocal, [2, 2], [5], d["S"][0]).double()
    m = _trainer_model(d, net, str(tmp_path), "ours")
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.125)
    m.optim.zero_grad()
    m.save(label="X")
    np.save(tmp_path / "S.npy", d["S"][0])
    code = f"""
import sys, types
for mod in ("hdf5storage", "gensim"):
    sys.modules[mod] = types.ModuleType(mod)
import numpy as np, scipy.sparse
np.int = int; np.float = float
sys.path.insert(0, "/root/reference")
import torch
torch.set_default_dtype(torch.float64)
import alegnn.utils.graphML as gml, alegnn.modules.architectures as archit, alegnn.modules.model as model
S = np.load(r"{tmp_path}/S.npy")
net = archit.SelectionGNN([1, 8, 8], [3, 3], True, torch.nn.ReLU, [20, 10], gml.MaxPoolLocal, [2, 2], [5], S)
opt = torch.optim.Adam(net.parameters(), lr=0.005)
m = model.Model(net, None, opt, None, None, 'cpu', 'ours', r"{tmp_path}")
m.load(label='X')
print("SUM", float(sum(p.sum() for p in net.parameters())))
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    want = float(sum(p.sum() for p in net.parameters()))
    got = float(out.stdout.strip().split("SUM")[-1])
    assert abs(got - want) < 1e-9 * max(1.0, abs(want))


def test_local_gnn_surface_matches_reference():
    from alegnn_amd.modules.architectures import LocalGNN
    d = load(os.path.join(GOLDEN, "selgnn_cfg1_sbm100.npz"))
    S = d["S"][0]
    net = LocalGNN([1, 8, 4], [3, 2], True, torch.nn.ReLU, [100, 100], gml.NoPool, [1, 1], [6, 1], S, order="Degree")
    assert list(net.state_dict().keys()) == ["GFL.0.weight", "GFL.0.bias", "GFL.3.weight", "GFL.3.bias",
                                             "Readout.0.weight", "Readout.0.bias", "Readout.2.weight", "Readout.2.bias"]
    assert tuple(net.Readout[0].weight.shape) == (6, 4) and not hasattr(net, "MLP")
    g = dict(np.load(os.path.join(GOLDEN, "graphtools_sbm100.npz")))
    assert list(net.order) == list(g["order_Degree"])
    with pytest.raises(AssertionError):
        net.singleNodeForward(torch.zeros(2, 1, 100), "3")          # architectures.py:1132-1134


@pytest.mark.parametrize("name", ["asym_E2_M6", "ring_M2", "sbm100_M10_Nin60", "asym37_M37"])
def test_node_variant_gf_surface_matches_reference(name):
    d = load(os.path.join(GOLDEN, f"nvgf_{name}.npz"))
    F, E, K, G, M = d["weight"].shape
    layer = gml.NodeVariantGF(G, F, K, M, E, True)
    assert list(layer.state_dict().keys()) == ["weight", "bias"] and tuple(layer.weight.shape) == (F, E, K, G, M)
    layer.addGSO(torch.tensor(d["S"]))
    assert layer.copyNodes.tolist() == d["copyNodes"].tolist()                     # graphML.py:2411-2468
    layer.addGSO([sp.csr_matrix(d["S"][e]) for e in range(E)])                    # sparse GSO in: same assignment
    assert layer.copyNodes.tolist() == d["copyNodes"].tolist()
    assert "node_taps=%d" % M in layer.extra_repr()


def test_graph_recurrent_surface_matches_reference():
    from alegnn_amd.modules.architectures import GraphRecurrentNN
    d = load(os.path.join(GOLDEN, "grnnarch_sbm100.npz"))
    net = GraphRecurrentNN(3, 6, 8, [3, 2], True, torch.tanh, torch.tanh, torch.nn.ReLU, [5, 2], d["S"][0])
    ref = {k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")}
    assert list(net.state_dict().keys()) == list(ref.keys())
    net.load_state_dict(ref, strict=True)
    hs = gml.HiddenState(3, 8, 3)
    assert [tuple(p.shape) for p in hs.parameters()] == [(8, 1, 3, 3), (8, 1, 3, 8), (8, 1), (8, 1)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):       # edge gating runs on the HIP path (gf_db.hip) like everything else
        gml.GatedGRNN(hs.aWeights, hs.bWeights, sp.identity(4, format="csr"), torch.zeros(2, 3, 3, 4), torch.zeros(2, 8, 4),
                      torch.tanh, q_hat=torch.ones(2, 3, 1, 4, 4))
    with pytest.raises(AssertionError):                              # an edge gate must be B x T x 1 x N x N (graphML.py:1372-1373)
        gml.GatedGRNN(hs.aWeights, hs.bWeights, sp.identity(4, format="csr"), torch.zeros(2, 3, 3, 4), torch.zeros(2, 8, 4),
                      torch.tanh, q_hat=torch.ones(2, 3, 1, 4, 5))
    eg = gml.EdgeGatedHiddenState(3, 8, 3)
    eg.addGSO(torch.eye(4).reshape(1, 4, 4))
    assert [k for k in eg.state_dict() if "GAT" in k] == ["inputGateGAT.mixer", "inputGateGAT.weight", "forgetGateGAT.mixer", "forgetGateGAT.weight"]
    assert tuple(eg.inputGateGAT.mixer.shape) == (1, 1, 2) and tuple(eg.inputGateGAT.weight.shape) == (1, 1, 1, 8)   # GraphAttentional(H,1,1)
    db = gml.HiddenState_DB(3, 8, 3, E=2)
    assert [tuple(p.shape) for p in db.parameters()] == [(8, 2, 3, 3), (8, 2, 3, 8), (8, 1), (8, 1)]
    with pytest.raises(AssertionError):
        db.addGSO(torch.zeros(2, 5, 1, 4, 4))                        # E mismatch (graphML.py:3521)
    gfdb = gml.GraphFilter_DB(3, 6, 4, 2)
    assert tuple(gfdb.weight.shape) == (6, 2, 4, 3) and tuple(gfdb.bias.shape) == (6, 1) and "no GSO stored" in gfdb.extra_repr()


def test_node_variant_gnn_surface_matches_reference():
    from alegnn_amd.modules.architectures import NodeVariantGNN
    d = load(os.path.join(GOLDEN, "nvgnn_sbm100.npz"))
    net = NodeVariantGNN([2, 8, 8], [3, 2], [10, 5], True, torch.nn.ReLU, [40, 10], gml.MaxPoolLocal, [2, 2], [4], d["S"][0])
    ref = {k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")}
    assert list(net.state_dict().keys()) == list(ref.keys())
    net.load_state_dict(ref, strict=True)
    assert net.N == [100, 40, 10] and isinstance(net.NVGFL[0], gml.NodeVariantGF) and net.NVGFL[3].M == 5


@pytest.mark.skipif(not os.path.isdir("/root/reference/alegnn"), reason="needs the reference checkout (build container only)")
def test_install_rebinds_the_reference_symbols():
    """alegnn_amd.install(gml): the reference's own architectures then build on the HIP layers (construction only here -- the
    forward needs a GPU; tests/test_gpu_parity.py runs the same layers against the reference's outputs)."""
    import subprocess
    import sys
    code = """
import sys, types
for mod in ("hdf5storage", "gensim"):
    sys.modules[mod] = types.ModuleType(mod)
import numpy as np, scipy.sparse
np.int = int; np.float = float
sys.path[:0] = ["/root/reference", %r]
import torch
import alegnn.utils.graphML as gml, alegnn.modules.architectures as archit
import alegnn_amd
from alegnn_amd.utils import graphML as amd
alegnn_amd.install(gml)
S = np.eye(12, k=1) + np.eye(12, k=-1)
net = archit.SelectionGNN([1, 8], [3], True, torch.nn.ReLU, [6], gml.NoPool, [1], [2], S)
assert type(net.GFL[0]) is amd.GraphFilter
nv = archit.NodeVariantGNN([1, 8], [3], [4], True, torch.nn.ReLU, [12], gml.NoPool, [1], [2], S)
assert type(nv.NVGFL[0]) is amd.NodeVariantGF and nv.NVGFL[0].copyNodes.shape[0] == 12
rn = archit.GraphRecurrentNN(2, 3, 4, [3, 2], True, torch.tanh, torch.tanh, torch.nn.ReLU, [2], S)
assert type(rn.hiddenState) is amd.HiddenState and type(rn.outputState) is amd.GraphFilter
assert gml.LSIGF is amd.LSIGF and gml.NVGF is amd.NVGF and gml.GatedGRNN is amd.GatedGRNN
print("OK")
""" % os.path.join(ROOT, "graph-neural-networks_amd")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


def test_trainer_single_node_uses_label_ids(tmp_path):
    """TrainerSingleNode / evaluateSingleNode (training.py:580-714, evaluation.py:91-168): the loss is taken at one target node
    per sample, looked up through data.getLabelID(split[, indices]) and archit.singleNodeForward(x, ids)."""
    from alegnn_amd.modules import evaluation, model, training

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(1, 1).double()

        def forward(self, x):                                        # B x 1 x N -> B x 1 x N
            return self.lin(x.permute(0, 2, 1)).permute(0, 2, 1)

        def singleNodeForward(self, x, nodes):
            y = self.forward(x)
            return y[torch.arange(x.shape[0]), :, torch.as_tensor(nodes)]

    class Data:
        def __init__(self):
            g = torch.Generator().manual_seed(3)
            self.x = {s: torch.randn(n, 1, 6, generator=g).double() for s, n in (("train", 10), ("valid", 4), ("test", 4))}
            self.ids = {s: torch.randint(0, 6, (v.shape[0],), generator=g).numpy() for s, v in self.x.items()}
            self.y = {s: 2.0 * v[torch.arange(v.shape[0]), 0, torch.as_tensor(self.ids[s])] + 1.0 for s, v in self.x.items()}
            self.nTrain = 10

        def getSamples(self, split, *a):
            return (self.x[split][a[0]], self.y[split][a[0]]) if a else (self.x[split], self.y[split])

        def getLabelID(self, split, *a):
            return self.ids[split][a[0]] if a else self.ids[split]

        def evaluate(self, yHat, y):
            return torch.sqrt(torch.mean((yHat.squeeze(-1) - y) ** 2))

    net, data = Net(), Data()
    optim = torch.optim.SGD(net.parameters(), lr=0.05)
    m = model.Model(net, lambda est, tgt: torch.nn.functional.mse_loss(est.squeeze(1), tgt), optim, training.TrainerSingleNode,
                    evaluation.evaluateSingleNode, 'cpu', 'sn', str(tmp_path))
    np.random.seed(0)
    tv = m.train(data, 40, 5, printInterval=0, doSaveVars=False)
    assert tv["lossTrain"][-1] < 0.05 * tv["lossTrain"][0]          # y = 2 x[target] + 1 is learnt only if the right node is read
    ev = m.evaluate(data, doSaveVars=False)
    assert ev["costBest"] < 0.2 and set(ev) == {"costBest", "costLast"}
    with pytest.raises(AssertionError):                              # an architecture without singleNodeForward is refused
        training.TrainerSingleNode(types.SimpleNamespace(archit=torch.nn.Linear(1, 1)), data, 1, 5)


def test_trainer_hip_graph_option_is_inert_off_gpu(tmp_path):
    """hipGraph=True only changes how the step is launched on a HIP device; on the CPU (the gloo tests, this container) the trainer
    runs the eager step and retraces the reference run all the same."""
    import ast
    from _util import ArrayData
    d = load(os.path.join(GOLDEN, "trainer_mlp.npz"))
    m = _trainer_model(d, _mlp(d["S"].shape[1]).double(), str(tmp_path), "mlp")
    np.random.seed(int(d["seed"]) + 1)
    tv = m.train(ArrayData(d, torch.float64), int(d["nEpochs"]), int(d["batchSize"]), doSaveVars=False, printInterval=0,
                 hipGraph=True, **ast.literal_eval(str(d["trainKw"])))
    assert m.trainer.useGraph and len(m.trainer._graphs) == 0
    assert np.allclose(tv["lossTrain"], d["lossTrain"], rtol=1e-9, atol=1e-12)


# ---- callers of LSIGF whose own logic is host-side composition ------------------------------------------
def _cpu_lsigf(h, S, x, b=None, activation=None):
    """The oracle's dense LSIGF behind alegnn_amd's call signature: lets the HOST logic of jARMA / the gated hidden-state modules be
    checked against the reference's goldens on a box without a GPU (the HIP kernels under the same calls are checked by -m gpu)."""
    from oracle import lsigf_oracle as orc
    gso = SparseGSO.from_any(S)
    N, nin = gso.N, x.shape[2]
    xp = torch.nn.functional.pad(x, (0, N - nin))
    y = orc.lsigf_dense(h, gso.to_dense(x.dtype), xp, b)[:, :, :nin]
    return torch.relu(y) if activation == "relu" else y


@pytest.mark.parametrize("path", sorted(__import__("glob").glob(os.path.join(GOLDEN, "jarma_*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_jarma_host_logic_reproduces_reference(path, monkeypatch):
    monkeypatch.setattr(gml, "LSIGF", _cpu_lsigf)
    d = load(path)
    t = {k: torch.tensor(d[k], requires_grad=True) for k in ("psi", "varphi", "phi", "x")}
    b = torch.tensor(d["b"], requires_grad=True) if "b" in d else None
    y = gml.jARMA(t["psi"], t["varphi"], t["phi"], torch.tensor(d["S"]), t["x"], b, tMax=int(d["tMax"]))
    (y * torch.tensor(d["dy"])).sum().backward()
    assert np.abs(y.detach().numpy() - d["y"]).max() < 1e-10 * np.abs(d["y"]).max()
    for k, v in t.items():
        assert np.abs(v.grad.numpy() - d["d" + k]).max() < 1e-9 * np.abs(d["d" + k]).max(), k
    if b is not None:
        assert np.abs(b.grad.numpy() - d["db"]).max() < 1e-9 * np.abs(d["db"]).max()


@pytest.mark.parametrize("path", sorted(__import__("glob").glob(os.path.join(GOLDEN, "gatedhs_*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_gated_hidden_state_host_logic_reproduces_reference(path, monkeypatch):
    monkeypatch.setattr(gml, "LSIGF", _cpu_lsigf)
    d = load(path)
    F, H, K, E = (int(v) for v in d["dims"])
    cls = gml.TimeGatedHiddenState if str(d["kind"]) == "time" else gml.NodeGatedHiddenState
    layer = cls(F, H, K, nonlinearity=torch.tanh, E=E, bias=True).double()
    layer.addGSO(torch.tensor(d["S"]))
    sd = {k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")}
    assert set(sd) == set(layer.state_dict())                       # the reference's checkpoint keys, exactly
    layer.load_state_dict(sd)
    x, z0 = torch.tensor(d["x"], requires_grad=True), torch.tensor(d["z0"], requires_grad=True)
    z, zT = layer(x, z0)
    assert list(zT.shape) == d["zT_shape"].tolist()
    (z * torch.tensor(d["dz"])).sum().backward()
    assert np.abs(z.detach().numpy() - d["z"]).max() < 1e-10
    assert np.abs(x.grad.numpy() - d["dx"]).max() < 1e-9 * max(1.0, np.abs(d["dx"]).max())
    for k, p in layer.named_parameters():
        assert np.abs(p.grad.numpy() - d["grad:" + k]).max() < 1e-9 * max(1.0, np.abs(d["grad:" + k]).max()), k


def test_gso_objects_copy_and_pickle():
    """Reference modules can be deep-copied and pickled (copy.deepcopy(model), torch.save(model)); the GSO holders carry ctypes plan
    handles and a finalizer, so they travel as host CSR and rebuild their device plans on first use."""
    import copy
    import pickle
    A = sp.random(50, 50, density=0.1, format="csr", random_state=np.random.RandomState(0))
    layer = gml.GraphFilter(4, 8, 3)
    layer.addGSO(A)
    layer._gso._plans[0] = ("fake-handle-array", [])                # as if device plans existed (no GPU here)
    try:
        for clone in (copy.deepcopy(layer), pickle.loads(pickle.dumps(layer))):
            assert clone._gso is not layer._gso and clone._gso._plans == {}
            assert (clone._gso.mats[0] != layer._gso.mats[0]).nnz == 0
            assert torch.equal(clone.weight, layer.weight)
    finally:
        layer._gso._plans.clear()
    ev = gml.EdgeVariantGF(2, 3, 2, 50, 50, 1, True, sparse=True)
    ev.addGSO(A)
    clone = pickle.loads(pickle.dumps(ev))
    assert clone._patterns[0].nnzp == ev._patterns[0].nnzp and clone._patterns[0]._plans == {}


def test_edge_variant_gnn_has_the_reference_surface():
    """Constructor, sub-module names and checkpoint keys of archit.EdgeVariantGNN (architectures.py:1721-1955)."""
    from alegnn_amd.modules.architectures import EdgeVariantGNN
    d = load(os.path.join(GOLDEN, "evgnn_asym37.npz"))
    net = EdgeVariantGNN([2, 4, 4], [3, 2], [20, 10], True, torch.nn.ReLU, [20, 10], gml.MaxPoolLocal, [2, 2], [3], d["S"][0])
    sd = {k[3:]: torch.tensor(v) for k, v in d.items() if k.startswith("sd:")}
    assert set(sd) == set(net.state_dict())
    net.double().load_state_dict(sd)
    assert net.N == [37, 20, 10] and net.EVGFL[0].M == 20 and net.EVGFL[3].M == 10 and net.EVGFL[3].N == 37
    with pytest.raises(AssertionError):
        EdgeVariantGNN([2, 4, 4], [3], [20, 10], True, torch.nn.ReLU, [20, 10], gml.MaxPoolLocal, [2, 2], [3], d["S"][0])



@pytest.mark.parametrize("path", golden_files("attention"), ids=case_id)
def test_learn_attention_gso_host_logic(path):
    """learnAttentionGSO (graphML.py:640-737) is pure host logic (elementwise + softmax on the [N,N] support): runs on the CPU and
    must reproduce the reference's coefficients and gradients exactly (float64)."""
    from alegnn_amd.utils import graphML as amd_gml
    d = load(path)
    x, a, W = (torch.tensor(d[k], requires_grad=True) for k in ("x", "a", "W"))
    q = amd_gml.learnAttentionGSO(x, a, W, torch.tensor(d["S"]))
    (q * torch.tensor(d["dq"])).sum().backward()
    assert relerr(q.detach().numpy(), d["q"]) < 1e-12
    for t, k in ((x, "dx"), (a, "da"), (W, "dW")):
        assert relerr(t.grad.numpy(), d[k]) < 1e-11, k


def test_locality_groups_are_cached_per_pattern_on_disk(tmp_path, monkeypatch):
    """gf_plan_create clusters the rows of large graphs into locality groups (host work, ~1 s at N = 4e4).  With GFHIP_PLAN_CACHE_DIR
    set the labels are stored once per sparsity pattern (groups_<hash>_<n>_<nnz>.bin: header + one int32 per row) and a damaged file
    is recomputed, never trusted.  (No GPU here: the call fails at the first device allocation, AFTER the host-side clustering.)"""
    import ctypes
    import time
    from alegnn_amd import _lib, graphgen
    monkeypatch.setenv("GFHIP_PLAN_CACHE_DIR", str(tmp_path))
    n = 40000
    A = graphgen.er(n, avg_degree=4.0, seed=5)
    rowptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    col = np.ascontiguousarray(A.indices, dtype=np.int32)
    val = np.ascontiguousarray(A.data)
    L = _lib.lib()

    def create():
        out = ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = L.gf_plan_create(n, A.nnz, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, int(val.dtype == np.float64), 0, ctypes.byref(out))
        dt = time.perf_counter() - t0
        if rc == 0:
            L.gf_plan_destroy(out)
        return rc, dt
    rc, t_first = create()
    files = list(tmp_path.iterdir())
    assert len(files) == 1 and files[0].name.startswith("groups_") and files[0].name.endswith(f"_{n}_{A.nnz}.bin")
    assert files[0].stat().st_size == 8 + 4 * n
    labels = np.fromfile(files[0], dtype=np.int32)
    assert labels[0] == n and labels[1] >= 2 and labels[2:].min() >= 0 and labels[2:].max() < labels[1]
    assert np.bincount(labels[2:]).max() <= 1.04 * n / labels[1] + 2          # balanced groups
    files[0].write_bytes(files[0].read_bytes()[: 8 + 4 * (n // 2)])          # truncate: must be recomputed and rewritten
    other = graphgen.er(n, avg_degree=4.0, seed=6)                            # another pattern -> another file
    out = ctypes.c_void_p()
    L.gf_plan_create(n, other.nnz, np.ascontiguousarray(other.indptr, dtype=np.int32).ctypes.data,
                     np.ascontiguousarray(other.indices, dtype=np.int32).ctypes.data, np.ascontiguousarray(other.data).ctypes.data,
                     int(other.data.dtype == np.float64), 0, ctypes.byref(out))
    assert len(list(tmp_path.iterdir())) == 2


def test_msweep_image_on_cpu(tmp_path):
    """The MSWEEP image of round 5 (csrc/gf_msweep_image.h: groups of four destination rows behind a (set, position) of a wave, their
    (source, slot) entries spread over the rounds in source order) is pure host code: tools/msweep_image_check.cpp builds it for random
    graphs, interprets it the way spmm_msweep_kernel executes it (one fmaf per entry, round by round) and compares bit for bit with the
    row-by-row product in ascending column order (graphML.py:158-161 per batch entry)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    exe = str(tmp_path / "msweep_image_check")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "graph-neural-networks_amd", "csrc"),
                        os.path.join(ROOT, "tools", "msweep_image_check.cpp"), "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout[-3000:]


def test_asm_stores_never_read_mfma_results():
    """The contraction / one-pass-backward kernels store through inline asm, which the compiler's hazard recogniser does not look into:
    the XDL-write -> VMEM-read wait states are only guaranteed while every stored value passes through a compiler-visible VALU
    instruction first.  tools/check_contract_isa.py compiles both files to ISA and checks the last writer of every asm store's data."""
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_contract_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_msweep_isa_register_contract():
    """spmm_msweep_kernel allocates its registers by hand (v24-v255, a0-a255 belong to the asm body and keep state between the passes of a
    launch): no instantiation may spill or use scratch, every one gets the whole register file, and no compiler-emitted instruction
    may name a register of the body (tools/check_msweep_isa.py compiles the file to ISA and scans it)."""
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_msweep_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 violations" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_fused_chain_can_be_switched_off_in_the_environment():
    """GFHIP_MSWEEP_FUSE=0 is the product-mode opt-out of the one-launch chain (gf_khop then runs one launch per hop): gf_msweep_status reports
    fusion_on = 0 in such a process and 1 otherwise; no knob (gf_tune) is involved."""
    import subprocess
    code = ("import sys; sys.path[:0] = [%r, %r]; import ctypes; from alegnn_amd import _lib; L = _lib.lib(); "
            "f = ctypes.c_uint32(7); on = ctypes.c_int32(7); assert L.gf_msweep_status(ctypes.byref(f), ctypes.byref(on)) == 0; print(f.value, on.value)"
            % (ROOT, os.path.join(ROOT, "graph-neural-networks_amd")))
    for env_val, want in (("0", "0 0"), ("1", "0 1"), (None, "0 1")):
        env = {k: v for k, v in os.environ.items() if k not in ("GFHIP_MSWEEP_FUSE", "GFHIP_EXPERIMENTS")}
        if env_val is not None:
            env["GFHIP_MSWEEP_FUSE"] = env_val
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and r.stdout.strip() == want, (env_val, r.stdout, r.stderr[-2000:])


def test_no_device_trap_or_abort_in_the_library_sources():
    """A training job must not die on a kernel-side trap: conditions the kernels rely on are checked by the host
    before the launch and come back as a status + gf_last_error() (e.g. gf_require_no_static_lds for the LDS-absolute panel kernels), run-time
    conditions take the abandon-and-repair path (gf_msweep.hip)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "graph-neural-networks_amd", "csrc")
    hits = []
    for f in sorted(glob.glob(os.path.join(csrc, "*"))):
        for i, line in enumerate(open(f, errors="replace"), 1):
            code = line.split("//")[0]
            if re.search(r"__builtin_trap|\babort\s*\(|\b__assert_fail\b|\bassert\s*\(", code):
                hits.append(f"{os.path.basename(f)}:{i}: {line.strip()[:100]}")
    assert not hits, hits
