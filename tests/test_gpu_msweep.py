"""spmm_msweep_kernel (round 5: source sweep with a batch entry's partial sums in the XCD's registers and an fp32 multi-block MFMA as
scatter-accumulate, gf_msweep_image.h / gf_msweep.hip) against scipy in float64 and BIT FOR BIT against the SELL-8 kernel (same per-row
summation order: ascending columns, one fmaf per entry).  Reference lines: the hop `x = torch.matmul(x, S)` of graphML.py:158-161."""
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
    tune(spmm_algo=0, spmm_bar=0, spmm_slack=5, spmm_group=1, spmm_pfd=0, spmm_census=0, spmm_tmo_ms=0, spmm_fuse=1, spmm_status_reset=1, spmm_xlayout=1,
         spmm_minwork=5)


def hop(plans, op, Xt, algo, **kw):
    tune(spmm_algo=algo, **kw)
    B, n, W = Xt.shape
    out = torch.full((B, n, W), float("nan"), device=DEV)
    _lib.check(_lib.lib().gf_spmm_hop(plans[0], op, Xt.data_ptr(), out.data_ptr(), B, W, stream()))
    torch.cuda.synchronize()
    return out


def er(n, deg, seed, directed=False, weighted=False):
    rng = np.random.RandomState(seed)
    r = np.repeat(np.arange(n), deg)
    c = rng.randint(0, n, size=r.size)
    A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(n, n))
    if not directed:
        A = A + A.T
    A = (A > 0).astype(np.float64)
    A.setdiag(0)
    A.eliminate_zeros()
    A = sp.lil_matrix(A)
    A[3, :] = 0                                 # an empty row (and, undirected, a sparse column)
    A = sp.csr_matrix(A)
    A.eliminate_zeros()
    if weighted:
        A.data = rng.uniform(-1.0, 1.0, size=A.data.size)
        return sp.csr_matrix(A)
    return sp.csr_matrix(A * 0.0625)


@pytest.mark.parametrize("bar", [1, 0])
@pytest.mark.parametrize("n,deg,B,directed,weighted", [
    (100000, 5, 24, False, False),      # config 4's shape: 25 sets per wave, three entries per XCD
    (100000, 5, 9, False, True),        # weighted: the value stream
    (40000, 5, 17, True, False),        # 10 sets, directed (forward and backward images differ)
    (60000, 4, 40, False, False),       # 15 sets
    (81000, 3, 8, False, True),         # 20 sets, weighted
    (102400, 3, 11, False, False),      # the largest single-pass graph
    (130000, 3, 9, False, False),       # two passes per batch entry (each sweeps the sources for half of the rows)
    (204800, 2, 8, False, True),        # the largest graph with an image
    (33000, 6, 8, True, True),
])
def test_msweep_hop_against_scipy_and_bitwise_against_sell(n, deg, B, directed, weighted, bar, knobs):
    A = er(n, deg, seed=n + B, directed=directed, weighted=weighted)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    rng = np.random.RandomState(1)
    X = rng.randn(B, n, 32).astype(np.float32)
    Xt = torch.tensor(X, device=DEV)
    for op, M in ((0, A.T.tocsr()), (1, A)):
        ref = hop(plans, op, Xt, 3)
        want = np.stack([M.astype(np.float64) @ X[b].astype(np.float64) for b in range(min(B, 4))])
        assert relerr(ref[:4].cpu().numpy(), want) < 2e-6
        for rep in range(3):                    # (launches in a row rotate through the barrier-counter slots)
            got = hop(plans, op, Xt, 5, spmm_bar=bar)
            assert torch.equal(got, ref), (op, n, B, bar, rep, int((got != ref).sum()), float((got - ref).abs().max()))


def test_msweep_hop_soak_at_config4_size(knobs):
    """Hand-scheduled registers, counted s_waitcnt, gathers in flight across the whole round loop: 20 launches at the bench size
    (N = 1e5, ~1e6 entries, 64 batch entries: 8 per XCD back to back) must all reproduce SELL-8's bits, with and without the barrier."""
    A = er(100000, 5, seed=11)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    Xt = torch.randn(64, 100000, 32, device=DEV)
    ref = hop(plans, 0, Xt, 3)
    for rep in range(20):
        got = hop(plans, 0, Xt, 5, spmm_bar=rep & 1)
        bad = int((got != ref).sum())
        assert bad == 0, (rep, bad)


@pytest.mark.parametrize("n,deg,B,K,weighted", [(60000, 4, 16, 4, False), (100000, 5, 24, 5, False), (52000, 5, 9, 3, True), (140000, 3, 8, 3, False)])
def test_msweep_fused_khop_chain_is_bitwise_the_per_hop_launches(n, deg, B, K, weighted, knobs):
    """gf_khop with the MFMA sweep runs the K-1 hops of an edge feature in ONE launch, batch entry by batch entry: hop h gathers the rows
    hop h-1 of the same launch stored (ordered by vmcnt(0) + the XCD barrier).  The tap stack must equal, bit for bit, the one the
    per-hop SELL-8 launches build -- in the default configuration (spmm_algo = 0: these graphs are above the default threshold), with
    the fusion off, and for both orientations."""
    A = er(n, deg, seed=n + K, weighted=weighted)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    L = _lib.lib()
    x0 = torch.randn(B, n, 32, device=DEV)

    def chain(op, **kw):
        tune(**kw)
        Z = torch.full((K, B, n, 32), float("nan"), device=DEV)
        Z[0].copy_(x0)
        _lib.check(L.gf_khop(plans, 1, op, Z.data_ptr(), B, 32, K, stream()))
        torch.cuda.synchronize()
        return Z

    for op in (0, 1):
        ref = chain(op, spmm_algo=3)
        assert L.gf_spmm_hop_kernel(plans[0], op, B, 32) == 0
        for kw in (dict(spmm_algo=0, spmm_fuse=1), dict(spmm_algo=0, spmm_fuse=0), dict(spmm_algo=5, spmm_fuse=1, spmm_bar=1)):
            for rep in range(2):
                got = chain(op, **kw)
                assert L.gf_spmm_hop_kernel(plans[0], op, B, 32) == 1
                assert torch.equal(got, ref), (op, kw, rep, int((got != ref).sum()))
    tune(spmm_fuse=1)


def test_msweep_chain_replays_from_a_hip_graph(knobs):
    """The trainer's hipGraph mode captures whole steps: the fused chain (ONE cooperative launch; its XCD barrier resets itself, so there
    is no state to zero between replays -- a version with a memset node in front replayed wrong from the second replay on) must be
    capturable and replay to the same bits."""
    n, B, K = 60000, 16, 4
    A = er(n, 4, seed=21)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    L = _lib.lib()
    tune(spmm_algo=0)
    Z = torch.full((K, B, n, 32), float("nan"), device=DEV)
    Z[0].normal_()
    _lib.check(L.gf_khop(plans, 1, 0, Z.data_ptr(), B, 32, K, stream()))
    torch.cuda.synchronize()
    ref = Z.clone()
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        _lib.check(L.gf_khop(plans, 1, 0, Z.data_ptr(), B, 32, K, side.cuda_stream))
    torch.cuda.current_stream(DEV).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _lib.check(L.gf_khop(plans, 1, 0, Z.data_ptr(), B, 32, K, stream()))
    for rep in range(3):
        Z[1:].fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(Z, ref), rep


def test_msweep_is_refused_where_it_does_not_apply(knobs):
    """spmm_algo = 5 never falls back silently: other widths, small batches and graphs without an image are errors."""
    A = er(40000, 5, seed=5)
    gso = SparseGSO([A])                                      # (the plans live as long as their SparseGSO)
    plans = gso.plans(DEV)
    L = _lib.lib()
    tune(spmm_algo=5)
    rcs = []
    for n, pl, B, W in ((40000, plans, 8, 16), (40000, plans, 4, 32)):
        X = torch.randn(B, n, W, device=DEV)
        Y = torch.empty_like(X)
        rcs.append(int(L.gf_spmm_hop(pl[0], 0, X.data_ptr(), Y.data_ptr(), B, W, stream())))
    gso_small = SparseGSO([er(12000, 5, seed=6)])             # below kMsMinNodes: no image
    small = gso_small.plans(DEV)
    X = torch.randn(8, 12000, 32, device=DEV)
    Y = torch.empty_like(X)
    rcs.append(int(L.gf_spmm_hop(small[0], 0, X.data_ptr(), Y.data_ptr(), 8, 32, stream())))
    torch.cuda.synchronize()
    assert all(rc != 0 for rc in rcs), rcs


def chain_status():
    import ctypes
    f, on = ctypes.c_uint32(0), ctypes.c_int32(0)
    _lib.check(_lib.lib().gf_msweep_status(ctypes.byref(f), ctypes.byref(on)))
    return int(f.value), int(on.value)


def khop_chain(plans, x0, K, op=0, **kw):
    tune(**kw)
    B, n, W = x0.shape
    Z = torch.full((K, B, n, W), float("nan"), device=DEV)
    Z[0].copy_(x0)
    _lib.check(_lib.lib().gf_khop(plans, 1, op, Z.data_ptr(), B, W, K, stream()))
    torch.cuda.synchronize()
    return Z


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("weighted", [False, True])
def test_fused_chain_with_a_bad_census_is_repaired_not_trapped(mode, weighted, knobs):
    """The fused chain takes its teams from the hardware (XCC id + a census of the launch's workgroups).  A census that does not show 32
    workgroups on each of 8 XCCs -- forced here: the last arriver calls it bad (1), or one workgroup claims the neighbouring XCC so that
    the counts come out 33 / 31 (2) -- must abandon the launch before its first store and leave the work to the repair kernel behind it:
    no trap, the same bits as SELL-8, the reason reported through gf_msweep_status, and the host stops fusing once it has seen it."""
    n, B, K = 60000, 16, 4
    A = er(n, 4, seed=31 + mode, weighted=weighted)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    x0 = torch.randn(B, n, 32, device=DEV)
    for op in (0, 1):
        ref = khop_chain(plans, x0, K, op, spmm_algo=3)
        tune(spmm_status_reset=1)
        assert chain_status() == (0, 1)
        ok = khop_chain(plans, x0, K, op, spmm_algo=0, spmm_fuse=1, spmm_census=0)
        assert torch.equal(ok, ref) and chain_status() == (0, 1)
        got = khop_chain(plans, x0, K, op, spmm_algo=0, spmm_fuse=1, spmm_census=mode)
        assert torch.equal(got, ref), (op, mode, int((got != ref).sum()))
        assert chain_status() == (1, 0)                     # census failure seen; fusion now off in this process
        again = khop_chain(plans, x0, K, op, spmm_census=0)   # ... so this call runs one launch per hop (still the sweep kernel)
        assert torch.equal(again, ref)
        tune(spmm_status_reset=1)
        back = khop_chain(plans, x0, K, op, spmm_census=0)    # and after the reset the same slot fuses again (the census words reset themselves)
        assert torch.equal(back, ref) and chain_status() == (0, 1)


def test_fused_chain_that_never_becomes_resident_runs_into_its_time_limit_and_is_repaired(knobs):
    """One workgroup of the cooperative grid never arrives at the census (forced): the others give up after the time limit (30 ms here,
    2 s in the product), poison the launch slot and leave; the repair kernel computes the chain.  Later launches on the poisoned slot are
    abandoned at once (and repaired) until the host has seen the report and stopped fusing."""
    n, B, K = 52000, 9, 3
    A = er(n, 5, seed=77)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    x0 = torch.randn(B, n, 32, device=DEV)
    ref = khop_chain(plans, x0, K, spmm_algo=3)
    tune(spmm_status_reset=1)
    got = khop_chain(plans, x0, K, spmm_algo=0, spmm_fuse=1, spmm_census=3, spmm_tmo_ms=30)
    assert torch.equal(got, ref), int((got != ref).sum())
    assert chain_status() == (2, 0)
    tune(spmm_status_reset=1)                               # the host forgets, the slot does not: still repaired, still the same bits
    got = khop_chain(plans, x0, K, spmm_census=0, spmm_tmo_ms=0)
    assert torch.equal(got, ref) and chain_status()[0] == 2


def test_fused_chain_beside_kernels_that_hold_compute_units(knobs):
    """The cooperative launch needs every CU; here other streams hold a few of them for ~20 ms each while the chain is launched.  The grid
    then becomes resident late (well inside the time limit): same bits, nothing to repair."""
    n, B, K = 60000, 16, 4
    A = er(n, 4, seed=5)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    x0 = torch.randn(B, n, 32, device=DEV)
    ref = khop_chain(plans, x0, K, spmm_algo=3)
    tune(spmm_algo=0, spmm_fuse=1, spmm_status_reset=1)
    side = [torch.cuda.Stream(device=DEV) for _ in range(8)]
    for rep in range(3):
        for s_ in side:
            with torch.cuda.stream(s_):
                torch.cuda._sleep(40_000_000)               # one spinning wave per stream
        got = khop_chain(plans, x0, K)
        assert torch.equal(got, ref), (rep, int((got != ref).sum()))
    assert chain_status() == (0, 1)


@pytest.mark.parametrize("n,deg,B,W,K,directed,weighted", [
    (100000, 5, 8, 64, 3, False, False),      # two slabs per entry: 16 (entry, slab) chains on 8 XCDs
    (60000, 4, 5, 64, 4, True, True),         # B < 8 but B * W / 32 >= 8; directed, weighted, 15 sets
    (52000, 5, 3, 96, 3, False, False),       # three slabs (a row is 384 bytes: not a power of two)
    (81000, 3, 4, 128, 3, False, True),       # four slabs, 20 sets, weighted
    (130000, 3, 5, 64, 3, False, False),      # two passes per (entry, slab)
])
def test_msweep_wide_rows_are_slabs_of_the_same_image(n, deg, B, W, K, directed, weighted, knobs):
    """W = 64 / 96 / 128 (the reference's layers are any width: graphML.py:2086-2107): a row is W / 32 slabs of 32 columns, the sweep runs the
    SAME image once per (batch entry, slab) with {row, column} addressing (buffer resource with the row as stride).  Single hops and the fused
    chain against SELL-8 of the same width, bit for bit, both orientations; the default configuration must pick the sweep."""
    A = er(n, deg, seed=n + B + W, directed=directed, weighted=weighted)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    L = _lib.lib()
    x0 = torch.randn(B, n, W, device=DEV)
    for op, M in ((0, A.T.tocsr()), (1, A)):
        ref1 = hop(plans, op, x0, 3)
        want = M.astype(np.float64) @ x0[0].cpu().numpy().astype(np.float64)
        assert relerr(ref1[0].cpu().numpy(), want) < 2e-6
        got1 = hop(plans, op, x0, 5)
        assert torch.equal(got1, ref1), (op, int((got1 != ref1).sum()), float((got1 - ref1).abs().max()))
        ref = khop_chain(plans, x0, K, op, spmm_algo=3)
        for kw in (dict(spmm_algo=0, spmm_fuse=1), dict(spmm_algo=0, spmm_fuse=0), dict(spmm_algo=5, spmm_fuse=1, spmm_bar=1)):
            got = khop_chain(plans, x0, K, op, **kw)
            assert L.gf_spmm_hop_kernel(plans[0], op, B, W) == 1
            assert torch.equal(got, ref), (op, kw, int((got != ref).sum()))
    # the abandon-and-repair path at this width
    tune(spmm_status_reset=1)
    got = khop_chain(plans, x0, K, 0, spmm_algo=0, spmm_fuse=1, spmm_census=1)
    assert torch.equal(got, khop_chain(plans, x0, K, 0, spmm_algo=3, spmm_census=0)) and chain_status()[0] == 1


def powerlaw(n, m, seed, weighted=False):
    """Row lengths with the tail of a Barabasi-Albert graph, P(length > k) = (m / k)^2, columns uniform: a few rows of hundreds to thousands of entries."""
    rng = np.random.RandomState(seed)
    deg = np.minimum(n // 8, (m / np.sqrt(np.maximum(rng.uniform(size=n), 1e-9))).astype(np.int64))
    deg[5] = 0
    r = np.repeat(np.arange(n), deg)
    c = rng.randint(0, n, size=r.size)
    A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(n, n))
    A.sum_duplicates()
    A.data[:] = rng.uniform(-1.0, 1.0, size=A.data.size) if weighted else 0.0625
    return sp.csr_matrix(A)


def msweep_info(plan, op):
    import ctypes
    out = (ctypes.c_int32 * 8)()
    _lib.check(_lib.lib().gf_debug_msweep_info(plan, op, out))
    return dict(zip(("sets", "passes", "rounds", "fill1000", "hub_rows", "hub_split_rows", "hub_limit", "hub_split"), list(out)))


@pytest.mark.parametrize("n,m,B,K,weighted,W", [(100000, 5, 16, 4, False, 32), (60000, 4, 9, 3, True, 32), (120000, 3, 8, 3, False, 32),
                                               (60000, 4, 5, 3, False, 64), (52000, 3, 3, 3, True, 96)])
def test_msweep_hub_rows_are_computed_outside_the_groups(n, m, B, K, weighted, W, knobs):
    """A power-law graph: rows far longer than a group's share would set the number of rounds every wave walks, so the image leaves them out of
    the groups and each wave computes its share of them from the CSR between its store phase and the hand-over -- the same ascending-column fmaf
    chain, so every row that is not SPLIT is bit for bit SELL-8's; the few rows longer than the split limit are summed as 32 partial chains +
    a fixed tree (relative 1e-6).  Orientation 1 (rows of S) has the hubs, orientation 0 (rows of S^T: Poisson lengths) none; single hops and
    the fused chain; 32-column rows and wide rows (slabs)."""
    A = powerlaw(n, m, seed=n + m, weighted=weighted)
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    info = msweep_info(plans[0], 1)
    assert info["sets"] > 0 and info["hub_rows"] > 0 and info["hub_split_rows"] > 0, info
    deg = np.diff(A.indptr)
    unsplit = torch.tensor(deg <= info["hub_split"], device=DEV)
    x0 = torch.randn(B, n, W, device=DEV)
    L = _lib.lib()
    for op in (1, 0):
        assert L.gf_spmm_hop_kernel(plans[0], op, B, W) == 1
        ref1 = hop(plans, op, x0, 3)
        got1 = hop(plans, op, x0, 5)
        rows = unsplit if op == 1 else torch.ones_like(unsplit)
        assert torch.equal(got1[:, rows], ref1[:, rows]), (op, int((got1[:, rows] != ref1[:, rows]).sum()))
        assert float((got1 - ref1).abs().max()) <= 1e-6 * float(ref1.abs().max()) * 8
        ref = khop_chain(plans, x0, K, op, spmm_algo=3)
        for kw in (dict(spmm_algo=0, spmm_fuse=1), dict(spmm_algo=0, spmm_fuse=0)):
            got = khop_chain(plans, x0, K, op, **kw)
            if op == 0:
                assert torch.equal(got, ref), (op, kw)
            else:                                          # (a split row's rounding difference propagates through the later hops)
                assert torch.equal(got[1][:, rows], ref[1][:, rows])
                assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), (op, kw)
        det = khop_chain(plans, x0, K, op, spmm_algo=0, spmm_fuse=1)
        assert torch.equal(det, khop_chain(plans, x0, K, op, spmm_algo=0, spmm_fuse=1))      # run-to-run: bit for bit (fixed tree)


@pytest.mark.parametrize("graph,n,B,K,Nin,act", [("er", 60000, 16, 4, 60000, None), ("er", 100000, 9, 3, 99000, "relu"), ("erw", 52000, 8, 3, 52000, "relu"),
                                                ("powerlaw", 100000, 8, 3, 100000, None)])
def test_layout_pass_inside_the_fused_chain_is_bitwise_the_separate_kernel(graph, n, B, K, Nin, act, knobs):
    """Round 6: on large graphs (32-column rows, one edge feature) the boundary layout pass -- x [B,G,Nin] -> node-major tap 0, zero rows beyond
    Nin (graphML.py:2131-2135), in the backward dy masked by the fused ReLU -- is a pre-phase of the fused chain launch: each XCD writes an entry's
    tap 0 right before it walks the entry.  A pure data movement: the whole layer (y, dx, dh, db through the C ABI's gf_lsigf_forward / _backward,
    plain and with the fused ReLU, Nin < N, a graph with hub rows) must come out bit for bit as with the separate layout kernel
    (spmm_xlayout = 0), also when the launch is abandoned and repaired."""
    from alegnn_amd import functional as F_
    A = powerlaw(n, 5, seed=3) if graph == "powerlaw" else er(n, 4, seed=n, weighted=graph == "erw")
    gso = SparseGSO([A])
    torch.manual_seed(7)
    h = (torch.randn(32, 1, K, 32, device=DEV) * 0.1).requires_grad_()
    bias = (torch.randn(32, 1, device=DEV) * 0.1).requires_grad_()
    x = torch.randn(B, 32, Nin, device=DEV, requires_grad=True)
    dy = torch.randn(B, 32, Nin, device=DEV)

    def run(**kw):
        tune(**kw)
        for t in (h, bias, x):
            t.grad = None
        y = F_.LSIGF(h, gso, x, bias, activation=act)
        y.backward(dy)
        torch.cuda.synchronize()
        return [t.detach().clone() for t in (y, x.grad, h.grad, bias.grad)]

    ref = run(spmm_algo=0, spmm_fuse=1, spmm_xlayout=0)
    assert _lib.lib().gf_spmm_hop_kernel(gso.plans(DEV)[0], 0, B, 32) == 1
    for rep in range(2):
        got = run(spmm_xlayout=1)
        for a, r, name in zip(got, ref, ("y", "dx", "dh", "db")):
            assert torch.equal(a, r), (name, rep, int((a != r).sum()))
    tune(spmm_status_reset=1)
    got = run(spmm_xlayout=1, spmm_census=1)                # abandoned launches: the repair kernel lays tap 0 out itself
    assert chain_status()[0] == 1
    for a, r, name in zip(got, ref, ("y", "dx", "dh", "db")):
        assert torch.equal(a, r), ("repaired", name, int((a != r).sum()))


def test_two_edge_features_on_a_large_graph_are_bitwise_sell(knobs):
    """E = 2 (graphML.py:154: tap 0 shared, K - 1 taps per edge feature): two fused chains per layer call, each on its own plan and gate; the
    whole layer (y, dx, dh, db) bit for bit what the SELL-8 hops give."""
    from alegnn_amd import functional as F_
    n = 60000
    gso = SparseGSO([er(n, 4, seed=1), er(n, 4, seed=2)])
    torch.manual_seed(0)
    h = (torch.randn(32, 2, 3, 32, device=DEV) * 0.1).requires_grad_()
    b = torch.zeros(32, 1, device=DEV, requires_grad=True)
    x = torch.randn(8, 32, n, device=DEV, requires_grad=True)
    dy = torch.randn(8, 32, n, device=DEV)
    outs = []
    for algo in (3, 0):
        tune(spmm_algo=algo)
        for t in (h, b, x):
            t.grad = None
        y = F_.LSIGF(h, gso, x, b)
        y.backward(dy)
        torch.cuda.synchronize()
        outs.append([t.detach().clone() for t in (y, x.grad, h.grad, b.grad)])
    assert _lib.lib().gf_spmm_hop_kernel(gso.plans(DEV)[0], 0, 8, 32) == 1
    for a, r, name in zip(outs[1], outs[0], ("y", "dx", "dh", "db")):
        assert torch.equal(a, r), name


@pytest.mark.parametrize("graph,n,B,W,K", [("er", 100000, 5, 32, 4), ("erw", 60000, 7, 32, 3), ("er", 66000, 3, 64, 3), ("powerlaw", 60000, 6, 32, 3),
                                           ("er", 52000, 2, 96, 3)])
def test_work_lists_of_five_to_seven_pairs_run_as_the_sweep(graph, n, B, W, K, knobs):
    """Round 6: the sweep takes work lists from 5 (batch entry, 32-column slab) pairs on (an XCD walks a pair in the same time alone or beside seven
    others; from 5 pairs that beats SELL-8: profiles/r06_l_share/); XCDs beyond the list leave at once.  Default path: the fused chain and the
    hop-by-hop launches bit for bit SELL-8 (rows the image splits: 1e-6), a repaired launch too; 4 pairs stay with SELL-8."""
    A = powerlaw(n, 5, seed=n) if graph == "powerlaw" else er(n, 4, seed=n + B, weighted=graph == "erw")
    gso = SparseGSO([A])
    plans = gso.plans(DEV)
    L = _lib.lib()
    x0 = torch.randn(B, n, W, device=DEV)
    assert L.gf_spmm_hop_kernel(plans[0], 0, 4 // (W // 32), W) == 0
    for op in (0, 1):
        assert L.gf_spmm_hop_kernel(plans[0], op, B, W) == 1
        sell = khop_chain(plans, x0, K, op, spmm_algo=3)
        for kw in (dict(spmm_fuse=1), dict(spmm_fuse=0), dict(spmm_fuse=1, spmm_census=1)):
            tune(spmm_status_reset=1)
            got = khop_chain(plans, x0, K, op, spmm_algo=0, **kw)
            assert chain_status()[0] == (1 if "spmm_census" in kw else 0)
            if msweep_info(plans[0], op)["hub_split_rows"] == 0:
                assert torch.equal(got, sell), (op, kw, int((got != sell).sum()))
            else:
                assert float((got - sell).abs().max()) <= 1e-5 * float(sell.abs().max()), (op, kw)
        tune(spmm_census=0, spmm_status_reset=1)
