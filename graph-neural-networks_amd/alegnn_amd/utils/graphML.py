"""Graph-filter layers with the reference's Python surface, backed by the gfx950 HIP library.

Drop-in for the symbols SelectionGNN binds from alegnn/utils/graphML.py:
    LSIGF          (graphML.py:83-176)     -> alegnn_amd.functional.LSIGF   (HIP)
    GraphFilter    (graphML.py:2036-2155)  -> same ctor, attributes, parameter names/shapes, addGSO/forward/extra_repr
    EdgeVariantGF  (graphML.py:2511-2712)  -> same ctor / parameters (dense weightEV), or per-edge storage (sparse=True)
    GatedGRNN      (graphML.py:1292-1527)  hidden-state recursion, one LSIGF over all B*T inputs + one per time step
    HiddenState    (graphML.py:3540-3681)  the module around it (GraphRecurrentNN's recurrent layer)
    TimeGatedHiddenState / NodeGatedHiddenState (graphML.py:3683-4031)  the gated recurrent layers (GatedGCRNNforRegression & co.)
    jARMA          (graphML.py:490-638)    ARMA filter by Jacobi iterations: sparse hops instead of dense [F,E,P,G,N,N] products
    NVGF           (graphML.py:293-387)    -> alegnn_amd.functional.NVGF  (HIP: LSIGF's tap stack, per-node bank)
    NodeVariantGF  (graphML.py:2317-2509)  the module around it (NodeVariantGNN's layer)
    LSIGF_DB / GRNN_DB / GraphFilter_DB / HiddenState_DB (graphML.py:977-1290, 3278-3538)  per-sample, delayed GSOs (gf_db.hip)
    EdgeGatedHiddenState (graphML.py:4033-4208) and the edge-gated branches of GatedGRNN (:1394-1419, :1434-1456)
    NoPool         (graphML.py:1850-1888)  identity pooling
    MaxPoolLocal   (graphML.py:1890-2028)  alpha-hop neighbourhood max, keep the first nOutputNodes nodes
Checkpoints are interchangeable with the reference: ``weight [F,E,K,G]``, ``bias [F,1]``; the GSO is a plain attribute
(not in the state_dict), exactly as at graphML.py:2099-2123.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from ..functional import EVGF_edges, LSIGF, NVGF, expand_node_taps, max_pool_local
from ..functional_db import GRNN_DB, LSIGF_DB, filter_per_sample
from ..gso import EdgePattern, SparseGSO
from . import graphTools

__all__ = ["LSIGF", "GraphFilter", "EdgeVariantGF", "NoPool", "MaxPoolLocal", "FusedReLU", "GatedGRNN", "HiddenState", "TimeGatedHiddenState",
           "NodeGatedHiddenState", "EdgeGatedHiddenState", "jARMA", "NVGF", "NodeVariantGF", "LSIGF_DB", "GRNN_DB", "GraphFilter_DB",
           "HiddenState_DB", "learnAttentionGSO", "GraphAttentional"]


class FusedReLU(nn.Identity):
    """Placeholder that keeps SelectionGNN's ``GFL`` layout [GraphFilter, sigma, rho] (and therefore the state_dict keys
    ``GFL.3.weight`` ...) when sigma = ReLU has been fused into the preceding GraphFilter's epilogue."""


class GraphFilter(nn.Module):
    """GraphFilter(in_features, out_features, filter_taps, edge_features=1, bias=True) -- graphML.py:2086-2107.

    ``addGSO`` accepts the reference's dense ``[E,N,N]`` tensor and, as a superset, a ``SparseGSO`` / scipy sparse
    matrix / list of them (mandatory once the dense tensor no longer fits: N = 1e5 is 40 GB dense).
    """

    def __init__(self, G, F, K, E=1, bias=True):
        super().__init__()
        self.G = G
        self.F = F
        self.K = K
        self.E = E
        self.S = None                      # no GSO assigned yet (graphML.py:2099)
        self._gso = None
        self.fused_activation = None       # 'relu': y = max(0, filter(x) + b) in one pass (set by SelectionGNN)
        self.weight = nn.parameter.Parameter(torch.Tensor(F, E, K, G))
        if bias:
            self.bias = nn.parameter.Parameter(torch.Tensor(F, 1))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.G * self.K)                  # graphML.py:2109-2114
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def addGSO(self, S):
        if sp.issparse(S) or isinstance(S, (list, tuple)):       # superset: sparse GSO, one matrix per edge feature
            S = SparseGSO.from_any(S)
        assert len(S.shape) == 3                                 # graphML.py:2118
        assert S.shape[0] == self.E                              # :2120
        self.N = S.shape[1]
        assert S.shape[2] == self.N                              # :2122
        self.S = S
        self._gso = SparseGSO.from_any(S)

    def forward(self, x):
        # x: batchSize x dimInFeatures x numberNodesIn (numberNodesIn <= N: zero-padded, graphML.py:2131-2135;
        # the output keeps numberNodesIn nodes, :2142-2143 -- both happen inside the HIP path, no copies here)
        assert self._gso is not None, "GraphFilter.forward called before addGSO"
        assert x.dim() == 3 and x.shape[2] <= self.N
        return LSIGF(self.weight, self._gso, x, self.bias, activation=self.fused_activation)

    def extra_repr(self):
        reprString = "in_features=%d, out_features=%d, " % (self.G, self.F) + "filter_taps=%d, " % (self.K) + \
                     "edge_features=%d, " % (self.E) + "bias=%s, " % (self.bias is not None)
        reprString += "GSO stored" if self.S is not None else "no GSO stored"
        return reprString


class NodeVariantGF(nn.Module):
    """NodeVariantGF(in_features, out_features, shift_taps, node_taps, edge_features=1, bias=True) -- graphML.py:2369-2393.
    Parameters ``weight [F,E,K,G,M]`` and ``bias [F,1]`` as in the reference.  The M node taps are spread over the N nodes by
    ``copyNodes`` (:2411-2468): node n >= M copies the taps of the lowest-numbered node < M in its 1-hop neighbourhood (wider
    neighbourhoods while some node has none)."""

    def __init__(self, G, F, K, M, E=1, bias=True):
        super().__init__()
        self.G = G
        self.F = F
        self.K = K
        self.M = M
        self.E = E
        self.S = None
        self._gso = None
        self.weight = nn.parameter.Parameter(torch.Tensor(F, E, K, G, M))
        if bias:
            self.bias = nn.parameter.Parameter(torch.Tensor(F, 1))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.G * self.K * self.M)             # :2395-2400
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def addGSO(self, S):
        if sp.issparse(S) or isinstance(S, (list, tuple)):
            S = SparseGSO.from_any(S)
        assert len(S.shape) == 3 and S.shape[0] == self.E           # :2404-2406
        self.N = S.shape[1]
        assert S.shape[2] == self.N
        self.S = S
        self._gso = SparseGSO.from_any(S)
        if self.M < self.N:
            hops = 1
            nbh = graphTools.computeNeighborhood(self._gso.mats, hops, nb=self.M)
            while any(len(v) == 0 for v in nbh):                    # :2425-2449: widen until every node sees a tap node
                hops += 1
                assert hops <= self.N, "NodeVariantGF.addGSO: some node is not connected to any of the first M nodes"
                wider = graphTools.computeNeighborhood(self._gso.mats, hops, nb=self.M)
                nbh = [wider[n] if len(v) == 0 else v for n, v in enumerate(nbh)]
            copyNodes = list(range(self.M)) + [min(nbh[m]) for m in range(self.M, self.N)]
            self.copyNodes = torch.tensor(copyNodes)
        else:
            self.copyNodes = torch.arange(min(self.M, self.N))      # :2461-2468
        # the nodes sharing each tap node, as CSR (the weight gradient is a gather over these groups)
        cn = self.copyNodes.numpy()
        order = np.argsort(cn, kind="stable")
        self._grp_idx = torch.tensor(order.astype(np.int32))
        self._grp_ptr = torch.tensor(np.concatenate(([0], np.cumsum(np.bincount(cn, minlength=self.M)))).astype(np.int32))

    def forward(self, x):
        assert self._gso is not None, "NodeVariantGF.forward called before addGSO"
        assert x.dim() == 3 and x.shape[2] <= self.N
        if self.M == self.N:
            h = self.weight                                         # :2482-2485
        else:
            if self.copyNodes.device != self.weight.device:
                dev = self.weight.device
                self.copyNodes, self._grp_ptr, self._grp_idx = self.copyNodes.to(dev), self._grp_ptr.to(dev), self._grp_idx.to(dev)
            h = expand_node_taps(self.weight, self.copyNodes, self._grp_ptr, self._grp_idx)
        return NVGF(h, self._gso, x, self.bias)                     # padding / truncation to Nin happen in the HIP path

    def extra_repr(self):
        reprString = "in_features=%d, out_features=%d, " % (self.G, self.F) + "shift_taps=%d, node_taps=%d, " % (
            self.K, self.M) + "edge_features=%d, " % (self.E) + "bias=%s, " % (self.bias is not None)
        reprString += "GSO stored" if self.S is not None else "no GSO stored"
        return reprString


class EdgeVariantGF(nn.Module):
    """EdgeVariantGF(in_features, out_features, shift_taps, selected_nodes, number_nodes, edge_features=1, bias=True)
    -- graphML.py:2574-2597.  Hybrid edge-variant filter: nodes < M get an edge-variant filter, the rest an LSI filter.

    sparse=False (default): parameters exactly as the reference -- ``weightEV [F,E,K,G,N,N]`` (only on-pattern entries are
    ever used or trained, graphML.py:2676), ``weightLSI [F,E,K,G]`` if M < N, ``bias [F,1]`` -- so checkpoints interchange.
    The HIP path never multiplies the dense tensor: the on-pattern entries are gathered (a view-like index, autograd
    scatters the gradient back) and the filter runs on per-edge storage.
    sparse=True (superset, mandatory when N^2 no longer fits: N = 5e4 is 3e13 bytes dense): the parameters ARE the
    per-edge storage, ``weightEVdiag [F,E,G,N]`` and ``weightEVedges[e] [F,K-1,G,nnzp_e]``, created by ``addGSO``.
    """

    def __init__(self, G, F, K, M, N, E=1, bias=True, sparse=False):
        super().__init__()
        self.G = G
        self.F = F
        self.K = K
        self.E = E
        self.M = M                         # number of selected nodes
        self.N = N                         # total number of nodes
        self.S = None
        self.sparse = bool(sparse)
        self._gso = None
        self._patterns = None
        if not self.sparse:
            self.weightEV = nn.parameter.Parameter(torch.Tensor(F, E, K, G, N, N))
        else:
            self.weightEVdiag = nn.parameter.Parameter(torch.Tensor(F, E, G, N))
            self.weightEVedges = nn.ParameterList()            # filled by addGSO (the pattern sizes them)
        if self.M < self.N:
            self.weightLSI = nn.parameter.Parameter(torch.Tensor(F, E, K, G))
        else:
            self.register_parameter('weightLSI', None)
        if bias:
            self.bias = nn.parameter.Parameter(torch.Tensor(F, 1))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.G * self.K * self.N)          # graphML.py:2599-2606
        if not self.sparse:
            self.weightEV.data.uniform_(-stdv, stdv)
        else:
            self.weightEVdiag.data.uniform_(-stdv, stdv)
            for w in self.weightEVedges:
                w.data.uniform_(-stdv, stdv)
        if self.weightLSI is not None:
            self.weightLSI.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def addGSO(self, S):
        if sp.issparse(S) or isinstance(S, (list, tuple)):
            S = SparseGSO.from_any(S)
        assert len(S.shape) == 3                                 # graphML.py:2610
        assert S.shape[0] == self.E                              # :2612
        self.N = S.shape[1]
        assert S.shape[2] == self.N                              # :2614
        self.S = S
        self._gso = SparseGSO.from_any(S)
        # pattern of taps k >= 1: (|S_e| + I > 1e-9) & hybrid mask (:2617-2643); tap 0: identity & mask -> nodes < M (:2653)
        self._patterns = [EdgePattern.from_gso(m, self.M) for m in self._gso.mats]
        self._index_cache = {}
        if self.sparse:
            dev, dt = self.weightEVdiag.device, self.weightEVdiag.dtype
            assert self.weightEVdiag.shape[3] == self.N
            sizes = [tuple(w.shape) for w in self.weightEVedges]
            want = [(self.F, self.K - 1, self.G, p.nnzp) for p in self._patterns]
            if sizes != want:                                    # first GSO, or a GSO with a different pattern
                stdv = 1. / math.sqrt(self.G * self.K * self.N)
                self.weightEVedges = nn.ParameterList(
                    [nn.parameter.Parameter(torch.empty(w, device=dev, dtype=dt).uniform_(-stdv, stdv)) for w in want])

    def _indices(self, e, device):
        key = (e, str(device))
        hit = self._index_cache.get(key)
        if hit is None:
            p = self._patterns[e]
            hit = (torch.arange(self.N, device=device), torch.from_numpy(p.rows).to(device), torch.from_numpy(p.cols).to(device),
                   (torch.arange(self.N, device=device) < self.M).to(torch.float32))
            self._index_cache[key] = hit
        return hit

    def forward(self, x):
        assert self._patterns is not None, "EdgeVariantGF.forward called before addGSO"
        assert x.dim() == 3 and x.shape[2] <= self.N
        u = None
        for e in range(self.E):
            ar, rows, cols, dmask = self._indices(e, x.device)
            if not self.sparse:
                w = self.weightEV[:, e]                                            # [F,K,G,N,N]
                wdiag = w[:, 0][:, :, ar, ar] * dmask                              # Phi_0: identity & hybrid mask
                wedge = w[:, 1:][:, :, :, rows, cols]                              # Phi_k on the pattern, k >= 1
            else:
                wdiag = self.weightEVdiag[:, e] * dmask
                wedge = self.weightEVedges[e]
            ue = EVGF_edges(self._patterns[e], wdiag, wedge, x, self.bias if e == 0 else None)   # bias once (:486-487)
            u = ue if u is None else u + ue
        if self.M < self.N:
            u = u + LSIGF(self.weightLSI, self._gso, x, self.bias)                 # bias a second time (:2686)
        return u

    def extra_repr(self):
        reprString = "in_features=%d, out_features=%d, " % (self.G, self.F) + "shift_taps=%d, " % (self.K) + \
                     "selected_nodes=%d, " % (self.M) + "number_nodes=%d, " % (self.N) + \
                     "edge_features=%d, " % (self.E) + "bias=%s, " % (self.bias is not None)
        reprString += "GSO stored" if self.S is not None else "no GSO stored"
        return reprString


class NoPool(nn.Module):
    """Pooling layer that does nothing, with MaxPoolLocal's interface -- graphML.py:1850-1888."""

    def __init__(self, nInputNodes, nOutputNodes, nHops):
        super().__init__()
        self.nInputNodes = nInputNodes
        self.nOutputNodes = nOutputNodes
        self.nHops = nHops
        self.neighborhood = None

    def addGSO(self, GSO):
        pass

    def forward(self, x):
        assert x.shape[2] == self.nInputNodes
        assert x.shape[2] >= self.nOutputNodes
        return x

    def extra_repr(self):
        return "in_dim=%d, out_dim=%d, number_hops = %d, no neighborhood needed" % (
            self.nInputNodes, self.nOutputNodes, self.nHops)


class MaxPoolLocal(nn.Module):
    """MaxPoolLocal(in_dim, out_dim, number_hops): v[b,f,n] = max over the number_hops-hop neighbourhood of node n
    (nodes < in_dim only), for the first out_dim nodes -- graphML.py:1890-2028.

    The neighbourhood index matrix is a non-persistent buffer: it follows ``.to(device)`` on its own (the reference
    recomputes it on the CPU inside every ``.to()``, architectures.py:477-479) and stays out of the state_dict.
    """

    def __init__(self, nInputNodes, nOutputNodes, nHops):
        super().__init__()
        self.nInputNodes = nInputNodes
        self.nOutputNodes = nOutputNodes
        self.nHops = nHops
        for name in ("neighborhood", "_rev_ptr", "_rev_i", "_rev_p"):
            self.register_buffer(name, None, persistent=False)

    def addGSO(self, S):
        assert len(S.shape) == 3                                 # graphML.py:1942
        self.N = S.shape[1]
        assert S.shape[2] == self.N
        device = self.neighborhood.device if self.neighborhood is not None else getattr(S, "device", None)
        if isinstance(S, SparseGSO):
            pattern_src = S.mats
        elif isinstance(S, torch.Tensor):
            pattern_src = np.array(S.detach().cpu())
        else:
            pattern_src = S
        nbh = graphTools.computeNeighborhood(pattern_src, self.nHops, self.nOutputNodes, self.nInputNodes, 'matrix')
        nbh = np.ascontiguousarray(nbh, dtype=np.int32)
        assert nbh.shape[0] == self.nOutputNodes                 # graphML.py:1962-1963
        assert int(nbh.max()) <= self.nInputNodes
        self.maxNeighborhoodSize = nbh.shape[1]
        # reverse lists for the backward gather: input node j -> (output node i, FIRST position of j in nbh[i])
        rows, pos, cols = [], [], []
        for i in range(nbh.shape[0]):
            seen = set()
            for p_, j in enumerate(nbh[i]):
                if j not in seen:
                    seen.add(int(j))
                    cols.append(int(j)); rows.append(i); pos.append(p_)
        order = np.lexsort((np.asarray(rows), np.asarray(cols)))          # by input node, then ascending output node
        cols_s = np.asarray(cols, dtype=np.int64)[order]
        rev_ptr = np.zeros(self.nInputNodes + 1, dtype=np.int32)
        np.add.at(rev_ptr, cols_s + 1, 1)
        rev_ptr = np.cumsum(rev_ptr).astype(np.int32)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32))
        bufs = dict(neighborhood=t(nbh), _rev_ptr=t(rev_ptr), _rev_i=t(np.asarray(rows)[order]), _rev_p=t(np.asarray(pos)[order]))
        for k, v in bufs.items():
            setattr(self, k, v.to(device) if device is not None else v)

    def forward(self, x):
        assert x.shape[2] == self.nInputNodes                    # graphML.py:1972-1976
        assert x.shape[2] >= self.nOutputNodes
        # one pass over x on the HIP path (the reference repeats x maxNeighborhood times, gathers and reduces: :2003-2018)
        return max_pool_local(x, self.neighborhood, self._rev_ptr, self._rev_i, self._rev_p)

    def extra_repr(self):
        reprString = "in_dim=%d, out_dim=%d, number_hops = %d, " % (self.nInputNodes, self.nOutputNodes, self.nHops)
        reprString += "neighborhood stored" if self.neighborhood is not None else "NO neighborhood stored"
        return reprString


# ---------------------------------------------------------------------------------------------------------------
# Graph recurrent layer: LSIGF per time step
# ---------------------------------------------------------------------------------------------------------------
def _check_gate(q, B, T, N, name):
    assert q.shape[0] == B or q.shape[0] == 1                       # graphML.py:1367, :1374
    if q.dim() > 1:
        assert q.shape[1] == T and q.shape[2] == 1 and (q.shape[3] == 1 or q.shape[3] == N), name
    if q.dim() > 4:
        assert q.shape[4] == N, name                                # :1372-1373 / :1379-1380  edge gate: B x T x 1 x N x N


def _dense_on(gso, like):
    """The GSO as the reference's dense [E,N,N] tensor on the device of `like` (edge gating multiplies it entrywise, :1397-1399)."""
    key = (str(like.device), like.dtype)
    cache = gso.__dict__.setdefault("_dense_dev", {})
    if key not in cache:
        cache[key] = gso.to_dense(like.dtype).to(like.device)
    return cache[key]


def GatedGRNN(a, b, S, x, z0, sigma, q_hat=None, q_check=None, xBias=None, zBias=None):
    """Hidden states z_t = sigma(q_hat_t * A(S) x_t + q_check_t * B(S) z_{t-1}), t = 1..T -- graphML.py:1292-1527.

    a [H,E,K,F] input-to-hidden taps, b [H,E,K,H] hidden-to-hidden taps, S the GSO (dense [E,N,N] or SparseGSO),
    x [B,T,F,N], z0 [B,H,N]; gates: None / ones(1) (no gating), [B|1,T,1,1] (time), [B|1,T,1,N] (node) or [B,T,1,N,N] (edge);
    xBias / zBias [H,1].  Returns z [B,T,H,N].
    Time / node gating: A(S)x for all B*T inputs is ONE LSIGF call (:1389-1391), the recursion one LSIGF call per time step (:1424).
    Edge gating (:1394-1419, :1434-1456): the gate multiplies the GSO entrywise, so every (b, t) has its own operator q * S_e; the
    filter with a per-sample operator is one HIP call for all B*T inputs (A) and one per time step (B) -- the reference multiplies all
    B*T signals with all B*T operators and keeps the diagonal (:1407-1413)."""
    H, E, K, F = a.shape
    assert b.shape[0] == H and b.shape[1] == E and b.shape[2] == K and b.shape[3] == H      # :1354-1357
    gso = SparseGSO.from_any(S)
    N = gso.N
    B, T = x.shape[0], x.shape[1]
    assert x.shape[2] == F and x.shape[3] == N                       # :1362-1363
    assert z0.shape[0] == B and z0.shape[1] == H and z0.shape[2] == N
    if q_hat is not None:
        _check_gate(q_hat, B, T, N, "q_hat")
    if q_check is not None:
        _check_gate(q_check, B, T, N, "q_check")
    edge_hat = q_hat is not None and q_hat.dim() > 4
    edge_check = q_check is not None and q_check.dim() > 4
    Sd = _dense_on(gso, x) if (edge_hat or edge_check) else None
    if edge_hat:
        edgeS = q_hat.reshape(B, T, E, N, N) * Sd                   # :1395-1399  (the reference's reshape needs E == 1 as well)
        Ax = filter_per_sample(a, edgeS, x, xBias)                  # :1400-1419  B x T x H x N
    else:
        Ax = LSIGF(a, gso, x.reshape(B * T, F, N), xBias).reshape(B, T, H, N)
        if q_hat is not None:
            Ax = q_hat * Ax                                         # :1392
    zt = z0
    states = []
    for t in range(T):
        if edge_check:
            edgeS = (q_check[:, t] * Sd).reshape(B, 1, E, N, N)     # :1437-1443  this step's gate: B x 1 x N x N
            Bz = filter_per_sample(b, edgeS, zt.reshape(B, 1, H, N), zBias).reshape(B, H, N)     # :1444-1459
        else:
            Bz = LSIGF(b, gso, zt.reshape(B, H, N), zBias)          # :1424
            if q_check is not None:                                 # :1426-1432  [B|1,1,1|N] broadcasts over H
                Bz = (q_check[:, t] if q_check.dim() > 1 else q_check) * Bz
        zt = sigma(Ax[:, t] + Bz)                                   # :1462
        states.append(zt)
    return torch.stack(states, dim=1)                               # B x T x H x N


class HiddenState(nn.Module):
    """HiddenState(signal_features, hidden_features, filter_taps, nonlinearity=torch.tanh, edge_features=1, bias=True)
    -- graphML.py:3540-3681.  Parameters ``aWeights [H,E,K,F]``, ``bWeights [H,E,K,H]``, ``xBias``/``zBias [H,1]``
    (same names and shapes: reference checkpoints load).  forward(x [B,T,F,N], z0 [B,H,N]) -> (z [B,T,H,N], z_T [B,1,H,N])."""

    def __init__(self, F, H, K, nonlinearity=torch.tanh, E=1, bias=True):
        super().__init__()
        self.F = F
        self.H = H
        self.K = K
        self.E = E
        self.S = None
        self._gso = None
        self.bias = bias
        self.sigma = nonlinearity
        self.aWeights = nn.parameter.Parameter(torch.Tensor(H, E, K, F))
        self.bWeights = nn.parameter.Parameter(torch.Tensor(H, E, K, H))
        if self.bias:
            self.xBias = nn.parameter.Parameter(torch.Tensor(H, 1))
            self.zBias = nn.parameter.Parameter(torch.Tensor(H, 1))
        else:
            self.register_parameter('xBias', None)
            self.register_parameter('zBias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.F * self.K)                      # graphML.py:3614-3620
        self.aWeights.data.uniform_(-stdv, stdv)
        self.bWeights.data.uniform_(-stdv, stdv)
        if self.bias:
            self.xBias.data.uniform_(-stdv, stdv)
            self.zBias.data.uniform_(-stdv, stdv)

    def addGSO(self, S):
        if sp.issparse(S) or isinstance(S, (list, tuple)):
            S = SparseGSO.from_any(S)
        assert len(S.shape) == 3 and S.shape[0] == self.E           # :3657-3659
        self.N = S.shape[1]
        assert S.shape[2] == self.N
        self.S = S
        self._gso = SparseGSO.from_any(S)

    def forward(self, x, z0):
        assert self._gso is not None                                # :3624
        assert len(x.shape) == 4 and x.shape[2] == self.F
        B, T, N = x.shape[0], x.shape[1], x.shape[3]
        assert len(z0.shape) == 3 and z0.shape[0] == B and z0.shape[1] == self.H and z0.shape[2] == N
        z = GatedGRNN(self.aWeights, self.bWeights, self._gso, x, z0, self.sigma, xBias=self.xBias, zBias=self.zBias)
        return z, z[:, T - 1:T].unsqueeze(1)                        # :3645-3648 (index_select(T-1) then unsqueeze(1))

    def extra_repr(self):
        reprString = "in_features=%d, hidden_features=%d, " % (self.F, self.H) + "filter_taps=%d, " % (self.K) + \
                     "edge_features=%d, " % (self.E) + "bias=%s, " % (self.bias) + "nonlinearity=%s" % (self.sigma)
        reprString += "GSO stored" if self.S is not None else "no GSO stored"
        return reprString


class _GatedHiddenStateBase(nn.Module):
    """Common part of TimeGatedHiddenState / NodeGatedHiddenState (graphML.py:3683-4031): the state taps (``aWeights``, ``bWeights``,
    ``xBias``, ``zBias``) and two ungated HiddenState layers whose outputs feed the input / forget gates.  Same parameter names and
    shapes as the reference: its checkpoints load."""

    def __init__(self, F, H, K, nonlinearity=torch.tanh, E=1, bias=True):
        super().__init__()
        self.F = F
        self.H = H
        self.K = K
        self.E = E
        self.S = None
        self._gso = None
        self.bias = bias
        self.sigma = nonlinearity
        self.aWeights = nn.parameter.Parameter(torch.Tensor(H, E, K, F))
        self.bWeights = nn.parameter.Parameter(torch.Tensor(H, E, K, H))
        self.inputGateGRNN = HiddenState(F, H, K, bias=bias)         # :3748 / :3920
        self.forgetGateGRNN = HiddenState(F, H, K, bias=bias)        # :3750 / :3922
        if self.bias:
            self.xBias = nn.parameter.Parameter(torch.Tensor(H, 1))
            self.zBias = nn.parameter.Parameter(torch.Tensor(H, 1))
        else:
            self.register_parameter('xBias', None)
            self.register_parameter('zBias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.F * self.K)                      # :3765-3771
        self.aWeights.data.uniform_(-stdv, stdv)
        self.bWeights.data.uniform_(-stdv, stdv)
        if self.bias:
            self.xBias.data.uniform_(-stdv, stdv)
            self.zBias.data.uniform_(-stdv, stdv)

    def _set_gso(self, S):
        if sp.issparse(S) or isinstance(S, (list, tuple)):
            S = SparseGSO.from_any(S)
        assert len(S.shape) == 3 and S.shape[0] == self.E           # :3822-3824
        self.N = S.shape[1]
        assert S.shape[2] == self.N
        self.S = S
        self._gso = SparseGSO.from_any(S)
        self.inputGateGRNN.addGSO(self._gso)
        self.forgetGateGRNN.addGSO(self._gso)

    def _check(self, x, z0):
        assert self._gso is not None
        assert len(x.shape) == 4 and x.shape[2] == self.F
        B, T, N = x.shape[0], x.shape[1], x.shape[3]
        assert len(z0.shape) == 3 and z0.shape[0] == B and z0.shape[1] == self.H and z0.shape[2] == N
        return B, T, N

    def _state(self, x, z0, qHat, qCheck, T):
        z = GatedGRNN(self.aWeights, self.bWeights, self._gso, x, z0, self.sigma, qHat, qCheck, xBias=self.xBias, zBias=self.zBias)
        return z, z[:, T - 1:T].unsqueeze(1)                        # index_select(T-1) then unsqueeze(1), as HiddenState

    def extra_repr(self):
        reprString = "in_features=%d, hidden_features=%d, " % (self.F, self.H) + "filter_taps=%d, " % (self.K) + \
                     "edge_features=%d, " % (self.E) + "bias=%s, " % (self.bias) + "nonlinearity=%s" % (self.sigma)
        reprString += "GSO stored" if self.S is not None else "no GSO stored"
        return reprString


class TimeGatedHiddenState(_GatedHiddenStateBase):
    """TimeGatedHiddenState(signal_features, hidden_features, filter_taps, nonlinearity=torch.tanh, edge_features=1, bias=True)
    -- graphML.py:3683-3855.  One scalar input gate and one forget gate per (sample, time step): a fully connected layer on the
    states of an ungated recurrent layer, ``inputGateFC`` / ``forgetGateFC`` = Linear(H*N, 1) created by ``addGSO`` (:3829-3830).
    forward(x [B,T,F,N], z0 [B,H,N]) -> (z [B,T,H,N], z_T [B,1,1,H,N]); every LSIGF inside runs on the HIP path."""

    def addGSO(self, S):
        self._set_gso(S)
        dt, dev = self.aWeights.dtype, self.aWeights.device
        self.inputGateFC = nn.Linear(self.H * self.N, 1, self.bias).to(device=dev, dtype=dt)
        self.forgetGateFC = nn.Linear(self.H * self.N, 1, self.bias).to(device=dev, dtype=dt)

    def forward(self, x, z0):
        B, T, N = self._check(x, z0)
        zHat, _ = self.inputGateGRNN(x, z0)                          # :3795-3798
        qHat = torch.sigmoid(self.inputGateFC(zHat.reshape((B, T, self.H * N)))).unsqueeze(2)       # B x T x 1 x 1
        zCheck, _ = self.forgetGateGRNN(x, z0)                       # :3801-3804
        qCheck = torch.sigmoid(self.forgetGateFC(zCheck.reshape((B, T, self.H * N)))).unsqueeze(2)
        return self._state(x, z0, qHat, qCheck, T)


class NodeGatedHiddenState(_GatedHiddenStateBase):
    """NodeGatedHiddenState(...) -- graphML.py:3857-4031.  One input gate and one forget gate per (sample, time step, node): a
    GraphFilter(H, 1, K) on the states of an ungated recurrent layer, ``inputGateGraphFilter`` / ``forgetGateGraphFilter`` created by
    ``addGSO`` (:4012-4013).  forward as TimeGatedHiddenState."""

    def addGSO(self, S):
        self._set_gso(S)
        dt, dev = self.aWeights.dtype, self.aWeights.device
        self.inputGateGraphFilter = GraphFilter(self.H, 1, self.K, bias=self.bias).to(device=dev, dtype=dt)
        self.forgetGateGraphFilter = GraphFilter(self.H, 1, self.K, bias=self.bias).to(device=dev, dtype=dt)
        self.inputGateGraphFilter.addGSO(self._gso)
        self.forgetGateGraphFilter.addGSO(self._gso)

    def forward(self, x, z0):
        B, T, N = self._check(x, z0)
        zHat, _ = self.inputGateGRNN(x, z0)                          # :3970-3973
        qHat = torch.sigmoid(self.inputGateGraphFilter(zHat.reshape((B * T, self.H, N)))).reshape((B, T, 1, N))
        zCheck, _ = self.forgetGateGRNN(x, z0)                       # :3976-3979
        qCheck = torch.sigmoid(self.forgetGateGraphFilter(zCheck.reshape((B * T, self.H, N)))).reshape((B, T, 1, N))
        return self._state(x, z0, qHat, qCheck, T)


_ZERO_TOLERANCE = 1e-9      # graphML.py:72
_INFINITE_NUMBER = 1e12     # graphML.py:73


def learnAttentionGSO(x, a, W, S, negative_slope=0.2):
    """Attention coefficients as a GSO -- graphML.py:640-737 (the gate network of EdgeGatedHiddenState, :4159-4168).

        alpha_ij^{pe} = softmax_j( LeakyReLU( a2^{pe} . W^{pe} x_i + a1^{pe} . W^{pe} x_j ) )   over the neighbourhood of i (S + I)
        (as coded at :706-712: the first half of the mixing vector multiplies the neighbour j)

    x [B,G,N], a [P,E,2F], W [P,E,F,G], S [E,N,N] dense or SparseGSO -> alpha [B,P,E,N,N] (dense: that is what an edge gate is).
    Host logic in torch: elementwise + softmax on the [N,N] support, no graph filter inside."""
    gso = SparseGSO.from_any(S)
    B, N = x.shape[0], x.shape[2]
    P, E = a.shape[0], a.shape[1]
    assert W.shape[0] == P and W.shape[1] == E                       # :681-682
    F = W.shape[2]
    assert a.shape[2] == 2 * F                                       # :684
    assert gso.E == E and gso.N == N                                 # :686-687
    Sd = _dense_on(gso, x)
    mask = ((Sd + torch.eye(N, dtype=x.dtype, device=x.device)).abs().sum(dim=0) > _ZERO_TOLERANCE).to(x.dtype)   # :692, :726-728
    Wx = torch.matmul(W.reshape(1, P, E, F, W.shape[3]), x.reshape(B, 1, 1, x.shape[1], N))      # B x P x E x F x N   (:701-703)
    a1Wx = torch.matmul(a[:, :, :F].reshape(1, P, E, 1, F), Wx)      # B x P x E x 1 x N   (:706-709)
    a2Wx = torch.matmul(a[:, :, F:].reshape(1, P, E, 1, F), Wx)
    e = nn.functional.leaky_relu(a1Wx + a2Wx.permute(0, 1, 2, 4, 3), negative_slope=negative_slope)   # :712-718
    alpha = nn.functional.softmax(e * mask - (1 - mask) * _INFINITE_NUMBER, dim=4)               # :730-733
    return alpha * mask                                              # :737


class GraphAttentional(nn.Module):
    """Parameter holder of the attention gate networks: ``mixer [K,E,2F]`` and ``weight [K,E,F,G]`` with the reference's names, shapes
    and initialisation (graphML.py:2899-2933), so that EdgeGatedHiddenState checkpoints interchange.  The attention ARCHITECTURES are
    out of scope; only ``learnAttentionGSO`` on these parameters is used here."""

    def __init__(self, G, F, K, E=1, nonlinearity=nn.functional.relu, concatenate=True):
        super().__init__()
        self.G, self.F, self.K, self.E = G, F, K, E
        self.S = None
        self.nonlinearity = nonlinearity
        self.concatenate = concatenate
        self.mixer = nn.parameter.Parameter(torch.Tensor(K, E, 2 * F))
        self.weight = nn.parameter.Parameter(torch.Tensor(K, E, F, G))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.G * self.K)                      # :2922-2924
        self.weight.data.uniform_(-stdv, stdv)
        self.mixer.data.uniform_(-stdv, stdv)

    def addGSO(self, S):
        assert len(S.shape) == 3 and S.shape[0] == self.E           # :2928-2930
        self.N = S.shape[1]
        assert S.shape[2] == self.N
        self.S = S

    def forward(self, x):
        raise NotImplementedError("GraphAttentional.forward (graph attention layer) is outside the accelerated path; "
                                  "EdgeGatedHiddenState only uses learnAttentionGSO on its parameters")


class EdgeGatedHiddenState(_GatedHiddenStateBase):
    """EdgeGatedHiddenState(signal_features, hidden_features, filter_taps, nonlinearity=torch.tanh, edge_features=1, bias=True)
    -- graphML.py:4033-4208.  One input gate and one forget gate per (sample, time step, EDGE): attention coefficients computed from
    the states of an ungated recurrent layer (``inputGateGAT`` / ``forgetGateGAT`` = GraphAttentional(H, 1, 1), created by ``addGSO``,
    :4189-4190) multiply the GSO entrywise inside GatedGRNN.  forward as TimeGatedHiddenState."""

    def addGSO(self, S):
        self._set_gso(S)
        dt, dev = self.aWeights.dtype, self.aWeights.device
        self.inputGateGAT = GraphAttentional(self.H, 1, 1).to(device=dev, dtype=dt)
        self.forgetGateGAT = GraphAttentional(self.H, 1, 1).to(device=dev, dtype=dt)
        self.inputGateGAT.addGSO(self._gso)
        self.forgetGateGAT.addGSO(self._gso)

    def forward(self, x, z0):
        B, T, N = self._check(x, z0)
        zHat, _ = self.inputGateGRNN(x, z0)                          # :4157-4158
        qHat = learnAttentionGSO(zHat.reshape((B * T, self.H, N)), self.inputGateGAT.mixer, self.inputGateGAT.weight,
                                 self._gso).reshape((B, T, 1, N, N))                              # :4159-4161
        zCheck, _ = self.forgetGateGRNN(x, z0)                       # :4164-4165
        qCheck = learnAttentionGSO(zCheck.reshape((B * T, self.H, N)), self.forgetGateGAT.mixer, self.forgetGateGAT.weight,
                                   self._gso).reshape((B, T, 1, N, N))                            # :4166-4168
        return self._state(x, z0, qHat, qCheck, T)


# ---------------------------------------------------------------------------------------------------------------
# Per-sample, delayed GSOs (the flocking models): GraphFilter_DB / HiddenState_DB around LSIGF_DB / GRNN_DB
# ---------------------------------------------------------------------------------------------------------------
class GraphFilter_DB(nn.Module):
    """GraphFilter_DB(in_features, out_features, filter_taps, edge_features=1, bias=True) -- graphML.py:3278-3393.
    Parameters ``weight [F,E,K,G]``, ``bias [F,1]`` (reference names / shapes / initialisation: checkpoints interchange);
    ``addGSO(S [B,T,E,N,N])`` stores the per-sample, per-time-step operators; ``forward(x [B,T,G,N]) -> [B,T,F,N]`` = LSIGF_DB on
    the HIP path (one launch per tap for all B*T operators)."""

    def __init__(self, G, F, K, E=1, bias=True):
        super().__init__()
        self.G, self.F, self.K, self.E = G, F, K, E
        self.S = None
        self.weight = nn.parameter.Parameter(torch.Tensor(F, E, K, G))
        if bias:
            self.bias = nn.parameter.Parameter(torch.Tensor(F, 1))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.G * self.K)                      # :3341-3345
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def addGSO(self, S):
        assert len(S.shape) == 5                                    # :3349
        assert S.shape[2] == self.E                                 # :3351
        self.N = S.shape[3]
        assert S.shape[4] == self.N
        self.S = S

    def forward(self, x):
        assert len(x.shape) == 4                                    # :3358
        B, T = x.shape[0], x.shape[1]
        assert self.S.shape[0] == B and self.S.shape[1] == T        # :3360-3362
        assert x.shape[3] == self.N
        return LSIGF_DB(self.weight, self.S, x, self.bias)          # :3366

    def extra_repr(self):
        reprString = "in_features=%d, out_features=%d, " % (self.G, self.F) + "filter_taps=%d, " % (self.K) + \
                     "edge_features=%d, " % (self.E) + "bias=%s, " % (self.bias is not None)
        reprString += "GSO stored" if self.S is not None else "no GSO stored"
        return reprString


class HiddenState_DB(nn.Module):
    """HiddenState_DB(signal_features, hidden_features, filter_taps, nonlinearity=torch.tanh, edge_features=1, bias=True)
    -- graphML.py:3395-3538.  Parameters ``aWeights [H,E,K,F]``, ``bWeights [H,E,K,H]``, ``xBias`` / ``zBias [H,1]``;
    ``addGSO(S [B,T,E,N,N])``; ``forward(x [B,T,F,N], z0 [B,H,N]) -> (z [B,T,H,N], z_T [B,1,1,H,N])`` = GRNN_DB on the HIP path."""

    def __init__(self, F, H, K, nonlinearity=torch.tanh, E=1, bias=True):
        super().__init__()
        self.F, self.H, self.K, self.E = F, H, K, E
        self.S = None
        self.bias = bias
        self.sigma = nonlinearity
        self.aWeights = nn.parameter.Parameter(torch.Tensor(H, E, K, F))
        self.bWeights = nn.parameter.Parameter(torch.Tensor(H, E, K, H))
        if self.bias:
            self.xBias = nn.parameter.Parameter(torch.Tensor(H, 1))
            self.zBias = nn.parameter.Parameter(torch.Tensor(H, 1))
        else:
            self.register_parameter('xBias', None)
            self.register_parameter('zBias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.F * self.K)                      # :3469-3475
        self.aWeights.data.uniform_(-stdv, stdv)
        self.bWeights.data.uniform_(-stdv, stdv)
        if self.bias:
            self.xBias.data.uniform_(-stdv, stdv)
            self.zBias.data.uniform_(-stdv, stdv)

    def addGSO(self, S):
        assert len(S.shape) == 5                                    # :3519
        assert S.shape[2] == self.E
        self.N = S.shape[3]
        assert S.shape[4] == self.N
        self.S = S

    def forward(self, x, z0):
        assert self.S is not None                                   # :3479
        assert len(x.shape) == 4                                    # :3489
        B, T = x.shape[0], x.shape[1]
        assert self.S.shape[0] == B and self.S.shape[1] == T        # :3491-3493
        assert x.shape[2] == self.F
        N = x.shape[3]
        assert len(z0.shape) == 3 and z0.shape[0] == B and z0.shape[1] == self.H and z0.shape[2] == N   # :3497-3500
        z = GRNN_DB(self.aWeights, self.bWeights, self.S, x, z0, self.sigma, xBias=self.xBias, zBias=self.zBias)
        return z, z[:, T - 1:T].unsqueeze(1)                        # :3506-3510

    def extra_repr(self):
        reprString = "in_features=%d, hidden_features=%d, " % (self.F, self.H) + "filter_taps=%d, " % (self.K) + \
                     "edge_features=%d, " % (self.E) + "bias=%s, " % (self.bias) + "nonlinearity=%s" % (self.sigma)
        reprString += "GSO stored" if self.S is not None else "no GSO stored"
        return reprString


# ---------------------------------------------------------------------------------------------------------------
# ARMA filter by Jacobi iterations: the residue is an LSIGF, the iterations are hops with the
# off-diagonal part of S
# ---------------------------------------------------------------------------------------------------------------
def _shift(off_t, v):
    """Stilde_e @ v for every edge feature e: v [B', E, G, N] -> [B', E, G, N], each product through the HIP filter (an LSIGF with
    K = 2 whose only non-zero taps are the identity on tap 1, on the plan of Stilde_e^T: LSIGF applies x S, the iteration needs S x).
    off_t: one single-feature SparseGSO (of Stilde_e^T) per edge feature."""
    Bp, E, G, N = v.shape
    sel = torch.zeros((G, 1, 2, G), dtype=v.dtype, device=v.device)
    sel[:, 0, 1, :] = torch.eye(G, dtype=v.dtype, device=v.device)
    return torch.stack([LSIGF(sel, off_t[e], v[:, e].contiguous()) for e in range(E)], dim=1)


def jARMA(psi, varphi, phi, S, x, b=None, tMax=5):
    """ARMA graph filter evaluated with tMax Jacobi iterations -- reference signature and semantics, graphML.py:490-638:

        u_f = sum_{e,g} sum_p [ sum_{tau=0..tMax} (-1)^tau varphi_p (Sbar_p^-1 Stilde)^tau Sbar_p^-1 x_g
                                + (-1)^(tMax+1) (Sbar_p^-1 Stilde)^(tMax+1) x_g ]  +  LSIGF(phi, S, x)  (+ b)
        Sbar_p^{fge} = Diag(S_e) - psi_p^{fge} I,   Stilde = DiagOff(S_e)

    psi, varphi [F,E,P,G], phi [F,E,K,G], S [E,N,N] (dense tensor or anything SparseGSO.from_any accepts), x [B,G,N].
    The reference materialises Sbar^-1 Stilde as a dense [F,E,P,G,N,N] tensor (:585-589); here Sbar^-1 is the diagonal it is
    ([F,E,P,G,N], elementwise) and every product with Stilde is a sparse hop of the [B,F,E,P,G] chain states on the HIP path."""
    F, E, P, G = psi.shape
    assert varphi.shape == psi.shape                                 # :549-552
    assert phi.shape[0] == F and phi.shape[1] == E and phi.shape[3] == G
    B = x.shape[0]
    assert x.shape[1] == G
    N = x.shape[2]
    gso = SparseGSO.from_any(S)
    assert gso.E == E and gso.N == N                                 # :560-561
    key = "_jarma_parts"
    parts = getattr(gso, key, None)
    if parts is None:                                                # the Jacobi splitting of S, once per GSO
        off = gso.offdiagonal().transposed()
        parts = ([SparseGSO([m]) for m in off.mats], gso.diagonal())
        setattr(gso, key, parts)
    off_t, diag = parts
    d = torch.as_tensor(diag, dtype=x.dtype, device=x.device)        # [E, N]
    Dinv = 1.0 / (d.reshape(1, E, 1, 1, N) - psi.reshape(F, E, P, G, 1))      # Sbar^-1 (diagonal), [F,E,P,G,N]   (:573-583)

    def shift(v):                                                    # v [B,F,E,P,G,N] -> Sbar^-1 Stilde v
        w = v.permute(0, 1, 3, 2, 4, 5).reshape(B * F * P, E, G, N)
        w = _shift(off_t, w).reshape(B, F, P, E, G, N).permute(0, 1, 3, 2, 4, 5)
        return Dinv.unsqueeze(0) * w

    x1 = Dinv.unsqueeze(0) * x.reshape(B, 1, 1, 1, G, N)             # Sbar^-1 x                       (:584-586)
    y = x.reshape(B, 1, 1, 1, G, N).expand(B, F, E, P, G, N)
    sign = 1.0
    H1 = varphi.reshape(1, F, E, P, G, 1) * x1                       # tau = 0 term of H1
    for tau in range(1, tMax + 1):                                   # :602-612
        x1 = shift(x1)
        y = shift(y)
        sign = -sign
        H1 = H1 + sign * varphi.reshape(1, F, E, P, G, 1) * x1
    y = shift(y)                                                     # (Sbar^-1 Stilde)^(tMax+1) x     (:627)
    H2 = -y if tMax % 2 == 0 else y                                  # :629
    u = (H1 + H2).sum(dim=(2, 3, 4))                                 # over e, p, g -> [B,F,N]          (:630-635)
    u = u + LSIGF(phi, gso, x)                                       # H3: the residue filter          (:598)
    if b is not None:
        u = u + b
    return u

