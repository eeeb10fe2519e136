"""Graph-filter layers with the reference's Python surface, backed by the gfx950 HIP library.

Drop-in for the symbols SelectionGNN binds from alegnn/utils/graphML.py:
    LSIGF          (graphML.py:83-176)     -> alegnn_amd.functional.LSIGF   (HIP)
    GraphFilter    (graphML.py:2036-2155)  -> same ctor, attributes, parameter names/shapes, addGSO/forward/extra_repr
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

from ..functional import LSIGF
from ..gso import SparseGSO
from . import graphTools

__all__ = ["LSIGF", "GraphFilter", "NoPool", "MaxPoolLocal"]


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
        return LSIGF(self.weight, self._gso, x, self.bias)

    def extra_repr(self):
        reprString = "in_features=%d, out_features=%d, " % (self.G, self.F) + "filter_taps=%d, " % (self.K) + \
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
        self.register_buffer("neighborhood", None, persistent=False)

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
        nbh = torch.as_tensor(nbh, dtype=torch.int64)
        assert nbh.shape[0] == self.nOutputNodes                 # graphML.py:1962-1963
        assert int(nbh.max()) <= self.nInputNodes
        self.maxNeighborhoodSize = nbh.shape[1]
        self.neighborhood = nbh.to(device) if device is not None else nbh

    def forward(self, x):
        assert x.shape[2] == self.nInputNodes                    # graphML.py:1972-1976
        assert x.shape[2] >= self.nOutputNodes
        xn = x[:, :, self.neighborhood]                          # B x F x nOutputNodes x maxNeighborhoodSize
        v, _ = torch.max(xn, dim=3)
        return v

    def extra_repr(self):
        reprString = "in_dim=%d, out_dim=%d, number_hops = %d, " % (self.nInputNodes, self.nOutputNodes, self.nHops)
        reprString += "neighborhood stored" if self.neighborhood is not None else "NO neighborhood stored"
        return reprString
