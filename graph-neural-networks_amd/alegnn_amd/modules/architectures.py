"""SelectionGNN with the reference's constructor (alegnn/modules/architectures.py:49-479), built on the HIP GraphFilter.

Same signature, same sub-module names (``GFL`` = [GraphFilter, sigma, rho] x L, ``MLP``), hence the same state_dict
keys (``GFL.0.weight`` ... ``MLP.0.bias``): checkpoints written by the reference's Model.save (model.py:106-117)
load here and vice versa.

Differences, all supersets or fixes:
  * ``GSO`` may also be a scipy sparse matrix, a list of them (one per edge feature) or a ``SparseGSO`` -- required
    for N >= 1e4 where the reference's dense E x N x N array stops fitting.
  * ``order='Degree' | 'EDS' | 'SpectralProxies'`` works (the reference evaluates ``Utils.graphTools.perm...`` with an
    undefined name ``Utils`` and raises NameError, architectures.py:210).
  * ``coarsening=True`` (Graclus, graphTools.py:1337-1614; single edge feature, as in the reference) accepts a sparse
    GSO too, and pads / reorders the input signal on the device in x's dtype (the reference round-trips through a
    float64 numpy array, architectures.py:429-434).
  * ``.to(device)`` moves parameters and buffers only; device plans are built lazily per device, and the identity
    node ordering skips the ``x[:, :, order]`` gather (architectures.py:437 copies x every call).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from ..gso import SparseGSO
from ..utils import graphML as gml
from ..utils import graphTools


class _WideToOneLinear(nn.Linear):
    """nn.Linear(in_features, 1) for the flattened N*F -> 1 readout of the regression architectures (movieGNN: dimLayersMLP = [1],
    architectures.py:298-319): same parameters and state_dict keys, forward as multiply + row sum.  rocBLAS / hipBLASLt pick a
    256-wide tile for the one-column GEMM and its gradients (N*F = 53824, batch 256: 365 us forward + backward against 145 us for
    the three memory-bound passes this is)."""

    def forward(self, x):
        if x.dim() != 2 or x.device.type != "cuda":
            return super().forward(x)
        y = (x * self.weight).sum(dim=1, keepdim=True)
        return y if self.bias is None else y + self.bias


def _readout_linear(in_features, out_features, bias):
    return _WideToOneLinear(in_features, 1, bias=bias) if out_features == 1 and in_features >= 4096 else nn.Linear(in_features, out_features, bias=bias)


class SelectionGNN(nn.Module):
    def __init__(self,
                 # Graph filtering
                 dimNodeSignals, nFilterTaps, bias,
                 # Nonlinearity
                 nonlinearity,
                 # Pooling
                 nSelectedNodes, poolingFunction, poolingSize,
                 # MLP in the end
                 dimLayersMLP,
                 # Structure
                 GSO,
                 # Ordering
                 order=None,
                 # Coarsening
                 coarsening=False):
        super().__init__()
        self._build_filters(dimNodeSignals, nFilterTaps, bias, nonlinearity, nSelectedNodes, poolingFunction, poolingSize,
                            GSO, order, coarsening)
        self.dimLayersMLP = dimLayersMLP

        fc = []
        if len(self.dimLayersMLP) > 0:                              # :299-317
            fc.append(_readout_linear(self.N[-1] * self.F[-1], dimLayersMLP[0], self.bias))
            for l in range(len(dimLayersMLP) - 1):
                fc.append(self.sigma())
                fc.append(nn.Linear(dimLayersMLP[l], dimLayersMLP[l + 1], bias=self.bias))
        self.MLP = nn.Sequential(*fc)

    def _build_filters(self, dimNodeSignals, nFilterTaps, bias, nonlinearity, nSelectedNodes, poolingFunction, poolingSize,
                       GSO, order, coarsening):
        """The part every GraphFilter-stack architecture shares (architectures.py:184-295, :941-1008): argument checks,
        GSO ingest / ordering, and GFL = [GraphFilter, sigma, rho] x L."""
        assert len(dimNodeSignals) == len(nFilterTaps) + 1         # architectures.py:184
        assert len(nSelectedNodes) == len(nFilterTaps)              # :187
        assert len(poolingSize) == len(nFilterTaps)                 # :189
        self.L = len(nFilterTaps)
        self.F = dimNodeSignals
        self.K = nFilterTaps
        self.bias = bias
        self.sigma = nonlinearity
        self.rho = poolingFunction
        self.coarsening = bool(coarsening)
        self.alpha = poolingSize
        self._order_name = order
        self._install_gso(GSO)
        if not self.coarsening:
            self.N = [self._gso.N] + nSelectedNodes                 # :256

        gfl = []
        for l in range(self.L):
            gfl.append(gml.GraphFilter(self.F[l], self.F[l + 1], self.K[l], self.E, self.bias))   # :277-278
            gfl[3 * l].addGSO(self._gsos[l] if self.coarsening else self._gso)                    # :282-285
            if self.sigma is nn.ReLU:              # sigma = ReLU runs in the filter's epilogue / backward mask
                gfl[3 * l].fused_activation = "relu"
                gfl.append(gml.FusedReLU())
            else:
                gfl.append(self.sigma())                                                          # :287
            if self.coarsening:
                gfl.append(self.rho(self.alpha[l]))                                               # :290 (e.g. nn.MaxPool1d)
            else:
                gfl.append(self.rho(self.N[l], self.N[l + 1], self.alpha[l]))                    # :292
                gfl[3 * l + 2].addGSO(self._gso)
        self.GFL = nn.Sequential(*gfl)

    # ---- GSO handling -----------------------------------------------------------------------------------------
    def _install_gso(self, GSO):
        """Normalise GSO to E x N x N, apply the node ordering, keep ``self.S`` (reference attribute) and the SparseGSO."""
        sparse_in = sp.issparse(GSO) or isinstance(GSO, (SparseGSO, list, tuple))
        if self.coarsening and self._install_coarsened(GSO, sparse_in):
            return
        self.coarsening = False       # more than one edge feature: selection pooling (architectures.py:224, :257)
        if sparse_in:
            # superset of the reference (its permFunction takes the dense [E,N,N] array, architectures.py:203-256): 'Degree' is O(nnz)
            # on the sparse matrices; 'EDS' / 'SpectralProxies' densify for the ORDER up to graphTools.kDenseOrderingMaxNodes nodes
            self._gso = SparseGSO.from_any(GSO)
            if self._order_name is not None:
                mats, order = graphTools.perm_sparse(self._gso.mats, self._order_name)
                self._gso = SparseGSO(mats)
                self.order = order
            else:
                self.order = list(range(self._gso.N))
            self.E = self._gso.E
            self.S = self._gso
            self._identity_order = list(self.order) == list(range(len(self.order)))
            self._order_index = None
            return
        if isinstance(GSO, torch.Tensor):
            GSO = GSO.detach().cpu().numpy()
        GSO = np.asarray(GSO)
        assert len(GSO.shape) == 2 or len(GSO.shape) == 3           # :192
        if len(GSO.shape) == 2:
            assert GSO.shape[0] == GSO.shape[1]
            GSO = GSO.reshape([1, GSO.shape[0], GSO.shape[1]])
        else:
            assert GSO.shape[1] == GSO.shape[2]
        self.E = GSO.shape[0]                                       # :202
        perm = graphTools.permIdentity if self._order_name is None else getattr(graphTools, 'perm' + self._order_name)
        S, self.order = perm(GSO)                                   # :253
        self.S = torch.tensor(S)                                    # :254-255 (keeps numpy's dtype, as the reference does)
        self._gso = SparseGSO.from_any(S)
        self._identity_order = list(self.order) == list(range(len(self.order)))
        self._order_index = None

    def _install_coarsened(self, GSO, sparse_in):
        """Graclus branch of the constructor / changeGSO (architectures.py:224-247, :393-415): L+1 coarsened GSOs, one
        per layer input, binary-tree node order, pooling size 2.  False when the GSO has several edge features."""
        if sparse_in:
            if isinstance(GSO, SparseGSO):
                mats = GSO.mats
            else:
                mats = [GSO] if sp.issparse(GSO) else list(GSO)
            if len(mats) != 1:
                return False
            W = sp.csr_matrix(mats[0])
        else:
            if isinstance(GSO, torch.Tensor):
                GSO = GSO.detach().cpu().numpy()
            GSO = np.asarray(GSO)
            assert len(GSO.shape) == 2 or len(GSO.shape) == 3       # :192
            if len(GSO.shape) == 3:
                assert GSO.shape[1] == GSO.shape[2]
                if GSO.shape[0] != 1:
                    return False
                GSO = GSO[0]
            assert GSO.shape[0] == GSO.shape[1]
            W = sp.csr_matrix(GSO)
        self.E = 1
        graphs, self.order = graphTools.coarsen(W, levels=self.L, self_connections=False)      # :227-228
        self._gsos = [SparseGSO.from_any(g) for g in graphs]
        self._gso = self._gsos[0]
        # reference attribute: list of 1 x N_l x N_l tensors (:232-245); a sparse GSO in stays sparse
        self.S = self._gsos if sparse_in else [torch.tensor(g.toarray().reshape(1, g.shape[0], g.shape[1])) for g in graphs]
        self.N = [g.shape[0] for g in graphs]
        self.alpha = [2] * self.L                                   # :247
        self._n_real = W.shape[0]
        self._identity_order = False
        self._order_index = None
        return True

    def changeGSO(self, GSO, nSelectedNodes=[], poolingSize=[]):
        """Swap the graph under the same filter taps -- architectures.py:322-420."""
        if self.coarsening:
            # the pooling modules do not depend on the graph, nSelectedNodes / poolingSize are ignored (:357, :364)
            self._install_gso(GSO)
            assert self.coarsening, "changeGSO: the model was built with coarsening, the new GSO must have one edge feature"
            for l in range(self.L):
                self.GFL[3 * l].addGSO(self._gsos[l])              # :413-415
            return
        self._install_gso(GSO)
        if len(poolingSize) > 0:
            assert len(poolingSize) == self.L
            self.alpha = poolingSize
        if len(nSelectedNodes) > 0:
            assert len(nSelectedNodes) == self.L
            self.N = [self._gso.N] + nSelectedNodes
            device = next(self.parameters()).device
            for l in range(self.L):
                self.GFL[3 * l + 2] = self.rho(self.N[l], self.N[l + 1], self.alpha[l])
                self.GFL[3 * l + 2].addGSO(self._gso)
                self.GFL[3 * l + 2].to(device)
        else:
            for l in range(self.L):
                self.GFL[3 * l + 2].addGSO(self._gso)
        for l in range(self.L):
            self.GFL[3 * l].addGSO(self._gso)

    # ---- forward ----------------------------------------------------------------------------------------------
    def _reorder(self, x):
        """x[:, :, order] (architectures.py:437), skipped for the identity; with coarsening also the fake-node padding."""
        if self.coarsening:
            if self._order_index is None or self._order_index.device != x.device:
                # fake nodes (index >= number of real nodes) read the zero column appended below
                idx = np.minimum(np.asarray(self.order, dtype=np.int64), self._n_real)
                self._order_index = torch.as_tensor(idx, dtype=torch.int64, device=x.device)
                self._order_plain = torch.as_tensor(np.asarray(self.order, dtype=np.int64), device=x.device)
            if x.shape[2] != self.N[0]:                             # :429-434 permCoarsening, kept on the device
                assert x.shape[2] == self._n_real
                x = torch.cat((x, x.new_zeros(x.shape[0], x.shape[1], 1)), dim=2)[:, :, self._order_index]
            else:
                x = x[:, :, self._order_plain]                      # :437
        elif not self._identity_order:
            if self._order_index is None or self._order_index.device != x.device:
                self._order_index = torch.as_tensor(self.order, dtype=torch.int64, device=x.device)
            x = x[:, :, self._order_index]                          # :437
        return x

    def _run_gfl(self, x):
        """self.GFL(x) (architectures.py:445) -- with runs of [GraphFilter, ReLU, NoPool] blocks on the same graph executed as ONE chain
        whose intermediate signals stay in the library's internal layout (functional.LSIGF_chain): per inner boundary the
        reference-layout round trip (graphML.py:170-171 + the next layer's re-layout) disappears.  Same values, bit for bit."""
        from ..functional import LSIGF_chain, lsigf_chain_supported
        mods = list(self.GFL)
        i = 0
        while i < len(mods):
            m = mods[i]
            run = []
            j = i
            while (isinstance(mods[j], gml.GraphFilter) and mods[j]._gso is not None and (not run or mods[j]._gso is run[0]._gso)):
                run.append(mods[j])
                # the block continues only through a fused ReLU and a pooling stage that keeps every node
                if not (mods[j].fused_activation == "relu" and j + 3 < len(mods) and isinstance(mods[j + 1], gml.FusedReLU)
                        and isinstance(mods[j + 2], gml.NoPool) and mods[j + 2].nInputNodes == mods[j + 2].nOutputNodes == mods[j]._gso.N):
                    break
                j += 3
            layers = [(f.weight, f.bias, f.fused_activation == "relu") for f in run]
            # (the chain bypasses the inner modules' __call__: a run in which any of them carries a hook goes module by module)
            inner = mods[i:i + 3 * (len(run) - 1) + 1]
            hooked = any(q._forward_hooks or q._forward_pre_hooks or q._backward_hooks or getattr(q, "_backward_pre_hooks", None) for q in inner)
            if len(run) >= 2 and not hooked and lsigf_chain_supported(run[0]._gso, x, layers):
                x = LSIGF_chain(layers, run[0]._gso, x)
                i += 3 * (len(run) - 1) + 1      # continue with the modules that follow the run's last filter (its FusedReLU is an identity)
            else:
                x = m(x)
                i += 1
        return x

    def splitForward(self, x):
        x = self._reorder(x)
        assert len(x.shape) == 3                                    # :440-443
        batchSize = x.shape[0]
        assert x.shape[1] == self.F[0]
        assert x.shape[2] == self.N[0]
        y = self._run_gfl(x)                                        # :445
        yFlat = y.reshape(batchSize, self.F[-1] * self.N[-1])       # :447
        return self.MLP(yFlat), y

    def forward(self, x):
        output, _ = self.splitForward(x)
        return output

    def to(self, *args, **kwargs):
        # The reference moves S and re-adds it to every layer (architectures.py:463-479); here the GSO lives on the host
        # as CSR and device plans are created on first use per device, so only parameters / buffers move.
        return super().to(*args, **kwargs)


class LocalGNN(SelectionGNN):
    """Graph filter stack with a per-node readout -- architectures.py:816-1182.  Same constructor, same sub-module names
    (``GFL``, ``Readout``) and state_dict keys as the reference; GSO handling, ``changeGSO`` and the fused ReLU are
    SelectionGNN's.  Output: B x dimReadout[-1] x N[-1]."""

    def __init__(self,
                 # Graph filtering
                 dimNodeSignals, nFilterTaps, bias,
                 # Nonlinearity
                 nonlinearity,
                 # Pooling
                 nSelectedNodes, poolingFunction, poolingSize,
                 # MLP in the end
                 dimReadout,
                 # Structure
                 GSO, order=None):
        nn.Module.__init__(self)
        self._build_filters(dimNodeSignals, nFilterTaps, bias, nonlinearity, nSelectedNodes, poolingFunction, poolingSize,
                            GSO, order, False)
        self.dimReadout = dimReadout
        fc = []
        if len(self.dimReadout) > 0:                                # :1011-1023: F[-1] -> dimReadout, applied node by node
            fc.append(nn.Linear(self.F[-1], dimReadout[0], bias=self.bias))
            for l in range(len(dimReadout) - 1):
                fc.append(self.sigma())
                fc.append(nn.Linear(dimReadout[l], dimReadout[l + 1], bias=self.bias))
        self.Readout = nn.Sequential(*fc)

    def splitForward(self, x):
        assert len(x.shape) == 3                                    # :1093-1095
        assert x.shape[1] == self.F[0]
        assert x.shape[2] == self.N[0]
        x = self._reorder(x)                                        # :1097
        yGFL = self.GFL(x)                                          # :1099
        y = self.Readout(yGFL.permute(0, 2, 1))                     # :1101-1102  B x N[-1] x dimReadout[-1]
        return y.permute(0, 2, 1), yGFL

    def singleNodeForward(self, x, nodes):
        """Output at one node per sample (MovieLens rating prediction, architectures.py:1117-1170).  ``nodes``: an int
        (same node for the whole batch), or a list / array of B node ids in the ORIGINAL numbering.  B x dimReadout[-1].
        The reference multiplies by a one-hot B x N[-1] x 1 matrix (:1160-1168); this is the same selection as a gather."""
        batchSize = x.shape[0]
        assert type(nodes) is int or type(nodes) is list or type(nodes) is np.ndarray      # :1132-1134
        position = np.empty(len(self.order), dtype=np.int64)
        position[np.asarray(self.order, dtype=np.int64)] = np.arange(len(self.order))       # where each node sits after ordering
        if type(nodes) is int:
            nodes = np.full(batchSize, position[nodes], dtype=np.int64)
        else:
            nodes = position[np.asarray(nodes, dtype=np.int64)]
        assert nodes.shape[0] == batchSize and int(nodes.max()) < self.N[-1]
        y = self.forward(x)                                         # B x R x N[-1]
        idx = torch.as_tensor(nodes, device=y.device).view(batchSize, 1, 1).expand(batchSize, y.shape[1], 1)
        return torch.gather(y, 2, idx).squeeze(2)


class NodeVariantGNN(SelectionGNN):
    """Node-variant filter stack -- architectures.py:1485-1719: ``NVGFL`` = [NodeVariantGF, sigma, rho] x L and ``MLP``; same
    constructor and state_dict keys as the reference.  GSO ingest / ordering are SelectionGNN's."""

    def __init__(self,
                 # Graph filtering
                 dimNodeSignals, nShiftTaps, nNodeTaps, bias,
                 # Nonlinearity
                 nonlinearity,
                 # Pooling
                 nSelectedNodes, poolingFunction, poolingSize,
                 # MLP in the end
                 dimLayersMLP,
                 # Structure
                 GSO, order=None):
        nn.Module.__init__(self)
        assert len(dimNodeSignals) == len(nShiftTaps) + 1           # :1582-1586
        assert len(nShiftTaps) == len(nNodeTaps)
        assert len(nSelectedNodes) == len(nShiftTaps)
        assert len(poolingSize) == len(nShiftTaps)
        self.L = len(nShiftTaps)
        self.F = dimNodeSignals
        self.K = nShiftTaps
        self.M = nNodeTaps
        self.bias = bias
        self.sigma = nonlinearity
        self.rho = poolingFunction
        self.alpha = poolingSize
        self.dimLayersMLP = dimLayersMLP
        self.coarsening = False
        self._order_name = order
        self._install_gso(GSO)
        self.N = [self._gso.N] + nSelectedNodes
        nvgfl = []
        for l in range(self.L):                                     # :1628-1641
            nvgfl.append(gml.NodeVariantGF(self.F[l], self.F[l + 1], self.K[l], self.M[l], self.E, self.bias))
            nvgfl[3 * l].addGSO(self._gso)
            nvgfl.append(self.sigma())
            nvgfl.append(self.rho(self.N[l], self.N[l + 1], self.alpha[l]))
            nvgfl[3 * l + 2].addGSO(self._gso)
        self.NVGFL = nn.Sequential(*nvgfl)
        fc = []
        if len(self.dimLayersMLP) > 0:                              # :1645-1662
            fc.append(_readout_linear(self.N[-1] * self.F[-1], dimLayersMLP[0], self.bias))
            for l in range(len(dimLayersMLP) - 1):
                fc.append(self.sigma())
                fc.append(nn.Linear(dimLayersMLP[l], dimLayersMLP[l + 1], bias=self.bias))
        self.MLP = nn.Sequential(*fc)

    def changeGSO(self, GSO, nSelectedNodes=[], poolingSize=[]):
        raise NotImplementedError("the reference's NodeVariantGNN has no changeGSO (the node taps are tied to the graph)")

    def splitForward(self, x):
        assert len(x.shape) == 3                                    # :1684-1688
        batchSize = x.shape[0]
        assert x.shape[1] == self.F[0]
        assert x.shape[2] == self.N[0]
        y = self.NVGFL(self._reorder(x))                            # :1690-1692
        return self.MLP(y.reshape(batchSize, self.F[-1] * self.N[-1])), y


class EdgeVariantGNN(SelectionGNN):
    """Edge-variant filter stack -- architectures.py:1721-1955 (BASELINE configs[4] as an architecture): ``EVGFL`` = [EdgeVariantGF,
    sigma, rho] x L and ``MLP``; same constructor and state_dict keys as the reference (dense ``weightEV [F,E,K,G,N,N]`` per layer).
    Superset: ``sparse=True`` keeps the taps per edge (the only form that exists at N = 5e4).  GSO ingest / ordering are
    SelectionGNN's."""

    def __init__(self,
                 # Graph filtering
                 dimNodeSignals, nShiftTaps, nFilterNodes, bias,
                 # Nonlinearity
                 nonlinearity,
                 # Pooling
                 nSelectedNodes, poolingFunction, poolingSize,
                 # MLP in the end
                 dimLayersMLP,
                 # Structure
                 GSO, order=None, sparse=False):
        nn.Module.__init__(self)
        assert len(dimNodeSignals) == len(nShiftTaps) + 1           # :1812-1818
        assert len(nFilterNodes) == len(nShiftTaps)
        assert len(nSelectedNodes) == len(nShiftTaps)
        assert len(poolingSize) == len(nShiftTaps)
        self.L = len(nShiftTaps)
        self.F = dimNodeSignals
        self.K = nShiftTaps
        self.M = nFilterNodes
        self.bias = bias
        self.sigma = nonlinearity
        self.rho = poolingFunction
        self.alpha = poolingSize
        self.dimLayersMLP = dimLayersMLP
        self.coarsening = False
        self._order_name = order
        self._install_gso(GSO)
        self.N = [self._gso.N] + nSelectedNodes
        evgfl = []
        for l in range(self.L):                                     # :1873-1886: every layer filters on the full graph (N[0])
            evgfl.append(gml.EdgeVariantGF(self.F[l], self.F[l + 1], self.K[l], self.M[l], self.N[0], self.E, self.bias, sparse=sparse))
            evgfl[3 * l].addGSO(self._gso)
            evgfl.append(self.sigma())
            evgfl.append(self.rho(self.N[l], self.N[l + 1], self.alpha[l]))
            evgfl[3 * l + 2].addGSO(self._gso)
        self.EVGFL = nn.Sequential(*evgfl)
        fc = []
        if len(self.dimLayersMLP) > 0:                              # :1890-1907
            fc.append(_readout_linear(self.N[-1] * self.F[-1], dimLayersMLP[0], self.bias))
            for l in range(len(dimLayersMLP) - 1):
                fc.append(self.sigma())
                fc.append(nn.Linear(dimLayersMLP[l], dimLayersMLP[l + 1], bias=self.bias))
        self.MLP = nn.Sequential(*fc)

    def changeGSO(self, GSO, nSelectedNodes=[], poolingSize=[]):
        raise NotImplementedError("the reference's EdgeVariantGNN has no changeGSO (the edge taps are tied to the graph)")

    def splitForward(self, x):
        assert len(x.shape) == 3                                    # :1918-1922
        batchSize = x.shape[0]
        assert x.shape[1] == self.F[0]
        assert x.shape[2] == self.N[0]
        y = self.EVGFL(self._reorder(x))                            # :1924-1926
        return self.MLP(y.reshape(batchSize, self.F[-1] * self.N[-1])), y


class GraphRecurrentNN(nn.Module):
    """Graph recurrent network -- architectures.py:4357-4672: HiddenState (z_t = sigma(A(S)x_t + B(S)z_{t-1})), an output
    GraphFilter on every z_t followed by rho, and a per-node readout.  Same constructor, sub-module names
    (``hiddenState``, ``outputState``, ``Readout``) and state_dict keys.  x: B x T x F x N -> B x T x dimReadout[-1] x N.

    The reference draws the initial state inside splitForward (``torch.randn`` on x's device, :4556); ``z0`` can be passed
    instead (superset) so that a run is reproducible across devices."""

    def __init__(self, dimInputSignals, dimOutputSignals, dimHiddenSignals, nFilterTaps, bias, nonlinearityHidden,
                 nonlinearityOutput, nonlinearityReadout, dimReadout, GSO):
        super().__init__()
        assert len(nFilterTaps) == 2                                # :4463
        self.F = dimInputSignals
        self.G = dimOutputSignals
        self.H = dimHiddenSignals
        self.K = nFilterTaps
        self.bias = bias
        self.sigma = nonlinearityHidden
        self.rho = nonlinearityOutput
        self.nonlinearityReadout = nonlinearityReadout
        self.dimReadout = dimReadout
        self._install(GSO)
        self.hiddenState = gml.HiddenState(self.F, self.H, self.K[0], nonlinearity=self.sigma, E=self.E, bias=self.bias)
        self.outputState = gml.GraphFilter(self.H, self.G, self.K[1], E=self.E, bias=self.bias)
        self.hiddenState.addGSO(self._gso)
        self.outputState.addGSO(self._gso)
        fc = []
        if len(self.dimReadout) > 0:                                # :4507-4519
            fc.append(nn.Linear(self.G, dimReadout[0], bias=self.bias))
            for l in range(len(dimReadout) - 1):
                fc.append(self.nonlinearityReadout())
                fc.append(nn.Linear(dimReadout[l], dimReadout[l + 1], bias=self.bias))
        self.Readout = nn.Sequential(*fc)

    def _install(self, GSO):
        if sp.issparse(GSO) or isinstance(GSO, (SparseGSO, list, tuple)):
            self._gso = SparseGSO.from_any(GSO)
            self.S = self._gso
        else:
            if isinstance(GSO, torch.Tensor):
                GSO = GSO.detach().cpu().numpy()
            GSO = np.asarray(GSO)
            assert len(GSO.shape) == 2 or len(GSO.shape) == 3       # :4479
            if len(GSO.shape) == 2:
                assert GSO.shape[0] == GSO.shape[1]
                GSO = GSO.reshape([1, GSO.shape[0], GSO.shape[1]])
            else:
                assert GSO.shape[1] == GSO.shape[2]
            self.S = torch.tensor(GSO)
            self._gso = SparseGSO.from_any(GSO)
        self.E = self._gso.E
        self.N = self._gso.N
        self.order = list(range(self.N))     # singleNodeForward looks nodes up in self.order, which the reference never sets

    def changeGSO(self, GSO):
        self._install(GSO)                                          # :4633-4660
        self.hiddenState.addGSO(self._gso)
        self.outputState.addGSO(self._gso)

    def splitForward(self, x, z0=None):
        assert len(x.shape) == 4                                    # :4540-4544
        B, T = x.shape[0], x.shape[1]
        assert x.shape[2] == self.F and x.shape[3] == self.N
        if z0 is None:
            z0 = torch.randn((B, self.H, self.N), device=x.device, dtype=x.dtype)       # :4556
        z, _ = self.hiddenState(x, z0)
        yOut = self.rho(self.outputState(z.reshape(B * T, self.H, self.N)))             # :4559-4561
        yOut = yOut.reshape(B, T, self.G, self.N)
        y = self.Readout(yOut.permute(0, 1, 3, 2))                  # B x T x N x dimReadout[-1]
        return y.permute(0, 1, 3, 2), yOut

    def forward(self, x, z0=None):
        output, _ = self.splitForward(x, z0)
        return output

    def singleNodeForward(self, x, nodes, z0=None):
        """B x T x dimReadout[-1] at one node per sample (:4576-4631)."""
        batchSize = x.shape[0]
        assert type(nodes) is int or type(nodes) is list or type(nodes) is np.ndarray
        nodes = np.full(batchSize, nodes, dtype=np.int64) if type(nodes) is int else np.asarray(nodes, dtype=np.int64)
        assert nodes.shape[0] == batchSize
        y = self.forward(x, z0)                                     # B x T x R x N
        idx = torch.as_tensor(nodes, device=y.device).view(batchSize, 1, 1, 1).expand(batchSize, y.shape[1], y.shape[2], 1)
        return torch.gather(y, 3, idx).squeeze(3)
