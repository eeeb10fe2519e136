"""GSO ingest: turn whatever the caller hands to ``addGSO`` into host CSR + lazily-built device plans.

The reference keeps the graph shift operator as a dense ``[E, N, N]`` tensor
(alegnn/utils/graphML.py:2116-2123, alegnn/modules/architectures.py:192-197, 253-256) -- 40 GB at N = 1e5.
``SparseGSO`` accepts that dense tensor (reference-compatible) and, as a superset, scipy / torch sparse
matrices; the device side is the opaque plan of ``gf_plan_create`` (include/gfhip.h).
"""
from __future__ import annotations

import ctypes
import weakref

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib


def _csr_from_dense(a: np.ndarray) -> sp.csr_matrix:
    m = sp.csr_matrix(a)
    m.eliminate_zeros()
    return m


class SparseGSO:
    """E sparse N x N shift operators S_e (host CSR, float64 or float32) + per-device plans."""

    def __init__(self, mats):
        mats = [sp.csr_matrix(m, copy=True) for m in mats]
        assert len(mats) >= 1
        n = mats[0].shape[0]
        for m in mats:
            assert m.shape == (n, n), "every edge feature needs an N x N operator"   # graphML.py:2119-2122
            if m.dtype not in (np.float32, np.float64):
                raise TypeError(f"GSO dtype {m.dtype} is not supported (float32 / float64)")
            m.sum_duplicates()
            m.sort_indices()
        self.mats = mats
        self.E = len(mats)
        self.N = n
        self.nnz = [int(m.nnz) for m in mats]
        self._plans = {}          # device index -> (ctypes array of plan pointers, [raw pointers])
        self._finalizer = weakref.finalize(self, SparseGSO._destroy_all, self._plans)

    # ---- copying / pickling: the host CSR travels, device plans are rebuilt on first use (reference modules can be deep-copied and
    #      pickled -- copy.deepcopy(model), torch.save(model) -- so ours must be; ctypes handles and the finalizer cannot travel) ------
    def __getstate__(self):
        return {"mats": self.mats}

    def __setstate__(self, state):
        self.__init__(state["mats"])

    def __deepcopy__(self, memo):
        return SparseGSO(self.mats)

    # ---- derived operators (callers whose recursion uses a part of S: jARMA's Jacobi splitting, graphML.py:565-575) -----------------
    def transposed(self) -> "SparseGSO":
        return SparseGSO([m.T.tocsr() for m in self.mats])

    def diagonal(self) -> np.ndarray:
        """[E, N] diagonals."""
        return np.stack([np.asarray(m.diagonal()) for m in self.mats])

    def offdiagonal(self) -> "SparseGSO":
        out = []
        for m in self.mats:
            o = m.tolil(copy=True)
            o.setdiag(0)
            o = o.tocsr()
            o.eliminate_zeros()
            out.append(o)
        return SparseGSO(out)

    # ---- construction --------------------------------------------------------------------------------
    # CONTRACT of the dense-tensor cache below: a GSO tensor handed to the functional calls is not modified in place through views that
    # bypass torch's version counter (S.data[...] = v, numpy arrays sharing its memory): such writes do not change the key and the
    # cached operator would be used.  Build a new tensor (or call SparseGSO.from_any on a clone) after such an edit.
    _dense_cache = {}            # (data_ptr, version, shape, dtype, device) -> SparseGSO: functional calls LSIGF(h, S_dense, x) in a loop
    _DENSE_CACHE_MAX = 8         # (GatedGRNN per time step, jARMA) must not rebuild host CSR + device plans on every call

    @classmethod
    def from_any(cls, S) -> "SparseGSO":
        if isinstance(S, SparseGSO):
            return S
        if isinstance(S, torch.Tensor) and S.layout == torch.strided:
            key = (S.data_ptr(), S._version, tuple(S.shape), S.dtype, str(S.device))
            hit = cls._dense_cache.get(key)
            if hit is not None and hit[0]() is S:                # same tensor object, unmodified since
                return hit[1]
            gso = cls._from_dense_tensor(S)
            if len(cls._dense_cache) >= cls._DENSE_CACHE_MAX:
                cls._dense_cache.pop(next(iter(cls._dense_cache)))
            cls._dense_cache[key] = (weakref.ref(S), gso)
            return gso
        return cls._from_other(S)

    @classmethod
    def _from_dense_tensor(cls, S) -> "SparseGSO":
        return cls._from_other(S.detach().cpu().numpy())

    @classmethod
    def _from_other(cls, S) -> "SparseGSO":
        if isinstance(S, torch.Tensor):
            if S.layout != torch.strided:                       # torch sparse COO / CSR, 2-D
                Sc = S.detach().cpu().to_sparse_coo().coalesce()
                assert Sc.dim() == 2, "sparse torch GSO must be 2-D (one edge feature)"
                idx = Sc.indices().numpy()
                return cls([sp.csr_matrix((Sc.values().numpy(), (idx[0], idx[1])), shape=tuple(Sc.shape))])
            S = S.detach().cpu().numpy()                        # (strided tensors arrive here as numpy already)
        if sp.issparse(S):
            return cls([S])
        if isinstance(S, (list, tuple)):
            return cls([m if sp.issparse(m) else _csr_from_dense(np.asarray(m)) for m in S])
        S = np.asarray(S)
        assert S.ndim == 2 or S.ndim == 3                       # architectures.py:192
        if S.ndim == 2:
            S = S[None]
        assert S.shape[1] == S.shape[2]                         # architectures.py:197 / graphML.py:2122
        if S.dtype not in (np.float32, np.float64):
            S = S.astype(np.float64)
        return cls([_csr_from_dense(S[e]) for e in range(S.shape[0])])

    # ---- reference-compatible views -----------------------------------------------------------------------
    @property
    def shape(self):
        return (self.E, self.N, self.N)

    def to_dense(self, dtype=torch.float32) -> torch.Tensor:
        """Dense [E,N,N] tensor (small N only) -- what the reference stores as ``GraphFilter.S``."""
        return torch.stack([torch.from_numpy(m.toarray()).to(dtype) for m in self.mats])

    def __repr__(self):
        return f"SparseGSO(E={self.E}, N={self.N}, nnz={self.nnz})"

    # ---- device plans ------------------------------------------------------------------------------------
    def plans(self, device: torch.device):
        """ctypes ``gf_plan*[E]`` for ``device`` (built on first use; one host->device upload per device)."""
        if device.type != "cuda":
            raise RuntimeError(f"alegnn_amd runs on MI355X only (HIP device required), got device '{device}'")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        hit = self._plans.get(idx)
        if hit is not None:
            return hit[0]
        L = _lib.lib()
        raw = []
        with torch.cuda.device(idx):
            for m in self.mats:
                rowptr = np.ascontiguousarray(m.indptr, dtype=np.int32)
                col = np.ascontiguousarray(m.indices, dtype=np.int32)
                val = np.ascontiguousarray(m.data)
                out = ctypes.c_void_p()
                rc = L.gf_plan_create(self.N, m.nnz, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data,
                                      1 if val.dtype == np.float64 else 0, 0, ctypes.byref(out))
                if rc != 0:
                    for p in raw:
                        L.gf_plan_destroy(p)
                    _lib.check(rc, "gf_plan_create")
                raw.append(out.value)
        arr = (ctypes.c_void_p * self.E)(*raw)
        self._plans[idx] = (arr, raw)
        return arr

    @staticmethod
    def _destroy_all(plans):
        try:
            L = _lib.lib()
        except Exception:
            return
        for _, raw in plans.values():
            for p in raw:
                L.gf_plan_destroy(p)
        plans.clear()


class EdgePattern:
    """Sparsity pattern of the edge-variant filter taps k >= 1 for ONE edge feature (host CSR of ones, sorted columns)
    + per-device ``gf_ev_plan`` handles.  Built by ``EdgeVariantGF.addGSO`` following graphML.py:2617-2643:
    ``(|S_e| + I > zeroTolerance)`` restricted to entries whose row < M or column < M (hybrid mask)."""

    ZERO_TOLERANCE = 1e-9                                       # graphML.py:27

    def __getstate__(self):
        return {"indptr": self.indptr, "indices": self.indices, "N": self.N}

    def __setstate__(self, state):
        n = state["N"]
        self.__init__(sp.csr_matrix((np.ones(len(state["indices"]), dtype=np.float32), state["indices"], state["indptr"]), shape=(n, n)))

    def __deepcopy__(self, memo):
        n = self.N
        return EdgePattern(sp.csr_matrix((np.ones(self.nnzp, dtype=np.float32), self.indices, self.indptr), shape=(n, n)))

    def __init__(self, pattern: sp.csr_matrix):
        pattern = sp.csr_matrix(pattern)
        pattern.sum_duplicates()
        pattern.sort_indices()
        self.N = pattern.shape[0]
        assert pattern.shape == (self.N, self.N)
        self.indptr = np.ascontiguousarray(pattern.indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(pattern.indices, dtype=np.int32)
        self.nnzp = int(self.indices.shape[0])
        self.rows = np.repeat(np.arange(self.N, dtype=np.int64), np.diff(self.indptr))
        self.cols = self.indices.astype(np.int64)
        self._plans = {}
        self._finalizer = weakref.finalize(self, EdgePattern._destroy_all, self._plans)

    @classmethod
    def from_gso(cls, S2d, M: int) -> "EdgePattern":
        S2d = sp.csr_matrix(S2d)
        N = S2d.shape[0]
        P = sp.coo_matrix((abs(S2d) + sp.identity(N, format="csr")) > cls.ZERO_TOLERANCE)
        keep = ((P.row < M) | (P.col < M)) & (P.data != 0)
        return cls(sp.csr_matrix((np.ones(int(keep.sum()), dtype=np.float32), (P.row[keep], P.col[keep])), shape=(N, N)))

    def plan(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError(f"alegnn_amd runs on MI355X only (HIP device required), got device '{device}'")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        hit = self._plans.get(idx)
        if hit is not None:
            return hit
        L = _lib.lib()
        out = ctypes.c_void_p()
        with torch.cuda.device(idx):
            _lib.check(L.gf_ev_plan_create(self.N, self.nnzp, self.indptr.ctypes.data, self.indices.ctypes.data,
                                           ctypes.byref(out)), "gf_ev_plan_create")
        self._plans[idx] = out.value
        return out.value

    @staticmethod
    def _destroy_all(plans):
        try:
            L = _lib.lib()
        except Exception:
            return
        for p in plans.values():
            L.gf_ev_plan_destroy(p)
        plans.clear()
