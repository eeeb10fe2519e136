"""Shared helpers for the test-suite (golden fixture loading, tolerances)."""
import ast
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Stated tolerances (SURVEY.md section 8c / BASELINE.md): the reference itself, fp32 vs fp64,
# differs by 1e-7..1.2e-6 relative to max|y| on these shapes.
FWD_RTOL = 1e-5      # max|y - y_ref64| <= FWD_RTOL * max|y_ref64|
GRAD_RTOL = 1e-4     # max|g - g_ref64| <= GRAD_RTOL * max|g_ref64|


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")))


def load(path):
    d = dict(np.load(path, allow_pickle=False))
    shape = tuple(int(v) for v in d["S_shape"])
    S = np.zeros(shape, dtype=np.float64)
    S[d["S_e"], d["S_r"], d["S_c"]] = d["S_v"]
    d["S"] = S
    if "cfg" in d:
        d["cfg"] = ast.literal_eval(str(d["cfg"]))
    return d


def relerr(a, ref):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    denom = np.max(np.abs(ref))
    return float(np.max(np.abs(a - ref)) / (denom if denom > 0 else 1.0))


def case_id(path):
    return os.path.splitext(os.path.basename(path))[0]


class ArrayData:
    """Minimal data object with the interface the trainers use (reference dataTools.py:172-219, :321-341): samples held
    as tensors, ``getSamples(split[, indices])``, ``evaluate`` = classification error rate.  Built from a trainer_*.npz."""

    def __init__(self, d, dtype):
        import torch
        self.samples = {s: (torch.tensor(d["x_" + s]).to(dtype), torch.tensor(d["y_" + s])) for s in ("train", "valid", "test")}
        self.nTrain = self.samples["train"][0].shape[0]
        self.dtype = dtype

    def getSamples(self, samplesType, *args):
        x, y = self.samples[samplesType]
        if len(args) == 1:
            x, y = x[args[0]], y[args[0]]
        return x, y

    def evaluate(self, yHat, y, tol=1e-9):
        import torch
        wrong = torch.sum(torch.abs(torch.argmax(yHat, dim=1) - y) > tol)
        return wrong.to(self.dtype) / len(y)


def large_gfilter_inputs(N, B, G, Nin, seed, kind=0):
    """Inputs of tests/golden/large/gfilter_*.npz, regenerated from the seed (numpy's legacy RandomState is stable across versions): the GSO as scipy CSR
    (kind 0: an undirected graph of mean degree ~6 with per-edge weights, no self loops; kind 1: directed with power-law row lengths), x [B,G,Nin], dy [B,F,Nin] is drawn by the caller's F.  The fixture
    stores the checksums that pin the regeneration (nnz, sum of the weights, sum of x)."""
    import scipy.sparse as sp
    rng = np.random.RandomState(seed)
    if kind == 1:        # directed, row lengths with a power-law tail (a few rows of hundreds to thousands of entries: the sweep's hub rows), weighted
        deg = np.minimum(N // 8, (3.0 / np.sqrt(np.maximum(rng.uniform(size=N), 1e-9))).astype(np.int64))
        r = np.repeat(np.arange(N), deg)
        A = sp.csr_matrix((np.ones(r.size), (r, rng.randint(0, N, size=r.size))), shape=(N, N))
        A.sum_duplicates()
        A.data[:] = rng.uniform(0.2, 1.0, size=A.data.size) / 24.0
        A = sp.csr_matrix(A)
    else:
        r = np.repeat(np.arange(N), 3)
        c = rng.randint(0, N, size=r.size)
        A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(N, N))
        A = sp.triu(A + A.T, k=1).tocsr()
        A.data[:] = rng.uniform(0.05, 0.2, size=A.data.size)
        A = sp.csr_matrix(A + A.T)
    x = rng.randn(B, G, Nin)
    return A, x
