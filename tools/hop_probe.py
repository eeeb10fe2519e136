#!/usr/bin/env python3
"""Launch the dominant kernel of a bench.py workload a few times, nothing else on the stream after set-up: the target of
`rocprofv3 --pmc ...` / `--kernel-trace --stats` (tools/pmc_collect.sh).  Usage: hop_probe.py <workload> [iters] [key=val ...]
An argument `k=v+k=v+...` is a VARIANT: the node-major K-hop chain (one gf_khop call on a tap stack, as the layer runs it; PROBE_SINGLE=1: one
hop out of and into separate buffers) is timed once per variant on the same plan (A/B on one box, one process)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import torch
import bench
from alegnn_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = _lib.lib()
variants = [a for a in sys.argv[3:] if "+" in a or a.startswith("v:")]
for kv in sys.argv[3:]:
    if kv in variants:
        continue
    k, v = kv.split("=")
    assert L.gf_tune(k.encode(), int(v)) == 0, k
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS[name])
if os.environ.get("PROBE_B"):  # same workload at another batch size (row width of the EVGF gathers = 4*B bytes)
    wl["B"] = int(os.environ["PROBE_B"])
if os.environ.get("PROBE_GRAPH") == "band":   # same N / degree, every neighbour within +-64 rows: the gather panel of a row block is L2-resident
    import numpy as np, scipy.sparse as sp
    from alegnn_amd import graphgen
    def band(N, avg_degree=10.0, seed=0, **kw):
        rng = np.random.RandomState(seed)
        r = np.repeat(np.arange(N), int(avg_degree) // 2)
        c = np.clip(r + rng.randint(1, 65, size=r.size), 0, N - 1)
        A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(N, N))
        A = ((A + A.T) > 0).astype(np.float64)
        A.setdiag(0); A.eliminate_zeros()
        return sp.csr_matrix(A / 20.0)
    graphgen.er = band
if os.environ.get("PROBE_GRAPH") == "powerlaw":   # same N, mean degree ~10, row lengths with a Barabasi-Albert tail (P(len > k) = (5 / k)^2), symmetrised
    import numpy as np, scipy.sparse as sp
    from alegnn_amd import graphgen
    def powerlaw(N, avg_degree=10.0, seed=0, **kw):
        rng = np.random.RandomState(seed)
        deg = np.minimum(N // 8, (0.25 * avg_degree / np.sqrt(np.maximum(rng.uniform(size=N), 1e-9))).astype(np.int64))
        r = np.repeat(np.arange(N), deg)
        A = sp.csr_matrix((np.ones(r.size), (r, rng.randint(0, N, size=r.size))), shape=(N, N))
        A = ((A + A.T) > 0).astype(np.float64)
        A.setdiag(0); A.eliminate_zeros()
        return sp.csr_matrix(A / 40.0)
    graphgen.er = powerlaw
w = bench.Workload(name, wl, dev, 0)
st = torch.cuda.current_stream().cuda_stream
if wl["kind"] == "evgf":
    with torch.no_grad():
        w.module(w.x.detach())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            w.module(w.x.detach())
        e1.record()
        torch.cuda.synchronize()
        print(f"evgf forward B={wl['B']}: {e0.elapsed_time(e1) / iters:.3f} ms")
else:
    layer = w.module if wl["kind"] == "filter" else w.module.GFL[3]
    B, N = wl["B"], layer.N
    W = wl["G"] if wl["kind"] == "filter" else wl["dimF"][1]
    W = int(os.environ.get("PROBE_W", W))      # same graph, another row width (the hop does not depend on the filter bank)
    K = wl["K"] if wl["kind"] == "filter" else wl["K"][1]
    plans = layer._gso.plans(dev)
    OP = int(os.environ.get("PROBE_OP", 0))
    if hasattr(L, "gf_debug_msweep_info"):
        info = (ctypes.c_int32 * 8)()
        if L.gf_debug_msweep_info(plans[0], OP, info) == 0:
            print("sweep image {sets, passes, rounds, fill x 1000, hub rows, split hub rows, hub limit, split limit}:", list(info), flush=True)
    if L.gf_lsigf_pipeline(plans, 1, W, W, K) == 2:
        Z = torch.randn(K, B * W // 4, N, 4, device=dev)
        for _ in range(iters):
            _lib.check(L.gf_khop_panel(plans, 1, 0, Z.data_ptr(), B, W, K, st))
    else:
        X0 = torch.randn(B, N, W, device=dev); X1 = torch.empty_like(X0)
        ms = ctypes.c_float()
        ref = None
        if not os.environ.get("PROBE_SINGLE"):   # the K-1 hops of one gf_khop call on a tap stack, as the layer runs them: ms per hop
            Z = torch.empty(K, B, N, W, device=dev)
            Z[0].copy_(X0)
            for var in (variants or [""]):
                for kv in var.replace("v:", "").split("+"):
                    if kv:
                        k, v = kv.split("=")
                        assert L.gf_tune(k.encode(), int(v)) == 0, k
                Z[1:].fill_(float("nan"))
                _lib.check(L.gf_time_khop(plans, 1, OP, Z.data_ptr(), B, W, K, max(iters, 3), st, ctypes.byref(ms)))
                torch.cuda.synchronize()
                same = "" if ref is None else f"  bitwise == first variant: {bool(torch.equal(ref, Z[1:]))}"
                if ref is None:
                    ref = Z[1:].clone()
                print(f"khop chain {name} K={K} nnz={w.nnz} {var}: {ms.value / (K - 1):.4f} ms per hop ({ms.value:.4f} per call){same}", flush=True)
            variants = ["__none__"]
        for var in ([] if variants == ["__none__"] else (variants or [""])):
            for kv in var.replace("v:", "").split("+"):
                if kv:
                    k, v = kv.split("=")
                    assert L.gf_tune(k.encode(), int(v)) == 0, k
            X1.fill_(float("nan"))
            _lib.check(L.gf_time_spmm_hop(plans[0], 0, X0.data_ptr(), X1.data_ptr(), B, W, max(iters, 3), st, ctypes.byref(ms)))
            torch.cuda.synchronize()
            same = "" if ref is None else f"  bitwise == first variant: {bool(torch.equal(ref, X1))}"
            if ref is None:
                ref = X1.clone()
            print(f"spmm hop {name} graph={os.environ.get('PROBE_GRAPH', 'default')} nnz={w.nnz} {' '.join(a for a in sys.argv[3:] if a not in variants)} {var}: {ms.value:.4f} ms{same}", flush=True)
torch.cuda.synchronize()
