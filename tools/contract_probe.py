#!/usr/bin/env python3
"""Time the node-major filter-bank contraction (gf_contract, forward bank) and the one-pass backward at a bench.py workload's shape, alone on the stream.
usage: GFHIP_EXPERIMENTS=1 [GFHIP_LIB=...] python tools/contract_probe.py [workload] [iters]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import torch
import bench
from alegnn_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = _lib.lib()
wl = bench.WORKLOADS[name]
B, N, G, F, K = wl["B"], wl["N"], wl["G"], wl["F"], wl["K"]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
Z = torch.randn(K, B, N, G, device=dev)
h = torch.randn(F, 1, K, G, device=dev) * 0.1
bias = torch.randn(F, device=dev)
y = torch.empty(B, F, N, device=dev)
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
ms = timed(lambda: _lib.check(L.gf_contract(Z.data_ptr(), h.data_ptr(), bias.data_ptr(), y.data_ptr(), B, N, N, G, F, 1, K, 0, st)))
gb = (K * B * N * G + B * N * F) * 4 / 1e9
print(f"gf_contract {name} B={B} N={N} {G}->{F} K={K} lib={os.environ.get('GFHIP_LIB', 'shipped')}: {ms:.4f} ms  ({gb:.2f} GB algorithmic -> {gb / ms:.2f} TB/s)", flush=True)
