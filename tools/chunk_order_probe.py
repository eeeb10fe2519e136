#!/usr/bin/env python3
"""Does the ORDER of the K-1 hops matter at config 4?  hop-major (what gf_khop does: hop k of all B batch entries, then hop k+1: every
hop reads a 1.64 GB tap from HBM) against batch-entry-major (all K-1 hops of a chunk of C batch entries back to back: the tap a hop
reads was written a moment ago by the previous launch -- does it come out of the 256 MB Infinity Cache?).  Same kernels, same plan,
same bytes; only the launch order differs.   usage: chunk_order_probe.py [workload=cfg4] [chunks "8,16,32"] [key=val ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import torch
import bench
from alegnn_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
chunks = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "8,16,32").split(",")]
L = _lib.lib()
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    assert L.gf_tune(k.encode(), int(v)) == 0, k
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS[name])
w = bench.Workload(name, wl, dev, 0)
layer = w.module
B, N, W, K = wl["B"], layer.N, wl["G"], wl["K"]
plans = layer._gso.plans(dev)
st = torch.cuda.current_stream().cuda_stream
Z = torch.randn(K, B, N, W, device=dev)
tap = B * N * W * 4
ent = N * W * 4


def hop(k, b0, nb):
    src = Z.data_ptr() + (k - 1) * tap + b0 * ent
    dst = Z.data_ptr() + k * tap + b0 * ent
    _lib.check(L.gf_spmm_hop(plans[0], 0, src, dst, nb, W, st))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def hop_major():
    for k in range(1, K):
        hop(k, 0, B)


ref_ms = timed(hop_major)
ref = Z.clone()
print(f"{name}: hop-major ({K - 1} launches of B = {B}): {ref_ms:.3f} ms = {ref_ms / (K - 1):.3f} ms per hop", flush=True)
for C in chunks:
    def entry_major():
        for b0 in range(0, B, C):
            for k in range(1, K):
                hop(k, b0, min(C, B - b0))
    Z[1:].fill_(float("nan"))
    ms = timed(entry_major)
    print(f"{name}: batch-entry-major, chunks of {C:3d} ({(K - 1) * ((B + C - 1) // C)} launches): {ms:.3f} ms = {ms / (K - 1):.3f} ms per hop"
          f"  ({ms / ref_ms:.3f} x)  bitwise == hop-major: {bool(torch.equal(Z, ref))}", flush=True)
