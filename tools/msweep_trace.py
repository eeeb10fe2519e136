#!/usr/bin/env python3
"""Phase times of spmm_msweep_kernel (first wave of every XCD, s_memtime stamps): per (batch entry, hop) the time from the entry's start to
the start of its round loop (zeroing, the previous stores' drain behind the first vmcnt(0), the first entry loads), the round loop, the
store issue, the XCD barrier.  usage: GFHIP_EXPERIMENTS=1 python tools/msweep_trace.py [key=val ...]   (knobs as tools/hop_probe.py)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import numpy as np, torch
import bench
from alegnn_amd import _lib
L = _lib.lib()
for kv in ["spmm_algo=5", "spmm_trace=1"] + sys.argv[1:]:
    k, v = kv.split("=")
    assert L.gf_tune(k.encode(), int(v)) == 0, k
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["cfg4"])
w = bench.Workload("cfg4", wl, dev, 0)
B, N, W, K = wl["B"], w.module.N, wl["G"], wl["K"]
plans = w.module._gso.plans(dev)
st = torch.cuda.current_stream().cuda_stream
Z = torch.randn(K, B, N, W, device=dev)
for _ in range(2):
    _lib.check(L.gf_khop(plans, 1, 0, Z.data_ptr(), B, W, K, st))
torch.cuda.synchronize()
buf = np.zeros((8, 64, 8), dtype=np.uint64)
L.gf_debug_msweep_trace.restype = ctypes.c_int
assert L.gf_debug_msweep_trace(ctypes.c_void_p(buf.ctypes.data)) == 0
raw = buf.astype(np.float64)
n0 = int((raw[0, :, 0] > 0).sum())
if n0 > 1 and raw[0, n0 - 1, 5] > raw[0, 0, 5]:   # s_memrealtime runs at 100 MHz: the s_memtime tick rate under this load
    print(f"s_memtime ticks per microsecond (XCD 0, {n0} slots): {(raw[0, n0 - 1, 4] - raw[0, 0, 4]) / (raw[0, n0 - 1, 5] - raw[0, 0, 5]) * 100.0:.1f}")
t = buf.astype(np.float64) / 2200.0         # s_memtime ticks = shader clocks (~2.2 GHz under this load) -> microseconds
ok = buf[:, :, 0] > 0
t[:, 1:, 0] = t[:, :-1, 4]                  # a body starts where the previous one's barrier opened; the first one at its loop start (its set-up is not stamped)
t[:, 0, 0] = t[:, 0, 1]
names = ["start->loop (zero, drain, entry loads)", "round loop", "store issue", "barrier"]
d = np.stack([t[:, :, i + 1] - t[:, :, i] for i in range(4)], -1)
print("knobs:", " ".join(sys.argv[1:]))
for x in range(8):
    n = int(ok[x].sum())
    print(f"XCD {x}: {n} (entry, hop) slots; first start {t[x, 0, 0] - t[ok][:, 0].min():8.1f} us;", "  ".join(f"{names[i].split(' ')[0]} {d[x, :n, i].mean():7.2f}" for i in range(4)), f" total {(t[x, n - 1, 4] - t[x, 0, 0]):9.1f} us")
print("mean over XCDs [us]:", {names[i]: round(float(d[ok][:, i].mean()), 2) for i in range(4)}, "sum", round(float(d[ok].sum(-1).mean()), 2))
print("first 6 slots of XCD 0:", np.round(d[0, :6], 1).tolist())
