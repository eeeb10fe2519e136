#!/usr/bin/env python3
"""Phase timeline of the chain kernel (needs the trace build: make -C graph-neural-networks_amd variant NAME=trace EXTRA=-DGF_CHAIN_TRACE,
run with GFHIP_EXPERIMENTS=1 GFHIP_LIB=.../libgfhip_trace.so).  Prints, for the second panel of the first 8 workgroups and every hop,
s_memtime deltas (100 MHz ticks -> us): gatherer wave 0 = [gather, wait B1, rewrite, wait B2], storer = [store issue, wait B1]."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import numpy as np, torch
from alegnn_amd import SparseGSO, _lib, graphgen
L = _lib.lib()
L.gf_chain_trace_set.argtypes = [ctypes.c_void_p]
N, B, W, K = 10_000, 256, 32, 5
deg = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
dev = torch.device("cuda:0")
A = graphgen.sbm(N, avg_degree=deg, seed=0)
gso = SparseGSO([A]); plans = gso.plans(dev)
P = B * W // 4
Z = torch.randn(K, P, N, 4, device=dev)
buf = torch.zeros(8 * 8 * 2 * 8, dtype=torch.int64, device=dev)
assert L.gf_tune(b"panel_chain", 2) == 0
st = torch.cuda.current_stream().cuda_stream
_lib.check(L.gf_khop_panel(plans, 1, 0, Z.data_ptr(), B, W, K, st)); torch.cuda.synchronize()
assert L.gf_chain_trace_set(buf.data_ptr()) == 0
_lib.check(L.gf_khop_panel(plans, 1, 0, Z.data_ptr(), B, W, K, st)); torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(8, 8, 2, 8).astype(np.float64) / 100.0      # s_memtime: 100 MHz constant clock -> us
for blk in range(8):
    for h in range(K - 1):
        g, s = t[blk, h, 0], t[blk, h, 1]
        print(f"wg {blk} hop {h}: gather {g[1]-g[0]:6.2f}  waitB1 {g[2]-g[1]:6.2f}  rewrite {g[3]-g[2]:5.2f}  waitB2 {g[4]-g[3]:5.2f} | "
              f"storer issue {s[1]-s[0]:6.2f}  waitB1 {s[2]-s[1]:6.2f} | hop total {g[4]-g[0]:6.2f} us")
