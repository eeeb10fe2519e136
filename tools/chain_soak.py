#!/usr/bin/env python3
"""Soak of the fused K-hop chain's hand-over protocol (census + XCD barriers, gf_msweep.hip): many cooperative launches from two streams of one process,
next to a third stream that keeps compute units busy in bursts, optionally in several processes at once (SOAK_PROCS); every launch's tap stack is compared
bit for bit with the per-hop SELL-8 result, the status word (launches abandoned and repaired) is read at the end.
usage: [SOAK_PROCS=2] python tools/chain_soak.py [seconds] [N] [B]       (product configuration: no gf_tune)"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
procs = int(os.environ.get("SOAK_PROCS", "1"))
if procs > 1 and not os.environ.get("SOAK_CHILD"):
    env = dict(os.environ, SOAK_CHILD="1")
    ps = [subprocess.Popen([sys.executable, __file__, *sys.argv[1:]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(procs)]
    outs = [p.communicate()[0] for p in ps]
    for i, o in enumerate(outs):
        print(f"[process {i}] " + "\n".join(l for l in o.splitlines() if l.startswith("soak")))
    sys.exit(max(p.returncode for p in ps))
import numpy as np, torch
from alegnn_amd import _lib, graphgen
from alegnn_amd.gso import SparseGSO
L = _lib.lib()
dev = torch.device("cuda:0")
K, W = 5, 32
gso = SparseGSO([graphgen.er(N, avg_degree=10.0, seed=0)])
plans = gso.plans(dev)
assert L.gf_spmm_hop_kernel(plans[0], 0, B, W) == 1
x0 = torch.randn(B, N, W, device=dev)
ref = torch.empty(K, B, N, W, device=dev); ref[0].copy_(x0)
st0 = torch.cuda.current_stream().cuda_stream
for k in range(1, K):                                  # the reference: one launch per hop (no hand-over inside a launch)
    _lib.check(L.gf_spmm_hop(plans[0], 0, ref[k - 1].data_ptr(), ref[k].data_ptr(), B, W, st0))
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
noise = torch.cuda.Stream()
Zs = [torch.empty(K, B, N, W, device=dev) for _ in streams]
launches = wrong = 0
t0 = time.time()
while time.time() - t0 < secs:
    for s, Z in zip(streams, Zs):
        with torch.cuda.stream(s):
            Z[1:].fill_(float("nan")); Z[0].copy_(x0)
            for _ in range(5):
                _lib.check(L.gf_khop(plans, 1, 0, Z.data_ptr(), B, W, K, s.cuda_stream))
                launches += 1
    with torch.cuda.stream(noise):
        if launches % 40 < 20:
            torch.cuda._sleep(2_000_000)               # ~1 ms of a kernel that occupies CUs
    torch.cuda.synchronize()
    for Z in Zs:
        wrong += int(not torch.equal(Z, ref))
flags, on = ctypes.c_uint32(0), ctypes.c_int32(0)
_lib.check(L.gf_msweep_status(ctypes.byref(flags), ctypes.byref(on)))
print(f"soak: {launches} fused chains (N={N} B={B} K={K}) from two streams in {time.time() - t0:.0f} s; tap stacks that differ from the per-hop result: {wrong}; "
      f"status flags {flags.value} (0 = no launch was abandoned), fusion on: {on.value}", flush=True)
sys.exit(1 if wrong else 0)
