#!/usr/bin/env python3
"""What happens to the fused K-hop chain (ONE cooperative launch that needs every CU) when another stream's kernels are resident: the one-rank
RCCL all-reduce of the gradient bucket (the collective a data-parallel step overlaps with its backward), and a kernel that holds a CU for 1 ms.
Records the chain's time alone / beside each, the all-reduce time, that the results stay bit for bit the same and that nothing was repaired.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/coop_beside_rccl.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-neural-networks_amd")]
import torch, torch.distributed as dist
import bench
from alegnn_amd import _lib
dist.init_process_group("nccl", rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
L = _lib.lib()
wl = dict(bench.WORKLOADS["cfg4"])
w = bench.Workload("cfg4", wl, dev, 0)
B, N, W, K = wl["B"], w.module.N, wl["G"], wl["K"]
plans = w.module._gso.plans(dev)
main = torch.cuda.current_stream(dev)
Z = torch.empty(K, B, N, W, device=dev)
Z[0].normal_()
bucket = torch.randn(5 * 32 * 32 + 32, device=dev)          # the layer's gradient bucket (20.6 KB)


def chain(n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(n):
        _lib.check(L.gf_khop(plans, 1, 0, Z.data_ptr(), B, W, K, main.cuda_stream))
    e1.record(main)
    e1.synchronize()
    return e0.elapsed_time(e1) / n


assert L.gf_spmm_hop_kernel(plans[0], 0, B, W) == 1
chain(2)
ref = Z.clone()
alone = [chain() for _ in range(3)]
side = torch.cuda.Stream(device=dev)
# (a) all-reduces enqueued back to back on a side stream while the chains run
with torch.cuda.stream(side):
    dist.all_reduce(bucket)
side.synchronize()
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
    a0.record(side)
    for _ in range(200):
        dist.all_reduce(bucket)
    a1.record(side)
beside_rccl = chain()
side.synchronize()
ar_ms = a0.elapsed_time(a1) / 200
same_a = bool(torch.equal(Z, ref))
# (b) a kernel that holds one CU for ~1 ms per launch, 20 launches back to back on the side stream
with torch.cuda.stream(side):
    for _ in range(20):
        torch.cuda._sleep(2_000_000)
beside_spin = chain()
side.synchronize()
same_b = bool(torch.equal(Z, ref))
f, on = ctypes.c_uint32(0), ctypes.c_int32(0)
L.gf_msweep_status(ctypes.byref(f), ctypes.byref(on))
print(f"fused chain alone: {min(alone):.3f} ms per call (runs {['%.3f' % a for a in alone]})")
print(f"beside 200 one-rank RCCL all-reduces of {bucket.numel() * 4} bytes on a side stream: {beside_rccl:.3f} ms per call, all-reduce {ar_ms * 1e3:.1f} us each, bitwise same: {same_a}")
print(f"beside 20 x ~1 ms single-wave kernels on a side stream: {beside_spin:.3f} ms per call (the cooperative grid waits for the occupied CU), bitwise same: {same_b}")
print(f"status flags {f.value}, fusion on {on.value}")
dist.destroy_process_group()
