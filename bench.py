#!/usr/bin/env python3
"""bench.py -- edges*taps/sec of one GraphFilter forward+backward (BASELINE.json metric) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload = "cfg2"): BASELINE.json configs[1] -- synthetic SBM N=10k, nnz~100k, batch 256 per GPU,
K=5, F 32->32, fp32.  A step = one pass of the hot path over one batch: GraphFilter forward + backward (dx, dh, db)
through the C ABI; for N>1 plus the ONE bucketed RCCL all-reduce of the tap/bias gradients (batch-DP, weak scaling:
per-GPU batch fixed).  Inputs are resident in HBM before the timed region.  value = B_global * nnz * K / t_step.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (one hop of the K-hop SpMM, in the pipeline the layer runs): achieved = algorithmic
                  bytes per launch (2*B*N*G*4 + nnz*8 + (N+1)*4) / average launch time measured here with
                  HIP events on the launch stream (gf_time_spmm_hop[_panel]); peak = 8 TB/s HBM3E; traffic = PMC HBM bytes
                  per launch from profiles/*_pmc.json for this workload and kernel (rocprofv3 --pmc in its own pass), else null.
  cpu_baseline -- the reference's CPU path (oracle restatement of graphML.py:152-175: dense S, torch.matmul loop, cat,
                  permute) timed on this box's host cores on a bounded batch sample; rank 0, N=1 only.
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "graph-neural-networks_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: graph model, N, avg degree, per-GPU batch, G, F, K
    "cfg2": dict(model="sbm", N=10_000, deg=10.0, B=256, G=32, F=32, K=5,
                 desc="SBM N=10k nnz~100k, batch 256/GPU, K=5, F 32->32 (BASELINE configs[1])"),
    "cfg4": dict(model="er", N=100_000, deg=10.0, B=128, G=32, F=32, K=5,
                 desc="ER N=100k nnz~1M, batch 128/GPU (1024 over 8 GPUs), K=5, F=32 (BASELINE configs[3])"),
    "tiny": dict(model="sbm", N=1000, deg=10.0, B=32, G=32, F=32, K=5, desc="plumbing check"),
}
HBM_PEAK_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=64, help="batch entries of the workload timed on the CPU")
    ap.add_argument("--detail", action="store_true", help="per-kernel timings to stderr")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2], help="0 auto | 1 node-major (L2 gathers) | 2 column panels (LDS gathers)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="gf_tune knob for experiments (repeatable)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs; there is no CPU fallback"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = world > 1 or "RANK" in os.environ       # under torch.distributed.run even N = 1 goes through RCCL
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from alegnn_amd import _lib, graphgen, parallel
    from alegnn_amd.utils import graphML as gml

    if args.pipeline:
        _lib.check(_lib.lib().gf_tune(b"pipeline", args.pipeline), "gf_tune pipeline")
    for kv in args.tune:
        key, val = kv.split("=")
        _lib.check(_lib.lib().gf_tune(key.encode(), int(val)), "gf_tune " + key)
    wl = WORKLOADS[args.workload]
    N, B, G, F, K = wl["N"], wl["B"], wl["G"], wl["F"], wl["K"]
    A = (graphgen.sbm if wl["model"] == "sbm" else graphgen.er)(N, avg_degree=wl["deg"], seed=0)
    nnz = int(A.nnz)
    torch.manual_seed(0)
    layer = gml.GraphFilter(G, F, K, 1, True)
    layer.addGSO(A)
    layer.to(dev)
    parallel.broadcast_parameters(layer)
    bucket = parallel.GradBucket(layer.parameters())
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.randn(B, G, N, device=dev, generator=gen).requires_grad_(True)     # synthetic signals, resident in HBM
    dy = torch.randn(B, F, N, device=dev, generator=gen)

    def step():
        bucket.zero_()
        x.grad = None
        y = layer(x)
        y.backward(dy)
        bucket.allreduce_mean()

    def sync_all():
        if distributed:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # the slowest rank defines the step time
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = (B * world) * nnz * K / (elapsed / args.steps)

    # ---- roofline of the dominant kernel: one SpMM hop of the pipeline the layer actually runs, HIP events on the
    #      launch stream (gf_time_spmm_hop*: hipEventRecord on that stream around `iters` back-to-back launches) -----------
    L = _lib.lib()
    plans = layer._gso.plans(dev)
    pipe = L.gf_lsigf_pipeline(plans, 1, G, F, K)
    ms = np.zeros(1, dtype=np.float32)
    stream = torch.cuda.current_stream().cuda_stream
    msp = ms.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    if pipe == 2:                                            # column panels, gathers from LDS
        X0 = torch.randn(B * G // 4, N, 4, device=dev)
        X1 = torch.empty_like(X0)
        _lib.check(L.gf_time_spmm_hop_panel(plans[0], 0, X0.data_ptr(), X1.data_ptr(), B * G // 4, 20, stream, msp))
        kname = "spmm_panel_kernel (one hop, op=S^T, column panels through LDS)"
    else:                                                    # node-major, gathers through L2
        X0 = torch.randn(B, N, G, device=dev)
        X1 = torch.empty_like(X0)
        _lib.check(L.gf_time_spmm_hop(plans[0], 0, X0.data_ptr(), X1.data_ptr(), B, G, 20, stream, msp))
        kname = "spmm_sell_kernel (one hop, op=S^T, node-major through L2)"
    hop_ms = float(ms[0])
    hop_bytes = 2 * B * N * G * 4 + nnz * 8 + (N + 1) * 4    # read X once, write X once, read the CSR once
    achieved = hop_bytes / (hop_ms * 1e-3) / 1e9
    traffic = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            pm = json.load(open(f))
            if pm.get("workload") == args.workload and pm.get("kernel", "").split("<")[0] == kname.split(" ")[0]:
                traffic = pm.get("hbm_bytes_per_launch")
        except Exception:
            pass
    roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 1),
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    algorithmic_bytes=hop_bytes, launch_ms=round(hop_ms, 5), pipeline=int(pipe))

    detail = None
    if args.detail and rank == 0:
        detail = kernel_breakdown(L, layer, plans, x.detach(), dy, B, N, G, F, K, dev)
        log("breakdown_ms", json.dumps(detail))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(A, layer, x.detach(), nnz, K, min(args.cpu_sample, B))

    if rank == 0:
        out = dict(metric="edges*taps/sec (GraphFilter fwd+bwd)", value=value, unit="edges*taps/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload=args.workload, description=wl["desc"], graph=wl["model"], N=N, nnz=nnz,
                               batch_per_gpu=B, global_batch=B * world, G=G, F=F, K=K, E=1,
                               parallelism=f"batch-dp{world}", grad_bucket_bytes=bucket.nbytes()),
                   roofline=roofline, cpu_baseline=cpu)
        if detail:
            out["breakdown_ms"] = detail
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


def kernel_breakdown(L, layer, plans, x, dy, B, N, G, F, K, dev):
    """Per-building-block time (torch events on the launch stream = torch's current stream), median of 10."""
    from alegnn_amd import _lib
    T = K
    Z = torch.empty((T, B, N, G), device=dev)
    P = torch.empty((T, B, N, F), device=dev)
    y = torch.empty((B, F, N), device=dev)
    dx = torch.empty((B, G, N), device=dev)
    dh = torch.empty_like(layer.weight)
    db = torch.empty((F,), device=dev)
    nb = L.gf_grad_taps_workspace_bytes(B, N, G, F, 1, K)
    ws = torch.empty(nb // 4 + 1, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    w, b = layer.weight.detach(), layer.bias.detach()
    if L.gf_lsigf_pipeline(plans, 1, G, F, K) == 2:
        def khop(buf, op, width):                            # as separate launches (the layer fuses the K-1 hops of a chain)
            tap = B * N * width
            for k in range(1, K):
                rc = L.gf_spmm_hop_panel(plans[0], op, buf.data_ptr() + 4 * tap * (k - 1), buf.data_ptr() + 4 * tap * k,
                                         B * width // 4, st)
                if rc:
                    return rc
            return 0
        calls = {
            "pack_x": lambda: L.gf_pack_panels(x.data_ptr(), Z.data_ptr(), B, G, N, N, st),
            "khop_fwd(K-1 panel hops)": lambda: khop(Z, 0, G),
            "contract_fwd": lambda: L.gf_contract_panel(Z.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, N, N, G, F, 1, K, 0, st),
            "pack_dy": lambda: L.gf_pack_panels(dy.data_ptr(), P.data_ptr(), B, F, N, N, st),
            "grad_taps": lambda: L.gf_grad_taps_panel(Z.data_ptr(), P.data_ptr(), dh.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, B, N, G, F, 1, K, st),
            "khop_bwd(K-1 panel hops)": lambda: khop(P, 1, F),
            "contract_bwd": lambda: L.gf_contract_panel(P.data_ptr(), w.data_ptr(), None, dx.data_ptr(), B, N, N, G, F, 1, K, 1, st),
            # what the layer actually runs: for G, F <= 32 the backward produces dx and dh in ONE pass over the adjoint stack
            # (bwd_fused_panel_kernel) instead of grad_taps + contract_bwd above
            "lsigf_forward(whole)": lambda: L.gf_lsigf_forward(plans, 1, x.data_ptr(), w.data_ptr(), b.data_ptr(), Z.data_ptr(), y.data_ptr(),
                                                              B, G, F, K, N, st),
            "lsigf_backward(whole)": lambda: L.gf_lsigf_backward(plans, 1, dy.data_ptr(), Z.data_ptr(), w.data_ptr(), P.data_ptr(), dx.data_ptr(),
                                                                dh.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, B, G, F, K, N, st),
        }
    else:
      calls = {
        "layout_in": lambda: L.gf_layout_bgn_to_bng(x.data_ptr(), Z.data_ptr(), B, G, N, N, st),
        "khop_fwd(K-1 hops)": lambda: L.gf_khop(plans, 1, 0, Z.data_ptr(), B, G, K, st),
        "contract_fwd": lambda: L.gf_contract(Z.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, N, N, G, F, 1, K, 0, st),
        "layout_dy": lambda: L.gf_layout_bgn_to_bng(dy.data_ptr(), P.data_ptr(), B, F, N, N, st),
        "grad_taps": lambda: L.gf_grad_taps(Z.data_ptr(), P.data_ptr(), dh.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, B, N, G, F, 1, K, st),
        "khop_bwd(K-1 hops)": lambda: L.gf_khop(plans, 1, 1, P.data_ptr(), B, F, K, st),
        "contract_bwd": lambda: L.gf_contract(P.data_ptr(), w.data_ptr(), None, dx.data_ptr(), B, N, N, G, F, 1, K, 1, st),
      }
    out = {}
    for name, fn in calls.items():
        ts = []
        for i in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(fn(), name)
            e1.record()
            e1.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        out[name] = round(float(np.median(ts)), 4)
    return out


def cpu_baseline(A, layer, x, nnz, K, sample):
    """The reference's own CPU path (dense S + matmul/cat/permute, oracle.graph_filter_step_dense) on `sample`
    batch entries of the workload, all host cores; plus the sparse-CSR CPU restatement on the same sample."""
    from oracle import lsigf_oracle as orc
    torch.set_num_threads(os.cpu_count() or 1)
    cores = torch.get_num_threads()
    w, b = layer.weight.detach().cpu(), layer.bias.detach().cpu()
    xs = x[:sample].cpu()
    N = A.shape[0]
    out = dict(cores=cores, kind="port", unit="edges*taps/s")
    St = torch.sparse_csr_tensor(torch.from_numpy(A.T.tocsr().indptr.astype(np.int64)),
                                 torch.from_numpy(A.T.tocsr().indices.astype(np.int64)),
                                 torch.from_numpy(A.T.tocsr().data.astype(np.float32)), size=(N, N))
    orc.graph_filter_step_sparse_torch(w, b, St, xs[:4])
    t0 = time.perf_counter()
    orc.graph_filter_step_sparse_torch(w, b, St, xs)
    out["sparse_port_value"] = sample * nnz * K / (time.perf_counter() - t0)
    if N <= 20_000:                                           # dense S is N^2*4 bytes: 400 MB at N=10k, 40 GB at 100k
        S = torch.from_numpy(A.toarray().astype(np.float32))[None]
        orc.graph_filter_step_dense(w, b, S, xs[:2])          # warm-up
        t0 = time.perf_counter()
        orc.graph_filter_step_dense(w, b, S, xs)
        dt = time.perf_counter() - t0
        out.update(value=sample * nnz * K / dt, seconds=round(dt, 3),
                   sample=f"literal dense restatement of graphML.py:152-175 (fwd+bwd, fp32), {sample} of the batch's entries")
    else:
        out.update(value=out["sparse_port_value"],
                   sample=f"sparse-CSR CPU restatement (dense S would be {N * N * 4 / 1e9:.0f} GB), {sample} batch entries")
    return out


if __name__ == "__main__":
    main()
