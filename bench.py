#!/usr/bin/env python3
"""bench.py -- edges*taps/sec of the graph-filter hot path (BASELINE.json metric) on N MI355X GPUs.

    python bench.py                       # N = 1, the north-star workload (cfg4), steps / warmup chosen so the timed region is >= 0.5 s
    python bench.py --workload cfg2 --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workloads = BASELINE.json configs (config.workload names the one that ran):
  cfg4 (default)  ER N=100k nnz~1M, 128 samples per GPU (configs[3]: batch 1024 over 8 GPUs), GraphFilter K=5, F 32->32.  This is the
                  size BASELINE.json's roofline target is stated on, and it fits one GPU.
  cfg2            SBM N=10k nnz~100k, batch 256, GraphFilter K=5, 32->32 (configs[1])
  cfg1            sourceLocGNN SelectionGNN: SBM N=100, F=[1,32,32], K=[5,5], MaxPoolLocal, MLP [5] (configs[0], examples/sourceLocGNN.py)
  cfg3            movieGNN SelectionGNN on a MovieLens-100k-sized graph: N=1682, kNN-10 weights, F=[1,64,32], K=[5,5], NoPool, MLP [1]
  cfg5            EdgeVariantGF (per-edge taps) SBM N=50k nnz~500k, K=3, 32->32, batch 16 (configs[4])
  db              HiddenState_DB / GRNN_DB (per-sample, per-time-step GSOs with delays) at the flocking shape:
                  B=20, T=100, N=50, 6 -> 32 features, K=3; a launch-bound recursion (T steps x ~15 launches), replayed as ONE HIP graph
A step = one pass of the hot path over one batch resident in HBM: forward + backward (dx, dh, db) through the C ABI; for N > 1 plus the
ONE bucketed RCCL all-reduce of the parameter gradients (batch-axis data parallelism, weak scaling: per-GPU batch fixed).
value = B_global * sum_layers(nnz * K) / t_step, t_step from the barrier + synchronize bracket the harness prescribes; the median of
per-step HIP-event times is reported next to it (ms_per_step_median).

Extra objects on the JSON line:
  roofline     -- the dominant kernel: achieved = algorithmic bytes per launch / its average launch time, timed here with
                  HIP events on the launch stream; peak = 8 TB/s HBM3E; traffic = PMC bytes per launch on the memory side of the L2s
                  (fabric / L2-miss traffic: Infinity-Cache hits are counted) from profiles/*_pmc.json (rocprofv3 --pmc in passes of
                  their own, tools/pmc_collect.sh) for this workload and kernel, only if taken on the running kernel sources, else null.
  distributed  -- (under torch.distributed.run) per-rank ms_per_step min / max, the all-reduce's HIP-event time per step and its share.
  cpu_baseline -- the reference's CPU path (oracle/: restatement of graphML.py:152-175 / :457-488 in torch / numpy) timed on this box's
                  host cores on a bounded sample of the same workload; rank 0, N = 1 only.
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

HBM_PEAK_GBS = 8000.0
WORKLOADS = {
    "cfg4": dict(kind="filter", graph="er", N=100_000, deg=10.0, B=128, G=32, F=32, K=5, steps=100, warmup=5, cpu_sample=32,
                 desc="ER N=100k nnz~1M, batch 128/GPU (1024 over 8 GPUs), K=5, F 32->32 (BASELINE configs[3], the north-star size)"),
    "cfg2": dict(kind="filter", graph="sbm", N=10_000, deg=10.0, B=256, G=32, F=32, K=5, steps=250, warmup=20, cpu_sample=64,
                 desc="SBM N=10k nnz~100k, batch 256/GPU, K=5, F 32->32 (BASELINE configs[1])"),
    "cfg1": dict(kind="selgnn", graph="sbm", N=100, deg=30.0, B=100, dimF=[1, 32, 32], K=[5, 5], sel=[10, 10], pool="MaxPoolLocal",
                 alpha=[6, 8], mlp=[5], steps=1500, warmup=50, cpu_sample=100, hip_graph=True,
                 desc="sourceLocGNN SelectionGNN: SBM N=100, F=[1,32,32], K=[5,5], MaxPoolLocal [10,10] alpha [6,8], MLP [5], batch 100/GPU (BASELINE configs[0])"),
    "cfg3": dict(kind="selgnn", graph="knn", N=1682, deg=10.0, B=256, dimF=[1, 64, 32], K=[5, 5], sel=[1682, 1682], pool="NoPool",
                 alpha=[1, 1], mlp=[1], steps=400, warmup=20, cpu_sample=64,
                 desc="movieGNN SelectionGNN on a MovieLens-100k-sized graph: N=1682 kNN-10 weights, F=[1,64,32], K=[5,5], NoPool, MLP [1], batch 256/GPU (BASELINE configs[2])"),
    "cfg5": dict(kind="evgf", graph="sbm", N=50_000, deg=10.0, B=16, G=32, F=32, K=3, steps=15, warmup=3, cpu_sample=2,
                 desc="EdgeVariantGF per-edge taps: SBM N=50k nnz~500k, K=3, F 32->32, batch 16/GPU (BASELINE configs[4])"),
    "db": dict(kind="db", graph="flock", N=50, B=20, T=100, G=6, H=32, K=3, steps=30, warmup=5, cpu_sample=20, hip_graph=True,
               desc="HiddenState_DB / GRNN_DB (graphML.py:1096-1290, 3395-3538) at the flocking example's shape: 50 agents, T=100 steps, a "
                    "communication-radius GSO per (sample, step), 6 input features, 32 hidden, K=3, batch 20/GPU; fwd+bwd replayed as ONE HIP graph"),
    "tiny": dict(kind="filter", graph="sbm", N=1000, deg=10.0, B=32, G=32, F=32, K=5, steps=50, warmup=5, cpu_sample=8, desc="plumbing check"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def flock_gsos(B, T, N, dev, seed=0):
    """S [B,T,1,N,N]: agents on a random walk in the unit square, an edge where two agents are within the communication radius, each
    operator divided by its largest degree (what the reference's flocking data set hands the _DB layers, dataTools.py: computeCommunicationGraph
    with 'normalizeGraph') -- synthetic positions, the same shape and sparsity regime (~10 neighbours per agent)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    pos = torch.rand(B, 1, N, 2, generator=g) + 0.01 * torch.cumsum(torch.randn(B, T, N, 2, generator=g), dim=1)
    d2 = ((pos[:, :, :, None, :] - pos[:, :, None, :, :]) ** 2).sum(-1)
    A = ((d2 < 0.26 ** 2) & ~torch.eye(N, dtype=torch.bool)).float()
    S = A / A.sum(-1).amax(-1).clamp(min=1.0)[..., None, None]
    return S[:, :, None].contiguous().to(dev), float(A.sum() / (B * T))


def make_graph(wl):
    from alegnn_amd import graphgen
    if wl["graph"] == "knn":
        return graphgen.knn_weighted(wl["N"], k=int(wl["deg"]), seed=0)
    return (graphgen.sbm if wl["graph"] == "sbm" else graphgen.er)(wl["N"], avg_degree=wl["deg"], seed=0)


class Workload:
    """module + inputs + the step; units = edges*taps of one sample (sum over the filter layers of nnz * K)."""

    def __init__(self, name, wl, dev, rank):
        from alegnn_amd.modules.architectures import SelectionGNN
        from alegnn_amd.utils import graphML as gml
        self.name, self.wl, self.dev = name, wl, dev
        N, B = wl["N"], wl["B"]
        torch.manual_seed(0)
        gen = torch.Generator(device=dev).manual_seed(1000 + rank)
        if wl["kind"] == "db":
            T = wl["T"]
            self.S, nnz = flock_gsos(B, T, N, dev, seed=rank)
            self.A, self.nnz = None, int(round(nnz))            # average non-zeros of one (sample, step) operator
            self.module = gml.HiddenState_DB(wl["G"], wl["H"], wl["K"], torch.tanh, 1, True).to(dev)
            self.module.addGSO(self.S)
            self.x = torch.randn(B, T, wl["G"], N, device=dev, generator=gen).requires_grad_(True)
            self.z0 = torch.zeros(B, wl["H"], N, device=dev)
            self.dy = torch.randn(B, T, wl["H"], N, device=dev, generator=gen)
            self.units = 2 * T * self.nnz * wl["K"]            # two filters (A(S)x and B(S)z) per step, T steps per sample
            return
        self.A = make_graph(wl)
        self.nnz = int(self.A.nnz)
        if wl["kind"] == "filter":
            self.module = gml.GraphFilter(wl["G"], wl["F"], wl["K"], 1, True)
            self.module.addGSO(self.A)
            self.module.to(dev)
            self.x = torch.randn(B, wl["G"], N, device=dev, generator=gen).requires_grad_(True)
            self.dy = torch.randn(B, wl["F"], N, device=dev, generator=gen)
            self.units = self.nnz * wl["K"]
        elif wl["kind"] == "evgf":
            self.module = gml.EdgeVariantGF(wl["G"], wl["F"], wl["K"], N, N, 1, True, sparse=True)
            self.module.addGSO(self.A)
            self.module.to(dev)
            self.x = torch.randn(B, wl["G"], N, device=dev, generator=gen).requires_grad_(True)
            self.dy = torch.randn(B, wl["F"], N, device=dev, generator=gen)
            self.units = self.nnz * wl["K"]
        else:
            self.module = SelectionGNN(wl["dimF"], wl["K"], True, torch.nn.ReLU, wl["sel"], getattr(gml, wl["pool"]), wl["alpha"],
                                       wl["mlp"], self.A).to(dev)
            self.x = torch.randn(B, wl["dimF"][0], N, device=dev, generator=gen).requires_grad_(True)
            self.dy = None
            self.units = self.nnz * sum(wl["K"])

    def step_fwd_bwd(self):
        self.x.grad = None
        if self.wl["kind"] == "db":
            self.module(self.x, self.z0)[0].backward(self.dy)
        elif self.dy is not None:
            self.module(self.x).backward(self.dy)
        else:
            self.module(self.x).square().sum().backward()          # the examples' losses need labels; any scalar loss drives the same path


def timed_hip_events(fn, n):
    """n calls, each between two HIP events on the launch stream (= torch's current stream: every C-ABI call is handed
    torch.cuda.current_stream()); returns the per-call milliseconds."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    return [e0.elapsed_time(e1) for e0, e1 in ev]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: per workload, >= 0.5 s of timed region)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=None, help="batch entries of the workload timed on the CPU")
    ap.add_argument("--comparator", action="store_true", help="also time the UNMODIFIED dense formulation (graphML.py:152-175 in torch ops, dense S) on "
                    "this GPU through PyTorch-ROCm for filter workloads of any size (default: only where dense S is small, N <= 20k)")
    ap.add_argument("--detail", action="store_true", help="per-building-block timings to stderr (filter workloads)")
    ap.add_argument("--no-graph", action="store_true", help="launch-bound workloads (cfg1): eager launches instead of one HIP-graph replay per step")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="gf_tune knob (needs GFHIP_EXPERIMENTS=1; repeatable)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs; there is no CPU fallback"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = world > 1 or "RANK" in os.environ       # under torch.distributed.run even N = 1 goes through RCCL
    # host side of W ranks on one box: an equal share of the cores each (graph generation, plan creation and the framework's own
    # helper threads would otherwise oversubscribe W-fold), and ONE clustering of the graph per machine instead of one per rank
    # (gf_plan_create's locality groups, ~2 s of one core at config 4: shared through GFHIP_PLAN_CACHE_DIR, see gf_plan.hip)
    cores = os.cpu_count() or 1
    if world > 1:
        torch.set_num_threads(max(1, cores // world))
    os.environ.setdefault("GFHIP_PLAN_CACHE_DIR", os.path.join(ROOT, "gpurun_out", "plan_cache"))
    os.makedirs(os.environ["GFHIP_PLAN_CACHE_DIR"], exist_ok=True)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL prints a version banner on STDOUT when it creates its first communicator; rank 0's stdout carries exactly one JSON line, so
        # the process's stdout (fd 1) points at stderr from here until that line is written.
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from alegnn_amd import _lib, parallel
    for kv in args.tune:
        key, val = kv.split("=")
        _lib.check(_lib.lib().gf_tune(key.encode(), int(val)), "gf_tune " + key)
    wl = WORKLOADS[args.workload]
    steps = args.steps if args.steps is not None else wl["steps"]
    warmup = args.warmup if args.warmup is not None else wl["warmup"]
    w = Workload(args.workload, wl, dev, rank)
    parallel.broadcast_parameters(w.module)
    # One flat gradient bucket + ONE all-reduce per step under data parallelism.  The step is THE SAME at every N: a single process
    # zeroes the bucket and lets backward accumulate into its views too (it only skips the collective), so the N = 1 line and the
    # N > 1 lines of a scaling curve time the same work.  Exception, stated on the line (config.grad_handling):
    # the EVGF workload on a single process, whose bucket would be the 4.7 GB of per-edge taps (3.4 ms per step of zeroing and
    # accumulating that no single-GPU training step pays); it is not part of the driver's scaling runs.
    params = [p for p in w.module.parameters() if p.requires_grad]
    use_bucket = distributed or wl["kind"] != "evgf"
    bucket = parallel.GradBucket(params) if use_bucket else None

    ar_events = []                                                 # (start, end) HIP events around the all-reduce, filled only while ar_probe is on
    ar_probe = [False]

    def step():
        if bucket is not None:
            bucket.zero_()
        else:
            for p in params:
                p.grad = None
        w.step_fwd_bwd()
        if bucket is not None:
            if ar_probe[0]:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                bucket.allreduce_mean()
                e1.record()
                ar_events.append((e0, e1))
            else:
                bucket.allreduce_mean()

    def sync_all():
        if distributed:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync_all()
    # A step of ~60 launches of a few microseconds each is bound by the host's launch rate: such workloads (config 1) replay the
    # step as ONE HIP graph -- the library only launches on the stream it is handed, allocates nothing and never synchronises, so
    # torch.cuda.graph captures it as is.  Single process only (the captured step has no collective).
    use_graph = bool(wl.get("hip_graph")) and not distributed and not args.no_graph
    eager_step = step
    if use_graph:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            eager_step()
        step = graph.replay
        for _ in range(warmup):
            step()
        sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    dist_info = None
    if distributed:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                               # per-rank wall time of the same K steps: a straggler shows up here
        per_rank_ms = [1e3 * float(t.item()) / steps for t in every]
        elapsed = max(float(t.item()) for t in every)              # the slowest rank defines the step time
        dist_info = dict(backend=dist.get_backend(), rccl_ranks=dist.get_world_size(), ms_per_step_rank_min=min(per_rank_ms),
                         ms_per_step_rank_max=max(per_rank_ms), ms_per_step_per_rank=[round(v, 4) for v in per_rank_ms])
    ms_per_step = 1e3 * elapsed / steps
    B = wl["B"]
    value = (B * world) * w.units / (elapsed / steps)
    per_step = timed_hip_events(step, max(20, min(steps, 50)))    # median of >= 20, HIP events
    ms_median = float(np.median(per_step))
    if dist_info is not None and bucket is not None:
        # the ONE collective of a step, timed with HIP events on the stream it is enqueued on (outside the timed region: 20 more steps):
        # what it costs on the step's critical path once the backward's last kernel has finished (nothing overlaps it by construction)
        ar_probe[0] = True
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ar_probe[0] = False
        ar_ms = sorted(e0.elapsed_time(e1) for e0, e1 in ar_events)
        t = torch.tensor([ar_ms[len(ar_ms) // 2], ar_ms[-1]], dtype=torch.float64, device=dev)
        mx = t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist_info.update(allreduce_ms_median_max_over_ranks=round(float(mx[0].item()), 4), allreduce_ms_worst=round(float(mx[1].item()), 4),
                         allreduce_bytes=bucket.nbytes(), allreduce_share_of_step=round(float(mx[0].item()) / ms_per_step, 5),
                         forced=bool(os.environ.get("GFHIP_FORCE_COLLECTIVES")))

    L = _lib.lib()
    roofline = ROOFLINES[wl["kind"]](L, w, wl)                      # every rank runs it: keeps the ranks in step
    mfma = mfma_filter(L, w, wl) if wl["kind"] == "filter" else None
    detail = None
    if args.detail and rank == 0 and wl["kind"] == "filter":
        detail = kernel_breakdown(L, w, wl)
        log("breakdown_ms", json.dumps(detail))
    comparator = dense_reference_on_gpu(w, wl, args.comparator) if rank == 0 and world == 1 else None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = CPU_BASELINES[wl["kind"]](w, wl, args.cpu_sample or wl["cpu_sample"])

    if rank == 0:
        cfg = dict(workload=args.workload, description=wl["desc"], graph=wl["graph"], N=wl["N"], nnz=w.nnz, batch_per_gpu=B,
                   global_batch=B * world, K=wl["K"], E=1, **({"T": wl["T"], "H": wl["H"]} if "T" in wl else {}), parallelism=f"batch-dp{world}", grad_bucket_bytes=(bucket.nbytes() if bucket is not None else sum(p.numel() * 4 for p in params)),
                   rccl_ranks=(dist.get_world_size() if distributed else 0), hip_graph=use_graph,
                   grad_handling=("flat bucket: zeroed every step, backward accumulates into its views" + (", one all-reduce" if distributed else ", no collective (one rank)")
                                  if bucket is not None else "p.grad = None before backward (no bucket)"),
                   devices=[torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())][:world])
        cfg.update({k: wl[k] for k in ("G", "F", "dimF", "sel", "pool", "alpha", "mlp") if k in wl})
        out = dict(metric="edges*taps/sec (GraphFilter fwd+bwd)", value=value, unit="edges*taps/s", n_gpus=world, steps=steps,
                   warmup=warmup, ms_per_step=ms_per_step, ms_per_step_median=ms_median, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f32", data="synthetic", config=cfg, roofline=roofline, cpu_baseline=cpu)
        if comparator is not None:
            comparator["speedup_of_this_library"] = round(comparator["ms_per_step"] / ms_per_step, 2)
            out["no_rewrite_comparator"] = comparator
        if dist_info is not None:
            out["distributed"] = dist_info
        if mfma is not None:
            out["mfma"] = mfma
        if detail:
            out["breakdown_ms"] = detail
        if distributed:
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        if distributed:
            os.dup2(2, 1)   # (whatever the teardown prints stays off stdout as well)
    if distributed:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------------------
# roofline of the dominant kernel of each workload kind
# ------------------------------------------------------------------------------------------------------------------------------
def kernel_source_sha():
    """sha256 over the library's sources (csrc/*.hip, gf_common.h, include/gfhip.h): what a PMC summary is valid for.  The GPU box has
    no .git, so provenance is tied to the bytes of the kernel sources, not to a commit id."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "graph-neural-networks_amd", "csrc", "*"))) + [os.path.join(ROOT, "include", "gfhip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_summary(workload, kernel, suffix="_pmc.json"):
    """Newest committed PMC summary (profiles/*<suffix>) for this workload and kernel, with whether it was taken on the kernel sources
    that are running now."""
    best, running = None, kernel_source_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix))):
        try:
            pm = json.load(open(f))
        except Exception:
            continue
        if pm.get("workload") == workload and pm.get("kernel") == kernel:
            if best is None or pm.get("kernel_src_sha") == running or best[1].get("kernel_src_sha") != running:
                best = (f, pm)          # a summary taken on the running sources wins over any other; among equals the last by name
    if best is None:
        return None, None
    f, pm = best
    src = dict(file=os.path.relpath(f, ROOT), kernel_src_sha=pm.get("kernel_src_sha"), running_src_sha=kernel_source_sha())
    src["matches_running_sources"] = src["kernel_src_sha"] == src["running_src_sha"]
    return pm, src


TRAFFIC_KIND = ("fabric bytes per launch = reads + writes that LEAVE the XCDs' L2s (TCC_EA0 requests: FETCH_SIZE x 2 per the guide's gfx950 "
                "correction + WRITE_SIZE); Infinity-Cache hits are counted, so this is L2-miss traffic, an upper bound on HBM traffic")


def pmc_traffic(workload, kernel):
    """(fabric bytes per launch, provenance): the bytes are reported only when the summary was taken on the kernel sources running now
    (else null: a stale counter next to a live timing is worse than none).  What the counters see is the memory side of the L2s --
    Infinity-Cache hits included (MI355X_MICROARCH.md) -- so the number is L2-miss (fabric) traffic, not HBM traffic."""
    pm, src = pmc_summary(workload, kernel)
    if pm is None:
        return None, None
    src["kind"] = TRAFFIC_KIND
    if pm.get("calibration"):                  # (round 6) the kernel's scalar prefetch splits a row's fetch into two 64-byte requests: calibrated as the guide asks
        src["calibration"] = pm["calibration"]
        src["bytes_by_rule_uncalibrated"] = pm.get("hbm_bytes_per_launch_by_rule")
    if pm.get("passes_complete") is False:
        return None, dict(src, warning=pm.get("warning"))
    return (pm.get("hbm_bytes_per_launch") if src["matches_running_sources"] else None), src


def hop_bytes(B, N, W, nnz):
    return 2 * B * N * W * 4 + nnz * 8 + (N + 1) * 4            # read X once, write X once, read the CSR once


def filter_hop_roofline(L, plans, name, B, N, W, K, nnz, dev):
    """One K-hop chain of a GraphFilter layer in the pipeline that layer runs, HIP events inside the library (gf_time_*:
    hipEventRecord on the launch stream around `iters` back-to-back launches)."""
    ms = ctypes.c_float()
    st = torch.cuda.current_stream().cuda_stream
    pipe = L.gf_lsigf_pipeline(plans, 1, W, W, K)
    from alegnn_amd import _lib
    if pipe == 2:                                                 # column panels: the K-1 hops of a panel are ONE launch (gf_chain.hip)
        Z = torch.randn(K, B * W // 4, N, 4, device=dev)
        _lib.check(L.gf_time_khop_panel(plans, 1, 0, Z.data_ptr(), B, W, K, 20, st, ctypes.byref(ms)))
        which = L.gf_khop_panel_uses_chain(plans[0], 0, B * W // 4)
        if which == 1:
            kern, hops = "spmm_chain_kernel", K - 1
            note = ("K-1 hops of every panel in one launch, panel resident in LDS: HBM sees 1 read + (K-1) writes of the signal per launch, "
                    "the algorithmic count (a read and a write per hop) is what `achieved` divides by")
        else:                                                     # few panels / small weighted GSO: one launch per hop
            kern, hops = ("spmm_panel_db_kernel" if which == 2 else "spmm_panel_kernel"), 1
            ms.value /= (K - 1)
            note = "one hop per launch (the K-1 launches of a chain timed together, launch_ms = their mean), gathers from an LDS-resident panel"
    else:                                                         # node-major, gathers through L2: the chain the layer walks (tap k-1 -> tap k)
        Z = torch.randn(K, B, N, W, device=dev)
        _lib.check(L.gf_time_khop(plans, 1, 0, Z.data_ptr(), B, W, K, 10, st, ctypes.byref(ms)))
        if L.gf_spmm_hop_kernel(plans[0], 0, B, W) == 1:
            kern, hops = "spmm_msweep_kernel", K - 1
            note = ("the K-1 hops of gf_khop in ONE launch, batch entry by batch entry: an XCD holds an entry's output rows in its register "
                    "files, its waves walk the source rows together (every row leaves HBM once per hop), fp32 MFMA scatter-accumulate.  This is the "
                    "launch timed here and in the PMC summary (template arguments <.., 0, 0>).  In the timed STEP the layer's launches of the same kernel "
                    "(<.., 0, 2> in a rocprofv3 kernel trace) also carry the boundary layout pass x / dy -> tap 0 as a pre-phase (round 6: 2 x B x N x W x 4 "
                    "more algorithmic bytes per launch, ~0.3 ms longer) in place of the two separate layout kernels")
        else:
            kern, hops = "spmm_sell_kernel", 1
            ms.value /= (K - 1)
            note = "one hop per launch (the K-1 launches of a chain timed together, launch_ms = their mean), node-major rows gathered through L2 / Infinity Cache"
    nbytes = hops * hop_bytes(B, N, W, nnz)
    achieved = nbytes / (ms.value * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(name, kern)
    return dict(bound="hbm", kernel=kern, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                algorithmic_bytes=nbytes, launch_ms=round(ms.value, 5), hops_per_launch=hops, pipeline=int(pipe), note=note)


def roofline_filter(L, w, wl):
    plans = w.module._gso.plans(w.dev)
    return filter_hop_roofline(L, plans, w.name, wl["B"], wl["N"], wl["G"], wl["K"], w.nnz, w.dev)


MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak of MI355X (MI355X_MICROARCH.md)


def mfma_filter(L, w, wl):
    """The filter-bank contraction [B*N, K*G] x [K*G, F] (graphML.py:170-171 + bias) of the forward: achieved fp32 MFMA flop/s from a live
    HIP-event timing of the kernel the layer runs, and the MFMA pipe's busy share from the committed counter summary (MfmaUtil =
    sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * SIMDs), tools/pmc_mfma.sh), reported only when it was taken on these sources."""
    from alegnn_amd import _lib
    layer, dev = w.module, w.dev
    B, N, G, F, K = wl["B"], wl["N"], wl["G"], wl["F"], wl["K"]
    plans = layer._gso.plans(dev)
    wt, b = layer.weight.detach(), layer.bias.detach()
    y = torch.empty((B, F, N), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    panel = L.gf_lsigf_pipeline(plans, 1, G, F, K) == 2
    Z = torch.randn((K, B * G // 4, N, 4) if panel else (K, B, N, G), device=dev)
    fn = (L.gf_contract_panel if panel else L.gf_contract)
    call = lambda: _lib.check(fn(Z.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), B, N, N, G, F, 1, K, 0, st), "contract")
    ms = float(np.median(timed_hip_events(call, 12)[2:]))
    flops = 2.0 * B * N * K * G * F
    kern = "contract_panel_kernel" if panel else "contract_mfma_kernel"
    pm, src = pmc_summary(w.name, kern, "_mfma_pmc.json")
    util = pm.get("mfma_util_pct") if (pm is not None and src["matches_running_sources"]) else None
    nbytes = 4 * (K * B * N * G + B * N * F + K * G * F)
    return dict(bound="mfma", kernel=kern, achieved=round(flops / (ms * 1e-3) / 1e12, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                frac=round(flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), launch_ms=round(ms, 4), flops=flops,
                mfma_busy_pct=util, mfma_busy_source=src, hbm_gbs=round(nbytes / (ms * 1e-3) / 1e9, 1),
                note="AI = 13 flop/B < machine balance 20: this contraction sits on the HBM roof (hbm_gbs), the MFMA pipe is busy mfma_busy_pct of the launch")


def roofline_selgnn(L, w, wl):
    """The hidden GraphFilter layer (F1 -> F2 on the full graph) carries the K-hop traffic: its chain at the width it runs."""
    layer = w.module.GFL[3]
    plans = layer._gso.plans(w.dev)
    W = wl["dimF"][1]
    r = filter_hop_roofline(L, plans, w.name, wl["B"], layer.N, W, wl["K"][1], int(layer._gso.mats[0].nnz), w.dev)
    r["layer"] = f"GFL[3]: GraphFilter {W}->{wl['dimF'][2]}, K={wl['K'][1]}, N={layer.N}"
    return r


def roofline_evgf(L, w, wl):
    """EVGF forward: V0 = diag taps, K-1 edge taps (ev_hop_lds4_kernel, the dominant kernel), the sum over (g, k).  Algorithmic
    bytes: per tap the edge weights F*G*nnzp*4 and the chain state read and written 2*B*F*G*N*4."""
    B, G, F, K, N = wl["B"], wl["G"], wl["F"], wl["K"], wl["N"]
    nnzp = int(w.module._patterns[0].nnzp)
    with torch.no_grad():
        xs = w.x.detach()
        ts = timed_hip_events(lambda: w.module(xs), 12)[2:]
    ms = float(np.median(ts))
    state = B * F * G * N * 4
    tap = F * G * nnzp * 4 + 2 * state
    fwd = (F * G * N * 4 + B * G * N * 4 + state) + (K - 1) * tap + (K * state + B * F * N * 4)   # diag tap + edge taps + sum
    achieved = fwd / (ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(w.name, "ev_hop_lds4_kernel")
    return dict(bound="hbm", kernel="gf_evgf_forward (ev_diag + (K-1) x ev_hop_lds4_kernel + ev_sum)", achieved=round(achieved, 1),
                peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                algorithmic_bytes=fwd, algorithmic_bytes_per_tap=tap, launch_ms=round(ms, 4), nnzp=nnzp,
                note="whole forward timed with HIP events on the launch stream; per-kernel split: profiles/*cfg5*_kernel_stats.csv")


def roofline_db(L, w, wl):
    """The per-(sample, step) hop of the delayed filter A(S)x (db_hop_kernel over all B*T operators in one launch): it reads every dense
    N x N operator once.  At flocking sizes (N = 50: 10 KB per operator) the launch is ~10 us -- launch-bound, as is the whole
    recursion (T steps x ~15 launches of a few microseconds); the roofline fraction is reported for completeness."""
    from alegnn_amd import _lib
    B, T, N = wl["B"], wl["T"], wl["N"]
    W = 8                                                         # the 6 input features padded to 8 (16-byte rows)
    X0 = torch.randn(B * T, N, W, device=w.dev)
    X1 = torch.empty_like(X0)
    st = torch.cuda.current_stream().cuda_stream
    NN = N * N
    call = lambda: _lib.check(L.gf_db_hop(w.S.data_ptr(), T * NN, NN, X0.data_ptr(), X1.data_ptr(), B, T, N, W, _lib.GF_OP_FWD, 1, st), "gf_db_hop")
    ms = float(np.median(timed_hip_events(call, 30)[5:]))
    nbytes = B * T * NN * 4 + 2 * B * T * N * W * 4
    achieved = nbytes / (ms * 1e-3) / 1e9
    return dict(bound="hbm", kernel="db_hop_kernel", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=None, traffic_source=None, algorithmic_bytes=nbytes, launch_ms=round(ms, 5),
                note="one delayed hop of all B*T per-sample operators; launch-bound at this size (see the kernel stats under profiles/)")


ROOFLINES = {"filter": roofline_filter, "selgnn": roofline_selgnn, "evgf": roofline_evgf, "db": roofline_db}


# ------------------------------------------------------------------------------------------------------------------------------
# CPU baselines: the oracle (port of the reference's CPU path), bounded samples
# ------------------------------------------------------------------------------------------------------------------------------
def _csr_t(A):
    At = A.T.tocsr()
    return torch.sparse_csr_tensor(torch.from_numpy(At.indptr.astype(np.int64)), torch.from_numpy(At.indices.astype(np.int64)),
                                   torch.from_numpy(At.data.astype(np.float32)), size=A.shape)


def cpu_filter(w, wl, sample):
    """CPU restatements of graphML.py:152-175 (fwd+bwd, fp32) on a bounded sample of the batch: every variant that can run is probed
    on two batch entries, the FASTEST one is then timed on a sample sized for ~5 s and reported (`variant` says which; the probes of
    the others are listed in `variants`).  Variants: the literal dense form (dense S fits: N <= 20k), the sparse-CSR restatement
    through torch.sparse at several thread counts (all cores is rarely the fastest: the CSR product oversubscribes a 2-socket host),
    and the scipy CSR restatement (one core)."""
    from oracle import lsigf_oracle as orc
    ncores = os.cpu_count() or 1
    threads_on_entry = torch.get_num_threads()               # (main() caps the threads per rank: put that back afterwards)
    wt, b = w.module.weight.detach().cpu(), w.module.bias.detach().cpu()
    N, K, B = wl["N"], wl["K"], wl["B"]
    xs = w.x.detach()[:min(B, max(2, sample))].cpu()
    St = _csr_t(w.A)
    variants = []

    def torch_sparse(th):
        def run(x):
            torch.set_num_threads(th)
            orc.graph_filter_step_sparse_torch(wt, b, St, x)
        return run

    def scipy_csr(x):
        xn = x.numpy()
        orc.lsigf_sparse(wt.numpy(), w.A, xn, b.numpy(), dtype=np.float32)
        orc.lsigf_sparse_grads(wt.numpy(), w.A, xn, b.numpy(), np.ones((xn.shape[0], wt.shape[0], N), np.float32), dtype=np.float32)

    cands = [(f"sparse-CSR restatement, torch.sparse_csr, {th} threads", th, torch_sparse(th))
             for th in sorted({min(ncores, t) for t in (8, 16, 32, 64)})]     # (all 256 hardware threads: 100x slower than 32, measured)
    cands.append(("sparse-CSR restatement, scipy (analytic backward), 1 core", 1, scipy_csr))
    if N <= 20_000:                                           # dense S is N^2*4 bytes: 400 MB at N=10k, 40 GB at 100k
        S = torch.from_numpy(w.A.toarray().astype(np.float32))[None]

        def dense(x):
            torch.set_num_threads(ncores)
            orc.graph_filter_step_dense(wt, b, S, x)
        cands.append((f"literal dense restatement (the reference's own form), {ncores} threads", ncores, dense))
    budget = time.perf_counter() + 12.0                        # probes stop here; the reported run adds ~5 s
    best = None
    for name, th, fn in cands:
        if time.perf_counter() > budget:
            break
        fn(xs[:1])                                             # warm-up (thread pools, first-touch)
        t0 = time.perf_counter()
        fn(xs[:2])
        dt = time.perf_counter() - t0
        v = 2 * w.nnz * K / dt
        variants.append(dict(variant=name, threads=th, value=v))
        if best is None or v > best[3]:
            best = (name, th, fn, v, dt / 2)
    name, th, fn, _, per_entry = best
    n = int(max(2, min(xs.shape[0], 5.0 / per_entry)))
    t0 = time.perf_counter()
    fn(xs[:n])
    dt = time.perf_counter() - t0
    torch.set_num_threads(threads_on_entry)
    return dict(value=n * w.nnz * K / dt, unit="edges*taps/s", cores=th, kind="port", variant=name, seconds=round(dt, 3), host_cores=ncores,
                sample=f"fastest of {len(variants)} CPU variants of graphML.py:152-175 (fwd+bwd, fp32): {name}; {n} of the batch's {B} entries",
                variants=variants)


def cpu_selgnn(w, wl, sample):
    """The GraphFilter layers of the architecture in the reference's dense form (the layers that define the metric's units), each at
    the node count it runs on; pooling / MLP are not part of the units and are left out.  Thread counts 16 / 64 / all are probed on a
    small sample and the fastest is the one timed and reported."""
    from oracle import lsigf_oracle as orc
    ncores = os.cpu_count() or 1
    threads_on_entry = torch.get_num_threads()
    sample = min(sample, wl["B"])
    S = torch.from_numpy(w.A.toarray().astype(np.float32))[None]
    nodes = [wl["N"]] + list(wl["sel"])
    layers = [(w.module.GFL[3 * l].weight.detach().cpu(), w.module.GFL[3 * l].bias.detach().cpu(), wl["dimF"][l], nodes[l]) for l in range(2)]

    def run(n, th):
        torch.set_num_threads(th)
        dt = 0.0
        for wt, b, G, Nl in layers:
            x = torch.randn(n, G, Nl)
            orc.graph_filter_step_dense(wt, b, S, x[:1])
            t0 = time.perf_counter()
            orc.graph_filter_step_dense(wt, b, S, x)
            dt += time.perf_counter() - t0
        return dt
    probes = {th: run(min(8, sample), th) for th in sorted({min(ncores, t) for t in (16, 64)} | {ncores})}
    th = min(probes, key=probes.get)
    dt = run(sample, th)
    torch.set_num_threads(threads_on_entry)
    return dict(cores=th, host_cores=ncores, kind="port", unit="edges*taps/s", value=sample * w.units / dt, seconds=round(dt, 3),
                sample=f"the two GraphFilter layers in the literal dense form of graphML.py:152-175 + :2125-2144 (fwd+bwd, fp32), {sample} samples, "
                       f"{th} threads (fastest of {sorted(probes)})")


def cpu_evgf(w, wl, sample):
    """scipy restatement of EVGF with per-edge taps (oracle/evgf_oracle.py), forward + analytic backward, one host core."""
    import scipy.sparse as sp
    from oracle import evgf_oracle as evo
    sample = min(sample, wl["B"])
    m = w.module
    pat = m._patterns[0]
    P = sp.csr_matrix((np.ones(pat.nnzp, dtype=np.float32), pat.indices, pat.indptr), shape=(pat.N, pat.N))
    wd = m.weightEVdiag.detach()[:, 0].cpu().numpy()
    we = m.weightEVedges[0].detach().cpu().numpy()
    xs = w.x.detach()[:sample].cpu().numpy()
    dys = w.dy[:sample].cpu().numpy()
    t0 = time.perf_counter()
    evo.evgf_sparse(P, wd, we, xs, None, dtype=np.float32)
    evo.evgf_sparse_grads(P, wd, we, xs, dys, dtype=np.float32)
    dt = time.perf_counter() - t0
    return dict(cores=1, kind="port", unit="edges*taps/s", value=sample * w.units / dt, seconds=round(dt, 3),
                sample=f"scipy restatement of graphML.py:457-488 with per-edge taps (forward + analytic backward, fp32), {sample} of the batch's {wl['B']} entries")


def cpu_db(w, wl, sample):
    """oracle/db_oracle.py's restatement of GRNN_DB (graphML.py:1096-1290) in CPU torch, forward + autograd backward, on `sample` of the
    batch's samples (all T steps)."""
    from oracle import db_oracle as dbo
    ncores = os.cpu_count() or 1
    threads_on_entry = torch.get_num_threads()
    n = max(1, min(sample, wl["B"]))
    m = w.module
    a, b = m.aWeights.detach().cpu().requires_grad_(True), m.bWeights.detach().cpu().requires_grad_(True)
    xb, zb = m.xBias.detach().cpu().requires_grad_(True), m.zBias.detach().cpu().requires_grad_(True)
    S, x, z0, dy = w.S[:n].cpu(), w.x.detach()[:n].cpu().requires_grad_(True), w.z0[:n].cpu(), w.dy[:n].cpu()
    best = None
    for th in sorted({min(ncores, t) for t in (1, 8, 32)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        dbo.grnn_db(a, b, S, x, z0, torch.tanh, xb, zb).backward(dy)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (th, dt)
    torch.set_num_threads(threads_on_entry)
    th, dt = best
    return dict(value=n * w.units / dt, unit="edges*taps/s", cores=th, host_cores=ncores, kind="port", seconds=round(dt, 3),
                sample=f"GRNN_DB restatement (oracle/db_oracle.py, CPU torch, fwd + autograd bwd), {n} of the batch's {wl['B']} samples, all {wl['T']} steps, "
                       f"{th} threads (fastest of 1 / 8 / 32)")


def dense_reference_on_gpu(w, wl, force):
    """SURVEY.md 8d's optional third column: the reference's own formulation -- dense S [1,N,N], one torch.matmul per tap, the taps grown by
    torch.cat, permute + matmul + permute for the bank (graphML.py:152-175) -- run UNMODIFIED in torch ops on this GPU (rocBLAS GEMMs),
    forward + backward, same inputs and parameters as the timed step.  Filter workloads; by default only where dense S is small."""
    if wl["kind"] != "filter" or (wl["N"] > 20_000 and not force):
        return None
    dev, N, K = w.dev, wl["N"], wl["K"]
    S = torch.zeros((1, N, N), dtype=torch.float32, device=dev)
    coo = w.A.tocoo()
    S[0, torch.as_tensor(coo.row, device=dev, dtype=torch.int64), torch.as_tensor(coo.col, device=dev, dtype=torch.int64)] = torch.as_tensor(coo.data, dtype=torch.float32, device=dev)
    h = w.module.weight.detach().clone().requires_grad_(True)
    b = w.module.bias.detach().clone().requires_grad_(True)
    x = w.x.detach().clone().requires_grad_(True)
    Fo, E, _, G = h.shape

    def step():
        for t in (h, b, x):
            t.grad = None
        B = x.shape[0]
        cur = x.reshape(B, 1, G, N)
        z = x.reshape(B, 1, 1, G, N).repeat(1, E, 1, 1, 1)
        for _ in range(1, K):                                   # :158-161
            cur = torch.matmul(cur, S.reshape(1, E, N, N))
            z = torch.cat((z, cur.reshape(B, E, 1, G, N)), dim=2)
        y = torch.matmul(z.permute(0, 4, 1, 2, 3).reshape(B, N, E * K * G), h.reshape(Fo, E * K * G).permute(1, 0)).permute(0, 2, 1) + b   # :170-175
        y.backward(w.dy if hasattr(w, "dy") else torch.ones_like(y))
    step()
    torch.cuda.synchronize()
    n = 3 if N <= 20_000 else 1
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    return dict(kind="reference formulation unmodified (dense S, torch.matmul / cat / permute) on this GPU via PyTorch-ROCm", ms_per_step=round(ms, 3),
                value=wl["B"] * w.nnz * K / (ms * 1e-3), unit="edges*taps/s", dense_S_bytes=int(N) * int(N) * 4, steps=n)


CPU_BASELINES = {"db": cpu_db, "filter": cpu_filter, "selgnn": cpu_selgnn, "evgf": cpu_evgf}


def kernel_breakdown(L, w, wl):
    """Per-building-block time of a GraphFilter layer (torch events on the launch stream = torch's current stream), median of 10."""
    from alegnn_amd import _lib
    layer, dev = w.module, w.dev
    B, N, G, F, K = wl["B"], wl["N"], wl["G"], wl["F"], wl["K"]
    plans = layer._gso.plans(dev)
    x, dy = w.x.detach(), w.dy
    Z = torch.empty((K, B, N, G), device=dev)
    P = torch.empty((K, B, N, F), device=dev)
    y = torch.empty((B, F, N), device=dev)
    dx = torch.empty((B, G, N), device=dev)
    dh = torch.empty_like(layer.weight)
    db = torch.empty((F,), device=dev)
    nb = L.gf_grad_taps_workspace_bytes(B, N, G, F, 1, K)
    ws = torch.empty(nb // 4 + 1, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    wt, b = layer.weight.detach(), layer.bias.detach()
    if L.gf_lsigf_pipeline(plans, 1, G, F, K) == 2:
        calls = {
            "pack_x": lambda: L.gf_pack_panels(x.data_ptr(), Z.data_ptr(), B, G, N, N, st),
            "khop_fwd(chain: K-1 hops)": lambda: L.gf_khop_panel(plans, 1, 0, Z.data_ptr(), B, G, K, st),
            "contract_fwd": lambda: L.gf_contract_panel(Z.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), B, N, N, G, F, 1, K, 0, st),
            "pack_dy": lambda: L.gf_pack_panels(dy.data_ptr(), P.data_ptr(), B, F, N, N, st),
            "khop_bwd(chain: K-1 hops)": lambda: L.gf_khop_panel(plans, 1, 1, P.data_ptr(), B, F, K, st),
            "grad_taps (separate)": lambda: L.gf_grad_taps_panel(Z.data_ptr(), P.data_ptr(), dh.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, B, N, G, F, 1, K, st),
            "contract_bwd (separate)": lambda: L.gf_contract_panel(P.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), B, N, N, G, F, 1, K, 1, st),
        }
    else:
        calls = {
            "layout_in": lambda: L.gf_layout_bgn_to_bng(x.data_ptr(), Z.data_ptr(), B, G, N, N, st),
            "khop_fwd(K-1 hops)": lambda: L.gf_khop(plans, 1, 0, Z.data_ptr(), B, G, K, st),
            "contract_fwd": lambda: L.gf_contract(Z.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), B, N, N, G, F, 1, K, 0, st),
            "layout_dy": lambda: L.gf_layout_bgn_to_bng(dy.data_ptr(), P.data_ptr(), B, F, N, N, st),
            "grad_taps": lambda: L.gf_grad_taps(Z.data_ptr(), P.data_ptr(), dh.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, B, N, G, F, 1, K, st),
            "khop_bwd(K-1 hops)": lambda: L.gf_khop(plans, 1, 1, P.data_ptr(), B, F, K, st),
            "contract_bwd": lambda: L.gf_contract(P.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), B, N, N, G, F, 1, K, 1, st),
        }
    calls["lsigf_forward(whole)"] = lambda: L.gf_lsigf_forward(plans, 1, x.data_ptr(), wt.data_ptr(), b.data_ptr(), Z.data_ptr(), y.data_ptr(),
                                                               B, G, F, K, N, st)
    calls["lsigf_backward(whole)"] = lambda: L.gf_lsigf_backward(plans, 1, dy.data_ptr(), Z.data_ptr(), wt.data_ptr(), P.data_ptr(), dx.data_ptr(),
                                                                 dh.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, B, G, F, K, N, st)
    out = {}
    for name, fn in calls.items():
        ts = timed_hip_events(lambda: _lib.check(fn(), name), 12)[2:]
        out[name] = round(float(np.median(ts)), 4)
    return out


if __name__ == "__main__":
    main()
