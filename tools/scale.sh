#!/bin/bash
# Weak-scaling record of bench.py on the GPUs of ONE node (batch-axis data parallelism, one RCCL all-reduce per step):
#   bash tools/scale.sh [workload] [gpu counts ...]        default: cfg4 on 1 2 4 8 (counts above the node's GPU count are skipped)
# Every run is the driver's own command line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...),
# N = 1 included, so that even the single-GPU line goes through RCCL.  Output: one JSON line per N in gpurun_out/scale/<workload>_N.json
# and a table (N, ms/step, edges*taps/s, x vs N = 1).  Also records, at N = 1, the step time with the collective forced
# (GFHIP_FORCE_COLLECTIVES=1: RCCL all-reduce of the gradient bucket with one rank) next to the plain single-process line.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
WL=${1:-cfg4}; shift; NS=${@:-1 2 4 8}
O=gpurun_out/scale; mkdir -p $O
NG=$(python -c "import torch; print(torch.cuda.device_count())")
CORES=$(nproc)
timeout 900 python bench.py --workload $WL --no-cpu-baseline > $O/${WL}_single_process.json 2> $O/${WL}_single_process.err
GFHIP_FORCE_COLLECTIVES=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --workload $WL --no-cpu-baseline > $O/${WL}_forced_collective_1.json 2> $O/${WL}_forced_collective_1.err
P=29520
for N in $NS; do
  [ "$N" -gt "$NG" ] && { echo "skip N=$N (node has $NG GPUs)"; continue; }
  P=$((P+1))
  OMP_NUM_THREADS=$((CORES / N > 0 ? CORES / N : 1)) timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $P \
      bench.py --gpus $N --workload $WL --no-cpu-baseline > $O/${WL}_$N.json 2> $O/${WL}_$N.err || tail -3 $O/${WL}_$N.err
done
python - "$O" "$WL" <<'PY'
import glob, json, sys
O, wl = sys.argv[1:3]
def load(f):
    try:
        return json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception:
        return None
base = None
print(f"{'run':28s} {'ms/step':>9s} {'edges*taps/s':>14s} {'x N=1':>7s}")
for tag in ["single_process", "forced_collective_1"] + [str(n) for n in (1, 2, 4, 8)]:
    d = load(f"{O}/{wl}_{tag}.json")
    if d is None:
        continue
    if tag == "1":
        base = d["value"]
    print(f"{wl + ' ' + tag:28s} {d['ms_per_step']:9.3f} {d['value']:14.4g} {(d['value'] / base if base and tag.isdigit() else float('nan')):7.2f}")
PY
