#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r16; rm -rf $O; mkdir -p $O
GFHIP_FORCE_COLLECTIVES=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; cat $O/bench_torchrun1.json; tail -3 $O/bench_torchrun1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ev -- python tools/evgf_bench.py > $O/evgf.json 2> $O/evgf.err
cat $O/evgf.json
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/evgf_kernel_stats_full.csv
python - <<'PY'
import csv
rows = list(csv.reader(open("gpurun_out/r16/evgf_kernel_stats_full.csv")))
for r in rows[:14]:
    print(r[0][:90], r[1:5])
PY
rm -rf $O/prof
