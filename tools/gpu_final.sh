#!/bin/bash
# round-end validation: full parity suite, smoke, bench (cfg2 default, cfg4), torchrun N=1 with forced collectives, rocprof kernel
# stats of the bench command, caller timings.  Every step under its own timeout.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r30; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
timeout 400 python bench.py --detail > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cat $O/bench_cfg2.json
timeout 300 python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cat $O/bench_cfg4.json
GFHIP_FORCE_COLLECTIVES=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; cat $O/bench_torchrun1.json; tail -2 $O/bench_torchrun1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if test -n "$f"; then python - "$f" $O/bench_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:150]
        w.writerow(r)
print(open(sys.argv[2]).read()[:1800])
PY
fi
rm -rf $O/prof
timeout 200 python tools/callers_bench.py > $O/callers.jsonl 2> $O/callers.err; cat $O/callers.jsonl
