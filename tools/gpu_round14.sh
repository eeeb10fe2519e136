#!/bin/bash
# full parity suite, bench (both pipelines), rocprof stats, PMC traffic of the panel hop
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r23; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 400 python bench.py --detail > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cat $O/bench_cfg2.json
timeout 400 python bench.py --detail --pipeline 1 --no-cpu-baseline > $O/bench_cfg2_pipe1.json 2> $O/bench_cfg2_pipe1.err; cat $O/bench_cfg2_pipe1.json
GFHIP_FORCE_COLLECTIVES=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; cat $O/bench_torchrun1.json; tail -3 $O/bench_torchrun1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats_full.csv
python - <<'PY'
import csv
rows = list(csv.reader(open("gpurun_out/r23/bench_kernel_stats_full.csv")))
with open("gpurun_out/r23/bench_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:150]
        w.writerow(r)
print(open("gpurun_out/r23/bench_kernel_stats.csv").read()[:2500])
PY
rm -rf $O/prof $O/bench_kernel_stats_full.csv
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -o pmc -- python tools/panel_probe.py cfg2 3 > $O/pmc$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
tot = {}
for d in sorted(glob.glob("gpurun_out/r23/pmc*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "spmm_panel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        tot[k] = sum(v) / len(v)
print(tot)
json.dump(tot, open("gpurun_out/r23/pmc_panel_raw.json", "w"))
PY
rm -rf $O/pmc*/
