#!/bin/bash
# Per-kernel averages of the bench command for a few workloads (rocprofv3 --kernel-trace --stats, no counters): the A/B tool for
# kernel changes -- compare with the committed *_kernel_stats.csv of the previous state.   usage: tools/ab_stats.sh <outdir> [workloads...]
O=${1:-gpurun_out/ab}; shift
mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for w in ${@:-cfg2 cfg4 cfg3}; do
  rm -rf $O/kt_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o bench -- python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err
  f=$(find $O/kt_$w -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/${w}_kernel_stats.csv; echo "== $w"; head -9 $f | cut -d, -f1-4 | cut -c1-150; else echo "== $w: no stats"; tail -5 $O/bench_$w.err; fi
  python - $O/bench_$w.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ms_per_step", round(d["ms_per_step"], 4), "value", f'{d["value"]:.4g}', "frac", d["roofline"]["frac"], "mfma", (d.get("mfma") or {}).get("launch_ms"))
except Exception as e:
    print("   no bench line:", e)
PY
  rm -rf $O/kt_$w
done
