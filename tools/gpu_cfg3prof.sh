#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out/r35; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o sel -- python tools/selgnn_bench.py cfg3 > $O/sel.log 2>&1
grep workload $O/sel.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
test -n "$f" && python - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
tot = sum(float(r[2]) for r in rows)
for r in rows[:16]:
    print(r[0][:95], r[1], round(float(r[3]) / 1e3, 1), "us", round(100 * float(r[2]) / tot, 1), "%")
PY
test -n "$f" && cp "$f" $O/selgnn_cfg3_kernel_stats.csv; rm -rf $O/prof
