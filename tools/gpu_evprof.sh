#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out/r33; rm -rf $O; mkdir -p $O
B=${1:-16}
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ev -- python tools/evgf_bench.py 50000 $B > $O/ev.log 2>&1
grep workload $O/ev.log | tee $O/evgf_cfg5.json
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
test -n "$f" && python - "$f" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:11]:
    print(r[0][:75], r[1], round(float(r[3]) / 1e3, 1), "us", r[4], "%")
PY
test -n "$f" && cp "$f" $O/evgf_kernel_stats.csv; rm -rf $O/prof
