#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r17; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "evgf or edge_variant" > $O/pytest_evgf.log 2>&1; echo "pytest exit $?" >> $O/pytest_evgf.log; tail -4 $O/pytest_evgf.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ev -- python tools/evgf_bench.py > $O/evgf.json 2> $O/evgf.err
cat $O/evgf.json
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/evgf_kernel_stats_full.csv
python - <<'PY'
import csv
rows = list(csv.reader(open("gpurun_out/r17/evgf_kernel_stats_full.csv")))
for r in rows[:9]:
    print(r[0][:90], r[1:5])
PY
rm -rf $O/prof
