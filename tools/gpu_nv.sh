#!/bin/bash
# NVGF kernels: parity tests, timings, per-kernel profile
O=gpurun_out/r29; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "node_variant" 2>&1 | tail -5
timeout 120 python -c "
import sys; sys.argv=['x']; sys.path.insert(0,'tools')
import callers_bench as c
c.nvgf(N=1000, B=64, G=32, F=32, K=5, M=100)
c.nvgf(N=10000, B=64, G=32, F=32, K=5, M=1000)
c.nvgf(N=10000, B=256, G=32, F=32, K=5, M=10000)
" 2>&1 | grep item | tee $O/nvgf.jsonl
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o nv -- python tools/nv_prof.py > $O/nv.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
test -n "$f" && python - "$f" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:10]:
    print(r[0][:80], r[1], round(float(r[3]) / 1e3, 1), "us", r[4], "%")
PY
