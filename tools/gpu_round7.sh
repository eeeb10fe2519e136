#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "spmm or lsigf_matches" > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
timeout 600 python tools/spmm_sweep.py cfg2 cfg4 > gpurun_out/sweep.log 2>&1
for w in cfg2 cfg4; do echo "== $w"; awk "/== $w/{f=1;next} /==/{f=0} f" gpurun_out/sweep.log | head -14; done; grep DEFAULT gpurun_out/sweep.log
