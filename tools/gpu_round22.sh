#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r22; rm -rf $O; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; cut -c1-600 $O/bench_default.json
timeout 300 python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-300 $O/bench_cfg4.json
timeout 300 python tools/panel_sweep.py mid5k n2k cfg3 2>&1 | grep -E "==|stagger=0|L2" | head -40 > $O/panel_small.log; cat $O/panel_small.log
