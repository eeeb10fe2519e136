#!/bin/bash
# round-9 GPU call: full parity suite (incl. EVGF), SpMM knob sweep (incl. prefetch workgroups), bench + rocprof stats, EVGF cfg5 timing
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r9; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 300 python tools/spmm_sweep.py cfg2 cfg4 > $O/sweep.log 2>&1; head -12 $O/sweep.log
timeout 300 python tools/evgf_bench.py > $O/evgf_cfg5.json 2> $O/evgf_cfg5.err; cat $O/evgf_cfg5.json; tail -3 $O/evgf_cfg5.err
timeout 400 python bench.py --detail > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cat $O/bench_cfg2.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
head -14 $O/bench_kernel_stats.csv
rm -rf $O/prof
