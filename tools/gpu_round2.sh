#!/bin/bash
# gpurun call 2: parity of the SELL kernel, knob sweep, rocprofv3 kernel stats of bench.py, PMC passes on the hop kernel.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 600 python tools/spmm_sweep.py cfg2 cfg4 > gpurun_out/sweep.log 2>&1; echo "sweep exit $?" >> gpurun_out/sweep.log
grep -E "==|DEFAULT" gpurun_out/sweep.log; for w in cfg2 cfg4; do awk "/== $w/{f=1;next} /==/{f=0} f" gpurun_out/sweep.log | head -6; done
P=gpurun_out/prof; rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $P/bench_stats.json 2> $P/bench_stats.err
for w in cfg2 cfg4; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | tr ' ' '+')
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $P/pmc_${w}_$tag -o pmc -- python tools/spmm_probe.py $w 3 > $P/pmc_${w}_$tag.log 2>&1
  done
done
find $P -name "*.csv" | head -40; du -sh $P
