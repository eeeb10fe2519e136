#!/bin/bash
# Final records of a round on the final sources, most important first: the default bench line with hash-matched PMC traffic, rocprofv3 kernel
# stats of the same command, GPU suite, smoke, MfmaUtil, the other workloads' counter summaries and lines.
#   bash tools/gpu_final.sh <tag> [budget seconds, default 3000]    -> gpurun_out/<tag>/ (copy to profiles/<tag>/)
# A stage is skipped (and says so) when its estimated duration no longer fits the budget: FULL=1 adds the cfg1 / cfg3 lines, the comparator
# line and tools/scale.sh (ONLY=extras: just those).  SUITE=subset runs the tests of the last change only (the driver runs the whole suite at round end).
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
T=${1:-r06_z_final}; LIMIT=${2:-3000}; O=gpurun_out/$T; mkdir -p $O
left() { [ $((SECONDS + ${2:-0})) -lt $LIMIT ] || { echo "[$SECONDS s] skipped (budget): $1"; return 1; }; echo "[$SECONDS s] $1"; }
line() { python3 -c "import json;d=json.load(open('$O/bench_$1.json'));r=d.get('roofline') or {};print('$1',d['ms_per_step'],d['value'],r.get('frac'),r.get('traffic'),r.get('launch_ms'))"; }
if [ "${ONLY:-}" != extras ]; then
# counter summaries on THESE sources first (bench.py reports traffic / MfmaUtil only from a summary whose source hash matches), then the lines
left "pmc cfg4" 150 && { timeout 280 bash tools/pmc_collect.sh cfg4 spmm_msweep_kernel r06 > $O/pmc_collect_cfg4.log 2>&1
  cp gpurun_out/pmc_cfg4/r06_cfg4_pmc.json profiles/r06_cfg4_pmc.json && cp profiles/r06_cfg4_pmc.json $O/; }
left "pmc mfma cfg4" 80 && { timeout 200 bash tools/pmc_mfma.sh cfg4 r06 > $O/pmc_mfma_cfg4.log 2>&1
  cp gpurun_out/pmc_mfma_cfg4/r06_cfg4_*_mfma_pmc.json profiles/ 2>/dev/null; cp gpurun_out/pmc_mfma_cfg4/r06_cfg4_*_mfma_pmc.json $O/ 2>/dev/null; }
left "bench cfg4" 70 && { timeout 300 python bench.py > $O/bench_cfg4.json 2> $O/bench_cfg4.err; line cfg4; }
left "kernel stats cfg4" 50 && { ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_cfg4 -o bench -- python $OLDPWD/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $OLDPWD/$O/prof_cfg4.log 2>&1 )
  f=$(find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_cfg4_kernel_stats.csv; rm -rf $O/prof_cfg4; }
left "pmc + line cfg2" 90 && { timeout 150 bash tools/pmc_collect.sh cfg2 spmm_chain_kernel r06 > $O/pmc_collect_cfg2.log 2>&1; cp gpurun_out/pmc_cfg2/r06_cfg2_pmc.json profiles/ && cp profiles/r06_cfg2_pmc.json $O/
  timeout 200 python bench.py --workload cfg2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; line cfg2; }
left "pmc + line cfg5" 110 && { timeout 200 bash tools/pmc_collect.sh cfg5 ev_hop_lds4_kernel r06 > $O/pmc_collect_cfg5.log 2>&1; cp gpurun_out/pmc_cfg5/r06_cfg5_pmc.json profiles/ && cp profiles/r06_cfg5_pmc.json $O/
  timeout 300 python bench.py --workload cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; line cfg5; }
left "tests of the last change" 80 && { timeout 300 python -m pytest tests/test_gpu_product_mode.py tests/test_gpu_msweep.py -x -q -m gpu -k "sweep_image or fused or replays or refused or wide" > $O/pytest_gpu_subset.log 2>&1; tail -3 $O/pytest_gpu_subset.log; }
left "smoke" 20 && { timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log; }
[ "${SUITE:-full}" = subset ] || { left "gpu suite" 280 && { timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; }; }
fi
[ "${FULL:-0}" = 1 ] || [ "${ONLY:-}" = extras ] || { echo "[$SECONDS s] done (FULL=1 adds cfg1, cfg3, comparator, scale)"; exit 0; }
for wl in cfg3 cfg1; do
  left "line $wl" 30 && { timeout 600 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; line $wl; }
done
left "scale" 100 && { bash tools/scale.sh cfg4 1 > $O/scale_cfg4.log 2>&1; tail -4 $O/scale_cfg4.log; mkdir -p $O/scale; cp gpurun_out/scale/cfg4_* $O/scale/ 2>/dev/null; }
left "comparator" 130 && { timeout 600 python bench.py --workload cfg4 --comparator --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_cfg4_with_comparator.json 2> $O/bench_cfg4_with_comparator.err; python3 -c "import json;d=json.load(open('$O/bench_cfg4_with_comparator.json'));print('cfg4 comparator',d.get('no_rewrite_comparator'))"; }
echo "[$SECONDS s] done"
