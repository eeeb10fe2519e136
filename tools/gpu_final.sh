#!/bin/bash
# Final records of a round on the final sources: GPU suite, smoke, the default bench line with hash-matched PMC traffic, rocprofv3 kernel
# stats of the same command, the other workloads' lines.   bash tools/gpu_final.sh <tag>    -> gpurun_out/<tag>/ (copy to profiles/<tag>/)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
T=${1:-r05_e_final}; O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
# counter summaries on THESE sources first (bench.py reports traffic / MfmaUtil only from a summary whose source hash matches), then the lines
bash tools/pmc_collect.sh cfg4 spmm_msweep_kernel r05 > $O/pmc_collect_cfg4.log 2>&1
cp gpurun_out/pmc_cfg4/r05_cfg4_pmc.json profiles/r05_cfg4_pmc.json; cp profiles/r05_cfg4_pmc.json $O/
bash tools/pmc_collect.sh cfg2 spmm_chain_kernel r05 > $O/pmc_collect_cfg2.log 2>&1; cp gpurun_out/pmc_cfg2/r05_cfg2_pmc.json profiles/ && cp profiles/r05_cfg2_pmc.json $O/
bash tools/pmc_collect.sh cfg5 ev_hop_lds4_kernel r05 > $O/pmc_collect_cfg5.log 2>&1; cp gpurun_out/pmc_cfg5/r05_cfg5_pmc.json profiles/ && cp profiles/r05_cfg5_pmc.json $O/
bash tools/pmc_mfma.sh cfg4 r05 > $O/pmc_mfma_cfg4.log 2>&1; cp gpurun_out/pmc_mfma_cfg4/r05_cfg4_*_mfma_pmc.json profiles/ 2>/dev/null; cp gpurun_out/pmc_mfma_cfg4/r05_cfg4_*_mfma_pmc.json $O/ 2>/dev/null
python bench.py > $O/bench_cfg4.json 2> $O/bench_cfg4.err; python3 -c "import json;d=json.load(open('$O/bench_cfg4.json'));print('cfg4',d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['traffic'],d['roofline']['launch_ms'])"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_cfg4 -o bench -- python $OLDPWD/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $OLDPWD/$O/prof_cfg4.log 2>&1 )
f=$(find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_cfg4_kernel_stats.csv; rm -rf $O/prof_cfg4
for wl in cfg2 cfg3 cfg1 cfg5; do
  timeout 900 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err
  python3 -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl',d['ms_per_step'],d['value'],d['roofline']['frac'] if d.get('roofline') else None, (d.get('no_rewrite_comparator') or {}).get('ms_per_step'))"
done
timeout 600 python bench.py --workload cfg4 --comparator --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_cfg4_with_comparator.json 2> $O/bench_cfg4_with_comparator.err; python3 -c "import json;d=json.load(open('$O/bench_cfg4_with_comparator.json'));print('cfg4 comparator',d.get('no_rewrite_comparator'))"
bash tools/scale.sh cfg4 1 > $O/scale_cfg4.log 2>&1; tail -4 $O/scale_cfg4.log; mkdir -p $O/scale; cp gpurun_out/scale/cfg4_* $O/scale/ 2>/dev/null
