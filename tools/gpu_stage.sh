#!/bin/bash
# One gpurun call = one stage.  Usage (on the GPU box, from the repo root): bash tools/gpu_stage.sh <stage> [outdir]
# Everything a stage prints goes to gpurun_out/<outdir>/ so it is merged back.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp GFHIP_EXPERIMENTS=1
S=${1:-chain}; O=gpurun_out/${2:-$S}; mkdir -p $O
case $S in
chain)   # first contact of the chain kernel: parity, then the cfg2 step with and without it, then the cfg4 L2-window bound
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or pipelines or edge_cases or hop_panel" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
  timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --detail > $O/bench_cfg2_chain.json 2> $O/bench_cfg2_chain.err; tail -2 $O/bench_cfg2_chain.err; cat $O/bench_cfg2_chain.json
  timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --tune panel_chain=0 > $O/bench_cfg2_perhop.json 2> $O/bench_cfg2_perhop.err; cat $O/bench_cfg2_perhop.json
  timeout 600 tools/l2window_bound > $O/l2window.log 2>&1; cat $O/l2window.log
  ;;
chain2)  # chain kernel iteration: parity of the chain tests + the cfg2 step
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or pipelines_agree or edge_cases" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
  timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --detail > $O/bench_cfg2_chain.json 2> $O/bench_cfg2_chain.err; grep breakdown $O/bench_cfg2_chain.err; cat $O/bench_cfg2_chain.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'])"
  ;;
probe)   # chain kernel time split
  timeout 600 python tools/chain_probe.py > $O/probe.log 2>&1; cat $O/probe.log
  ;;
variants)  # A/B library builds: probe each alegnn_amd/libgfhip_*.so
  for so in graph-neural-networks_amd/alegnn_amd/libgfhip_*.so; do echo "== $so"; PROBE_DEGS=0,10 GFHIP_LIB=$PWD/$so timeout 300 python tools/chain_probe.py 10000 256 2>&1 | grep -v amdgpu.ids; done > $O/variants.log 2>&1; cat $O/variants.log
  ;;
evvariants)  # A/B library builds on the EVGF workload
  for so in graph-neural-networks_amd/alegnn_amd/libgfhip*.so; do echo "== $so"; GFHIP_LIB=$PWD/$so timeout 200 python bench.py --workload cfg5 --no-cpu-baseline --steps 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['launch_ms'])"; done > $O/evvariants.log 2>&1; cat $O/evvariants.log
  ;;
trace)   # phase timeline of the chain kernel (trace build)
  for d in 0 10 20; do echo "== degree $d"; GFHIP_LIB=$PWD/graph-neural-networks_amd/alegnn_amd/libgfhip_trace.so timeout 300 python tools/chain_trace.py $d 2>&1 | grep -v amdgpu.ids | head -12; done > $O/trace.log 2>&1; cat $O/trace.log
  ;;
ceiling) # memory-system ceilings + chain store modes
  timeout 120 tools/hbm_ceiling > $O/hbm_ceiling.log 2>&1; cat $O/hbm_ceiling.log
  for sm in 0 1 2; do echo "== spmm_store=$sm"; PROBE_DEGS=10 timeout 200 python tools/chain_probe.py 10000 256 spmm_store=$sm 2>&1 | grep deg; done > $O/store_modes.log 2>&1; cat $O/store_modes.log
  ;;
chain3)  # chain parity + probe
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or pipelines_agree or edge_cases" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
  timeout 600 python tools/chain_probe.py > $O/probe.log 2>&1; cat $O/probe.log
  ;;
suite)   # the whole GPU suite (what the driver runs at round end) + smoke
  timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -8 $O/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
  ;;
bench)   # the driver's bench line + the other BASELINE workloads (CPU baselines included)
  for w in cfg4 cfg2 cfg1 cfg3 cfg5; do timeout 900 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; tail -2 $O/bench_$w.err | cut -c1-300; cut -c1-1500 $O/bench_$w.json; done
  ;;
pmc)     # PMC traffic of the dominant kernels (own passes)
  bash tools/pmc_collect.sh cfg4 spmm_sell_kernel r02 > $O/pmc_cfg4.log 2>&1; tail -1 $O/pmc_cfg4.log | cut -c1-800
  bash tools/pmc_collect.sh cfg2 spmm_chain_kernel r02 > $O/pmc_cfg2.log 2>&1; tail -1 $O/pmc_cfg2.log | cut -c1-800
  ;;
fullsize) # full-size parity tests + the two SelectionGNN bench lines + PMC traffic
  timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --durations=5 > $O/pytest_fullsize.log 2>&1; tail -12 $O/pytest_fullsize.log
  for w in cfg1 cfg3; do timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; tail -2 $O/bench_$w.err | cut -c1-300; cut -c1-1800 $O/bench_$w.json; done
  bash tools/gpu_stage.sh pmc $2
  ;;
stats)   # rocprofv3 --kernel-trace --stats of the bench command, per workload (own runs: no counters here)
  for w in cfg4 cfg2 cfg5 cfg3; do
    rm -rf $O/kt_$w; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o bench -- python bench.py --workload $w --no-cpu-baseline > $O/bench_prof_$w.json 2> $O/bench_prof_$w.err
    f=$(find $O/kt_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_${w}_kernel_stats.csv && head -8 $f | cut -c1-200
    rm -rf $O/kt_$w
  done
  bash tools/pmc_collect.sh cfg5 ev_hop_lds4_kernel r02 > $O/pmc_cfg5.log 2>&1; tail -1 $O/pmc_cfg5.log | cut -c1-600
  ;;
*) echo "unknown stage $S"; exit 2;;
esac
