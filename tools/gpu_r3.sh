#!/bin/bash
# Round-3 GPU stages.  Usage (on the GPU box, from the repo root): bash tools/gpu_r3.sh <stage> [outdir]
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
S=${1:-sweep}; O=gpurun_out/${2:-r3_$S}; mkdir -p $O
case $S in
sweep)   # go / no-go of the XCD-synchronous source sweep (tools/xcd_sweep.hip) + fresh counters of the production hop at HEAD
  timeout 300 tools/xcd_sweep accum=1 > $O/sweep_accum1.log 2>&1; grep -c "^cfg" $O/sweep_accum1.log; grep "^#" $O/sweep_accum1.log; grep "^cfg" $O/sweep_accum1.log | sort -k9 -n | head -14; grep "^cfg" $O/sweep_accum1.log | grep "lag=-1"
  timeout 200 tools/xcd_sweep accum=0 lag=-1,2,4 > $O/sweep_accum20.log 2>&1; grep "^cfg" $O/sweep_accum20.log | sort -k9 -n | head -8
  best=$(grep "^cfg" $O/sweep_accum1.log | awk '$NF+0 < 1e-3' | sort -k9 -n | head -1 | awk '{print $2}')
  echo "best cfg of accum=1: $best"
  for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_sweep -o pmc -- tools/xcd_sweep accum=1 only=$best iters=2 > $O/pmc_sweep.log 2>&1
    python - "$O" "$c" <<'PY'
import csv, glob, sys, collections
O, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pmc_sweep/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sweep_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("PMC sweep_kernel", {k: sum(v) / len(v) for k, v in agg.items()})
PY
    rm -rf $O/pmc_sweep
  done 2>&1 | tee $O/pmc_sweep_summary.log
  rocprofv3 --list-avail > $O/list_avail.log 2>&1
  GFHIP_EXPERIMENTS=1 bash tools/pmc_collect.sh cfg4 spmm_sell_kernel r03 > $O/pmc_cfg4.log 2>&1; tail -1 $O/pmc_cfg4.log | cut -c1-1200
  cp gpurun_out/pmc_cfg4/r03_cfg4_pmc.json $O/ 2>/dev/null
  ;;
pmcsweep)  # L2 hit rate / fabric bytes of chosen sweep configurations (ids of the accum=1 enumeration)
  for id in ${CFGS:-0 3 10}; do
    for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
      timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_sweep -o pmc -- tools/xcd_sweep accum=1 only=$id iters=2 > $O/pmc_sweep.log 2>&1
      grep "^cfg" $O/pmc_sweep.log
      python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pmc_sweep/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sweep_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
if "TCC_HIT_sum" in m: print("   L2 hit rate %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])), m)
if "FETCH_SIZE" in m: print("   fabric reads %.2f GB" % (m["FETCH_SIZE"] * 2048 / 1e9))
PY
      rm -rf $O/pmc_sweep
    done
  done 2>&1 | tee $O/pmc_sweep_cfgs.log
  ;;
suite)   # the whole GPU suite as the driver runs it (product configuration: no GFHIP_EXPERIMENTS) + smoke
  timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
  # the same kernels in the PRODUCT configuration (gf_tune refused): full-size parity + the per-sample GSO family
  GFHIP_EXPERIMENTS=0 timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_db.py -x -q -m gpu > $O/pytest_product_mode.log 2>&1; tail -3 $O/pytest_product_mode.log
  ;;
bench)   # the driver's bench line + the other BASELINE workloads (CPU baselines included)
  for w in cfg4 cfg2 cfg1 cfg3 cfg5; do timeout 900 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; tail -2 $O/bench_$w.err | cut -c1-300; cut -c1-1800 $O/bench_$w.json; done
  ;;
pmc)     # counters of the dominant kernels on the sources of this snapshot (own passes, no tracing domains)
  GFHIP_EXPERIMENTS=1 bash tools/pmc_collect.sh cfg4 spmm_sell_kernel r03 > $O/pmc_cfg4.log 2>&1; tail -1 $O/pmc_cfg4.log | cut -c1-600; cp gpurun_out/pmc_cfg4/r03_cfg4_pmc.json $O/
  GFHIP_EXPERIMENTS=1 bash tools/pmc_collect.sh cfg2 spmm_chain_kernel r03 > $O/pmc_cfg2.log 2>&1; tail -1 $O/pmc_cfg2.log | cut -c1-600; cp gpurun_out/pmc_cfg2/r03_cfg2_pmc.json $O/
  GFHIP_EXPERIMENTS=1 bash tools/pmc_collect.sh cfg5 ev_hop_lds4_kernel r03 > $O/pmc_cfg5.log 2>&1; tail -1 $O/pmc_cfg5.log | cut -c1-600; cp gpurun_out/pmc_cfg5/r03_cfg5_pmc.json $O/
  bash tools/pmc_mfma.sh cfg4 r03 > $O/pmc_mfma_cfg4.log 2>&1; tail -3 $O/pmc_mfma_cfg4.log | cut -c1-500; cp gpurun_out/pmc_mfma_cfg4/*.json $O/
  bash tools/pmc_mfma.sh cfg2 r03 > $O/pmc_mfma_cfg2.log 2>&1; tail -3 $O/pmc_mfma_cfg2.log | cut -c1-500; cp gpurun_out/pmc_mfma_cfg2/*.json $O/
  ;;
stats)   # rocprofv3 --kernel-trace --stats of the bench command, per workload (own runs: no counters here)
  for w in cfg4 cfg2 cfg5 cfg3; do
    rm -rf $O/kt_$w; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o bench -- python bench.py --workload $w --no-cpu-baseline > $O/bench_prof_$w.json 2> $O/bench_prof_$w.err
    f=$(find $O/kt_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_${w}_kernel_stats.csv && head -8 $f | cut -c1-200
    rm -rf $O/kt_$w
  done
  ;;
scale)   # multi-GPU rehearsal as far as one GPU allows: single process vs torchrun N=1 vs forced collective
  bash tools/scale.sh cfg4 1 > $O/scale_cfg4.log 2>&1; cat $O/scale_cfg4.log; cp gpurun_out/scale/*.json $O/ 2>/dev/null
  bash tools/scale.sh cfg2 1 > $O/scale_cfg2.log 2>&1; cat $O/scale_cfg2.log; cp gpurun_out/scale/cfg2*.json $O/ 2>/dev/null
  ;;
diag)    # where do the gather kernels lose time?  TCP / TLB / SQ counters of four targets, own passes per group.
         # (Round 3: the TA_* / GRBM group aborted rocprofv3 under the python targets and every pass then ran into its timeout -- 26 GPU-minutes
         #  lost; that group is gone and the per-pass timeout is 100 s.  tools/tlb_counters.sh is the bounded version that was used.)
  run_pmc() {  # $1 = tag, $2 = kernel substring, rest = command
    tag=$1; kn=$2; shift 2
    for grp in "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
               "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
               "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
               "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
               "TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_HIT_sum TCC_MISS_sum"; do
      rm -rf $O/pm; timeout 100 rocprofv3 --pmc $grp --output-format csv -d $O/pm -o pmc -- "$@" > $O/pm.log 2>&1
      python - "$O" "$tag" "$kn" <<'PY'
import csv, glob, sys, collections
O, tag, kn = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kn in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{tag:28s} {k:42s} {sum(v[len(v)//2:]) / max(1, len(v) - len(v)//2):16.0f}   ({len(v)} launches, second half averaged)")
PY
    done
  }
  export GFHIP_EXPERIMENTS=1
  run_pmc "microbench 4 MB panel" gather_kernel tools/gather_ceiling one 32768 | tee $O/diag_microbench_4MB.log
  run_pmc "microbench 8 MB panel" gather_kernel tools/gather_ceiling one 65536 | tee $O/diag_microbench_8MB.log
  run_pmc "hop ER" spmm_sell_kernel python tools/hop_probe.py cfg4 4 | tee $O/diag_hop_er.log
  PROBE_GRAPH=band run_pmc "hop band" spmm_sell_kernel python tools/hop_probe.py cfg4 4 | tee $O/diag_hop_band.log
  run_pmc "sweep lag=8 depth=2" sweep_kernel tools/xcd_sweep wg=2 accum=1 lag=8 depth=2 iters=3 | tee $O/diag_sweep.log
  rm -rf $O/pm
  ;;
*) echo "unknown stage $S"; exit 2;;
esac
