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
*) echo "unknown stage $S"; exit 2;;
esac
