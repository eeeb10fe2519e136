#!/bin/bash
# Round-5 GPU sessions (one gpurun call each): bash tools/gpu_r5.sh <stage> ; logs under gpurun_out/r05_<stage>/
cd "$(dirname "$0")/.."; export TMPDIR=/tmp GFHIP_EXPERIMENTS=1
S=$1; O=gpurun_out/r05_$S; mkdir -p $O
pmc() {  # pmc <tag> <counters...> -- <hop_probe args>: one rocprofv3 pass, per-kernel averages
  local tag=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf $O/pm; timeout 200 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $O/pm -o pmc -- python tools/hop_probe.py "$@" > $O/pm_$tag.log 2>&1 || echo "pmc pass failed: $tag"
  python3 - "$O" "$tag" <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1:3]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = "msweep" if "msweep" in kn else ("sell" if "spmm_sell" in kn else None)
        if k: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{tag:10s} {k:7s} {c:42s} {sum(v)/len(v):18.0f}  ({len(v)} launches)")
PY
  rm -rf $O/pm
}
case $S in
a)  # first contact: MFMA layout, parity, timing variants, counters
  tools/mfma4x4_probe 2>&1 | tee $O/mfma_probe.log
  timeout 900 python -m pytest tests/test_gpu_msweep.py -x -q 2>&1 | tail -15 | tee $O/pytest_msweep.log
  V="v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1+spmm_srcmask=0 v:spmm_algo=5+spmm_bar=0+spmm_srcmask=0 v:spmm_algo=5+spmm_bar=1+spmm_srcmask=1048448 v:spmm_algo=5+spmm_bar=0+spmm_srcmask=1048448 v:spmm_algo=3+spmm_srcmask=0"
  timeout 300 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "spmm hop" | tee $O/hop_er.log
  for sl in 0 20 35; do
    timeout 300 python tools/hop_probe.py cfg4 10 spmm_slack=$sl v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 v:spmm_algo=5+spmm_bar=0 2>&1 | grep "spmm hop" | tee $O/hop_er_slack$sl.log
  done
  pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 | tee $O/pmc.log
  pmc l2nb TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:spmm_algo=5+spmm_bar=0 | tee -a $O/pmc.log
  pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -- cfg4 3 v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 | tee -a $O/pmc.log
  pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY -- cfg4 3 v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 | tee -a $O/pmc.log
  pmc wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- cfg4 3 v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 | tee -a $O/pmc.log
  ;;
esac
