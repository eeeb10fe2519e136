#!/bin/bash
# Counters of the XCD-sweep prototype (tools/xcd_sweep) next to the production hop: is the prototype waiting for memory, for LDS or for itself?
# Standalone binary targets only (fast, bounded): bash tools/sweep_counters.sh
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out/r3_sweepctr; mkdir -p $O
run() {  # $1 tag, $2 kernel substring, rest: command
  tag=$1; kn=$2; shift 2
  for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY"; do
    rm -rf $O/pm; timeout 90 rocprofv3 --pmc $grp --output-format csv -d $O/pm -o pmc -- "$@" > $O/pm.log 2>&1 || echo "group failed: $grp"
    python3 - "$O" "$tag" "$kn" <<'PY'
import csv, glob, sys, collections
O, tag, kn = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kn in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{tag:26s} {k:34s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
PY
  done
}
run "sweep lag=8 depth=2 accum=1" sweep_kernel tools/xcd_sweep wg=2 accum=1 lag=8 depth=2 iters=3 2>&1 | tee $O/sweep_lag8.log
run "sweep unsynced accum=1" sweep_kernel tools/xcd_sweep wg=2 accum=1 lag=-1 depth=2 iters=3 2>&1 | tee $O/sweep_unsynced.log
run "sweep lag=8 gather-only" sweep_kernel tools/xcd_sweep wg=2 accum=0 lag=8 depth=2 iters=3 2>&1 | tee $O/sweep_lag8_gather_only.log
run "microbench 8 MB panel" gather_kernel tools/gather_ceiling one 65536 2>&1 | tee $O/microbench_8MB.log
rm -rf $O/pm
