#!/bin/bash
# Where do the L2 misses of the shipped sweep come from?  Fabric read requests / L2 hits / vector-cache latency of the fused chain at config 4:
# default, without the scalar prefetch (lead longer than the sweep), without stores (single hops out of HBM, timing / traffic only).
cd "$(dirname "$0")/.."; export TMPDIR=/tmp GFHIP_EXPERIMENTS=1
O=gpurun_out/r05_h_knobs; mkdir -p $O
pmc() {  # pmc <tag> <counters...> -- <hop_probe args>
  local tag=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf $O/pm; timeout 100 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $O/pm -o pmc -- python tools/hop_probe.py "$@" > $O/pm_$tag.log 2>&1 || echo "pmc pass failed: $tag"
  python3 - "$O" "$tag" <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1:3]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msweep" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"{tag:22s} {c:34s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
PY
  rm -rf $O/pm
}
for v in "default:spmm_pfd=16" "no_prefetch:spmm_pfd=24" "lead12:spmm_pfd=12"; do
  pmc l2_${v%%:*} TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- cfg4 3 ${v#*:}
  pmc tcp_${v%%:*} TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -- cfg4 3 ${v#*:}
done 2>&1 | tee $O/miss_split.log
PROBE_SINGLE=1 pmc l2_single_nostore TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- cfg4 3 spmm_algo=5 spmm_store=3 2>&1 | tee -a $O/miss_split.log
PROBE_SINGLE=1 pmc l2_single TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- cfg4 3 spmm_algo=5 2>&1 | tee -a $O/miss_split.log
rm -f $O/pm_*.log
