#!/bin/bash
# Counters of the node-major hop kernels (SELL-8 and STREAM) on one plan: one rocprofv3 pass per counter group, both kernels in the
# same process.   usage: [PROBE_GRAPH=band] bash tools/pmc_hop.sh <outdir> [stream variant, default spmm_algo=2+spmm_wps=4+spmm_tk=2+spmm_nc=1]
cd "$(dirname "$0")/.."; export TMPDIR=/tmp GFHIP_EXPERIMENTS=1
O=$1; mkdir -p $O; SV=${2:-spmm_algo=2+spmm_wps=4+spmm_tk=2+spmm_nc=1}; KN=${3:-spmm_stream}
G=${PROBE_GRAPH:-er}
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  rm -rf $O/pm; timeout 100 rocprofv3 --pmc $grp --output-format csv -d $O/pm -o pmc -- python tools/hop_probe.py cfg4 3 v:spmm_algo=3 v:$SV > $O/pm.log 2>&1 || echo "group failed: $grp"
  python3 - "$O" "$G" "$KN" <<'PY'
import csv, glob, sys, collections
O, G, KN = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        tag = "stream" if KN in kn else ("sell" if "spmm_sell" in kn else None)
        if tag:
            agg[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (tag, k), v in sorted(agg.items()):
    print(f"{G:5s} {tag:7s} {k:42s} {sum(v)/len(v):18.0f}  ({len(v)} launches)")
PY
done 2>&1 | tee $O/pmc_hop_$G.log
rm -rf $O/pm
