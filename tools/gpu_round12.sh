#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
P=gpurun_out/r12; rm -rf $P; mkdir -p $P
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
         "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_LDS SQ_INSTS_WAVE32_LDS" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $P/p$i -o pmc -- python tools/panel_probe.py cfg2 3 > $P/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r12/p*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "spmm_panel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[2], {k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
grep -il "error\|invalid\|not found" $P/*.log | head
rm -rf $P/p*/
