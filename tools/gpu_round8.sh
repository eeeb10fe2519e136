#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
P=gpurun_out/prof8; rm -rf $P; mkdir -p $P
V="spmm_algo=0 spmm_bt=2 spmm_spw=1 spmm_store=2 spmm_ucap=8"
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" \
         "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_REQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
         "GRBM_GUI_ACTIVE GRBM_SPI_BUSY SPI_CSN_BUSY SPI_CSN_WAVE SPI_RA_REQ_NO_ALLOC_CSN"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $P/p$i -o pmc -- python tools/spmm_probe.py cfg2 3 $V > $P/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof8/p*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "spmm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[2], {k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
grep -il "error\|invalid\|not found" $P/*.log | head
