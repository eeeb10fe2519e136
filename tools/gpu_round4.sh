#!/bin/bash
# gpurun call 3: parity, knob sweep with store policies, PMC L2 hit-rate passes for a few variants.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python tools/spmm_sweep.py cfg2 cfg4 > gpurun_out/sweep.log 2>&1; echo "sweep exit $?" >> gpurun_out/sweep.log
for w in cfg2 cfg4; do echo "== $w"; awk "/== $w/{f=1;next} /==/{f=0} f" gpurun_out/sweep.log | head -8; grep -A0 DEFAULT gpurun_out/sweep.log; done
P=gpurun_out/prof4; rm -rf $P; mkdir -p $P
i=0
for v in "spmm_algo=0 spmm_bt=1 spmm_spw=2 spmm_store=1" "spmm_algo=0 spmm_bt=1 spmm_spw=2 spmm_store=0" "spmm_algo=0 spmm_bt=2 spmm_spw=2 spmm_store=1"; do
  i=$((i+1))
  for w in cfg2 cfg4; do
    for c in "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES TA_BUSY_avr TA_TA_BUSY_sum"; do
      tag=$(echo $c | tr ' ' '+')
      timeout 300 rocprofv3 --pmc $c --output-format csv -d $P/v${i}_${w}_$tag -o pmc -- python tools/spmm_probe.py $w 3 $v > $P/v${i}_${w}_$tag.log 2>&1
    done
  done
  echo "v$i = $v" >> $P/variants.txt
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof4/v*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "spmm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[2], {k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
cat $P/variants.txt
