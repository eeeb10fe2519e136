#!/bin/bash
# HBM traffic of the panel hop (config 2) from the PMC counters, separate passes, no tracing domains
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out/r38; rm -rf $O; mkdir -p $O
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -o pmc -- python tools/panel_probe.py cfg2 3 > $O/pmc$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
tot = {}
for d in sorted(glob.glob("gpurun_out/r38/pmc*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "spmm_panel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        tot[k] = sum(v) / len(v)
print(json.dumps(tot))
json.dump(tot, open("gpurun_out/r38/pmc_panel_raw.json", "w"))
PY
rm -rf $O/pmc*/
