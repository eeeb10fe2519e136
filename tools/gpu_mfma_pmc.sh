#!/bin/bash
# MFMA utilisation of the contraction and of the one-pass backward kernel: SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE
# (counter pass on its own, no tracing domains).
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out/r36; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc.log 2>&1
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
test -n "$f" && python - "$f" <<'PY' | tee $O/mfma_util.json
import collections, csv, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    for tag in ("contract_panel_kernel", "bwd_fused_panel_kernel", "spmm_panel_kernel", "pack_panels_kernel"):
        if tag in k:
            agg[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for tag, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    out[tag] = dict(m, launches=len(next(iter(c.values()))))
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc
