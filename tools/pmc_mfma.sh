#!/bin/bash
# MFMA pipe utilisation of the contraction kernels of a workload: rocprofv3 --pmc passes of their own (no tracing domains) over a short
# bench.py run, reduced to gpurun_out/pmc_mfma_<workload>/<tag>_<workload>_<kernel>_mfma_pmc.json in the form bench.py's `mfma` object reads.
#   usage: bash tools/pmc_mfma.sh <workload> <tag>        e.g.  cfg4 r03
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
WL=${1:-cfg4}; TAG=${2:-r03}
O=gpurun_out/pmc_mfma_$WL; rm -rf $O; mkdir -p $O
i=0
for c in "MfmaUtil" "SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -o pmc -- python bench.py --workload $WL --no-cpu-baseline --steps 3 --warmup 1 > $O/pmc$i.log 2>&1
done
python - "$WL" "$TAG" "$O" <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, ".")
wl, tag, O = sys.argv[1:4]
import bench
per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(f"{O}/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(d)):
        for k in ("contract_mfma_kernel", "contract_panel_kernel", "bwd_fused_panel_kernel"):
            if k in r["Kernel_Name"]:
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn in ("contract_mfma_kernel", "contract_panel_kernel", "bwd_fused_panel_kernel"):
    if kn not in per:
        continue
    m = {c: sum(v) / len(v) for c, v in per[kn].items()}
    out = dict(workload=wl, kernel=kn, kernel_src_sha=bench.kernel_source_sha(), launches_averaged=len(next(iter(per[kn].values()))), raw=m,
               mfma_util_pct=m.get("MfmaUtil"), mfma_flops_per_launch=m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512,
               command=f"rocprofv3 --pmc <group> -- python bench.py --workload {wl} --no-cpu-baseline --steps 3 --warmup 1   (tools/pmc_mfma.sh)",
               definition="MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM) * 100 (rocprofv3 derived metric)")
    json.dump(out, open(f"{O}/{tag}_{wl}_{kn}_mfma_pmc.json", "w"), indent=1)
    print(json.dumps(out))
PY
rm -rf $O/pmc*/
