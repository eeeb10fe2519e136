#!/bin/bash
# HBM traffic of a workload's dominant kernel from the PMC counters: one rocprofv3 --pmc pass per counter group (no tracing
# domains in the same run), reduced to profiles/<tag>_<workload>_pmc.json in the form bench.py's roofline.traffic reads.
#   usage: bash tools/pmc_collect.sh <workload> <kernel-name-substring> <tag>        e.g.  cfg4 spmm_sell_kernel r02
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
WL=${1:-cfg4}; KN=${2:-spmm_sell_kernel}; TAG=${3:-r03}
O=gpurun_out/pmc_$WL; rm -rf $O; mkdir -p $O
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do   # (the TCC block has 4 counter slots: FETCH_SIZE takes 3, WRITE_SIZE 2)
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -o pmc -- python tools/hop_probe.py $WL 3 > $O/pmc$i.log 2>&1
done
python - "$WL" "$KN" "$TAG" "$O" <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, ".")
wl, kn, tag, O = sys.argv[1:5]
tot, nl = {}, 0
for d in sorted(glob.glob(f"{O}/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if kn in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        tot[k] = sum(v) / len(v); nl = len(v)
import bench
w = bench.WORKLOADS[wl]
out = dict(workload=wl, kernel=kn, kernel_src_sha=bench.kernel_source_sha(), launches_averaged=nl, raw=tot,
           command=f"rocprofv3 --pmc <group> -- python tools/hop_probe.py {wl} 3   (tools/pmc_collect.sh: one pass per counter group)",
           correction="FETCH_SIZE is reported in KiB and, on gfx950, at exactly half the bytes of a 16-B/lane streaming read (MI355X_MICROARCH.md, "
                      "HBM section: TCC_EA0_RDREQ x 64 B for 128-byte requests): read bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 is taken as is")
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    out["hbm_read_bytes_per_launch"] = tot["FETCH_SIZE"] * 1024 * 2
    out["hbm_write_bytes_per_launch"] = tot["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
if "TCC_HIT_sum" in tot:
    out["l2_hit_rate"] = round(tot["TCC_HIT_sum"] / max(1.0, tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]), 3)
json.dump(out, open(f"{O}/{tag}_{wl}_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O/pmc*/
