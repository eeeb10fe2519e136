#!/bin/bash
# HBM traffic of a workload's dominant kernel from the PMC counters: one rocprofv3 --pmc pass per counter group (no tracing
# domains in the same run), reduced to profiles/<tag>_<workload>_pmc.json in the form bench.py's roofline.traffic reads.
#   usage: bash tools/pmc_collect.sh <workload> <kernel-name-substring> <tag>        e.g.  cfg4 spmm_sell_kernel r02
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
WL=${1:-cfg4}; KN=${2:-spmm_sell_kernel}; TAG=${3:-r03}
O=gpurun_out/pmc_$WL; rm -rf $O; mkdir -p $O
i=0
: > $O/passes.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do   # (the TCC block has 4 counter slots: FETCH_SIZE takes 3, WRITE_SIZE 2)
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -o pmc -- python tools/hop_probe.py $WL 3 > $O/pmc$i.log 2>&1
  echo "pmc$i $? $c" >> $O/passes.txt           # (rocprofv3 on this image often dies at tool teardown, after the CSV is complete: the exit code is recorded, the rows are counted below)
done
if [ "$KN" = spmm_msweep_kernel ]; then   # calibration pass: the same launch without the scalar prefetch (whose 64-byte requests the x2 rule counts as 128-byte ones)
  GFHIP_EXPERIMENTS=1 timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_nopf -o pmc -- python tools/hop_probe.py $WL 3 spmm_pfd=-1 > $O/pmc_nopf.log 2>&1
  echo "pmc_nopf $? TCC_EA0_RDREQ_sum(spmm_pfd=-1)" >> $O/passes.txt
fi
python - "$WL" "$KN" "$TAG" "$O" <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, ".")
wl, kn, tag, O = sys.argv[1:5]
tot, nl, rows, nopf = {}, 0, {}, None
for d in sorted(glob.glob(f"{O}/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if kn in r["Kernel_Name"] and "repair" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "pmc_nopf" in d:
            nopf = sum(v) / len(v)
            rows["nopf:" + k] = len(v)
        else:
            tot[k] = sum(v) / len(v); nl = len(v); rows[k] = len(v)
passes = [dict(zip(("pass", "rocprofv3_exit", "counters"), l.split(None, 2))) for l in open(f"{O}/passes.txt").read().splitlines() if l.strip()]
complete = bool(rows) and len(set(rows.values())) == 1 and min(rows.values()) > 0   # every counter has one row per dispatch of the kernel, the same number in every pass
import bench
w = bench.WORKLOADS[wl]
out = dict(workload=wl, kernel=kn, kernel_src_sha=bench.kernel_source_sha(), launches_averaged=nl, raw=tot, passes=passes, rows_per_counter=rows, passes_complete=complete,
           command=f"rocprofv3 --pmc <group> -- python tools/hop_probe.py {wl} 3   (tools/pmc_collect.sh: one pass per counter group)",
           correction="FETCH_SIZE is reported in KiB and, on gfx950, at exactly half the bytes of a 16-B/lane streaming read (MI355X_MICROARCH.md, "
                      "HBM section: TCC_EA0_RDREQ x 64 B for 128-byte requests): read bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 is taken as is")
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    out["hbm_read_bytes_per_launch"] = tot["FETCH_SIZE"] * 1024 * 2
    out["hbm_write_bytes_per_launch"] = tot["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
    if nopf is not None and "TCC_EA0_RDREQ_sum" in tot:
        # Calibration on this kernel's own access pattern (the guide: the x2 holds for 128-byte requests, other widths are to be calibrated): the scalar
        # prefetch fetches a row as TWO 64-byte requests where a gather's miss is ONE 128-byte request -- same bytes, one request more, and the x2
        # rule books that request as another 128 bytes.  The same launch without the prefetch issues RDREQ_nopf requests for the same rows.
        extra = max(0.0, tot["TCC_EA0_RDREQ_sum"] - nopf)
        out["rdreq_without_prefetch_per_launch"] = nopf
        out["hbm_bytes_per_launch_by_rule"] = out["hbm_bytes_per_launch"]
        out["hbm_read_bytes_per_launch"] -= extra * 128.0
        out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
        out["calibration"] = (f"{extra:.0f} of the launch's {tot['TCC_EA0_RDREQ_sum']:.0f} fabric read requests are the second 64-byte half of a prefetched row "
                              "(TCC_EA0_RDREQ with minus without the scalar prefetch, separate pass): booked at 128 bytes by the x2 rule, they add no bytes; "
                              "hbm_bytes_per_launch is the rule's figure minus 128 bytes for each, hbm_bytes_per_launch_by_rule the uncorrected one")
if not complete:
    out["warning"] = "counter rows per pass differ or are missing: a pass died before its CSV was complete; do not use these numbers"
if "TCC_HIT_sum" in tot:
    out["l2_hit_rate"] = round(tot["TCC_HIT_sum"] / max(1.0, tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]), 3)
json.dump(out, open(f"{O}/{tag}_{wl}_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O/pmc*/
