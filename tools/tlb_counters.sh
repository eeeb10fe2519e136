cd /root/repo; export TMPDIR=/tmp GFHIP_EXPERIMENTS=1
O=gpurun_out/r3_tlb; mkdir -p $O
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum SQ_BUSY_CYCLES"; do
  rm -rf $O/pm; timeout 150 rocprofv3 --pmc $grp --output-format csv -d $O/pm -o pmc -- python tools/hop_probe.py cfg4 4 > $O/pm.log 2>&1 || echo "group failed: $grp"
  python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmm_sell_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"hop ER cfg4  {k:40s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
PY
done 2>&1 | tee $O/hop_tlb_latency_counters.log
rm -rf $O/pm
