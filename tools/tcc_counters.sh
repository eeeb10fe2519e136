cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r3_tcc; mkdir -p $O
run() { tag=$1; kn=$2; shift 2
  for grp in "TCC_REQ_sum TCC_TAG_STALL_sum" "TCC_BUSY_sum TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum TCC_BUBBLE_sum" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"; do
    rm -rf $O/pm; timeout 60 rocprofv3 --pmc $grp --output-format csv -d $O/pm -o pmc -- "$@" > $O/pm.log 2>&1 || echo "group failed: $grp"
    python3 - "$O" "$tag" "$kn" <<'PY'
import csv, glob, sys, collections
O, tag, kn = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kn in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{tag:26s} {k:40s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
PY
  done
}
run "sweep lag=8 accum=1" sweep_kernel tools/xcd_sweep wg=2 accum=1 lag=8 depth=2 iters=3
run "sweep unsynced" sweep_kernel tools/xcd_sweep wg=2 accum=1 lag=-1 depth=2 iters=3
run "microbench 8MB 16w/CU" "gather_kernel<0, 8>" tools/gather_ceiling one 65536 1024
rm -rf $O/pm
