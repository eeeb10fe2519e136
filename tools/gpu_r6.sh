#!/bin/bash
# Round-6 GPU sessions (one gpurun call each): bash tools/gpu_r6.sh <stage> ; logs under gpurun_out/r06_<stage>/ (lab notebook: stages are kept as run)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp GFHIP_EXPERIMENTS=1
S=$1; O=gpurun_out/r06_$S; mkdir -p $O
LIBD=$PWD/graph-neural-networks_amd/alegnn_amd
pmc() {  # pmc <tag> <counters...> -- <hop_probe args>: one rocprofv3 pass, per-kernel averages (rows per launch are checked: a pass that died early says so)
  local tag=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf $O/pm; timeout 300 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $O/pm -o pmc -- python tools/hop_probe.py "$@" > $O/pm_$tag.log 2>&1; local rc=$?
  python3 - "$O" "$tag" "$rc" <<'PY'
import csv, glob, sys, collections
O, tag, rc = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = "msweep" if "msweep_kernel" in kn else ("sell" if "spmm_sell" in kn else None)
        if k: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
n = {len(v) for v in agg.values()}
print(f"{tag:10s} rocprofv3 exit {rc}; launches per counter: {sorted(n)}" + ("" if len(n) == 1 else "  <-- UNEVEN: pass incomplete"))
for (k, c), v in sorted(agg.items()):
    print(f"{tag:10s} {k:7s} {c:42s} {sum(v)/len(v):18.0f}  ({len(v)} launches)")
PY
  rm -rf $O/pm
}
case $S in
a)  # where do the 12 ns per gather instruction go?  timing-only variants of the loop (no MFMAs / no DPP / global_load / half the MFMAs), each in
    # four regimes: as shipped; no prefetch; real sources without stores; sources confined to 1 MB (every gather hits L2) without and with stores
  R="v:spmm_algo=0+spmm_pfd=16+spmm_store=2+spmm_srcmask=0 v:spmm_pfd=0+spmm_store=2+spmm_srcmask=0 v:spmm_pfd=0+spmm_store=3+spmm_srcmask=0 v:spmm_pfd=0+spmm_store=3+spmm_srcmask=1048448 v:spmm_pfd=0+spmm_store=2+spmm_srcmask=1048448"
  for lib in "" nomfma nodpp nomfmadpp halfmfma; do
    if [ -n "$lib" ]; then export GFHIP_LIB=$LIBD/libgfhip_$lib.so; else unset GFHIP_LIB; fi
    echo "== lib=${lib:-shipped}" | tee -a $O/decompose.log
    timeout 300 python tools/hop_probe.py cfg4 5 $R 2>&1 | grep "khop chain" | sed 's/bitwise.*//' | tee -a $O/decompose.log
  done
  for lib in glob globnomfmadpp; do   # global_load has no range check: only with the sources confined
    export GFHIP_LIB=$LIBD/libgfhip_$lib.so
    echo "== lib=$lib" | tee -a $O/decompose.log
    timeout 300 python tools/hop_probe.py cfg4 5 v:spmm_algo=0+spmm_pfd=0+spmm_store=3+spmm_srcmask=1048448 v:spmm_pfd=0+spmm_store=2+spmm_srcmask=1048448 2>&1 | grep "khop chain" | sed 's/bitwise.*//' | tee -a $O/decompose.log
  done
  unset GFHIP_LIB
  timeout 300 python tools/msweep_trace.py 2>&1 | tail -12 | tee $O/trace_default.log
  ;;
b)  # the census / repair protocol: tests, and what the census costs on the default chain; ring depth 5 vs 10 on the shipped loop
  timeout 1200 python -m pytest tests/test_gpu_msweep.py -x -q 2>&1 | tail -15 | tee $O/pytest_msweep.log
  timeout 300 python tools/hop_probe.py cfg4 10 v:spmm_algo=0+spmm_depth=0 v:spmm_depth=5 v:spmm_depth=0 v:spmm_depth=5 v:spmm_depth=0+spmm_fuse=0 v:spmm_fuse=1 2>&1 | grep "khop chain" | tee $O/khop.log
  timeout 300 python tools/msweep_trace.py 2>&1 | tail -12 | tee $O/trace_default.log
  ;;
c)  # LDS ring (a whole round of gathers in flight per wave): parity, then time against the VGPR ring in the regimes of stage a
  timeout 900 python -m pytest tests/test_gpu_msweep.py -x -q -k "lds_ring" 2>&1 | tail -15 | tee $O/pytest_ring.log
  R="v:spmm_algo=0+spmm_ring=0+spmm_pfd=16+spmm_store=2+spmm_srcmask=0 v:spmm_ring=1 v:spmm_ring=0 v:spmm_ring=1 v:spmm_ring=1+spmm_store=3 v:spmm_ring=1+spmm_store=3+spmm_srcmask=1048448 v:spmm_ring=1+spmm_store=2+spmm_srcmask=1048448 v:spmm_ring=0+spmm_srcmask=0+spmm_fuse=0 v:spmm_ring=1+spmm_fuse=0"
  timeout 300 python tools/hop_probe.py cfg4 10 $R 2>&1 | grep "khop chain" | tee $O/khop.log
  timeout 300 python tools/msweep_trace.py spmm_ring=1 2>&1 | tail -12 | tee $O/trace_ring.log
  ;;
d)  # near scalar prefetch at a controlled rate: one row every k-th step (k = 1 shipped, 2, 3, 4), lead 1 .. 16 iterations
  for lib in "" pfk2 pfk3 pfk4; do
    if [ -n "$lib" ]; then export GFHIP_LIB=$LIBD/libgfhip_$lib.so; else unset GFHIP_LIB; fi
    echo "== lib=${lib:-shipped}" | tee -a $O/khop.log
    V="v:spmm_algo=0+spmm_pfd=16"; for l in 1 2 3 4 6 8 12 16 0; do V="$V v:spmm_pfd=$l"; done
    timeout 300 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "khop chain" | sed 's/bitwise.*//' | tee -a $O/khop.log
  done
  ;;
e)  # counters: vector-cache requests in flight per CU with 10 gathers per wave in VGPRs vs 25 per wave in LDS (the window is not the limit)
  for v in "d10:spmm_algo=0+spmm_ring=0+spmm_pfd=0" "ring25:spmm_algo=0+spmm_ring=1+spmm_pfd=0" "d5:spmm_algo=0+spmm_ring=0+spmm_depth=5+spmm_pfd=0"; do
    tag=${v%%:*}; var=${v#*:}
    pmc ${tag}_tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -- cfg4 3 v:$var | tee -a $O/pmc.log
    pmc ${tag}_l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:$var | tee -a $O/pmc.log
    pmc ${tag}_ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum -- cfg4 3 v:$var | tee -a $O/pmc.log
    pmc ${tag}_grbm GRBM_GUI_ACTIVE GRBM_COUNT -- cfg4 3 v:$var | tee -a $O/pmc.log
  done
  ;;
f)  # wide rows, monotonic barrier, zero-in-store: msweep suite, then time (cfg4 chain; W = 64 against SELL-8)
  timeout 1500 python -m pytest tests/test_gpu_msweep.py -x -q 2>&1 | tail -15 | tee $O/pytest_msweep.log
  timeout 300 python tools/hop_probe.py cfg4 10 v:spmm_algo=0 v:spmm_algo=3 v:spmm_algo=0 v:spmm_algo=0+spmm_fuse=0 v:spmm_fuse=1 2>&1 | grep "khop chain" | tee $O/khop.log
  timeout 300 python tools/msweep_trace.py 2>&1 | tail -12 | tee $O/trace_default.log
  PROBE_W=64 PROBE_B=64 timeout 300 python tools/hop_probe.py cfg4 5 v:spmm_algo=3 v:spmm_algo=0 v:spmm_algo=3 v:spmm_algo=0 2>&1 | grep "khop chain" | tee $O/khop_w64.log
  ;;
g)  # does a NEAR scalar prefetch allocate in L2?  fabric read requests with one row per second step, lead 3 iterations (variant pfk2) vs no prefetch
  export GFHIP_LIB=$LIBD/libgfhip_pfk2.so
  pmc pf0_l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:spmm_algo=0+spmm_pfd=0 | tee -a $O/pmc.log
  pmc pf3_l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:spmm_algo=0+spmm_pfd=3 | tee -a $O/pmc.log
  pmc pf3_tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -- cfg4 3 v:spmm_algo=0+spmm_pfd=3 | tee -a $O/pmc.log
  unset GFHIP_LIB
  ;;
h)  # the cooperative chain beside RCCL's kernels / a CU-holding kernel on another stream
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/coop_beside_rccl.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -8 | tee $O/coop_beside_rccl.log
  ;;
i)  # near scalar prefetch at p rows per q steps (12.7 M rows per hop need 0.74 rows per step and wave; a wave may have 15 scalar loads outstanding)
  for lib in "" pf12 pf23 pf34 pf45 pf56; do
    if [ -n "$lib" ]; then export GFHIP_LIB=$LIBD/libgfhip_$lib.so; else unset GFHIP_LIB; fi
    echo "== lib=${lib:-shipped}" | tee -a $O/khop.log
    V="v:spmm_algo=0+spmm_pfd=16"; for l in 1 2 3 4 5 16 0; do V="$V v:spmm_pfd=$l"; done
    timeout 300 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "khop chain" | sed 's/bitwise.*//' | tee -a $O/khop.log
  done
  unset GFHIP_LIB
  ;;
j)  # A/B, interleaved, separate processes: round 5's prefetch (2 s_loads per step, lead 16: variant pfold + spmm_pfd=16) against 3 rows per 4 steps, lead 3 (default)
  for rep in 1 2 3 4; do
    GFHIP_LIB=$LIBD/libgfhip_pfold.so timeout 300 python tools/hop_probe.py cfg4 10 v:spmm_algo=0+spmm_pfd=16 v:spmm_pfd=16 2>&1 | grep "khop chain" | sed 's/^/old  /;s/bitwise.*//' | tee -a $O/ab.log
    timeout 300 python tools/hop_probe.py cfg4 10 v:spmm_algo=0+spmm_pfd=3 v:spmm_pfd=3 2>&1 | grep "khop chain" | sed 's/^/new  /;s/bitwise.*//' | tee -a $O/ab.log
  done
  timeout 1200 python -m pytest tests/test_gpu_msweep.py -x -q 2>&1 | tail -3 | tee $O/pytest_msweep.log
  ;;
l)  # image slack (rounds beyond the mean group length) x prefetch lead, now that the prefetch is near: a looser image keeps a row's uses closer together
  for sl in 5 10 15 20 30; do
    echo "== slack $sl" | tee -a $O/khop.log
    timeout 300 python tools/hop_probe.py cfg4 10 spmm_slack=$sl v:spmm_algo=0+spmm_pfd=3 v:spmm_pfd=2 v:spmm_pfd=3 v:spmm_pfd=4 v:spmm_pfd=0 2>&1 | grep "khop chain" | sed 's/bitwise.*//' | tee -a $O/khop.log
  done
  ;;
m)  # two processes on one GPU, both launching fused chains
  timeout 900 python -m pytest tests/test_gpu_parallel.py -x -q -k "two_processes" 2>&1 | tail -8 | tee $O/pytest_two_processes.log
  ;;
n)  # hub rows: parity, then a power-law graph at config 4's size against SELL-8
  timeout 900 python -m pytest tests/test_gpu_msweep.py -x -q -k "hub_rows" 2>&1 | tail -12 | tee $O/pytest_hub.log
  PROBE_GRAPH=powerlaw timeout 300 python tools/hop_probe.py cfg4 5 v:spmm_algo=3 v:spmm_algo=0 v:spmm_algo=3 v:spmm_algo=0 v:spmm_algo=0+spmm_fuse=0 2>&1 | grep "khop chain\|image" | tee $O/khop_powerlaw.log
  PROBE_OP=1 PROBE_GRAPH=powerlaw timeout 300 python tools/hop_probe.py cfg4 5 v:spmm_algo=3 v:spmm_algo=0 v:spmm_algo=3 v:spmm_algo=0 2>&1 | grep "khop chain\|image" | tee -a $O/khop_powerlaw.log
  timeout 300 python tools/hop_probe.py cfg4 10 v:spmm_algo=0 v:spmm_algo=0 2>&1 | grep "khop chain" | tee $O/khop_er.log
  bash tools/gpu_r6.sh m
  ;;
q)  # SELL-8 against the sweep by size / degree / batch / width on the final kernel (DESIGN 3.1g "by size")
  timeout 1400 python tools/msweep_sizes.py 50000,10,0,256 60000,10,0,128 80000,10,0,128 100000,10,0,128 100000,10,1,128 100000,10,0,100 100000,10,0,9 100000,20,0,64 100000,4,0,128 \
      120000,10,0,128 150000,10,0,128 200000,10,0,64 100000,10,0,64,64 100000,10,0,32,128 2>&1 | grep "^N=" | tee $O/sizes.log
  ;;
r)  # prefetch lead by graph degree (rounds): auto = (rounds + 7) / 14 against fixed leads
  for c in 100000,4,0,128 100000,10,0,128 100000,20,0,64 60000,10,0,128; do
    for l in 0 1 2 3 6 -1; do echo -n "pfd=$l  "; timeout 300 python tools/msweep_sizes.py $c spmm_pfd=$l 2>&1 | grep "^N="; done
  done | tee $O/lead_by_degree.log
  ;;
u)  # (NO-GO, code not kept: git stash of this session) prefetch pattern per hop as a RUNTIME mask (s_bitcmp + branch per step, one row slot per step):
    # first hop of an entry (rows from HBM) x later hops (rows from the Infinity Cache), against the previous commit's library: every pattern slower
  for rep in 1 2; do
    GFHIP_LIB=$LIBD/libgfhip_prev.so timeout 300 python tools/hop_probe.py cfg4 10 v:spmm_algo=0 v:spmm_algo=0 2>&1 | grep "khop chain" | sed 's/^/prev /;s/bitwise.*//' | tee -a $O/patterns.log
    V="v:spmm_algo=0+spmm_pfpat0=119+spmm_pfpat1=119"
    for p1 in 119 255 127 85; do for p0 in 255 119 85 17 0; do V="$V v:spmm_pfpat0=$p0+spmm_pfpat1=$p1"; done; done
    timeout 400 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "khop chain" | sed 's/^/new  /;s/bitwise.*//' | tee -a $O/patterns.log
  done
  ;;
w)  # the layout pass inside the fused chain launch: parity, then the step with and without it
  timeout 900 python -m pytest tests/test_gpu_msweep.py -x -q -k "layout_pass" 2>&1 | tail -12 | tee $O/pytest_layout.log
  for v in 1 0 1 0; do echo -n "spmm_xlayout=$v  "; timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --tune spmm_xlayout=$v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done | tee $O/bench_xlayout.log
  ;;
y)  # (NO-GO, code not kept) scalar prefetch of the wave's own entry stream (4 rounds ahead) on top of the row prefetch: A/B, interleaved processes, both positions
  for rep in 1 2 3; do
    for lib in nopfe "" pfe23; do
      if [ -n "$lib" ]; then export GFHIP_LIB=$LIBD/libgfhip_$lib.so; else unset GFHIP_LIB; fi
      timeout 300 python tools/hop_probe.py cfg4 10 v:spmm_algo=0 v:spmm_algo=0 2>&1 | grep "khop chain" | sed "s/^/${lib:-pfe34} /;s/bitwise.*//" | tee -a $O/ab.log
    done
  done
  unset GFHIP_LIB
  timeout 900 python -m pytest tests/test_gpu_msweep.py -x -q 2>&1 | tail -3 | tee $O/pytest_msweep.log
  ;;
z)  # hub rows: the cost model's length limit against fixed ones (power-law graph at config 4's size, B = 128)
  for lim in 0 12 16 20 27 35 45 60 90 150; do
    echo -n "spmm_hublim=$lim  " | tee -a $O/hublim.log
    PROBE_GRAPH=powerlaw timeout 300 python tools/hop_probe.py cfg4 5 spmm_hublim=$lim v:spmm_algo=0 v:spmm_algo=0 2>&1 | grep "khop chain\|image" | sed 's/khop chain cfg4 K=5 //;s/bitwise.*//' | tr '\n' ' ' | tee -a $O/hublim.log; echo | tee -a $O/hublim.log
  done
  ;;
share)  # (NO-GO, code not kept: profiles/r06_l_share/shared_last_round.patch) the last round of a work list that is no multiple of 8, shared by groups of XCDs:
    # chain time per hop with / without, and short work lists against SELL-8 (kept: the sweep from 5 pairs on, knob spmm_minwork)
  for B in 9 10 12 17 20 100 129; do
    PROBE_B=$B timeout 300 python tools/hop_probe.py cfg4 5 v:spmm_share=0 v:spmm_share=1 v:spmm_share=0 v:spmm_share=1 2>&1 | grep "khop chain" | sed "s/^/B=$B  /;s/khop chain cfg4 K=5 //" | tee -a $O/share.log
  done
  for B in 1 2 3 4 5 6 7; do
    PROBE_B=$B timeout 300 python tools/hop_probe.py cfg4 5 spmm_minwork=1 v:spmm_algo=3 v:spmm_algo=5+spmm_share=1 v:spmm_algo=5+spmm_share=0 v:spmm_algo=3 v:spmm_algo=5+spmm_share=1 2>&1 | grep "khop chain" | sed "s/^/B=$B  /;s/khop chain cfg4 K=5 //" | tee -a $O/share_short.log
  done
  for W in 64 128; do for B in 1 3 5; do
    PROBE_W=$W PROBE_B=$B timeout 300 python tools/hop_probe.py cfg4 5 spmm_minwork=1 v:spmm_algo=3 v:spmm_algo=5+spmm_share=1 v:spmm_algo=5+spmm_share=0 2>&1 | grep "khop chain" | sed "s/^/W=$W B=$B  /;s/khop chain cfg4 K=5 //" | tee -a $O/share_short.log
  done; done
  ;;
esac
