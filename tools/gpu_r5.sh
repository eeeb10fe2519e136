#!/bin/bash
# Round-5 GPU sessions (one gpurun call each): bash tools/gpu_r5.sh <stage> ; logs under gpurun_out/r05_<stage>/
# (lab notebook: the stages are kept as they were run; knobs of the early stages -- spmm_pf2, spmm_pfe, spmm_pfs, spmm_pfa -- no longer exist)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp GFHIP_EXPERIMENTS=1
S=$1; O=gpurun_out/r05_$S; mkdir -p $O
case $S in a|b|c|d|e|f|g) export PROBE_SINGLE=1;; esac   # (the first sessions timed single hops; tools/hop_probe.py now times the chain)
pmc() {  # pmc <tag> <counters...> -- <hop_probe args>: one rocprofv3 pass, per-kernel averages
  local tag=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf $O/pm; timeout 200 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $O/pm -o pmc -- python tools/hop_probe.py "$@" > $O/pm_$tag.log 2>&1 || echo "pmc pass failed: $tag"
  python3 - "$O" "$tag" <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1:3]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = "msweep" if "msweep" in kn else ("sell" if "spmm_sell" in kn else None)
        if k: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{tag:10s} {k:7s} {c:42s} {sum(v)/len(v):18.0f}  ({len(v)} launches)")
PY
  rm -rf $O/pm
}
case $S in
a)  # first contact: MFMA layout, parity, timing variants, counters
  tools/mfma4x4_probe 2>&1 | tee $O/mfma_probe.log
  timeout 900 python -m pytest tests/test_gpu_msweep.py -x -q 2>&1 | tail -15 | tee $O/pytest_msweep.log
  V="v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1+spmm_srcmask=0 v:spmm_algo=5+spmm_bar=0+spmm_srcmask=0 v:spmm_algo=5+spmm_bar=1+spmm_srcmask=1048448 v:spmm_algo=5+spmm_bar=0+spmm_srcmask=1048448 v:spmm_algo=3+spmm_srcmask=0"
  timeout 300 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "spmm hop" | tee $O/hop_er.log
  for sl in 0 20 35; do
    timeout 300 python tools/hop_probe.py cfg4 10 spmm_slack=$sl v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 v:spmm_algo=5+spmm_bar=0 2>&1 | grep "spmm hop" | tee $O/hop_er_slack$sl.log
  done
  pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 | tee $O/pmc.log
  pmc l2nb TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:spmm_algo=5+spmm_bar=0 | tee -a $O/pmc.log
  pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -- cfg4 3 v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 | tee -a $O/pmc.log
  pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY -- cfg4 3 v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 | tee -a $O/pmc.log
  pmc wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- cfg4 3 v:spmm_algo=3 v:spmm_algo=5+spmm_bar=1 | tee -a $O/pmc.log
  ;;
b)  # scalar prefetch of first touches, XCD stagger
  K="spmm_algo=5+spmm_srcmask=0"
  V="v:spmm_algo=3"
  for bar in 1 0; do
    V="$V v:$K+spmm_bar=$bar+spmm_pfd=0+spmm_pf2=0+spmm_stag=0"
    for pfd in 1 2 4; do V="$V v:$K+spmm_bar=$bar+spmm_pfd=$pfd+spmm_pf2=0+spmm_stag=0"; done
    V="$V v:$K+spmm_bar=$bar+spmm_pfd=2+spmm_pf2=1+spmm_stag=0 v:$K+spmm_bar=$bar+spmm_pfd=4+spmm_pf2=1+spmm_stag=0"
    V="$V v:$K+spmm_bar=$bar+spmm_pfd=0+spmm_pf2=0+spmm_stag=2 v:$K+spmm_bar=$bar+spmm_pfd=0+spmm_pf2=0+spmm_stag=4 v:$K+spmm_bar=$bar+spmm_pfd=2+spmm_pf2=1+spmm_stag=2 v:$K+spmm_bar=$bar+spmm_pfd=2+spmm_pf2=1+spmm_stag=4"
  done
  timeout 600 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "spmm hop" | tee $O/hop_er.log
  pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -- cfg4 3 v:$K+spmm_bar=1+spmm_pfd=2+spmm_pf2=1+spmm_stag=0 | tee $O/pmc.log
  pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:$K+spmm_bar=1+spmm_pfd=2+spmm_pf2=1+spmm_stag=0 | tee -a $O/pmc.log
  ;;
c)  # prefetch lead sweep; time without stores / with L2-resident sources
  K="spmm_algo=5+spmm_srcmask=0+spmm_stag=0+spmm_store=2"
  V="v:spmm_algo=3"
  for bar in 1 0; do
    for pfd in 4 6 8 12 16 24; do V="$V v:$K+spmm_bar=$bar+spmm_pfd=$pfd+spmm_pf2=1"; done
  done
  V="$V v:$K+spmm_bar=1+spmm_pfd=0+spmm_store=3 v:$K+spmm_bar=1+spmm_pfd=8+spmm_pf2=1+spmm_store=3 v:$K+spmm_bar=1+spmm_pfd=0+spmm_store=3+spmm_srcmask=1048448 v:$K+spmm_bar=0+spmm_pfd=0+spmm_store=3+spmm_srcmask=1048448 v:$K+spmm_bar=1+spmm_pfd=8+spmm_pf2=1+spmm_store=3+spmm_srcmask=1048448"
  timeout 600 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "spmm hop" | tee $O/hop_er.log
  ;;
d)  # counters: pure-hit regime (sources confined to 1 MB, no stores) and the real run with the scalar prefetch
  K="spmm_algo=5+spmm_stag=0+spmm_bar=1"
  for v in "hit:$K+spmm_pfd=0+spmm_store=3+spmm_srcmask=1048448" "real:$K+spmm_pfd=8+spmm_pf2=1+spmm_store=2+spmm_srcmask=0"; do
    tag=${v%%:*}; var=${v#*:}
    pmc ${tag}_sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY -- cfg4 3 v:$var | tee -a $O/pmc.log
    pmc ${tag}_sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -- cfg4 3 v:$var | tee -a $O/pmc.log
    pmc ${tag}_sq3 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 -- cfg4 3 v:$var | tee -a $O/pmc.log
    pmc ${tag}_tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -- cfg4 3 v:$var | tee -a $O/pmc.log
    pmc ${tag}_ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum -- cfg4 3 v:$var | tee -a $O/pmc.log
    pmc ${tag}_l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:$var | tee -a $O/pmc.log
  done
  pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT -- cfg4 3 v:$K+spmm_pfd=8+spmm_pf2=1+spmm_store=2+spmm_srcmask=0 | tee -a $O/pmc.log
  ;;
e)  # row prefetch with immediate offsets + entry-stream prefetch
  K="spmm_algo=5+spmm_srcmask=0+spmm_stag=0+spmm_store=2+spmm_bar=1"
  V="v:spmm_algo=3 v:$K+spmm_pfd=0+spmm_pfe=0"
  for pfd in 6 8 12; do for pfe in 0 2 4; do V="$V v:$K+spmm_pfd=$pfd+spmm_pfe=$pfe"; done; done
  V="$V v:$K+spmm_pfd=8+spmm_pfe=3+spmm_bar=0 v:$K+spmm_bar=1+spmm_pfd=8+spmm_pfe=3+spmm_store=3 v:$K+spmm_pfd=8+spmm_pfe=3+spmm_store=3+spmm_srcmask=1048448 v:$K+spmm_pfd=0+spmm_pfe=0+spmm_store=3+spmm_srcmask=1048448"
  timeout 600 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "spmm hop" | tee $O/hop_er.log
  ;;
f)  # ring depth 10 (two rounds per loop iteration), with and without the scalar row prefetch
  timeout 900 python -m pytest tests/test_gpu_msweep.py -x -q 2>&1 | tail -5 | tee $O/pytest_msweep.log
  K="spmm_algo=5+spmm_srcmask=0+spmm_stag=0+spmm_store=2+spmm_bar=1"
  V="v:spmm_algo=3"
  for d in 5 10; do for pfd in 0 2 3 4 6; do V="$V v:$K+spmm_depth=$d+spmm_pfd=$pfd"; done; done
  V="$V v:$K+spmm_depth=10+spmm_pfd=0+spmm_bar=0 v:$K+spmm_depth=10+spmm_pfd=4+spmm_bar=0 v:$K+spmm_bar=1+spmm_depth=10+spmm_pfd=0+spmm_store=3 v:$K+spmm_depth=10+spmm_pfd=4+spmm_store=3 v:$K+spmm_depth=10+spmm_pfd=0+spmm_store=3+spmm_srcmask=1048448"
  timeout 600 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "spmm hop" | tee $O/hop_er.log
  ;;
h)  # the K-1 hops fused into one launch (entry-major), against one launch per hop
  timeout 600 python -m pytest tests/test_gpu_msweep.py -x -q > $O/pytest_msweep.log 2>&1; tail -3 $O/pytest_msweep.log
  K="spmm_algo=5+spmm_srcmask=0+spmm_stag=0+spmm_store=2+spmm_depth=10"
  V="v:spmm_algo=3"
  for fuse in 0 1; do for pfd in 0 4 6; do for bar in 1 0; do V="$V v:$K+spmm_fuse=$fuse+spmm_pfd=$pfd+spmm_bar=$bar"; done; done; done
  V="$V v:$K+spmm_fuse=1+spmm_pfd=6+spmm_bar=1+spmm_stag=1 v:$K+spmm_fuse=1+spmm_pfd=6+spmm_bar=1+spmm_stag=3 v:$K+spmm_fuse=1+spmm_pfd=6+spmm_bar=1+spmm_stag=0+spmm_depth=5"
  PROBE_CHAIN=1 timeout 600 python tools/hop_probe.py cfg4 5 $V 2>&1 | grep "khop chain" | tee $O/khop_er.log
  ;;
i)  # the flaky refusal test x5; slack; NT gathers; leads
  for r in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_msweep.py -x -q -k "refused or 40000" > $O/pytest_refused_$r.log 2>&1; tail -2 $O/pytest_refused_$r.log | head -1; done
  K="spmm_algo=5+spmm_srcmask=0+spmm_stag=0+spmm_store=2+spmm_depth=10+spmm_fuse=1+spmm_bar=1"
  for sl in 0 5 10 15; do
    PROBE_CHAIN=1 timeout 300 python tools/hop_probe.py cfg4 5 spmm_slack=$sl v:spmm_algo=3 v:$K+spmm_pfd=6 v:$K+spmm_pfd=8 2>&1 | grep "khop chain" | tee -a $O/khop_slack.log
  done
  GFHIP_LIB=$PWD/graph-neural-networks_amd/alegnn_amd/libgfhip_nt.so PROBE_CHAIN=1 timeout 300 python tools/hop_probe.py cfg4 5 v:spmm_algo=3 v:$K+spmm_pfd=6 v:$K+spmm_pfd=0 2>&1 | grep "khop chain" | tee $O/khop_nt.log
  ;;
j)  # entry quads as 128-byte lines
  timeout 600 python -m pytest tests/test_gpu_msweep.py -x -q > $O/pytest_msweep.log 2>&1; tail -2 $O/pytest_msweep.log
  K="spmm_algo=5+spmm_srcmask=0+spmm_stag=0+spmm_store=2+spmm_depth=10+spmm_fuse=1+spmm_bar=1"
  PROBE_CHAIN=1 timeout 300 python tools/hop_probe.py cfg4 5 v:spmm_algo=3 v:$K+spmm_pfd=0 v:$K+spmm_pfd=6 v:$K+spmm_pfd=8 v:$K+spmm_pfd=8+spmm_bar=0 v:$K+spmm_pfd=8+spmm_depth=5 2>&1 | grep "khop chain" | tee -a $O/khop.log
  timeout 300 python tools/hop_probe.py cfg4 10 v:spmm_algo=3 v:$K+spmm_pfd=8 v:$K+spmm_pfd=8+spmm_store=3 v:$K+spmm_pfd=0+spmm_store=3+spmm_srcmask=1048448 2>&1 | grep "spmm hop" | tee $O/hop.log
  pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -- cfg4 3 v:$K+spmm_pfd=8 | tee $O/pmc.log
  ;;
k)  # where an entry's time goes
  K="spmm_srcmask=0 spmm_store=2 spmm_depth=10 spmm_fuse=1 spmm_bar=1 spmm_pfd=8"
  timeout 300 python tools/msweep_trace.py $K spmm_stag=0 2>&1 | tail -12 | tee $O/trace_stag0.log
  timeout 300 python tools/msweep_trace.py $K spmm_stag=3 2>&1 | tail -12 | tee $O/trace_stag3.log
  timeout 300 python tools/msweep_trace.py $K spmm_stag=0 spmm_store=3 2>&1 | tail -12 | tee $O/trace_nostore.log
  ;;
l)  # row bands -> sets (low rows stored last: still in L2 when the next hop of the fused chain starts)
  timeout 600 python -m pytest tests/test_gpu_msweep.py -x -q > $O/pytest_msweep.log 2>&1; tail -2 $O/pytest_msweep.log
  timeout 300 python tools/hop_probe.py cfg4 5 v:spmm_algo=3 v:spmm_algo=0 v:spmm_algo=0+spmm_fuse=0 v:spmm_algo=0+spmm_fuse=1+spmm_bar=0 2>&1 | grep "khop chain" | tee $O/khop.log
  pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:spmm_algo=0 v:spmm_algo=0+spmm_fuse=0 | tee $O/pmc.log
  ;;
z)  # the shipped kernel: phase stamps and counters on the final sources (no knobs: the product defaults)
  timeout 300 python tools/msweep_trace.py 2>&1 | tail -12 | tee $O/trace_default.log
  timeout 300 python tools/msweep_trace.py spmm_store=3 2>&1 | tail -12 | tee $O/trace_nostore.log
  pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- cfg4 3 v:spmm_algo=0 | tee $O/pmc.log
  pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -- cfg4 3 v:spmm_algo=0 | tee -a $O/pmc.log
  pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM -- cfg4 3 v:spmm_algo=0 | tee -a $O/pmc.log
  pmc mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -- cfg4 3 v:spmm_algo=0 | tee -a $O/pmc.log
  ;;
esac
