#!/bin/bash
# Round-4 GPU stages.  Usage (on the GPU box, from the repo root): bash tools/gpu_r4.sh <stage> [outdir]
cd "$(dirname "$0")/.."; export TMPDIR=/tmp GFHIP_EXPERIMENTS=1
S=${1:-a}; O=gpurun_out/${2:-r4_$S}; mkdir -p $O
VARS="v:spmm_algo=3 v:spmm_algo=2+spmm_sd=16+spmm_wps=4 v:spmm_algo=2+spmm_sd=8+spmm_wps=4 v:spmm_algo=2+spmm_wps=6 v:spmm_algo=2+spmm_wps=8"
case $S in
a)  # stream kernel: parity, then ER / band timings of every variant on one plan; masked-gather microbenchmark
  timeout 120 tools/gather_mask > $O/gather_mask.log 2>&1; cat $O/gather_mask.log
  timeout 900 python -m pytest tests/test_gpu_stream.py -x -q > $O/pytest_stream.log 2>&1; tail -5 $O/pytest_stream.log
  timeout 300 python tools/hop_probe.py cfg4 10 $VARS > $O/hop_er.log 2>&1; grep "spmm hop" $O/hop_er.log
  PROBE_GRAPH=band timeout 300 python tools/hop_probe.py cfg4 10 $VARS > $O/hop_band.log 2>&1; grep "spmm hop" $O/hop_band.log
  ;;
b)  # ticket hand-out variants (atomic-bound in stage a: one counter per XCD served ~32 tickets/us)
  timeout 900 python -m pytest tests/test_gpu_stream.py -x -q > $O/pytest_stream.log 2>&1; tail -5 $O/pytest_stream.log
  V="v:spmm_algo=3"
  for tk in 0 1 2; do for nc in 16 4; do for sh in spmm_sd=16+spmm_wps=4 spmm_wps=6 spmm_wps=8; do
    [ $tk = 0 ] && [ $nc = 4 ] && continue
    V="$V v:spmm_algo=2+spmm_tk=$tk+spmm_nc=$nc+$sh"
  done; done; done
  timeout 400 python tools/hop_probe.py cfg4 10 $V > $O/hop_er.log 2>&1; grep "spmm hop" $O/hop_er.log
  PROBE_GRAPH=band timeout 400 python tools/hop_probe.py cfg4 10 $V > $O/hop_band.log 2>&1; grep "spmm hop" $O/hop_band.log
  ;;
c)  # lean stream kernel (buffer loads, LAST mask, scalar-atomic tickets)
  timeout 300 python -m pytest tests/test_gpu_stream.py -x -q > $O/pytest_stream.log 2>&1; tail -5 $O/pytest_stream.log
  V="v:spmm_algo=3"
  for sh in spmm_wps=4 spmm_wps=5 spmm_wps=8; do for tk in "spmm_tk=0" "spmm_tk=2+spmm_nc=16" "spmm_tk=2+spmm_nc=4" "spmm_tk=2+spmm_nc=1"; do
    V="$V v:spmm_algo=2+$tk+$sh"
  done; done
  timeout 120 python tools/hop_probe.py cfg4 10 $V > $O/hop_er.log 2>&1; grep "spmm hop" $O/hop_er.log
  PROBE_GRAPH=band timeout 120 python tools/hop_probe.py cfg4 10 $V > $O/hop_band.log 2>&1; grep "spmm hop" $O/hop_band.log
  ;;
d)  # counters: why is the stream kernel no faster than SELL-8?
  PROBE_GRAPH=band bash tools/pmc_hop.sh $O > /dev/null 2>&1; cat $O/pmc_hop_band.log
  bash tools/pmc_hop.sh $O > /dev/null 2>&1; cat $O/pmc_hop_er.log
  ;;
e)  # band graph WITHOUT locality groups (window-sorted natural order: true L2 residency) -- both kernels, + hit-rate counters
  V="v:spmm_algo=3 v:spmm_algo=2+spmm_wps=4+spmm_tk=2+spmm_nc=1 v:spmm_algo=2+spmm_wps=4+spmm_tk=2+spmm_nc=16 v:spmm_algo=2+spmm_wps=5+spmm_tk=2+spmm_nc=1 v:spmm_algo=2+spmm_wps=4+spmm_tk=0"
  PROBE_GRAPH=band timeout 120 python tools/hop_probe.py cfg4 10 spmm_group=0 $V > $O/hop_band_nogroups.log 2>&1; grep "spmm hop" $O/hop_band_nogroups.log
  PROBE_GRAPH=band timeout 120 python tools/hop_probe.py cfg4 10 spmm_group=0 spmm_store=3 v:spmm_algo=3 > $O/hop_band_nogroups_nostore.log 2>&1; grep "spmm hop" $O/hop_band_nogroups_nostore.log
  rm -rf $O/pm; PROBE_GRAPH=band timeout 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pm -o pmc -- python tools/hop_probe.py cfg4 3 spmm_group=0 v:spmm_algo=3 v:spmm_algo=2+spmm_wps=4+spmm_tk=2+spmm_nc=1 > $O/pm.log 2>&1
  python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        tag = "stream" if "spmm_stream" in kn else ("sell" if "spmm_sell" in kn else None)
        if tag:
            agg[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (tag, k), v in sorted(agg.items()):
    print(f"band-nogroups {tag:7s} {k:30s} max {max(v):16.0f} n={len(v)}")
PY
  rm -rf $O/pm
  ;;
f)  # launch order of the K-1 hops (Infinity-Cache residency of the tap just written) + gather rates out of L2 / IC / HBM
  timeout 200 python tools/chunk_order_probe.py cfg4 8,16,32,64 > $O/chunk_order.log 2>&1; cat $O/chunk_order.log | grep cfg4
  timeout 200 python tools/chunk_order_probe.py cfg4 8,16 spmm_store=0 > $O/chunk_order_plain_stores.log 2>&1; grep cfg4 $O/chunk_order_plain_stores.log
  for rows in 32768 131072 524288 2097152; do timeout 60 tools/gather_ceiling one $rows 4096; done > $O/gather_l2_ic_hbm.log 2>&1; cat $O/gather_l2_ic_hbm.log
  ;;
g)  # prefetch runs: band graph (with / without locality groups), stream kernel with and without them, against SELL-8; ER unchanged?
  timeout 200 python -m pytest tests/test_gpu_stream.py -x -q > $O/pytest_stream.log 2>&1; tail -3 $O/pytest_stream.log
  V="v:spmm_algo=3 v:spmm_algo=2+spmm_wps=4+spmm_tk=2+spmm_nc=1 v:spmm_algo=2+spmm_wps=4+spmm_tk=2+spmm_nc=4 v:spmm_algo=2+spmm_wps=5+spmm_tk=2+spmm_nc=1 v:spmm_algo=2+spmm_wps=4+spmm_tk=0"
  for spf in 1 0; do for grp in 1 0; do
    PROBE_GRAPH=band timeout 100 python tools/hop_probe.py cfg4 10 spmm_group=$grp spmm_spf=$spf $V > $O/hop_band_g${grp}_pf$spf.log 2>&1; grep "spmm hop" $O/hop_band_g${grp}_pf$spf.log
  done; done
  timeout 100 python tools/hop_probe.py cfg4 10 $V > $O/hop_er.log 2>&1; grep "spmm hop" $O/hop_er.log
  ;;
h)  # prefetch lead x front width (waves per SIMD)
  V="v:spmm_algo=3"; for w in 4 2 1; do for nc in 1 4; do V="$V v:spmm_algo=2+spmm_wps=$w+spmm_tk=2+spmm_nc=$nc"; done; done
  for spf in 0 2 4; do for grp in 0 1; do
    PROBE_GRAPH=band timeout 100 python tools/hop_probe.py cfg4 10 spmm_group=$grp spmm_spf=$spf $V > $O/hop_band_g${grp}_pf$spf.log 2>&1; grep "spmm hop" $O/hop_band_g${grp}_pf$spf.log
  done; done
  timeout 100 python tools/hop_probe.py cfg4 10 $V > $O/hop_er.log 2>&1; grep "spmm hop" $O/hop_er.log
  ;;
db)  # the _DB family's bench line: graph replay vs eager launches, kernel stats of the eager step
  timeout 300 python bench.py --workload db > $O/bench_db.json 2> $O/bench_db.err; tail -3 $O/bench_db.err; cat $O/bench_db.json
  timeout 300 python bench.py --workload db --no-graph --no-cpu-baseline > $O/bench_db_eager.json 2> $O/bench_db_eager.err; tail -2 $O/bench_db_eager.err; cat $O/bench_db_eager.json | cut -c1-300
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o db -- python $OLDPWD/bench.py --workload db --no-graph --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1)
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_db_kernel_stats.csv && rm -rf $O/prof; python tools/show_stats.py $O | head -24
  ;;
suite)   # the whole GPU suite as the driver runs it + smoke
  unset GFHIP_EXPERIMENTS
  timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
  ;;
final)   # the round's numbers on the final sources: bench lines, kernel stats, PMC traffic of the dominant kernels
  unset GFHIP_EXPERIMENTS
  timeout 400 python bench.py > $O/bench_cfg4.json 2> $O/bench_cfg4.err; tail -2 $O/bench_cfg4.err; cut -c1-400 $O/bench_cfg4.json
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o cfg4 -- python $OLDPWD/bench.py --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2>&1)
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_cfg4_kernel_stats.csv; rm -rf $O/prof
  GFHIP_EXPERIMENTS=1 bash tools/pmc_collect.sh cfg4 spmm_sell_kernel r04 > $O/pmc_cfg4.log 2>&1; tail -1 $O/pmc_cfg4.log | cut -c1-500; cp gpurun_out/pmc_cfg4/r04_cfg4_pmc.json $O/ 2>/dev/null
  for wl in cfg2 cfg3 cfg5 cfg1 db; do
    timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; python -c "import json,sys; d=json.load(open('$O/bench_$wl.json')); print('$wl', round(d['ms_per_step'],4), 'ms/step', d['roofline']['kernel'][:40], d['roofline']['frac'])"
  done
  python tools/show_stats.py $O | head -12
  ;;
final2)  # PMC of the other workloads' dominant kernels + MFMA utilisation (source-hash tied)
  GFHIP_EXPERIMENTS=1 bash tools/pmc_collect.sh cfg2 spmm_chain_kernel r04 > $O/pmc_cfg2.log 2>&1; tail -1 $O/pmc_cfg2.log | cut -c1-300; cp gpurun_out/pmc_cfg2/r04_cfg2_pmc.json $O/ 2>/dev/null
  GFHIP_EXPERIMENTS=1 bash tools/pmc_mfma.sh cfg4 r04 > $O/mfma_cfg4.log 2>&1; tail -1 $O/mfma_cfg4.log | cut -c1-300
  cp gpurun_out/pmc_mfma_*/r04_*_mfma_pmc.json $O/ 2>/dev/null; ls $O
  ;;
pw)      # config-3-class weighted panel hop: decomposition + LDS counters
  timeout 200 python tools/panel_w_probe.py 1682 64 256 v:panel_np=1 v:panel_np=4 v:panel_rotate=0 v:panel_grid=512 v:panel_grid=256 > $O/probe.log 2>&1; cat $O/probe.log
  timeout 100 python tools/panel_w_probe.py 1682 32 256 > $O/probe_w32.log 2>&1; grep -A1 "^weighted\|^equal" $O/probe_w32.log
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    rm -rf $O/pm; (export TMPDIR=/tmp PROBE_ONLY=1; timeout 100 rocprofv3 --pmc $grp --output-format csv -d $O/pm -o pmc -- python tools/panel_w_probe.py 1682 64 256 v:panel_db=1 > $O/pm.log 2>&1 || echo "group failed: $grp")
    python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmm_panel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:42s} {sum(v)/len(v):18.0f}  ({len(v)} launches)")
PY
  done 2>&1 | tee $O/panel_counters.log
  rm -rf $O/pm
  ;;
ho)      # node-major layer hand-over: bitwise tests + the step it saves at the config-4 class
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "handover" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
  timeout 300 python tools/handover_bench.py 100000 128 2 > $O/bench2.log 2>&1; tail -1 $O/bench2.log
  timeout 300 python tools/handover_bench.py 100000 64 3 > $O/bench3.log 2>&1; tail -1 $O/bench3.log
  ;;
pdb)     # double-buffered panel kernel against the per-hop kernel
  timeout 200 python tools/panel_w_probe.py 1682 64 256 v:panel_db=1 v:panel_db=1+panel_loaders=1 v:panel_db=1+panel_loaders=3 v:panel_db=1+panel_thr=768 > $O/probe.log 2>&1; grep -v "amdgpu.ids" $O/probe.log
  timeout 100 python tools/panel_w_probe.py 1682 32 256 v:panel_db=1 > $O/probe_w32.log 2>&1; grep -A2 "^weighted\|^equal" $O/probe_w32.log
  timeout 100 python tools/panel_w_probe.py 1000 32 256 v:panel_db=1 v:panel_db=1+panel_thr=1024 > $O/probe_n1000.log 2>&1; grep -A3 "^weighted\|^equal" $O/probe_n1000.log
  timeout 100 python tools/panel_w_probe.py 2500 32 256 v:panel_db=1 > $O/probe_n2500.log 2>&1; grep -A2 "^weighted\|^equal" $O/probe_n2500.log
  ;;
pvar)    # A/B builds of gf_panel.hip (tools/panel_variant.sh): entry-chunk size and LDS read batching
  for v in "" kgc4 batch kgc4b; do
    echo "== variant ${v:-shipped}"
    (export GFHIP_EXPERIMENTS=1; [ -n "$v" ] && export GFHIP_LIB=$PWD/graph-neural-networks_amd/alegnn_amd/libgfhip_$v.so; timeout 100 python tools/panel_w_probe.py 1682 64 256 v:panel_db=1 2>&1 | grep -A2 "^weighted\|^equal")
  done > $O/variants.log 2>&1; cat $O/variants.log
  ;;
pdbt)    # tests of the double-buffered panel kernel + the panel-pipeline tests it now takes part in + cfg3 bench
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "panel or khop or handover or pipelines or selection or golden" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
  timeout 300 python bench.py --workload cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; python -c "import json; d=json.load(open('$O/bench_cfg3.json')); print(d['ms_per_step'], d['roofline'])"
  GFHIP_EXPERIMENTS=1 timeout 300 python - <<'PY' > $O/bench_cfg3_off.log 2>&1
import sys; sys.path[:0]=['.','graph-neural-networks_amd']
from alegnn_amd import _lib
_lib.check(_lib.lib().gf_tune(b"panel_db", 0))
sys.argv=['bench.py','--workload','cfg3','--no-cpu-baseline']
exec(open('bench.py').read())
PY
  tail -1 $O/bench_cfg3_off.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('panel_db=0:', d['ms_per_step'], d['roofline']['launch_ms'])"
  ;;
pkh)     # the same probe on a tap stack (4 hops back to back, as bench.py times them), variants alternated; then cfg3 with and without
  export PROBE_KHOP=1
  timeout 200 python tools/panel_w_probe.py 1682 64 256 v:panel_db=0 v:panel_db=1 v:panel_db=0 v:panel_db=1 v:panel_db=1+panel_loaders=6 v:panel_db=1+panel_loaders=2 > $O/probe.log 2>&1; grep -v "amdgpu.ids" $O/probe.log | grep -A7 "^weighted\|^equal"
  timeout 100 python tools/panel_w_probe.py 1682 32 256 v:panel_db=0 v:panel_db=1 v:panel_db=0 v:panel_db=1 > $O/probe_w32.log 2>&1; grep -A5 "^weighted\|^equal" $O/probe_w32.log
  unset PROBE_KHOP
  for rep in 1 2; do for db in 0 1; do
  GFHIP_EXPERIMENTS=1 DB=$db timeout 300 python - <<'PY' 2>/dev/null | tail -1 | python -c "import json,sys,os; d=json.loads(sys.stdin.readline()); print('cfg3 panel_db=%s:' % os.environ.get('DB'), round(d['ms_per_step'],4), 'ms/step, hop', d['roofline']['launch_ms'])"
import sys, os; sys.path[:0]=['.','graph-neural-networks_amd']
from alegnn_amd import _lib
_lib.check(_lib.lib().gf_tune(b"panel_db", int(os.environ["DB"])))
sys.argv=['bench.py','--workload','cfg3','--no-cpu-baseline']
exec(open('bench.py').read())
PY
  done; done
  ;;
f64)     # one-pass backward with F = 64: oracle parity, goldens, cfg3 with and without
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "oracle or golden or selection or handover or trainer" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
  for f in 1 0; do
    GFHIP_EXPERIMENTS=1 timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --tune bwd_fuse64=$f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cfg3 bwd_fuse64=$f:', round(d['ms_per_step'],4), 'ms/step')"
  done
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o cfg3 -- python $OLDPWD/bench.py --workload cfg3 --no-cpu-baseline --steps 50 --warmup 3 > /dev/null 2>&1)
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_cfg3_kernel_stats.csv; rm -rf $O/prof
  python tools/show_stats.py $O | head -12
  ;;
wgred)   # workgroup-level reduction of the tap-gradient partials: parity, then the benches it touches
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
  for wl in cfg3 cfg2 cfg4; do
    timeout 300 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$wl', round(d['ms_per_step'],4), 'ms/step')"
  done
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o cfg3 -- python $OLDPWD/bench.py --workload cfg3 --no-cpu-baseline --steps 50 --warmup 3 > /dev/null 2>&1)
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_cfg3_kernel_stats.csv; rm -rf $O/prof
  python tools/show_stats.py $O | head -12
  ;;
cd8)     # ring depth of contract_panel_kernel<., 8, D> (Cin = 64): kernel averages under rocprofv3, cfg3
  for v in "" d8_3 d8_4; do
    (export GFHIP_EXPERIMENTS=1; [ -n "$v" ] && export GFHIP_LIB=$PWD/graph-neural-networks_amd/alegnn_amd/libgfhip_$v.so
     cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_$v -o cfg3 -- python $OLDPWD/bench.py --workload cfg3 --no-cpu-baseline --steps 50 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('variant ${v:-shipped}:', round(d['ms_per_step'],4), 'ms/step (under rocprof)')")
    f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); grep "contract_panel_kernel" $f | cut -d, -f1-4 | cut -c1-120
    rm -rf $O/prof_$v
  done 2>&1 | tee $O/d8.log
  ;;
final3)  # kernel stats of the other workloads, config-5 PMC, the torchrun / RCCL N = 1 rehearsal -- all on the final sources
  for wl in cfg2 cfg3 cfg5 db; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o $wl -- python $OLDPWD/bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2>&1)
    f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_${wl}_kernel_stats.csv; rm -rf $O/prof
  done
  python tools/show_stats.py $O | head -40
  GFHIP_EXPERIMENTS=1 bash tools/pmc_collect.sh cfg5 ev_hop_lds4_kernel r04 > $O/pmc_cfg5.log 2>&1; tail -1 $O/pmc_cfg5.log | cut -c1-300; cp gpurun_out/pmc_cfg5/r04_cfg5_pmc.json $O/ 2>/dev/null
  timeout 600 bash tools/scale.sh cfg4 1 > $O/scale.log 2>&1; tail -6 $O/scale.log; mkdir -p $O/scale; cp gpurun_out/scale/cfg4_*.json $O/scale/ 2>/dev/null
  ;;
pwc)     # counters of the double-buffered panel kernel (the default at N = 1682), tap-stack mode
  export PROBE_KHOP=1
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
             "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    rm -rf $O/pm; (export TMPDIR=/tmp PROBE_ONLY=1; timeout 100 rocprofv3 --pmc $grp --output-format csv -d $O/pm -o pmc -- python tools/panel_w_probe.py 1682 64 256 v:panel_db=1 > $O/pm.log 2>&1 || echo "group failed: $grp")
    python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmm_panel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("<")[0].split("::")[-1], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (kn, k), v in sorted(agg.items()):
    print(f"{kn:24s} {k:36s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
PY
  done 2>&1 | tee $O/panel_db_counters.log
  rm -rf $O/pm
  ;;
swbar)  # XCD barriers: time and hit rate vs barriers per batch entry
  V="v:spmm_algo=3"; for l in ${LAGS:-0 1 2 4}; do V="$V v:spmm_algo=4+spmm_lag=$l"; done
  timeout 120 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "spmm hop"
  for l in ${PMLAGS:-1 4}; do rm -rf $O/pm; timeout 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pm -o pmc -- python tools/hop_probe.py cfg4 3 v:spmm_algo=4+spmm_lag=$l > $O/pm.log 2>&1
  python3 - "$O" $l <<'PY'
import csv, glob, sys, collections
O, l = sys.argv[1:3]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmm_sweep" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"sweep barriers/entry={l}", {k: max(v) for k, v in sorted(agg.items())})
PY
  done; rm -rf $O/pm
  ;;
swrobust)  # the sweep kernel's LDS variant: repeated bitwise comparisons at several sizes
  for a in "100000 5 128 spmm_lag=1" "100000 5 41 spmm_lag=0" "60000 8 64 spmm_lag=1" "30000 9 24"; do timeout 100 python tools/sweep_debug.py $a 2>&1 | grep -a "mismatching"; done
  timeout 300 python -m pytest tests/test_gpu_sweep.py -x -q 2>&1 | tail -1
  V="v:spmm_algo=3 v:spmm_algo=4+spmm_lag=0 v:spmm_algo=4+spmm_lag=1"
  timeout 120 python tools/hop_probe.py cfg4 10 $V 2>&1 | grep "spmm hop"
  ;;
sw)  # first contact of the sweep kernel: parity, then ER / band timings against SELL-8, then hit rate
  timeout 300 python -m pytest tests/test_gpu_sweep.py -x -q > $O/pytest_sweep.log 2>&1; tail -8 $O/pytest_sweep.log
  V=${SWV:-"v:spmm_algo=3 v:spmm_algo=4+spmm_lag=0+spmm_sd=1 v:spmm_algo=4+spmm_lag=0+spmm_sd=0 v:spmm_algo=4+spmm_lag=4+spmm_sd=0"}
  timeout 120 python tools/hop_probe.py cfg4 10 $V > $O/hop_er.log 2>&1; grep "spmm hop" $O/hop_er.log || tail -5 $O/hop_er.log
  rm -rf $O/pm; timeout 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pm -o pmc -- python tools/hop_probe.py cfg4 3 v:spmm_algo=3 v:spmm_algo=4+spmm_lag=${PMC_LAG:-16} > $O/pm.log 2>&1
  python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        tag = "sweep" if "spmm_sweep" in kn else ("sell" if "spmm_sell" in kn else None)
        if tag:
            agg[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (tag, k), v in sorted(agg.items()):
    print(f"ER {tag:7s} {k:30s} max {max(v):16.0f} n={len(v)}")
PY
  rm -rf $O/pm
  ;;
esac
