#!/bin/bash
# A/B of two library builds on the SAME box, interleaved (box-to-box variance of the per-kernel averages is 3-6 %): the shipped
# libgfhip.so against a variant (GFHIP_LIB, honoured only with GFHIP_EXPERIMENTS=1).   usage: tools/ab_same_box.sh <variant.so> <outdir> [workloads...]
V=$1; O=${2:-gpurun_out/abx}; shift 2
mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for rep in 1 2; do
  for w in ${@:-cfg2 cfg4 cfg3}; do
    for side in new base; do
      if [ $side = base ]; then export GFHIP_EXPERIMENTS=1 GFHIP_LIB=$V; else unset GFHIP_EXPERIMENTS GFHIP_LIB; fi
      rm -rf $O/kt
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_${w}_${side}_$rep.json 2> $O/err.log
      f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_${side}_${rep}_kernel_stats.csv
      rm -rf $O/kt
    done
  done
done
unset GFHIP_EXPERIMENTS GFHIP_LIB
python - $O <<'PY'
import csv, glob, sys, json, collections, re
O = sys.argv[1]
tab = collections.defaultdict(dict)
for f in sorted(glob.glob(O + '/*_kernel_stats.csv')):
    w, side, rep = re.match(r'.*/(\w+?)_(new|base)_(\d)_kernel_stats.csv', f).groups()
    for r in list(csv.DictReader(open(f)))[:6]:
        name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Name']).split('(')[0][:40]
        tab[(w, name)].setdefault(side, []).append(float(r['AverageNs']) / 1e3)
for (w, name), d in sorted(tab.items()):
    if 'new' in d and 'base' in d:
        n, b = sum(d['new']) / len(d['new']), sum(d['base']) / len(d['base'])
        print(f"{w} {name:42s} base {b:9.1f} us  new {n:9.1f} us  {100 * (n / b - 1):+6.1f} %   (runs: base {d['base']}, new {d['new']})")
    else:
        print(w, name, {k: [round(x, 1) for x in v] for k, v in d.items()})
for f in sorted(glob.glob(O + '/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'], 4))
    except Exception as e:
        print(f, 'no line')
PY
