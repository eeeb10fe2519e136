#!/bin/bash
# Same-box A/B of the config-4 hop: the shipped library against a variant build (`make variant NAME=x EXTRA=-D...` in graph-neural-networks_amd),
# three interleaved repetitions of tools/hop_probe.py.   usage: tools/ab_hop_variant.sh <variant.so> [workload]
V=$1; W=${2:-cfg4}
for rep in 1 2 3; do
  for lib in "" $V; do
    if [ -n "$lib" ]; then export GFHIP_EXPERIMENTS=1 GFHIP_LIB=$lib; else export GFHIP_EXPERIMENTS=1; unset GFHIP_LIB; fi
    echo -n "lib=${lib:-shipped} "; python tools/hop_probe.py $W 10 2>/dev/null | grep "spmm hop"
  done
done
