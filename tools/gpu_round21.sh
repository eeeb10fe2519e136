#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r21; rm -rf $O; mkdir -p $O
timeout 300 python tools/selgnn_bench.py > $O/selgnn.json 2> $O/selgnn.err; cat $O/selgnn.json; tail -3 $O/selgnn.err
