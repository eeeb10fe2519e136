#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r26; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "graph_capturable" 2>&1 | tail -15
