#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r11; rm -rf $O; mkdir -p $O
timeout 300 python tools/panel_sweep.py cfg2 cfg2w > $O/panel_sweep.log 2>&1; cat $O/panel_sweep.log
