#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r11; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel or pipelines" > $O/pytest_panel.log 2>&1; echo "pytest exit $?" >> $O/pytest_panel.log
tail -4 $O/pytest_panel.log
timeout 300 python tools/panel_sweep.py cfg2 > $O/panel_sweep.log 2>&1; cat $O/panel_sweep.log
