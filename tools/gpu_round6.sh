#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 300 ./tools/gather_ceiling > gpurun_out/gather_ceiling2.log 2>&1; head -12 gpurun_out/gather_ceiling2.log
