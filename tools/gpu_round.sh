#!/bin/bash
# One gpurun call: environment facts, GPU parity tests, smoke, bench (+ per-kernel breakdown). Logs -> gpurun_out/.
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
{
  echo "== host"; nproc; free -g | head -2
  echo "== gpu"; python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.device_count(), torch.version.hip)"
} > gpurun_out/env.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --detail > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err; echo "bench exit $?" >> gpurun_out/bench_cfg2.err
timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 3 --detail --no-cpu-baseline > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err; echo "bench exit $?" >> gpurun_out/bench_cfg4.err
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; cat gpurun_out/bench_cfg2.json; tail -3 gpurun_out/bench_cfg2.err; cat gpurun_out/bench_cfg4.json; tail -3 gpurun_out/bench_cfg4.err
