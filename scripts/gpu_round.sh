#!/bin/bash
# One GPU-box round: tests, smoke, bench.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
python bench.py --steps 5 --warmup 1 > gpurun_out/bench.log 2>&1; tail -5 gpurun_out/bench.log
