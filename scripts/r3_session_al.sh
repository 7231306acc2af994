#!/bin/bash
# round 3, session AL: the whole GPU suite and the profile passes on the final code; traffic_latest.json
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r3al_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3al_pytest_gpu.log
tail -10 gpurun_out/r3al_pytest_gpu.log
rm -rf /tmp/pytest-of-* /tmp/krepp_*
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
bash scripts/profile.sh r3d > gpurun_out/prof_r3d.log 2>&1
tail -3 gpurun_out/prof_r3d.log
rm -rf /tmp/krepp_bench_*
python bench.py --steps 10 --warmup 2 > gpurun_out/r3al_bench.json 2> gpurun_out/r3al_bench.err
python scripts/traffic.py gpurun_out/prof_r3d gpurun_out/r3al_bench.json gpurun_out/traffic_r3d.json
tail -c 600 gpurun_out/r3al_bench.json
