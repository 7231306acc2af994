#!/bin/bash
# round 3, session I: the whole GPU suite on the current code, the profile passes behind profiles/round3_a_*, place timing on the 1000-genome tree
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3i_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3i_pytest_gpu.log
tail -14 gpurun_out/r3i_pytest_gpu.log
bash scripts/profile.sh r3a > gpurun_out/prof_r3a.log 2>&1
tail -5 gpurun_out/prof_r3a.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r3i_bench.json 2> gpurun_out/r3i_bench.err
python scripts/traffic.py gpurun_out/prof_r3a gpurun_out/r3i_bench.json gpurun_out/traffic_r3a.json
KR_PLACE_TIMING=1 python scripts/time_place_big.py 400000 > gpurun_out/r3i_place.log 2>&1; grep -v "^\[place" gpurun_out/r3i_place.log | tail -4; grep "place/device\|\[place\]" gpurun_out/r3i_place.log | tail -12
