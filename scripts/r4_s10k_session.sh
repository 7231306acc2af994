#!/bin/bash
# round 4: the 10,000-genome index (configs[3]'s per-GPU part): profile passes + a bench line
ulimit -c 0
mkdir -p gpurun_out
bash scripts/profile.sh s10k --workload syn10000 > gpurun_out/s10k_profile.log 2>&1
python3 bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/s10k_bench.json 2> gpurun_out/s10k_bench.err
tail -c 1500 gpurun_out/s10k_bench.json
