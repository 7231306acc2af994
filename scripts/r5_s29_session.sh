#!/bin/bash
# round 5, session 29: HEAD at the end of the round -- the whole GPU suite, smoke, the profile set (digests current), bench lines
TAG=r5d
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/${TAG}_tests.txt | head -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 scripts/traffic.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1
cp gpurun_out/${TAG}_traffic.json profiles/traffic_latest.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
tail -c 300 gpurun_out/${TAG}_bench_driver.json
python3 bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/${TAG}_s10k_bench.json 2> gpurun_out/${TAG}_s10k_bench.err
tail -c 200 gpurun_out/${TAG}_s10k_bench.json
