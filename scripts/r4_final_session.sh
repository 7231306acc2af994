#!/bin/bash
# round 4, last GPU session: profile passes + traffic + the two bench lines + the whole GPU suite on the final code.
# usage (on the GPU box): bash scripts/r4_final_session.sh <tag>
TAG=${1:-r6}
ulimit -c 0
mkdir -p gpurun_out
bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 scripts/traffic.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
rm -rf /tmp/krepp_bench_*
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputests.log 2>&1
tail -3 gpurun_out/${TAG}_gputests.log
tail -c 600 gpurun_out/${TAG}_bench.json
