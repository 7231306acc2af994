#!/bin/bash
# round 5, session 10: the profile set at HEAD (rocprofv3 stats + PMC on both indexes), the bench lines it must reproduce
# (default flags, the driver's flags, --workload syn10000), and the CLI end to end on both indexes
TAG=r5a
ulimit -c 0
mkdir -p gpurun_out
bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 scripts/traffic.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1
cp gpurun_out/${TAG}_traffic.json profiles/traffic_latest.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
tail -c 400 gpurun_out/${TAG}_bench_driver.json
bash scripts/profile.sh ${TAG}_s10k --workload syn10000 > gpurun_out/${TAG}_s10k_profile.log 2>&1
python3 bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/${TAG}_s10k_bench.json 2> gpurun_out/${TAG}_s10k_bench.err
tail -c 300 gpurun_out/${TAG}_s10k_bench.json
python scripts/time_cli.py 16000000 > gpurun_out/${TAG}_cli_toy25.txt 2>&1
KR_TIME_CLI_TRACE=1 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/${TAG}_cli_syn1000.txt 2>&1
grep "rc 0\|kr_" gpurun_out/${TAG}_cli_syn1000.txt | cut -c1-200
