#!/bin/bash
# round 4: compressed query files on the GPU box (reader alone + CLI), then the two bench lines once more (final bench.py)
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python3 scripts/time_gz.py 3000000 > gpurun_out/r7_time_gz.txt 2>&1
rm -rf /tmp/krepp_gz_*
python3 bench.py > gpurun_out/r7_bench.json 2> gpurun_out/r7_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r7_bench_driver.json 2> gpurun_out/r7_bench_driver.err
timeout 600 python3 -m pytest tests/test_gpu_rccl_cli.py -x -q > gpurun_out/r7_cli_tests.log 2>&1
tail -40 gpurun_out/r7_time_gz.txt; tail -3 gpurun_out/r7_cli_tests.log
