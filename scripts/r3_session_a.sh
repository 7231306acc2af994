#!/bin/bash
# round 3, first GPU session: the new tests first, then the whole GPU suite, then the default bench
mkdir -p gpurun_out
python -m pytest tests/test_gpu_place_k27.py tests/test_gpu_rccl_cli.py tests/test_gpu_bench.py -m gpu -x -q > gpurun_out/r3a_new_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3a_new_tests.log
tail -15 gpurun_out/r3a_new_tests.log
python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r3a_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3a_pytest_gpu.log
tail -30 gpurun_out/r3a_pytest_gpu.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; tail -c 1500 gpurun_out/r3a_bench.json
