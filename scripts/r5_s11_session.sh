#!/bin/bash
# round 5, session 11: fuzzers on the GPU (text against the host formatter; configurations, alphabet, long sequences against the
# oracle), the kernels of a CLI batch by name, place by phase
ulimit -c 0
mkdir -p gpurun_out
python scripts/fuzz_text.py 10 > gpurun_out/r5_s11_fuzz_text.txt 2>&1; tail -3 gpurun_out/r5_s11_fuzz_text.txt
python scripts/fuzz_reads.py 6 > gpurun_out/r5_s11_fuzz_reads.txt 2>&1; tail -2 gpurun_out/r5_s11_fuzz_reads.txt
python scripts/sweep_configs.py 11 > gpurun_out/r5_s11_sweep_configs.txt 2>&1; tail -2 gpurun_out/r5_s11_sweep_configs.txt
python scripts/fuzz_long.py 5 > gpurun_out/r5_s11_fuzz_long.txt 2>&1; tail -2 gpurun_out/r5_s11_fuzz_long.txt
python scripts/sweep_libs.py > gpurun_out/r5_s11_sweep_libs.txt 2>&1; tail -2 gpurun_out/r5_s11_sweep_libs.txt
KR_TIME_CLI_CONFIGS=0 KR_TIME_CLI_TRACE=1 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s11_cli_trace.txt 2>&1
grep "rc 0\|kr_\|kernels of" gpurun_out/r5_s11_cli_trace.txt | cut -c1-200
KR_PLACE_TIMING=1 python scripts/time_place_big.py > gpurun_out/r5_s11_place_timing.txt 2>&1
grep "place/device\|tabular: 400000" gpurun_out/r5_s11_place_timing.txt | head -40 | cut -c1-200
