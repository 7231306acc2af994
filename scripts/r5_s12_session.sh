#!/bin/bash
# round 5, session 12: the FASTQ parser with SSE2 record checks, chunks parsed from a mapping of the file -- CLI end to end again
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_place.py tests/test_gpu_rccl_cli.py tests/test_gpu_text.py -m gpu -x -q > gpurun_out/r5_s12_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s12_tests.txt
python scripts/time_gz.py > gpurun_out/r5_s12_reader.txt 2>&1; tail -25 gpurun_out/r5_s12_reader.txt | cut -c1-200
python scripts/time_cli.py 16000000 > gpurun_out/r5_s12_cli_toy25.txt 2>&1
grep "elapsed" gpurun_out/r5_s12_cli_toy25.txt | grep -o "^[a-z]* \[[^]]*\] {[^}]*}\|elapsed: [0-9.]* sec ([0-9]* reads/s" | paste - - | head -20
KR_TIME_CLI_CONFIGS=0,6,7,8 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s12_cli_syn1000.txt 2>&1
grep "rc 0\|parse" gpurun_out/r5_s12_cli_syn1000.txt | cut -c1-220
KR_FASTX_MMAP=0 KR_TIME_CLI_CONFIGS=8 python scripts/time_cli_syn1000.py 8e6 2>&1 | grep "rc 0\|parse" | cut -c1-220
