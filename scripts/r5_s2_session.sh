#!/bin/bash
# round 5, session 2: the device-side report text (kr_dev_text.inc): parity tests, then the CLI end to end on both indexes
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_text.py tests/test_place.py::test_cli_dist_on_a_file_large_enough_for_the_parallel_reader tests/test_gpu_rccl_cli.py -x -q --durations=5 > gpurun_out/r5_s2_tests.txt 2>&1
tail -15 gpurun_out/r5_s2_tests.txt
python scripts/time_cli.py 16000000 > gpurun_out/r5_s2_cli_toy25.txt 2>&1
grep -v "^place" gpurun_out/r5_s2_cli_toy25.txt | head -40
python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s2_cli_syn1000.txt 2>&1
cat gpurun_out/r5_s2_cli_syn1000.txt
