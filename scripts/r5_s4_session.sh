#!/bin/bash
# round 5, session 4: CLI with detached batches / multi-chunk batches / exit without tear-down; per-kernel times with the direct
# de-duplication on both indexes
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_syn1000.py::test_syn1000_10gb_index_vs_oracle_and_full_batch_properties --deselect tests/test_gpu_bench.py --durations=5 > gpurun_out/r5_s4_tests.txt 2>&1
tail -9 gpurun_out/r5_s4_tests.txt
python scripts/time_cli.py 16000000 > gpurun_out/r5_s4_cli_toy25.txt 2>&1
grep "^dist" gpurun_out/r5_s4_cli_toy25.txt | head -30
KR_TIME_CLI_CONFIGS=0,3,4,6,8 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s4_cli_syn1000.txt 2>&1
cat gpurun_out/r5_s4_cli_syn1000.txt
bash scripts/ktimes.sh s10k --workload syn10000 > gpurun_out/r5_s4_ktimes_syn10000.txt 2>&1
grep -v "relayout\|build_" gpurun_out/r5_s4_ktimes_syn10000.txt
bash scripts/ktimes.sh s1k > gpurun_out/r5_s4_ktimes_syn1000.txt 2>&1
grep -v "relayout\|build_" gpurun_out/r5_s4_ktimes_syn1000.txt
KR_DD_DIRECT=0 bash scripts/ktimes.sh s10k0 --workload syn10000 > gpurun_out/r5_s4_ktimes_syn10000_dd0.txt 2>&1
grep "dedup\|select\|llh" gpurun_out/r5_s4_ktimes_syn10000_dd0.txt
