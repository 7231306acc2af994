#!/bin/bash
# round 5, session 13: kr_place_stream by ranges of reads -- parity (every place test) and rate
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_place.py tests/test_gpu_place_k27.py "tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_never_leaves_the_device" "tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_matches_the_oracle" -m gpu -x -q > gpurun_out/r5_s13_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s13_tests.txt; tail -5 gpurun_out/r5_s13_tests.txt | cut -c1-200
python scripts/sweep_place.py > gpurun_out/r5_s13_sweep_place.txt 2>&1; tail -2 gpurun_out/r5_s13_sweep_place.txt
for k in 4 1 2 8; do
  echo "== KR_PLACE_RANGES=$k"
  KR_PLACE_RANGES=$k python scripts/time_place_big.py > gpurun_out/r5_s13_place_ranges$k.txt 2>&1
  cut -c1-200 gpurun_out/r5_s13_place_ranges$k.txt | head -6
done
python scripts/time_cli.py 16000000 2>&1 | grep "^place" | grep -o "^place [^{]*\|elapsed: [0-9.]* sec ([0-9]* reads/s" | paste - - | head
