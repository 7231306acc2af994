#!/bin/bash
# round 5, session 8: select kernel by groups of lanes (A/B), place counters on lines of their own + reads per visit (A/B)
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_syn1000.py::test_syn1000_10gb_index_vs_oracle_and_full_batch_properties --deselect tests/test_gpu_bench.py > gpurun_out/r5_s8_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s8_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_main --workload $w > gpurun_out/r5_s8_ktimes_${w}_main.txt 2>&1
  echo "== $w main (select by groups)"; grep "select" gpurun_out/r5_s8_ktimes_${w}_main.txt
  cp krepp_amd/lib/variants/selgrp0/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_selgrp0 --workload $w > gpurun_out/r5_s8_ktimes_${w}_selgrp0.txt 2>&1
  echo "== $w one read at a time"; grep "select" gpurun_out/r5_s8_ktimes_${w}_selgrp0.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
python scripts/time_place_big.py > gpurun_out/r5_s8_place_main.txt 2>&1
echo "== place main"; cat gpurun_out/r5_s8_place_main.txt | cut -c1-220
cp krepp_amd/lib/variants/plrc16/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
python scripts/time_place_big.py > gpurun_out/r5_s8_place_plrc16.txt 2>&1
echo "== place, 16 reads per visit"; cat gpurun_out/r5_s8_place_plrc16.txt | cut -c1-220
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
