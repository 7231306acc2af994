#!/bin/bash
# round 5, session 21: straight-line epilogue with key batches and exact handling of a position hit twice, spilled reads compacted up to
# 768 live events: parity (whole suite), then the accumulate kernel's time with and without the compaction
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_sequences.py tests/test_gpu_syn1000.py -x -q -m gpu > gpurun_out/r5_s21_tests.txt 2>&1
tail -3 gpurun_out/r5_s21_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_compact --workload $w > gpurun_out/r5_s21_${w}_compact.txt 2>&1
  echo "== $w compacted"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s21_${w}_compact.txt
  cp krepp_amd/lib/variants/nocompact/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_nocompact --workload $w > gpurun_out/r5_s21_${w}_nocompact.txt 2>&1
  echo "== $w not compacted"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s21_${w}_nocompact.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
python bench.py > gpurun_out/r5_s21_bench.json 2> gpurun_out/r5_s21_bench.err; cut -c1-300 gpurun_out/r5_s21_bench.json
python bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/r5_s21_bench_syn10000.json 2> gpurun_out/r5_s21_bench_syn10000.err; cut -c1-300 gpurun_out/r5_s21_bench_syn10000.json
timeout 1200 python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_long_sequences.py --deselect tests/test_gpu_syn1000.py > gpurun_out/r5_s21_tests2.txt 2>&1
tail -3 gpurun_out/r5_s21_tests2.txt
