#!/bin/bash
# round 5, session 20: events of spilled reads compacted into the LDS (KR_ACC_COMPACT_SPILLED): parity, then the accumulate kernel's time
# with and without on both indexes
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_text.py tests/test_gpu_long_sequences.py tests/test_gpu_filter_slots.py \
  tests/test_gpu_syn1000.py tests/test_gpu_place_k27.py -x -q -m gpu > gpurun_out/r5_s20_tests.txt 2>&1
tail -3 gpurun_out/r5_s20_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_compact --workload $w > gpurun_out/r5_s20_${w}_compact.txt 2>&1
  echo "== $w compacted"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s20_${w}_compact.txt
  cp krepp_amd/lib/variants/nocompact/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_nocompact --workload $w > gpurun_out/r5_s20_${w}_nocompact.txt 2>&1
  echo "== $w not compacted"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s20_${w}_nocompact.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
python bench.py > gpurun_out/r5_s20_bench.json 2> gpurun_out/r5_s20_bench.err; cat gpurun_out/r5_s20_bench.json | cut -c1-400
