#!/bin/bash
# round 5, session 18: straight-line epilogue for up to 128 keys (two rounds of the per-key step) -- parity, A/B against 64, path statistics
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_text.py tests/test_gpu_long_sequences.py tests/test_gpu_filter_slots.py tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle tests/test_gpu_place_k27.py -x -q > gpurun_out/r5_s18_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s18_tests.txt; tail -3 gpurun_out/r5_s18_tests.txt | cut -c1-200
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_k128 --workload $w > gpurun_out/r5_s18_ktimes_${w}_k128.txt 2>&1
  echo "== $w 128 keys"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s18_ktimes_${w}_k128.txt
  cp krepp_amd/lib/variants/keys64/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_k64 --workload $w > gpurun_out/r5_s18_ktimes_${w}_k64.txt 2>&1
  echo "== $w 64 keys"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s18_ktimes_${w}_k64.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
cp krepp_amd/lib/variants/stats/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
S="--no-cpu-baseline --no-host-inclusive --steps 1 --warmup 0 --check-reads 1000 --skip-host-path-check --distinct-batches 1"
for w in syn1000 syn10000; do KR_ITEM_PLACEMENT_TRIALS=0 KR_DEBUG_SKIP=512 python bench.py --workload $w $S 2>&1 | grep "kr stats\] paths" | cut -c1-330; done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
