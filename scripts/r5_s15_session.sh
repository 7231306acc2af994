#!/bin/bash
# round 5, session 15: the straight-line epilogue with counts instead of planes -- parity, then A/B by kernel time on both indexes
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_text.py tests/test_gpu_long_sequences.py tests/test_gpu_filter_slots.py tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle -x -q > gpurun_out/r5_s15_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s15_tests.txt; tail -3 gpurun_out/r5_s15_tests.txt | cut -c1-200
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_main --workload $w > gpurun_out/r5_s15_ktimes_${w}_counts.txt 2>&1
  echo "== $w counts"; grep "acc_kernel_t<true, 5, false, 7\|scan_pipe" gpurun_out/r5_s15_ktimes_${w}_counts.txt
  cp krepp_amd/lib/variants/planes/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_planes --workload $w > gpurun_out/r5_s15_ktimes_${w}_planes.txt 2>&1
  echo "== $w planes"; grep "acc_kernel_t<true, 5, false, 7\|scan_pipe" gpurun_out/r5_s15_ktimes_${w}_planes.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
cp krepp_amd/lib/variants/stats/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
S="--no-cpu-baseline --no-host-inclusive --steps 1 --warmup 0 --check-reads 1000 --skip-host-path-check --distinct-batches 1"
for w in syn1000 syn10000; do KR_ITEM_PLACEMENT_TRIALS=0 KR_DEBUG_SKIP=512 python bench.py --workload $w $S 2>&1 | grep "kr stats\] paths\|kr stats\] reads [0-9]" | cut -c1-330; done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
