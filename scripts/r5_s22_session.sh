#!/bin/bash
# round 5, session 22: in-place compaction for reads whose keys do not fit one batch; the LDS event capacity at 768 and 1024 instead of 512;
# what the reads with spilled events still cost (scripts/acc_spilled_reads_ablation.diff: 8 = such a read does nothing after its events
# are collected, 9 = nothing after the compaction)
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle -x -q -m gpu > gpurun_out/r5_s22_tests.txt 2>&1
tail -3 gpurun_out/r5_s22_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  for v in main ev768 ev1024 abl8 abl9; do
    if [ $w = syn10000 ] && [ ${v#abl} != $v ]; then continue; fi
    if [ $v = main ]; then cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
    bash scripts/ktimes.sh ${w}_$v --workload $w > gpurun_out/r5_s22_${w}_$v.txt 2>&1
    echo "== $w $v"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s22_${w}_$v.txt
  done
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
