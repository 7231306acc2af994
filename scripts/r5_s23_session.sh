#!/bin/bash
# round 5, session 23: 1,024 events in the LDS, one compaction pass (what does not fit goes back to the global scratch), spilled tiles loaded
# four at a time; parity, then what is left (scripts/acc_spilled_reads_ablation.diff: 8 = a read with spilled events does nothing after its
# events are collected, 9 = nothing after the compaction, 10 = no read does anything after marks / compaction)
ulimit -c 0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_syn1000.py tests/test_gpu_long_sequences.py -x -q -m gpu > gpurun_out/r5_s23_tests.txt 2>&1
tail -3 gpurun_out/r5_s23_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  for v in main abl8 abl9 abl10; do
    if [ $v = main ]; then cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
    bash scripts/ktimes.sh ${w}_$v --workload $w > gpurun_out/r5_s23_${w}_$v.txt 2>&1
    echo "== $w $v"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s23_${w}_$v.txt
  done
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
