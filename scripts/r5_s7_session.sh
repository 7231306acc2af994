#!/bin/bash
# round 5, session 7: record slots in larger chunks (the accumulate kernel's shared counter), de-duplication table size
ulimit -c 0
mkdir -p gpurun_out
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_main --workload $w > gpurun_out/r5_s7_ktimes_${w}_main.txt 2>&1
  echo "== $w main"; grep "acc_kernel_t<true, 5, false, 7\|dedup_kernel\|select\|scan_pipe\|clear" gpurun_out/r5_s7_ktimes_${w}_main.txt
  cp krepp_amd/lib/variants/rec2048/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_rec2048 --workload $w > gpurun_out/r5_s7_ktimes_${w}_rec2048.txt 2>&1
  echo "== $w rec2048"; grep "acc_kernel_t<true, 5, false, 7\|dedup_kernel\|select\|scan_pipe" gpurun_out/r5_s7_ktimes_${w}_rec2048.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
  for sh in 2 3; do
    KR_DD_SHIFT=$sh bash scripts/ktimes.sh ${w}_sh$sh --workload $w > gpurun_out/r5_s7_ktimes_${w}_sh$sh.txt 2>&1
    echo "== $w dd_shift $sh"; grep "dedup\|select\|llh" gpurun_out/r5_s7_ktimes_${w}_sh$sh.txt
  done
done
