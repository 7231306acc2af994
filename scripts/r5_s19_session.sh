#!/bin/bash
# round 5, session 19: what the GENERAL epilogue costs, and for which reads (scripts/acc_general_epilogue_ablation.patch: 5 = skipped for every
# read the straight-line epilogue turns away, 6 = skipped for those whose events spilled out of the LDS, 7 = skipped for the others)
ulimit -c 0
mkdir -p gpurun_out
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_full --workload $w > gpurun_out/r5_s19_${w}_full.txt 2>&1
  echo "== $w whole kernel"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s19_${w}_full.txt
  for a in 5 6 7; do
    cp krepp_amd/lib/variants/abl$a/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
    bash scripts/ktimes.sh ${w}_abl$a --workload $w > gpurun_out/r5_s19_${w}_abl$a.txt 2>&1
    echo "== $w general epilogue skipped, mode $a"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s19_${w}_abl$a.txt
  done
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
