#!/bin/bash
# round 5, session 17: where the straight-line epilogue's time is -- the accumulate kernel with the epilogue cut short at four points
# (scripts/acc_epilogue_ablation.patch builds the variants: results wrong by design, launch times are what is read)
ulimit -c 0
mkdir -p gpurun_out
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_full --workload $w > gpurun_out/r5_s17_${w}_full.txt 2>&1
  echo "== $w whole kernel"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s17_${w}_full.txt
  for a in 1 2 3 4; do
    cp krepp_amd/lib/variants/abl$a/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
    bash scripts/ktimes.sh ${w}_abl$a --workload $w > gpurun_out/r5_s17_${w}_abl$a.txt 2>&1
    echo "== $w cut after step $a"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s17_${w}_abl$a.txt
  done
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
# a traced process on the scan's fast level, for the kernel-stats summary at HEAD (up to four tries)
for t in 1 2 3 4; do
  bash scripts/profile_trace_only.sh r5c$t > gpurun_out/r5c${t}_trace.log 2>&1
  tail -1 gpurun_out/r5c${t}_trace.log
done
