#!/bin/bash
# round 6, session 42: de-duplication kernel with only its coalesced loads run ahead (variants/dpipe2) against HEAD (no pipeline)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s42
cp krepp_amd/lib/libkrepp_amd.so /tmp/base.so
for v in base dpipe2; do
  if [ $v = base ]; then cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
  echo "#### $v"; bash scripts/r6_s40_kernel_times.sh 2>&1 | grep -E "==|dedup_kernel|select_lane|llh_kernel "
done
cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so
