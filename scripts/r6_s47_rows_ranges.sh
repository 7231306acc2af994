#!/bin/bash
# round 6, session 47: the row kernel with the reads' record ranges staged in LDS (HEAD) against fetched per step (variants/rowsld0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s47
cp krepp_amd/lib/libkrepp_amd.so /tmp/base.so
for v in rowsld0 base; do
  if [ $v = base ]; then cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
  echo "#### $v"; bash scripts/r6_s46_rows_kernels.sh 2>&1 | grep -E "rows_write|select_lane|dedup_kernel"
  grep '^{"metric"' gpurun_out/s46/bench.log | tail -n 1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); hi=o['value_host_inclusive']
print('  resident', round(o['value']/1e6,2), 'host', round(hi['value']/1e6,2), 'steady', round(hi['steady_state']['value']/1e6,2), hi.get('kernel_ms_in_this_leg'))"
done
cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rows or many_records" 2>&1 | tail -n 2
