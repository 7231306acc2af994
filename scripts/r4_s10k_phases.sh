#!/bin/bash
# round 4: where the accumulate kernel's time goes on the 10,000-genome index (KR_DEBUG_SKIP ablations + the kernel's own statistics)
ulimit -c 0
mkdir -p gpurun_out
B="--workload syn10000 --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 1000 --skip-host-path-check"
for v in ${SKIPS:-0 512 16 2}; do
  KR_DEBUG_SKIP=$v python3 bench.py $B > gpurun_out/s10k_skip_$v.json 2> gpurun_out/s10k_skip_$v.err
  echo "KR_DEBUG_SKIP=$v"; python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/s10k_skip_$v.json') if l.startswith('{\"metric\"')][-1]); print({k:(round(x,2) if isinstance(x,float) else x) for k,x in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')})"
  grep "kr stats" gpurun_out/s10k_skip_$v.err | tail -4
done
