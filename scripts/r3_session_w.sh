#!/bin/bash
# Round 3, session W: do physically contiguous allocations (hipDeviceMallocContiguous) remove the scan's slow launch levels?
ulimit -c 0
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8 KR_HBM_VERBOSE=1
mkdir -p gpurun_out
for mode in 3 0 3 2 1 3 0; do
  KR_HBM_CONTIGUOUS=$mode python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --stream-variance 6 \
     > gpurun_out/r3w_m$mode.json 2> gpurun_out/r3w_m$mode.err
  echo "== KR_HBM_CONTIGUOUS=$mode"
  python3 - gpurun_out/r3w_m$mode.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('timed steps:', round(d['value'] / 1e6, 2), 'M reads/s', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['check']['rows_equal'])
PY
  grep -E "stream-variance\] stream [0-9]+:|no contiguous" gpurun_out/r3w_m$mode.err | sed 's/\[stream-variance\] //'
  rm -rf /tmp/krepp_bench_*
done
