#!/bin/bash
# Round 3, session AJ: per-launch scan times of the timed steps after the placement trials (does the kept list keep its level?)
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do
  KR_ITEM_PLACEMENT_VERBOSE=1 python3 bench.py --steps 10 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3aj_$i.json 2> gpurun_out/r3aj_$i.err
  grep "item list" gpurun_out/r3aj_$i.err | sed 's/\[krepp_amd\] item list //' | tr '\n' ';'; echo
  python3 - gpurun_out/r3aj_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('  ', round(d['value'] / 1e6, 2), 'M reads/s', d['kernel_ms']['scan_per_launch'], d['config']['item_list_placement'])
PY
  rm -rf /tmp/krepp_bench_*
done
