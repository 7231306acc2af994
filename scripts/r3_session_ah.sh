#!/bin/bash
# Round 3, session AH: a stream tries a few allocations of its item list and keeps the one the scan ran fastest on
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_syn1000.py tests/test_gpu_parity.py -x -q -k "slotted or large_batch or golden or lanes or where_a" 2>&1 | tail -3 | cut -c1-200
rm -rf /tmp/pytest-of-* /tmp/krepp_*
run() {
  echo "== $1 (KR_ITEM_PLACEMENT_TRIALS=$2)"
  KR_ITEM_PLACEMENT_TRIALS=$2 KR_ITEM_PLACEMENT_VERBOSE=1 python3 bench.py --steps 6 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3ah_$1.json 2> gpurun_out/r3ah_$1.err
  grep "item list" gpurun_out/r3ah_$1.err | sed 's/\[krepp_amd\] //' | tr '\n' ';'; echo
  python3 - gpurun_out/r3ah_$1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('  ', round(d['value'] / 1e6, 2), 'M reads/s', round(d['ms_per_step'], 2), 'ms/step', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['config']['item_list_placement'], d['check']['rows_equal'])
PY
  rm -rf /tmp/krepp_bench_*
}
for i in 1 2 3 4 5 6; do run on$i 3; done
for i in 1 2 3; do run off$i 0; done
