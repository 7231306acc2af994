#!/bin/bash
# Round 3, session AK: scan launch times around a large allocation and a large free (no placement trials)
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
KR_ITEM_PLACEMENT_TRIALS=0 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --churn-gb 48 > gpurun_out/r3ak.json 2> gpurun_out/r3ak.err
grep churn gpurun_out/r3ak.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r3ak.json').read().strip().splitlines()[-1])
print(d['kernel_ms']['scan_per_launch'])
PY
