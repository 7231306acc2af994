#!/bin/bash
# Round 3, session AI: placement trials under the tests that use large batches; the default bench (host-inclusive leg included); two ranks on one GPU
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_syn1000.py tests/test_gpu_bench.py -x -q --durations=4 2>&1 | tail -9 | cut -c1-200
rm -rf /tmp/pytest-of-* /tmp/krepp_*
KR_ITEM_PLACEMENT_VERBOSE=1 python3 bench.py --steps 10 > gpurun_out/r3ai_bench.json 2> gpurun_out/r3ai_bench.err
grep -c "item list" gpurun_out/r3ai_bench.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r3ai_bench.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 2), 'M reads/s', round(d['ms_per_step'], 2), 'ms/step', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['config']['item_list_placement'], d['check']['rows_equal'])
print('host-inclusive', round(d['value_host_inclusive']['value'] / 1e6, 2), 'M reads/s; warmup', d['warmup'], 'frac', round(d['roofline']['frac'], 3), 'traffic_frac', d['roofline'].get('traffic_frac'))
PY
