#!/bin/bash
# round 3, session AO: last check of the round -- the whole GPU suite, smoke, the default bench as the driver runs it
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r3ao_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3ao_pytest_gpu.log
tail -9 gpurun_out/r3ao_pytest_gpu.log
rm -rf /tmp/pytest-of-* /tmp/krepp_*
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3ao_bench.json 2> gpurun_out/r3ao_bench.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r3ao_bench.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 2), 'M reads/s', round(d['ms_per_step'], 2), 'ms/step', {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d['kernel_ms'].items() if k != 'source'})
print(d['config']['item_list_placement'], 'host-inclusive', round(d['value_host_inclusive']['value'] / 1e6, 2), 'frac', round(d['roofline']['frac'], 3), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'].get('commit'), d['check'])
PY
