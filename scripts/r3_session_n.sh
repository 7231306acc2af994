#!/bin/bash
# round 3, session N: place on the 25-reference index after the heuristic (weights precomputed only on deep trees): phases and kernels
ulimit -c 0
mkdir -p gpurun_out
KR_PLACE_TIMING=1 python scripts/time_place.py 400000 > gpurun_out/r3n_place_toy.log 2>&1
grep -v "^\[place" gpurun_out/r3n_place_toy.log | tail -14; grep "place/device" gpurun_out/r3n_place_toy.log | sed -n '4,9p'
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=$PWD/gpurun_out/r3n_place_trace
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/time_place.py 400000 > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3n_place_trace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'kr_' in r['Name'] and float(r['AverageNs']) > 2e4:
        print(r['Name'][:80], r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6, 3), 'max_ms', round(float(r['MaxNs'])/1e6, 3))
PY
find $OUT -name "*.csv" -size +5M -delete
python scripts/time_place_big.py 400000 2>&1 | tail -3
python -m pytest tests/test_place.py tests/test_gpu_place_k27.py -m gpu -x -q 2>&1 | tail -2
