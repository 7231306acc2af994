#!/bin/bash
# round 3, session M: kernel trace of place on the 1000-genome tree (after the weights / compaction changes)
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=$PWD/gpurun_out/r3m_place_trace
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/time_place_big.py 400000 > $OUT/run.log 2>&1
grep -v "^\[\|^W\|^E\|^I" $OUT/run.log | tail -4
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3m_place_trace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'kr_' in r['Name']:
        print(r['Name'][:80], r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6, 3), 'max_ms', round(float(r['MaxNs'])/1e6, 3))
PY
find $OUT -name "*.csv" -size +5M -delete
KR_PLACE_TIMING=1 python scripts/time_place_big.py 400000 > gpurun_out/r3m_place.log 2>&1; grep "place/device" gpurun_out/r3m_place.log | tail -4
python scripts/time_place.py 400000 2>&1 | tail -12
