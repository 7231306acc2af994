#!/bin/bash
# One PMC pass with an arbitrary counter list.  usage: scripts/pmc.sh <tag> "<counters>" [bench args...]
TAG=$1; CTRS=$2; shift 2
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8 # before the profiler's preload initialises the runtime
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 --check-reads 1000 "$@" > $OUT/bench.log 2>&1
python3 scripts/summarize_prof.py $OUT $TAG > /dev/null
find $OUT -name "*.csv" -size +20M -delete
python3 - <<PY
import json
d=json.load(open('$OUT/summary_$TAG.json'))
for k,v in d.get('pmc',{}).items():
    print(k.split('kr_')[-1][:28], {a:f"{b:.4g}" for a,b in v.items()})
PY
