#!/bin/bash
# kernel trace + SQ counters only.  usage: scripts/profile_quick.sh <tag> [bench args...]
TAG=$1; shift
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8 # before the profiler's preload initialises the runtime
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 --skip-host-path-check $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.log 2>&1
python3 scripts/summarize_prof.py $OUT $TAG > /dev/null
find $OUT -name "*.csv" -size +20M -delete
python3 - <<PY
import json
d=json.load(open('$OUT/summary_$TAG.json'))
for k,v in d.get('pmc',{}).items():
    print(k[-50:], {a:f"{b:.4g}" for a,b in v.items()})
PY
grep -h '"metric"' $OUT/bench_trace.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('reads/s',d['value'], 'ms/step',d['ms_per_step'], d['kernel_ms'])"
