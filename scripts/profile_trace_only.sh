#!/bin/bash
# The kernel-trace + stats pass of scripts/profile.sh alone (the PMC passes are not repeated): usage scripts/profile_trace_only.sh <tag> [bench args]
# Under the profiler bench.py runs without item-list placement trials, so about every other traced process sits on the scan's
# 46 ms level (DESIGN / docs/design/03): run it again for a process on the level the untraced lines are measured on.
TAG=$1; shift
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
export KR_ITEM_PLACEMENT_TRIALS=0
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 --skip-host-path-check $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
python3 scripts/summarize_prof.py $OUT $TAG
find $OUT -name "*.csv" -size +20M -delete
grep "kr_scan_pipe" $OUT/summary_$TAG.txt | head -1 | cut -c1-40,190-300
