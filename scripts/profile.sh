#!/bin/bash
# rocprofv3 passes over the bench command (kernel trace + stats, then PMC passes on their own).
# usage: scripts/profile.sh <tag> [bench args...]
TAG=$1; shift
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8 # before the profiler's preload initialises the runtime
export KR_ITEM_PLACEMENT_TRIALS=0 # (bench.py asks for placement trials during its set-up: under the profiler every launch runs on the one list, so that the tool's per-kernel averages are over like launches)
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
# (--skip-host-path-check: every launch of the profiled kernels is a full-size one, so that the tool's averages are comparable with bench.py's)
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 --skip-host-path-check $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 bench.py $ARGS > $OUT/bench_l2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.log 2>&1
python3 scripts/summarize_prof.py $OUT $TAG
find $OUT -name "*.csv" -size +20M -delete
ls -R $OUT | head -50
