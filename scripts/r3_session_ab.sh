#!/bin/bash
# Round 3, session AB: whole parity file on the current code (select as at the start of the day, dedup with the 16-byte probe), kernel times
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_long_sequences.py tests/test_place.py -x -q -m gpu 2>&1 | tail -5 | cut -c1-300
rm -rf /tmp/pytest-of-* /tmp/krepp_*
OUT=$PWD/gpurun_out/r3ab_trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 > $OUT/bench.log 2>&1
python3 scripts/kstats.py $OUT | grep -E "select|dedup_kernel|llh_kernel|scan_pipe|acc_kernel_t<true, 5, false, 7>" | cut -c1-200
rm -rf /tmp/krepp_bench_*
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reads/s', round(d['value']/1e6,2), 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items() if isinstance(v,float)}, d['check'])"
