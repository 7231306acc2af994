#!/bin/bash
# round 3, session AP: the scan's episodes (40.9 <-> 44.4 ms within one stream) with and without the other kernels around it
ulimit -c 0
mkdir -p gpurun_out
export KR_ITEM_PLACEMENT_TRIALS=0
run() {
  name=$1; shift
  env "$@" > gpurun_out/r3ap_$name.json 2> gpurun_out/r3ap_$name.err
  python3 - gpurun_out/r3ap_$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['ms_per_step'], 2), 'ms/step; scan per launch:', d['kernel_ms']['scan_per_launch'])
PY
  rm -rf /tmp/krepp_bench_*
}
B="python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000"
run default X=1 $B
run scan_only KR_DEBUG_SKIP=2 $B
run two_streams X=1 $B --pipeline-streams 2
run default_again X=1 $B
