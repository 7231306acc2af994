#!/bin/bash
# round 3, session AS: syn10000 accumulate time with and without the two-segment launch in the chain
ulimit -c 0
mkdir -p gpurun_out
export KR_ITEM_PLACEMENT_TRIALS=0
run() {
  name=$1; shift
  env "$@" python3 bench.py --workload syn10000 --steps 8 --warmup 2 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3as_$name.json 2> gpurun_out/r3as_$name.err
  python3 - gpurun_out/r3as_$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'] / 1e6, 2), 'M reads/s', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['check']['rows_equal'])
PY
  rm -rf /tmp/krepp_bench_*
}
run default X=1
run no_lean2 KR_DEBUG_NO_LEAN2=1
run default_b X=1
run no_lean2_b KR_DEBUG_NO_LEAN2=1
