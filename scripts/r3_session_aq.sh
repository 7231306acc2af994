#!/bin/bash
# round 3, session AQ: accumulate kernel with the next read's metadata waiting in LDS
ulimit -c 0
mkdir -p gpurun_out
export KR_ITEM_PLACEMENT_TRIALS=0
python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | cut -c1-200
rm -rf /tmp/pytest-of-* /tmp/krepp_*
run() {
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3aq_$1.json 2> gpurun_out/r3aq_$1.err
  python3 - gpurun_out/r3aq_$1.json $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'] / 1e6, 2), 'M reads/s', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['check']['rows_equal'])
PY
  rm -rf /tmp/krepp_bench_*
}
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
run ahead_a
cp krepp_amd/lib/variants/meta0/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
run off_a
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
run ahead_b
cp krepp_amd/lib/variants/meta0/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
run off_b
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
