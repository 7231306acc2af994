#!/bin/bash
# round 3, session AR: select kernel with two reads per pass of a group of lanes
ulimit -c 0
export TMPDIR=/tmp KR_ITEM_PLACEMENT_TRIALS=0
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_long_sequences.py -x -q 2>&1 | tail -3 | cut -c1-200
python3 -m pytest tests/test_gpu_syn1000.py -x -q -k "slotted" 2>&1 | tail -2 | cut -c1-200
rm -rf /tmp/pytest-of-* /tmp/krepp_*
trace() {
  OUT=$PWD/gpurun_out/r3ar_$1; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 > $OUT/bench.log 2>&1
  echo "== $1"; python3 scripts/kstats.py $OUT | grep -E "select|dedup_kernel" | cut -c1-200
  grep -h '"metric"' $OUT/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   rows_equal', d['check']['rows_equal'], 'llh_select', round(d['kernel_ms']['llh_select'],2))"
  rm -rf /tmp/krepp_bench_*
}
trace pair
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/pair0/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
trace single
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
